"""Photon maps for the tests of the gather (materials.rs:376-384).  The library traces photons too (rtmi_trace_photons, the tracing
half of World::compute_photon_map, world.rs:186-273: tests/test_gpu_photon_trace.py); the gather's tests do not depend on it and
make a map themselves: photons sent from a point in seeded random directions and
stored where the ORACLE says they first hit (direction = minus the direction of travel, power = colour / n, as emit_photon + store +
scale_power leave a direct photon, photon_map.rs:140-142 and world.rs:201), plus a seeded scatter of them around, so that estimates see photons from
all sides.  Both the oracle and the device then get the same array."""
import numpy as np

import oracle_lib as orc


def direct_photons(world, origin, n, seed, colour=(1.0, 0.95, 0.9)):
    flat = world.flatten()  # (no map attached yet)
    rng = np.random.default_rng(seed)
    pos, dirs, powers = [], [], []
    o = np.asarray(origin, dtype=np.float64)
    tries = 0
    while len(pos) < n and tries < 50 * n:
        tries += 1
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        if d[1] > 0.3:  # (mostly downwards: the scene lies below the light)
            continue
        rec = orc.hit_record(flat, o, d)
        if not rec.hit:
            continue
        pos.append(np.array(rec.point[:3]) + rng.normal(scale=0.01, size=3))
        dirs.append(-d)  # TravellingPhoton::store, photon_map.rs:141
        powers.append(np.asarray(colour) * rng.uniform(0.5, 1.5) * 250.0 / n)
    ph = orc.make_photons(np.array(pos), np.array(dirs), np.array(powers))
    return ph


def attach(world, photons, n_nearest, max_search_radius=0.0):
    world.set_photon_arrays(np.ascontiguousarray(photons["position"]), np.ascontiguousarray(photons["direction"]),
                            np.ascontiguousarray(photons["power_scale"]), np.ascontiguousarray(photons["power_rgb"]), n_nearest, max_search_radius)
