"""The oracle's restatement of the tracing half of World::compute_photon_map (world.rs:186-273, materials.rs:414-464,
photon_map.rs:144-172, the lights' emit_photon) - CPU only.

Pinned by the reference's own unit tests where it has them: TravellingPhoton::reflect / refract (photon_map.rs:475-525).
The reference has no test of scatter ("How to test random functions?", photon_map.rs:470-473), of the lights' emit_photon
or of trace_photon, and draws from thread_rng: for those the parity is UNPINNED by fixtures - the tests below check what
the cited code implies for any stream (unit directions, the cosine law, conservation of the emitted power, the kinds the
flags let through, the order of the collected hits)."""
import ctypes as C
import math

import numpy as np
import pytest

import oracle_lib as orc
import photon_trace_cases as cases
from raytracing_amd import abi


def test_photon_reflect():  # photon_map.rs:475-481
    assert np.allclose(orc.photon_reflect((1, -1, -1), (0, 1, 0)), (1, 1, -1), atol=1e-5)


def test_photon_refract_equal_index():  # photon_map.rs:483-495
    d = np.array([1.0, -1.0, 0.0]) / math.sqrt(2.0)
    assert np.allclose(orc.photon_refract(d, (0, 1, 0), 1.0, 1.0), d, atol=1e-5)


def test_photon_refract_lo_hi():  # photon_map.rs:497-510
    d = np.array([1.0, -1.0, 0.0]) / math.sqrt(2.0)
    r = orc.photon_refract(d, (0, 1, 0), 1.0, 2.0)
    assert abs(np.dot(r, r) - 1.0) < 1e-5 and np.allclose(r, (0.353553, -0.935414, 0), atol=1e-5)


def test_photon_refract_hi_lo():  # photon_map.rs:512-525
    d = np.array([1.0, -1.0, 0.0]) / math.sqrt(2.0)
    r = orc.photon_refract(d, (0, 1, 0), 1.3, 1.0)
    assert abs(np.dot(r, r) - 1.0) < 1e-5 and np.allclose(r, (0.919239, -0.393700, 0), atol=1e-5)


def test_refraction_beyond_the_critical_angle_has_no_answer():  # photon_map.rs:165: sqrt of a negative number
    d = np.array([1.0, -0.2, 0.0])
    d /= np.linalg.norm(d)
    assert orc.photon_refract(d, (0, 1, 0), 1.5, 1.0) is None


def test_cosine_distribution():  # cosine_distribution.rs:16-28: unit vectors in the hemisphere of the direction, E[cos] = 2/3
    n = np.array([0.0, 0.6, 0.8])
    v = np.array([orc.cosine_sample(n, 5, i) for i in range(4000)])
    assert np.allclose(np.linalg.norm(v, axis=1), 1.0, atol=1e-12)
    c = v @ n
    assert c.min() >= -1e-12 and abs(c.mean() - 2.0 / 3.0) < 0.02


def test_direct_photons_of_the_example_room():
    """examples/photon-map.rs' room is closed: every photon of the disc light hits something (rays == emitted with the
    direct map alone, world.rs:228-232), all but those landing on the glass ball (no diffuse term) are stored, each with
    1 / n of the light's photon power 1000 * 0.25 * 2 / pi per channel (lights.rs:471, world.rs:200), all travelling down."""
    f = cases.box().flatten()
    n = 4000
    ph, st = orc.trace_photons(f, abi.make_photon_opts(n, seed=3, direct=True, diffuse=False, caustic=False))
    assert st["n_rays"] == n and 0.85 * n < st["n_stored"] <= n and st["n_truncated"] == 0
    per = 1000.0 * 0.25 * 2.0 / math.pi / n
    assert np.allclose(cases.power_of(ph), per, rtol=1.01 / 255.0)  # (compress_color truncates: 255 * c / c can come out as 254)
    assert (ph["direction"][:, 1] > 0).all()  # stored: MINUS the direction of travel (photon_map.rs:140-142)
    assert np.allclose(np.linalg.norm(ph["direction"], axis=1), 1.0, atol=1e-5)


def test_kinds_follow_the_flags():
    """world.rs:256-265: the default flags (direct off) store no first hits - with them on, the count grows by exactly the
    direct photons; caustic photons are the ones behind the glass ball."""
    f = cases.box().flatten()
    n = 3000
    a = orc.trace_photons(f, abi.make_photon_opts(n, seed=3, direct=False, diffuse=True, caustic=True))[1]
    b = orc.trace_photons(f, abi.make_photon_opts(n, seed=3, direct=True, diffuse=True, caustic=True))[1]
    d = orc.trace_photons(f, abi.make_photon_opts(n, seed=3, direct=True, diffuse=False, caustic=False))[1]
    assert a["n_rays"] == b["n_rays"]  # the same paths: the flags only choose what is kept ...
    assert b["n_stored"] - a["n_stored"] == d["n_stored"]  # ... (enable_diffuse stays on in a and b: it also feeds the roulette)
    c = orc.trace_photons(f, abi.make_photon_opts(n, seed=3, direct=False, diffuse=False, caustic=True))
    assert 0 < c[1]["n_stored"] < 0.2 * n


def test_order_prefix_and_count_query():
    f = cases.lights_room().flatten()
    opts = abi.make_photon_opts(1500, seed=9, direct=True)
    full, st = orc.trace_photons(f, opts)
    head, st2 = orc.trace_photons(f, opts, capacity=100)
    assert st == st2 and len(full) == st["n_stored"] and np.array_equal(full[:100], head)


def test_conservation_of_power_in_expectation():
    """Russian roulette (materials.rs:443-461) keeps a photon's power through specular bounces and rescales it by
    reflectance / mean reflectance on a diffuse one: the stored power per generation falls like the mean diffuse
    reflectance (0.5 x colour) - the sum over all stored photons stays below emitted / (1 - 0.5)."""
    f = cases.box().flatten()
    n = 3000
    ph, st = orc.trace_photons(f, abi.make_photon_opts(n, seed=21, direct=True, diffuse=True, caustic=True))
    emitted = 3.0 * 1000.0 * 0.25 * 2.0 / math.pi
    total = cases.power_of(ph).sum()
    assert 0.9 * emitted < total < 2.0 * emitted


def test_errors_of_the_reference_come_back_as_codes():
    from raytracing_amd import scenes
    w = scenes.photon_room(16, 16)[0]  # its glass sphere is Phong::default() glass: diffuse 0.9 + specular 0.9 + transparency 1
    ph, st = orc.trace_photons(w.flatten(), abi.make_photon_opts(2000, seed=1))
    assert st["n_stored"] <= -2  # materials.rs:443-445 panics there
    from raytracing_amd import _front as rt
    w2 = rt.World.empty()
    w2.add_light(rt.AmbientLight(rt.color(1, 1, 1)))
    w2.add_item(rt.sphere())
    w2.finalize_scene()
    assert orc.trace_photons(w2.flatten(), abi.make_photon_opts(10))[1]["n_stored"] == -1  # world.rs:212 panics


def test_entry_point_refuses_bad_arguments_without_a_gpu():
    L = abi.lib()
    res = abi.PhotonResult()
    assert L.rtmi_trace_photons(None, None, None, 0, C.byref(res)) == -1
    assert "trace_photons" in L.rtmi_last_error().decode()
