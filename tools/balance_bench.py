"""The reference's only timing program, rust/examples/benchmark_photon_map_balancing.rs:9-47 ("Balancing": n photons uniform
in [-1, 1]^3, the time of PhotonMap::from_vec), on this library: the time of rtmi_scene_create for a world of one sphere
that carries such a map (validation + PhotonMap::balance on host threads + the upload of 32 bytes per photon).
The reference prints its timings and records none.  Usage: python tools/balance_bench.py [n ...]
(RTMI_LIB=tests/emul/librtmi_emul_fast.so RTMI_NO_TORCH=1 times the HOST code alone in a container without a GPU)"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import abi, _front as rt  # noqa: E402

ns = [int(x) for x in sys.argv[1:]] or [1_000, 10_000, 100_000, 1_000_000, 10_000_000]
rng = np.random.default_rng(1)
out = {"what": "PhotonMap::from_vec as benchmark_photon_map_balancing.rs measures it, inside rtmi_scene_create", "threads": os.environ.get("RTMI_HOST_THREADS", "default (<= 16)"), "rows": []}
for n in ns:
    w = rt.World.empty()
    w.add_light(rt.PointLight(rt.point(0, 5, 0), rt.color(1, 1, 1)))
    w.add_item(rt.sphere())
    w.finalize_scene()
    pos = rng.uniform(-1.0, 1.0, (n, 3)).astype(np.float32)
    w.set_photon_arrays(pos, np.tile(np.array([0, 1, 0], dtype=np.float32), (n, 1)), np.ones(n, dtype=np.float32), np.full((n, 3), 255, dtype=np.uint8), 1, 0.0)
    best = None
    for _ in range(3 if n <= 1_000_000 else 2):
        t0 = time.perf_counter()
        sc = abi.Scene(w)
        dt = time.perf_counter() - t0
        sc.close()
        best = dt if best is None else min(best, dt)
    out["rows"].append({"n": n, "ms": round(best * 1e3, 2), "Mphotons_per_s": round(n / best / 1e6, 2)})
print(json.dumps(out))
