"""Debug helper: random rays into a scene, list where the f32 device path's first hit differs from the oracle's."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as orc
from raytracing_amd import abi, scenes
name = sys.argv[1]
kw = {"teapot": {"lite": True}, "dice": {"nx": 1, "ny": 1}}.get(name, {})
world, _ = scenes.BASELINE_SCENES[name](32, 16, **kw)
f = world.flatten()
rng = np.random.default_rng(7)
n = 10000
for _ in range({"ch11a": 1, "soft": 2, "teapot": 3, "dice": 4}.get(name, 1)):
    o = rng.uniform(-4, 4, (n, 3)) + np.array([0, 2.0, -3.0])
    tgt = rng.uniform(-2, 2, (n, 3)) + np.array([0, 0.5, 0.5])
d = tgt - o
d /= np.linalg.norm(d, axis=1, keepdims=True)
sc = abi.Scene(world)
_, r32 = abi.trace_rays(sc, o, d, opts=abi.make_opts(precision=0))
bad = 0
for i in range(n):
    ref = orc.hit_record(f, o[i], d[i])
    g = r32[i]
    if g.hit != ref.hit or (ref.hit and g.leaf != ref.leaf):
        bad += 1
        if bad <= 25:
            ts, lv = orc.world_intersect(f, o[i], d[i])
            print(i, "ref", ref.hit, ref.leaf, ref.t, "gpu", g.hit, g.leaf, g.t, "o", o[i], "d", d[i], "xs", list(zip(np.round(ts, 6), lv))[:6])
print("bad", bad, "of", n)
for l in range(f.n_leaves):
    print(l, f.leaf_kind(l), np.round(f.leaf_inv(l), 4).tolist())
