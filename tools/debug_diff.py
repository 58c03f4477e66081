"""Debug helper: lists pixels where the HIP path and the oracle disagree and compares their hit records."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as orc
from raytracing_amd import abi, scenes

name = sys.argv[1]; prec = int(sys.argv[2]); kw = eval(sys.argv[3]) if len(sys.argv) > 3 else {}
scheme = name.startswith("scheme:")
if scheme:
    world, cam = scenes.SCHEME_GOLDENS[name[7:]][0]()
else:
    world, cam = scenes.BASELINE_SCENES[name](192, 96, deterministic=True, **kw)
ref, rst = orc.render(world, cam)
sc = abi.Scene(world)
img, st = abi.render_host(sc, cam, abi.make_opts(precision=prec))
d = np.abs(orc.to_u8(img.astype(np.float64)).astype(int) - orc.to_u8(ref).astype(int)).max(axis=2)
ys, xs = np.nonzero(d > 1)
print("bad", len(ys), "of", d.size)
flat = world.flatten()
for y, x in list(zip(ys, xs))[:int(os.environ.get("NSHOW", "12"))]:
    o, dd = orc.camera_ray(cam, int(x), int(y))
    rec = orc.hit_record(flat, o, dd)
    rgb, recs = abi.trace_rays(sc, [o], [dd], opts=abi.make_opts(precision=prec))
    g = recs[0]
    print("px", x, y, "gpu", img[y, x], "ref", ref[y, x])
    print("   orc: hit", rec.hit, "leaf", rec.leaf, "kind", flat.leaf_kind(rec.leaf) if rec.hit else None, "t %.9g" % rec.t, "in", rec.inside, "n", list(rec.normal), "n1n2", rec.n1, rec.n2, "sh", rec.shadowed)
    print("   gpu: hit", g.hit, "leaf", g.leaf, "kind", flat.leaf_kind(g.leaf) if g.hit else None, "t %.9g" % g.t, "in", g.inside, "n", list(g.normal), "n1n2", g.n1, g.n2, "sh", g.shadowed, "rgb", rgb[0])
