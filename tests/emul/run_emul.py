"""Runs rtmi_trace_rays / rtmi_camera_rays of the HOST EMULATION build (tests/emul/librtmi_emul.so, ASan + UBSan) over
random rays into every BASELINE scene and compares the f64 results with the oracle.  Usage (from the repo root):

    make -C tests/emul && LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:verify_asan_link_order=0 \
        RTMI_NO_TORCH=1 RTMI_LIB=$PWD/tests/emul/librtmi_emul.so python tests/emul/run_emul.py [n_rays]

Any out-of-bounds access of a scene record, stack or queue aborts with an ASan report.  This is how a changed kernel is
checked on the CPU before it is allowed on a GPU box."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
assert "emul" in os.environ.get("RTMI_LIB", ""), "set RTMI_LIB to the emulation library"
import oracle_lib as orc  # noqa: E402
from raytracing_amd import abi, scenes  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(7)
print(abi.describe())
import photon_cases  # noqa: E402


def photon_room(h, v, **kw):
    """scenes.photon_room with a map of 3000 photons attached: the gather of materials.rs:376-384 in trace_kernel<R, true>"""
    world, cam = scenes.photon_room(h, v, **kw)
    photon_cases.attach(world, photon_cases.direct_photons(world, (-3, 6, -4), 3000, 3), 24, 0.6)
    return world, cam


EXTRA = {"gallery": scenes.lights_gallery, "yard": scenes.yard, "yard-group": scenes.yard, "photons": photon_room, "photons-only": photon_room,
         "crowd": scenes.crowd, "meshes": scenes.meshes}
for name, kw in [("ch07", {}), ("ch11a", {}), ("ch11b", {}), ("soft", {}), ("dof", {}), ("teapot", {"lite": True}), ("dice", {"nx": 1, "ny": 1}), ("gallery", {}), ("crowd", {}), ("meshes", {}),
                 ("yard", {}), ("yard-group", {"bounded": False}), ("photons", {}), ("photons-only", {"direct": False})]:
    world, cam = (EXTRA[name] if name in EXTRA else scenes.BASELINE_SCENES[name])(32, 16, **kw)
    f = world.flatten()
    sc = abi.Scene(world)
    o = rng.uniform(-4, 4, (n, 3)) + np.array([0, 2.0, -3.0])
    tgt = rng.uniform(-2, 2, (n, 3)) + np.array([0, 0.5, 0.5])
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    c64, r64 = abi.trace_rays(sc, o, d, opts=abi.make_opts(precision=1))
    c32, r32 = abi.trace_rays(sc, o, d, opts=abi.make_opts(precision=0))
    bad = 0
    same32 = 0
    col32 = 0
    for i in range(n):
        ref = orc.hit_record(f, o[i], d[i])
        g = r64[i]
        ok = g.hit == ref.hit and (not ref.hit or (g.leaf == ref.leaf and abs(g.t - ref.t) < 1e-7 * max(1.0, abs(ref.t)) and (g.n1, g.n2) == (ref.n1, ref.n2)))
        if name not in ("soft", "gallery"):  # (scenes with randomly sampled lights: colours are compared by the image tests)
            rc = orc.color_at(f, o[i], d[i], world.max_reflection_depth)[1]
            if ok:
                ok = bool(np.all(np.abs(c64[i] - rc) < 1e-9))
            # the f32 product path (flat scenes: the LDS block and pair tables of the render kernels): colour to 1/255, n1 / n2 of the first hit
            col32 += int(bool(np.all(np.abs(c32[i] - rc) < 4e-3)) and (not ref.hit or r32[i].leaf != ref.leaf or (abs(r32[i].n1 - ref.n1) < 1e-6 and abs(r32[i].n2 - ref.n2) < 1e-6)))
        else:
            col32 += 1
        bad += 0 if ok else 1
        same32 += int(r32[i].hit == ref.hit and (not ref.hit or r32[i].leaf == ref.leaf or abs(r32[i].t - ref.t) < 1e-5 * max(1.0, ref.t)))
    px = rng.integers(0, 32, 64).astype(np.int32); py = rng.integers(0, 16, 64).astype(np.int32)
    abi.camera_rays(cam, px, py, np.arange(64, dtype=np.int32), opts=abi.make_opts(precision=0))
    sc.close()
    print("%-7s rays %d  f64 mismatches vs oracle %d  f32 same leaf %.2f %%  f32 colour within 4e-3 and same n1/n2 %.2f %%" % (name, n, bad, 100.0 * same32 / n, 100.0 * col32 / n), flush=True)
    assert bad == 0 and same32 >= 0.995 * n and col32 >= 0.98 * n
# The host side of rtmi_render (cached device frame, pinned staging, pieces copied on by threads) and of rtmi_render_multi
# (per-share buffers, streams and events kept with scenes[0]) under ASan.  The render kernels themselves are not emulated
# (their launcher is a no-op here), so only memory handling is checked: sizes, reuse across calls, a change of frame
# size, several "devices" worth of shares, release at scene destruction.
class HostBuffer:
    def __init__(self, shape):
        self.a = np.zeros(shape, dtype=np.float32)

    def data_ptr(self):
        return self.a.ctypes.data


world, cam = scenes.chapter11a(64, 40)
world2, cam2 = scenes.chapter11a(96, 56)
sc = abi.Scene(world)
for c in (cam, cam, cam2, cam):
    img, st = abi.render_host(sc, c)
    assert img.shape == (c.vsize(), c.hsize(), 3)
replicas = [abi.Scene(world) for _ in range(3)]
for c, n in ((cam, 3), (cam, 3), (cam2, 2), (cam, 1)):
    out = HostBuffer((c.vsize(), c.hsize(), 3))
    abi.render_multi(replicas[:n], c, abi.make_opts(), out, tile_rows=8)
for r in replicas:
    r.close()
sc.close()
print("host paths of rtmi_render / rtmi_render_multi clean")
print("emulation run clean")
