"""Renders frames with the library RTMI_LIB names and prints a hash of each image and its counters:
run it under two builds to see whether a change kept the images bit for bit."""
import sys, os, hashlib, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracing_amd import abi, scenes
cases = [("ch11a", 3840, 2160, 1), ("ch11a", 3840, 2160, 8), ("ch11a", 640, 360, 1)]
if len(sys.argv) > 1 and sys.argv[1] == "all":
    cases += [("dice", 960, 540, 1), ("teapot", 960, 540, 1), ("gallery", 400, 300, 1), ("soft", 640, 360, 1), ("dof", 640, 360, 1)]
stream = torch.cuda.current_stream().cuda_stream
for name, h, v, nr in cases:
    if name == "ch11a": world, cam = scenes.chapter11a(h, v)
    elif name == "dice": world, cam = scenes.chapter16(h, v)
    elif name == "teapot": world, cam = scenes.chapter15(h, v)
    elif name == "soft": world, cam = scenes.soft_shadow(h, v)
    elif name == "dof": world, cam = scenes.depth_of_field(h, v)
    else: world, cam = scenes.lights_gallery(h, v, deterministic=False)
    sc = abi.Scene(world)
    for r in ([0] if nr == 1 else [0, nr - 1]):
        rows = abi.rank_rows(v, r, nr, 8)
        buf = torch.zeros((rows, h, 3), dtype=torch.float32, device="cuda")
        st = torch.zeros(abi.N_STATS, dtype=torch.int64, device="cuda")
        opts = abi.make_opts(rank=r, nranks=nr, tile_rows=8)
        abi.render_device(sc, cam, opts, buf, st, stream)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3): abi.render_device(sc, cam, opts, buf, None, stream)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        hsh = hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()[:16]
        s = st.cpu().tolist()
        print(name, h, v, "rank %d/%d" % (r, nr), hsh, "ms %.3f" % ms, "stats", s[:8], s[9:], flush=True)
