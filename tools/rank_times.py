"""Per-rank render time of the sharded headline frame, one rank at a time on one GPU: what each GPU of an N-GPU run does."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracing_amd import abi, scenes
world, cam = scenes.chapter11a(3840, 2160)
sc = abi.Scene(world)
stream = torch.cuda.current_stream().cuda_stream
for nr in (1, 2, 4, 8):
    times = []
    for r in range(nr):
        rows = abi.rank_rows(2160, r, nr, 8)
        buf = torch.zeros((rows, 3840, 3), dtype=torch.float32, device="cuda")
        opts = abi.make_opts(rank=r, nranks=nr, tile_rows=8)
        for _ in range(2):
            abi.render_device(sc, cam, opts, buf, None, stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            abi.render_device(sc, cam, opts, buf, None, stream)
        e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1) / 5)
    print("nranks %d: per-rank ms min %.3f max %.3f  -> ideal speedup %.2f" % (nr, min(times), max(times), t1 / max(times) if nr > 1 else 1.0) if nr > 1 or not globals().__setitem__("t1", max(times)) else "")
