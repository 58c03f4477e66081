import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracing_amd import abi, scenes, tiles
world, cam = scenes.chapter11a(200, 100)
sc = abi.Scene(world)
full, st1 = abi.render_host(sc, cam, abi.make_opts())
full2, st2 = abi.render_host(sc, cam, abi.make_opts())
d = np.abs(full - full2).max(axis=2)
print("run-to-run: differing px", (d > 0).sum(), "max", d.max(), st1["samples"], st2["samples"])
h, v = cam.hsize(), cam.vsize()
for nr in (1, 2, 4):
    out = np.zeros((v, h, 3), dtype=np.float32)
    tot = 0
    for r in range(nr):
        rows = abi.rank_rows(v, r, nr, 8)
        buf = torch.zeros((rows, h, 3), dtype=torch.float32, device="cuda")
        stats = torch.zeros(abi.N_STATS, dtype=torch.int64, device="cuda")
        abi.render_device(sc, cam, abi.make_opts(rank=r, nranks=nr, tile_rows=8), buf, stats, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        tot += int(stats[4])
        out[tiles.rank_row_index(v, r, nr, 8).numpy()] = buf.cpu().numpy()
    d = np.abs(out - full).max(axis=2)
    ys, xs = np.nonzero(d > 0)
    print("nranks", nr, "differing px", len(ys), "max", d.max(), "samples", tot, list(zip(ys[:10], xs[:10])), d[ys[:10], xs[:10]])
