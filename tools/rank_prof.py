import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracing_amd import abi, scenes
nr = int(sys.argv[1]); cap = int(sys.argv[2]) if len(sys.argv) > 2 else 0
world, cam = scenes.chapter11a(3840, 2160)
sc = abi.Scene(world)
stream = torch.cuda.current_stream().cuda_stream
for r in range(nr):
    rows = abi.rank_rows(2160, r, nr, 8)
    buf = torch.zeros((rows, 3840, 3), dtype=torch.float32, device="cuda")
    opts = abi.make_opts(rank=r, nranks=nr, tile_rows=8, max_samples=cap)
    for _ in range(3):
        abi.render_device(sc, cam, opts, buf, None, stream)
    torch.cuda.synchronize()
