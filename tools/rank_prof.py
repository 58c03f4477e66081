import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracing_amd import abi, scenes
nr = int(sys.argv[1]); cap = int(sys.argv[2]) if len(sys.argv) > 2 else 0
name = sys.argv[3] if len(sys.argv) > 3 else "ch11a"
H = int(sys.argv[4]) if len(sys.argv) > 4 else 3840
V = int(sys.argv[5]) if len(sys.argv) > 5 else 2160
world, cam = {"ch11a": scenes.chapter11a, "soft": scenes.soft_shadow, "dof": scenes.depth_of_field, "dice": scenes.chapter16,
              "teapot": scenes.chapter15, "gallery": scenes.lights_gallery}[name](H, V)
sc = abi.Scene(world)
stream = torch.cuda.current_stream().cuda_stream
for r in range(nr):
    rows = abi.rank_rows(V, r, nr, 8)
    buf = torch.zeros((rows, H, 3), dtype=torch.float32, device="cuda")
    opts = abi.make_opts(rank=r, nranks=nr, tile_rows=8, max_samples=cap)
    for _ in range(3):
        abi.render_device(sc, cam, opts, buf, None, stream)
    torch.cuda.synchronize()
