"""tools/inflight.py <nranks> : what rank 0's share of the headline frame costs alone - throughput of back-to-back frames with 1 and 2 frames in flight (one HIP stream and work slot per frame in flight)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracing_amd import abi, scenes
nr = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H, V = 3840, 2160
world, cam = scenes.chapter11a(H, V)
sc = abi.Scene(world)
rows = abi.rank_rows(V, 0, nr, 8)
opts = abi.make_opts(rank=0, nranks=nr, tile_rows=8)
ks = [int(x) for x in sys.argv[2].split(',')] if len(sys.argv) > 2 else [1, 2, 1, 2]
all_streams = [torch.cuda.Stream() for _ in range(max(ks))]
for k in ks:
    streams = all_streams[:k]
    bufs = [torch.zeros((rows, H, 3), dtype=torch.float32, device="cuda") for _ in range(k)]
    for i in range(2 * k):
        abi.render_device(sc, cam, opts, bufs[i % k], None, streams[i % k].cuda_stream)
    torch.cuda.synchronize()
    n = 24
    t0 = time.perf_counter()
    for i in range(n):
        abi.render_device(sc, cam, opts, bufs[i % k], None, streams[i % k].cuda_stream)
    torch.cuda.synchronize()
    print("nranks %d frames in flight %d: %.3f ms/frame" % (nr, k, (time.perf_counter() - t0) / n * 1e3), flush=True)
