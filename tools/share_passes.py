"""tools/share_passes.py : first-pass / second-pass time of rank 0's share of the headline frame at N = 1, 2, 4, 8 (rtmi_render_device_timed)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import abi, scenes
world, cam = scenes.chapter11a(3840, 2160)
sc = abi.Scene(world)
stream = torch.cuda.current_stream().cuda_stream
for nr in (1, 2, 4, 8):
    rows = abi.rank_rows(2160, 0, nr, 8)
    buf = torch.zeros((rows, 3840, 3), dtype=torch.float32, device="cuda")
    st = torch.zeros(abi.N_STATS, dtype=torch.int64, device="cuda")
    opts = abi.make_opts(rank=0, nranks=nr, tile_rows=8)
    a = b = 0.0
    for i in range(4):
        st.zero_()
        m1, m2 = abi.render_device_timed(sc, cam, opts, buf, st, stream)
        if i: a += m1 / 3; b += m2 / 3
    s = dict(zip(abi.STAT_NAMES, st.cpu().tolist()))
    print("nranks %d: first pass %.2f ms, second pass %.2f ms, samples %d, discarded %d" % (nr, a, b, s["samples"], s["discarded"]), flush=True)
