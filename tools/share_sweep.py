"""tools/share_sweep.py [nranks ...] : rank 0's share of the headline frame alone on the GPU under different settings of the
first pass's scheduling knobs (rtmi.hip launch_render reads them from the environment at every call): the whole call between
events (mean of 5, one after the other), its two passes, the samples it discards."""
import os, sys, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from raytracing_amd import abi, scenes
world, cam = scenes.chapter11a(3840, 2160)
sc = abi.Scene(world)
stream = torch.cuda.current_stream().cuda_stream
KNOBS = ("RTMI_SPREAD_BELOW", "RTMI_INLANE_MIN_VISITS", "RTMI_INLANE_MAX", "RTMI_INLANE_NEED", "RTMI_KEEP_LANES", "RTMI_GRID_PER_CU", "RTMI_VISITS_PER_WAVE")
SETTINGS = [{}]
for vpw, mx in itertools.product((1, 2, 3), (32, 64, 128, 256, 4096)):
    SETTINGS.append({"RTMI_SPREAD_BELOW": 0, "RTMI_INLANE_MIN_VISITS": 1, "RTMI_INLANE_MAX": mx, "RTMI_INLANE_NEED": mx, "RTMI_VISITS_PER_WAVE": vpw})
for nr in [int(x) for x in sys.argv[1:]] or [2, 4, 8]:
    rows = abi.rank_rows(2160, 0, nr, 8)
    buf = torch.zeros((rows, 3840, 3), dtype=torch.float32, device="cuda")
    st = torch.zeros(abi.N_STATS, dtype=torch.int64, device="cuda")
    opts = abi.make_opts(rank=0, nranks=nr, tile_rows=8)
    for s in SETTINGS:
        for k in KNOBS:
            os.environ.pop(k, None)
        for k, v in s.items():
            os.environ[k] = str(v)
        for _ in range(2):
            abi.render_device(sc, cam, opts, buf, None, stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            abi.render_device(sc, cam, opts, buf, None, stream)
        e1.record(); torch.cuda.synchronize()
        st.zero_()
        m1, m2 = abi.render_device_timed(sc, cam, opts, buf, st, stream)
        d = dict(zip(abi.STAT_NAMES, st.cpu().tolist()))
        print("nranks %d  call %.3f ms  (first %.2f second %.2f)  samples %d discarded %d  %s" % (
            nr, e0.elapsed_time(e1) / 5, m1, m2, d["samples"], d["discarded"], " ".join("%s=%s" % (k[5:], v) for k, v in s.items()) or "(defaults)"), flush=True)
