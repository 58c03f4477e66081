"""Hashes of small adaptive frames of every scene kind in both precisions; run under RTMI_SPREAD_BELOW=0 and =1000000 to compare the two first-pass modes."""
import sys, os, hashlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from raytracing_amd import abi, scenes
for name, build, h, v in (("ch11a", scenes.chapter11a, 320, 180), ("dice", scenes.chapter16, 320, 180), ("teapot", scenes.chapter15, 320, 180),
                          ("soft", scenes.soft_shadow, 160, 90), ("dof", scenes.depth_of_field, 160, 90), ("gallery", scenes.lights_gallery, 200, 150)):
    world, cam = build(h, v)
    sc = abi.Scene(world)
    for prec in (0, 1):
        buf = torch.zeros((v, h, 3), dtype=torch.float32, device="cuda")
        st = torch.zeros(abi.N_STATS, dtype=torch.int64, device="cuda")
        abi.render_device(sc, cam, abi.make_opts(seed=3, precision=prec), buf, st, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        print(name, "f64" if prec else "f32", hashlib.sha256(buf.cpu().numpy().tobytes()).hexdigest()[:16], st.cpu().tolist()[:5], flush=True)
    sc.close()
