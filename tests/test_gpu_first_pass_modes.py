"""The two ways of the first pass (render_kernel: a pixel's first samples one after the other on its lane, or spread over
lanes of their own; rtmi.hip chooses by the size of the rank's share of the frame) add the same samples up in the same
order.  Both are forced here through RTMI_SPREAD_BELOW and must agree bit for bit, in f64 (no contraction) and in f32
(contraction inside one expression only, so that every instantiation of the sampling code rounds alike): otherwise a
frame would depend on how it is sharded.  Before -ffp-contract=on the CSG dice differed in the last bits in f32."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

CHILD = r"""
import sys, json, hashlib
sys.path.insert(0, %r)
import torch
from raytracing_amd import abi, scenes
out = {}
for name, build, h, v in (("ch11a", scenes.chapter11a, 320, 180), ("dice", scenes.chapter16, 240, 140), ("teapot", scenes.chapter15, 240, 140),
                          ("soft", scenes.soft_shadow, 160, 90), ("dof", scenes.depth_of_field, 160, 90), ("gallery", scenes.lights_gallery, 200, 150)):
    world, cam = build(h, v)
    sc = abi.Scene(world)
    for prec in (0, 1):
        buf = torch.zeros((v, h, 3), dtype=torch.float32, device="cuda")
        st = torch.zeros(abi.N_STATS, dtype=torch.int64, device="cuda")
        abi.render_device(sc, cam, abi.make_opts(seed=3, precision=prec), buf, st, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        img = buf.cpu().numpy()
        out["%%s_%%s" %% (name, "f64" if prec else "f32")] = {"sha": hashlib.sha256(img.tobytes()).hexdigest(), "samples": st.cpu().tolist()[4],
                                                           "mean": float(img.mean())}
    sc.close()
print(json.dumps(out))
""" % ROOT


def _render(spread_below):
    env = dict(os.environ, RTMI_SPREAD_BELOW=str(spread_below))
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_both_first_pass_modes_agree():
    lanes, spread = _render(0), _render(1000000)
    for k in lanes:
        assert lanes[k]["sha"] == spread[k]["sha"], k
        assert lanes[k]["samples"] == spread[k]["samples"], k
