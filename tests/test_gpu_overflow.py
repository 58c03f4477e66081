"""The adaptive second pass with a work-queue slot far too small for the frame (RTMI_WORK_MB).  Queues, unit
lists and the heavy table may overflow as they like: the image and the ray counters stay what they are with room to
spare, bit for bit - a pixel's sums are the reference's sequential ones (camera.rs:285-286) wherever its samples are
taken, also on the lanes that find no queue slot and finish their pixels alone.

The same comparison guards something a build could lose silently: a sample's colour must not depend on which kernel
computes it (resample_kernel in rounds, units_kernel in units), although the compiler is free to fuse multiply-adds
differently in every instantiation of the sampling code - the small slots move most samples from one to the other."""
import hashlib
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
for name, build, h, v in (("ch11a", scenes.chapter11a, 320, 180), ("ch11a_f64", scenes.chapter11a, 320, 180), ("soft", scenes.soft_shadow, 160, 90), ("dice", scenes.chapter16, 240, 140),
                          ("teapot", scenes.chapter15, 240, 140), ("ch11a_big", scenes.chapter11a, 1280, 720)):
    world, cam = build(h, v)
    sc = abi.Scene(world)
    buf = torch.zeros((v, h, 3), dtype=torch.float32, device="cuda")
    st = torch.zeros(abi.N_STATS, dtype=torch.int64, device="cuda")
    abi.render_device(sc, cam, abi.make_opts(seed=7, precision=1 if name.endswith("_f64") else 0), buf, st, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    s = st.cpu().tolist()
    img = buf.cpu().numpy()
    out[name] = {"sha": hashlib.sha256(img.tobytes()).hexdigest(), "stats": s[:abi.STAT_NAMES.index("discarded")],
                 "mean": float(img.mean()), "coarse": img.reshape(v // 10, 10, h // 10, 10, 3).mean(axis=(1, 3)).round(4).tolist()}
    sc.close()
print(json.dumps(out))
""" % ROOT


def _render(work_mb):
    env = dict(os.environ)
    env.pop("RTMI_WORK_MB", None)
    if work_mb:
        env["RTMI_WORK_MB"] = str(work_mb)
    r = subprocess.run([sys.executable, "-c", CHILD], capture_output=True, text=True, timeout=280, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


RAYS = slice(0, 5)  # primary, reflection, refraction, shadow, samples


def _same(a, b, scene):
    assert a[scene]["sha"] == b[scene]["sha"], scene
    if scene in ("teapot", "dice"):  # a wave walks the scene tree together: the work counters depend on the schedule, the rays do not
        assert a[scene]["stats"][RAYS] == b[scene]["stats"][RAYS], scene
    else:
        assert a[scene]["stats"] == b[scene]["stats"], scene


def test_small_work_slots_change_nothing():
    full = _render(0)
    # 2 MiB: 10922 pixels per queue, 360 units per pass.  The small frames' queues fit; their unit lists do not.  The
    # 1280x720 frame queues many times that: the rest are finished by their first-pass lanes, sample by sample - the
    # same sequential sums (camera.rs:285-286), so the frame is the same bit for bit there too.
    small = _render(2)
    for scene in full:
        _same(small, full, scene)
    # 16 MiB: every queue fits again, the unit lists (2900 units) overflow on the big frame
    mid = _render(16)
    for scene in full:
        _same(mid, full, scene)
