"""Host-side contracts of the device entry points on a real GPU: concurrent renders of one scene on different streams (each
stream owns a work-queue slot of the per-device pool), the stream limit, the C++ example end to end, and the PCIe-inclusive
host entry point agreeing with the device one."""
import os
import subprocess

import numpy as np
import pytest

from raytracing_amd import abi, scenes

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_concurrent_streams_give_the_sequential_images():
    import torch
    world, cam_a = scenes.chapter11a(320, 160)
    _, cam_b = scenes.chapter11a(256, 128)
    sc = abi.Scene(world)
    ref_a, _ = abi.render_host(sc, cam_a, abi.make_opts(seed=1))
    ref_b, _ = abi.render_host(sc, cam_b, abi.make_opts(seed=2))
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    out_a = torch.zeros((160, 320, 3), dtype=torch.float32, device="cuda")
    out_b = torch.zeros((128, 256, 3), dtype=torch.float32, device="cuda")
    for _ in range(3):  # interleaved launches: both adaptive, both using their stream's queue slot
        abi.render_device(sc, cam_a, abi.make_opts(seed=1), out_a, None, s1.cuda_stream)
        abi.render_device(sc, cam_b, abi.make_opts(seed=2), out_b, None, s2.cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(out_a.cpu().numpy(), ref_a) and np.array_equal(out_b.cpu().numpy(), ref_b)
    # a fifth distinct stream (the null stream was used by render_host) is refused loudly, not raced
    extra = [torch.cuda.Stream() for _ in range(3)]
    with pytest.raises(abi.RtmiError, match="streams"):
        for s in extra:
            abi.render_device(sc, cam_b, abi.make_opts(seed=2), out_b, None, s.cuda_stream)
    torch.cuda.synchronize()
    sc.close()


def test_cpp_example_renders_a_ppm(tmp_path):
    exe = str(tmp_path / "render_ppm")
    libdir = os.path.join(ROOT, "raytracing_amd")
    subprocess.run(["g++", "-std=c++17", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "render_ppm.cpp"),
                    "-L" + libdir, "-lrtmi", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    out = tmp_path / "o.ppm"
    r = subprocess.run([exe, str(out), "128"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    head = out.read_text().split("\n", 3)
    assert head[0] == "P3" and head[1] == "128 64" and head[2] == "255"
    vals = np.array(out.read_text().split()[4:], dtype=int)
    assert vals.size == 128 * 64 * 3 and vals.min() >= 0 and vals.max() <= 255 and vals.std() > 10
