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
    sc.close()


def test_any_number_of_streams_over_time_and_four_at_once():
    """A work-queue slot belongs to a stream only while that stream's work is in flight: a process may render on any number
    of streams one after the other (a host that makes a stream per frame), and on up to four at the same time; a fifth
    concurrent one is refused loudly, never raced, and is welcome again once the others have finished."""
    import torch
    world, cam = scenes.chapter11a(256, 128)
    sc = abi.Scene(world)
    ref, _ = abi.render_host(sc, cam, abi.make_opts(seed=2))
    for _ in range(12):  # 12 short-lived streams, one after the other
        s = torch.cuda.Stream()
        out = torch.zeros((128, 256, 3), dtype=torch.float32, device="cuda")
        abi.render_device(sc, cam, abi.make_opts(seed=2), out, None, s.cuda_stream)
        s.synchronize()
        assert np.array_equal(out.cpu().numpy(), ref)
        del s
    # many at once: a frame that keeps the GPU busy for milliseconds, seven streams launched back to back
    _, big = scenes.chapter11a(1280, 720)
    streams = [torch.cuda.Stream() for _ in range(7)]
    outs = [torch.zeros((720, 1280, 3), dtype=torch.float32, device="cuda") for _ in streams]
    accepted = []
    for s, o in zip(streams, outs):
        try:
            abi.render_device(sc, big, abi.make_opts(seed=3), o, None, s.cuda_stream)
            accepted.append(o)
        except abi.RtmiError as e:
            assert "streams are rendering" in str(e)
    torch.cuda.synchronize()
    assert 4 <= len(accepted) <= 7  # (more than four only if early ones had finished by then)
    for o in accepted[1:]:
        assert torch.equal(o, accepted[0])
    # ... and afterwards every stream is welcome again
    for s in streams:
        out = torch.zeros((128, 256, 3), dtype=torch.float32, device="cuda")
        abi.render_device(sc, cam, abi.make_opts(seed=2), out, None, s.cuda_stream)
        s.synchronize()
        assert np.array_equal(out.cpu().numpy(), ref)
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
