"""rtmi_render_device_noalloc (include/rtmi.h): the entry point that allocates nothing, through the C ABI, against
rtmi_render_device on the same inputs - bit for bit, image and counters - on two calls in a row, the second with a
smaller input (the static work memory keeps the first call's contents).  Collected last (test_zz_*): the round's end
runs pytest -x."""
import numpy as np
import pytest

from raytracing_amd import abi, scenes

pytestmark = pytest.mark.gpu

KEYS = ("primary", "reflection", "refraction", "shadow", "samples", "leaf_tests", "node_visits", "overflow", "flops", "aabb_tests")


def _both(world, cam, **opts):
    import torch
    sc = abi.Scene(world)
    h = cam.hsize()
    o = abi.make_opts(**opts)
    rows = abi.rank_rows(cam.vsize(), o.rank, max(o.nranks, 1), o.tile_rows)
    res = []
    for fn in (abi.render_device, abi.render_device_noalloc):
        out = torch.full((rows, h, 3), -1.0, dtype=torch.float32, device="cuda")
        st = torch.zeros(abi.N_STATS, dtype=torch.int64, device="cuda")
        torch.cuda.synchronize()
        free_before = torch.cuda.mem_get_info()[0]
        fn(sc, cam, o, out, st, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        if fn is abi.render_device_noalloc:
            assert torch.cuda.mem_get_info()[0] == free_before  # nothing was allocated (or freed) on the device
        res.append((out.cpu().numpy(), dict(zip(abi.STAT_NAMES, st.cpu().tolist()))))
    sc.close()
    return res


RAY_KEYS = ("primary", "reflection", "refraction", "shadow", "samples", "overflow")


def _same(res, keys=KEYS):
    (a, sa), (b, sb) = res
    assert np.isfinite(b).all() and (b >= 0).all()
    assert np.array_equal(a, b)
    for k in keys:
        assert sa[k] == sb[k], k
    assert sb["overflow"] == 0
    return sb


def test_two_calls_in_a_row_the_second_smaller():
    # 1: chapter-11a as shipped at 640x360 (230,400 pixels, nearly all of them queued: small frames take only the first
    #    samples on the first-pass lanes), heavy pixels promoted to units
    st = _same(_both(*scenes.chapter11a(640, 360), seed=5))
    assert st["samples"] > 12 * 640 * 360
    # 2: a smaller frame of another scene (tree traversal, area of the array used before is now stale)
    # (a triangle mesh: a wave walks its BVH together, so the work counters depend on how the queues - of another size
    # here - group the samples; image and rays do not)
    st = _same(_both(*scenes.chapter15(200, 120, lite=True), seed=6), RAY_KEYS)
    assert st["samples"] > 12 * 200 * 120
    # 3: f64 validation mode, and a row-tile share
    _same(_both(*scenes.soft_shadow(256, 128), seed=7, precision=1))
    _same(_both(*scenes.chapter11a(640, 360), seed=5, rank=1, nranks=3, tile_rows=16))
    # 4: one sample per pixel: no work memory is touched at all
    _same(_both(*scenes.chapter11a(640, 360, deterministic=True)))


def test_calls_beyond_the_limit_are_refused():
    import torch
    world, cam = scenes.chapter11a(1280, 720)  # 921,600 pixels > 524,288
    sc = abi.Scene(world)
    out = torch.zeros((720, 1280, 3), dtype=torch.float32, device="cuda")
    with pytest.raises(abi.RtmiError, match="static work memory"):
        abi.render_device_noalloc(sc, cam, abi.make_opts(), out, None, torch.cuda.current_stream().cuda_stream)
    # ... and accepted in two row-tile shares, equal to the pooled entry point's
    for r in range(2):
        o = abi.make_opts(rank=r, nranks=2)
        rows = abi.rank_rows(720, r, 2, 8)
        a = torch.zeros((rows, 1280, 3), dtype=torch.float32, device="cuda")
        b = torch.zeros_like(a)
        abi.render_device(sc, cam, o, a, None, torch.cuda.current_stream().cuda_stream)
        abi.render_device_noalloc(sc, cam, o, b, None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(a, b)
    sc.close()
