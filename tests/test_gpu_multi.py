"""rtmi_render_multi (include/rtmi.h): Camera::render over several scene replicas of one process - one per GPU on a
multi-GPU host; here all on the one GPU of the test box, which exercises the sharding, the gather of the compact row
buffers and the kernel that puts the rows in place (the peer copy of shares rendered on another device cannot run here).
The gathered frame is the single-device frame bit for bit, and the counters add up."""
import numpy as np
import pytest

from raytracing_amd import abi, scenes

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,tile_rows", [(1, 8), (2, 8), (3, 16), (8, 8)])
def test_shares_gathered_equal_the_whole_frame(n, tile_rows):
    import torch
    world, cam = scenes.chapter11a(328, 150)  # (150 rows: the last tile is partial)
    replicas = [abi.Scene(world) for _ in range(n)]
    h, v = cam.hsize(), cam.vsize()
    whole = torch.zeros((v, h, 3), dtype=torch.float32, device="cuda")
    wst = torch.zeros(abi.N_STATS, dtype=torch.int64, device="cuda")
    abi.render_device(replicas[0], cam, abi.make_opts(seed=4), whole, wst, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out = torch.full((v, h, 3), -1.0, dtype=torch.float32, device="cuda")
    st = abi.render_multi(replicas, cam, abi.make_opts(seed=4), out, tile_rows=tile_rows)
    torch.cuda.synchronize()
    assert torch.equal(out, whole)
    ref = dict(zip(abi.STAT_NAMES, wst.cpu().tolist()))
    for k in ("primary", "reflection", "refraction", "shadow", "samples", "flops"):
        assert st[k] == ref[k], k
    assert st["kernel_ms"] > 0
    for s in replicas:
        s.close()


def test_shares_on_distinct_devices_equal_the_whole_frame():
    """The peer-copy branch of rtmi_render_multi (hipMemcpyPeerAsync of a share rendered on another device, ordered by that
    share's event): needs two GPUs in one process - skipped on the one-GPU test box, where no run has exercised it yet."""
    import torch
    n = min(torch.cuda.device_count(), 4)
    if n < 2:
        pytest.skip("one GPU visible: the cross-device gather cannot run here")
    world, cam = scenes.chapter11a(328, 150)
    replicas = []
    for d in range(n):
        with torch.cuda.device(d):
            replicas.append(abi.Scene(world))
    h, v = cam.hsize(), cam.vsize()
    with torch.cuda.device(0):
        whole = torch.zeros((v, h, 3), dtype=torch.float32, device="cuda:0")
        abi.render_device(replicas[0], cam, abi.make_opts(seed=4), whole, None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        out = torch.full((v, h, 3), -1.0, dtype=torch.float32, device="cuda:0")
        for _ in range(2):  # (the second call runs on the kept streams, events and staging buffers)
            out.fill_(-1.0)
            st = abi.render_multi(replicas, cam, abi.make_opts(seed=4), out, tile_rows=8)
            torch.cuda.synchronize()
            assert torch.equal(out, whole)
            assert st["kernel_ms"] > 0
    for d, s in enumerate(replicas):
        with torch.cuda.device(d):
            s.close()
