"""Multi-rank path on CPU (gloo, world_size 2 and 3): row-tile ownership, compact buffers, padded gather and assembly.

No GPU here, so each rank fills its compact buffer from an oracle render of the same small frame (the oracle is only the
stand-in for the device kernel's output; tests/test_gpu_parity.py checks that rtmi_render_device writes exactly this
compact layout).  The gathered frame must equal the unsharded one bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import oracle_lib as orc
from raytracing_amd import abi, scenes, tiles

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world_size, port, hsize, vsize, tile_rows, q):
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        world, cam = scenes.chapter11a(hsize, vsize, deterministic=True)
        full, _ = orc.render(world, cam, abi.make_opts(), n_threads=2)
        full = torch.from_numpy(full.astype(np.float32))
        g = tiles.RowGather(vsize, hsize, tile_rows=tile_rows)
        assert g.rows == abi.rank_rows(vsize, rank, world_size, tile_rows)  # the C ABI's partition (rtmi_rank_rows)
        local = g.new_local()
        idx = tiles.rank_row_index(vsize, rank, world_size, tile_rows)
        local[: idx.numel()] = full[idx]
        frame = g.gather(local)
        if rank == 0:
            q.put(bool(torch.equal(frame, full)))
            frame2 = g.gather(local)  # buffers are reused: a second frame through the same object
            q.put(bool(torch.equal(frame2, full)))
        else:
            assert frame is None
            g.gather(local)
        # ray statistics are summed over ranks like bench.py does
        t = torch.tensor([float(g.rows)], dtype=torch.float64)
        dist.all_reduce(t)
        assert int(t.item()) == vsize
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size,vsize,tile_rows", [(2, 40, 8), (2, 37, 8), (3, 50, 16), (2, 5, 8)])
def test_gather_rows_equals_unsharded_frame(world_size, vsize, tile_rows):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world_size, port, 24, vsize, tile_rows, q)) for r in range(world_size)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert q.get(timeout=5) is True
    assert q.get(timeout=5) is True


def test_row_maps_partition_the_frame():
    for v in (1, 7, 8, 9, 64, 100, 2160):
        for n in (1, 2, 3, 4, 8):
            for tr in (8, 16, 64, 0, 5):
                seen = torch.cat([tiles.rank_row_index(v, r, n, tr) for r in range(n)])
                assert sorted(seen.tolist()) == list(range(v))
                for r in range(n):
                    assert tiles.rank_rows(v, r, n, tr) == abi.rank_rows(v, r, n, tr)


def test_single_rank_gather_is_identity():
    g = tiles.RowGather(20, 6)
    local = torch.arange(20 * 6 * 3, dtype=torch.float32).reshape(20, 6, 3)
    assert torch.equal(g.gather(local), local)
    with pytest.raises(ValueError):
        g.gather(local[:10])


def _photon_worker(rank, world_size, port, n_emit, q):
    """The photon map's N > 1 path: every rank traces its share of the photons (here the oracle stands in for
    rtmi_trace_photons; tests/test_gpu_photon_trace.py checks that the device call honours first / count the same way) and
    all ranks end up with the whole map, in the order of the one-call map."""
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    try:
        import photon_trace_cases as cases
        flat = cases.box().flatten()
        first, count = tiles.photon_share(n_emit, rank, world_size)
        mine, _ = orc.trace_photons(flat, abi.make_photon_opts(n_emit, seed=3, direct=True, first=first, count=count))
        local = torch.from_numpy(np.ascontiguousarray(mine).view(np.uint8).reshape(-1, 32).copy())
        whole = tiles.all_gather_photons(local)
        ref, _ = orc.trace_photons(flat, abi.make_photon_opts(n_emit, seed=3, direct=True))
        q.put(bool(np.array_equal(whole.numpy().reshape(-1).view(orc.PHOTON_DTYPE), ref)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world_size,n_emit", [(2, 5000), (3, 1001)])
def test_photon_shares_gather_to_the_whole_map(world_size, n_emit):
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_photon_worker, args=(r, world_size, port, n_emit, q)) for r in range(world_size)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    for _ in range(world_size):
        assert q.get(timeout=5) is True


def test_photon_shares_partition_the_photons():
    for n in (1, 2, 7, 1000, 2_000_001):
        for k in (1, 2, 3, 8):
            shares = [tiles.photon_share(n, r, k) for r in range(k)]
            assert shares[0][0] == 0 and sum(c for _, c in shares) == n
            assert all(shares[i][0] + shares[i][1] == shares[i + 1][0] for i in range(k - 1))
