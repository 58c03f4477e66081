"""Row-tile sharding of one frame over the ranks of a node.

The frame is cut into tiles of `tile_rows` rows (a multiple of 8, the wave tile height); tile t belongs to rank
t % nranks.  Each rank renders its tiles into a compact buffer (its rows in ascending global order), which is what
rtmi_render_device writes when opts.nranks > 1.  There is no data-path collective inside the render: pixels are
independent (camera.rs:250-256 `par_iter` over pixels) and the counter RNG is keyed by the global pixel index, so
the image does not depend on the sharding.  The only exchange is the gather of finished rows to rank 0, the step the
reference performs when it writes the `Canvas`.

Pure torch + torch.distributed: runs over RCCL ("nccl") on the GPUs and over "gloo" in the CPU tests.
"""
import torch
import torch.distributed as dist


def norm_tile_rows(tile_rows):
    """Same normalisation as rtmi_rank_rows (rtmi.hip norm_tile_rows): a positive multiple of 8, default 8."""
    if tile_rows is None or tile_rows <= 0:
        return 8
    return (int(tile_rows) + 7) // 8 * 8


def rank_row_index(vsize, rank, nranks, tile_rows=8):
    """Global row numbers of the rows rank `rank` owns, in the order they sit in its compact buffer (int64 tensor)."""
    tr = norm_tile_rows(tile_rows)
    rows = []
    for t in range(rank, (vsize + tr - 1) // tr, nranks):
        rows.extend(range(t * tr, min(vsize, (t + 1) * tr)))
    return torch.tensor(rows, dtype=torch.int64)


def rank_rows(vsize, rank, nranks, tile_rows=8):
    """Row count of a rank's compact buffer; equals rtmi_rank_rows (checked in tests)."""
    return int(rank_row_index(vsize, rank, nranks, tile_rows).numel())


def max_rank_rows(vsize, nranks, tile_rows=8):
    return max(rank_rows(vsize, r, nranks, tile_rows) for r in range(nranks))


class RowGather:
    """Gathers the ranks' compact row buffers into the whole frame on `dst`.

    Buffers are padded to the largest rank's row count so that one fixed-size gather serves every rank; the row maps
    are built once and kept on the device of the buffers.  `gather(local)` returns the assembled [vsize, hsize, C]
    frame on dst (None elsewhere)."""

    def __init__(self, vsize, hsize, channels=3, tile_rows=8, dtype=torch.float32, device="cpu", group=None, dst=0):
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.nranks = dist.get_world_size(group) if dist.is_initialized() else 1
        self.dst = dst
        self.vsize, self.hsize, self.channels = vsize, hsize, channels
        self.tile_rows = norm_tile_rows(tile_rows)
        self.rows = rank_rows(vsize, self.rank, self.nranks, self.tile_rows)
        self.padded_rows = max_rank_rows(vsize, self.nranks, self.tile_rows)
        self.device = torch.device(device)
        self.dtype = dtype
        self.parts = None
        self.frame = None
        self.index = None
        if self.rank == dst and self.nranks > 1:
            self.index = [rank_row_index(vsize, r, self.nranks, self.tile_rows).to(self.device) for r in range(self.nranks)]
            self.frame = torch.zeros((vsize, hsize, channels), dtype=dtype, device=self.device)
            if self.nranks > 1:
                self.parts = [torch.zeros((self.padded_rows, hsize, channels), dtype=dtype, device=self.device) for _ in range(self.nranks)]

    def new_local(self):
        """A zeroed compact buffer of the padded size (rows beyond `self.rows` stay zero and are dropped on assembly)."""
        return torch.zeros((self.padded_rows, self.hsize, self.channels), dtype=self.dtype, device=self.device)

    def assemble(self, parts):
        for r, part in enumerate(parts):
            idx = self.index[r]
            self.frame.index_copy_(0, idx, part[: idx.numel()])
        return self.frame

    def gather(self, local):
        if tuple(local.shape) != (self.padded_rows, self.hsize, self.channels):
            raise ValueError("local buffer must be [%d, %d, %d], got %s" % (self.padded_rows, self.hsize, self.channels, tuple(local.shape)))
        if self.nranks == 1:
            return local  # one rank owns every row, already in order
        dist.gather(local, self.parts if self.rank == self.dst else None, dst=self.dst, group=self.group)
        if self.rank != self.dst:
            return None
        return self.assemble(self.parts)


# ------------------------------------------------------------------------------------------------
# photon map: ranks trace disjoint shares of the photons (rtmi_photon_opts.first / count) and exchange the arrays
# ------------------------------------------------------------------------------------------------
def photon_share(n_emit, rank, nranks):
    """(first, count) of rank's share of n_emit photons: contiguous blocks in rank order, so the shares one after the other
    are the photons of the whole call in the reference's order (world.rs:195-201)."""
    nranks = max(int(nranks), 1)
    first = n_emit * rank // nranks
    return first, n_emit * (rank + 1) // nranks - first


def all_gather_photons(local, group=None):
    """Every rank's stored photons (a uint8 tensor [n_r, 32] of rtmi_photon records, on the device the backend moves: the
    GPU under RCCL) -> the whole map [sum n_r, 32] on every rank, shares in rank order.  Two collectives: the counts, then
    the records padded to the largest share - every rank needs the whole map, because every rank renders with it."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    n = dist.get_world_size(group)
    count = torch.tensor([local.shape[0]], dtype=torch.int64, device=local.device)
    counts = [torch.zeros_like(count) for _ in range(n)]
    dist.all_gather(counts, count, group=group)
    counts = [int(c.item()) for c in counts]
    padded = torch.zeros((max(max(counts), 1), 32), dtype=torch.uint8, device=local.device)
    padded[: local.shape[0]] = local
    parts = [torch.zeros_like(padded) for _ in range(n)]
    dist.all_gather(parts, padded, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0)
