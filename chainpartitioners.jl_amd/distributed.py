"""Row-tiled K-part DP across the GPUs of one node: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI) completes every layer's cost vector with ONE all_gather of the per-rank row tiles; nothing else is
exchanged on the data path.

Reference semantics: all cst[j', k] of a layer depend only on layer k-1 (DynamicSplitter.jl:33-46), so the rows
j' = 1..n+1 are tiled contiguously over the ranks.  Each rank holds a replica of the CSR
arrays and link arrays; `ptr` stays sharded; unravel_splits (DynamicSplitter.jl:89-99) is K single-integer
MAX all_reduces that ask the owner of each row.

Per-layer volume: (n+1) * 8 bytes in total (config 5: 400 MB; each rank contributes 1/G and receives (G-1)/G of it).
"""
from __future__ import annotations

import numpy as np
import torch

from . import models as M


def tile_bounds(n, world):
    """Equal contiguous row tiles of j' = 1..n+1 (half-open, 1-based); the last tiles may be shorter or empty."""
    t = -(-(n + 1) // world)
    return [(min(1 + g * t, n + 2), min(1 + (g + 1) * t, n + 2)) for g in range(world)], t


class TiledDP:
    """One rank's share of the DP.  `step_layer(k)` computes this rank's tile of layer k into `cur`; the caller then
    completes `cur` (all_gather) and calls `swap()`."""

    def __init__(self, hip, handle, n, K, method, rank, world, device):
        mdl = method.f
        self.hip, self.handle, self.n, self.K, self.rank, self.world = hip, handle, n, K, rank, world
        self.mm = mdl.marshal(w_table=n + 1)
        self.tiles, self.tile = tile_bounds(n, world)
        lo, hi = self.tiles[rank]
        self.lo, self.hi = lo, hi
        self.dp = hip.dp_begin(handle, K, method.combine, method.order, self.mm, lo, hi)
        dt = torch.int64 if mdl.dtype == M.CP_I64 else torch.float64
        padded = self.tile * world                      # equal slices for all_gather_into_tensor
        self.prev = torch.zeros(padded, dtype=dt, device=device)
        self.cur = torch.zeros(padded, dtype=dt, device=device)

    def step_layer(self, k):
        self.hip.dp_layer(self.dp, k, self.prev.data_ptr(), self.cur.data_ptr())

    def my_slice(self, buf):
        a = (self.lo - 1)
        return buf[a:a + self.tile]

    def swap(self):
        self.prev, self.cur = self.cur, self.prev

    def ptr_at(self, k, jp):
        return self.hip.dp_ptr_at(self.dp, k, jp)

    def close(self):
        self.hip.dp_destroy(self.dp)


def partition_stripe_tiled(hip, handle, n, K, method, *, device, group=None):
    """partition_stripe(A, K, Dynamic{Total,Bottleneck}{Splitter,Chunker}(f)) with the DP rows tiled over the ranks of
    `group` (default: the world).  Every rank returns the same (K+1) split vector (1-based numpy int64).

    Ordering: the handle's kernels are enqueued on torch's CURRENT stream (cp_set_stream), the stream the collective is
    ordered against, so layer k+1 cannot read `prev` before the gather of layer k has landed -- whatever stream context the
    caller runs under.  The gather is in place: every rank's tile of `cur` is the send buffer (no clone, no device-wide sync)."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    host_staged = dist.get_backend(group) == "gloo"      # CPU rehearsal of the same exchange (tests); RCCL works on HBM
    stream = torch.cuda.current_stream(device)
    hip.set_stream(handle, stream.cuda_stream)
    T = TiledDP(hip, handle, n, K, method, rank, world, device)
    try:
        T.step_layer(1)                                 # whole layer on every rank, no exchange
        T.swap()
        for k in range(2, K + 1):
            T.step_layer(k)                             # (returns with the tile written: cp_dp_layer waits for its stream)
            mine = T.my_slice(T.cur)
            if host_staged:
                parts = [torch.empty(T.tile, dtype=mine.dtype) for _ in range(world)]
                dist.all_gather(parts, mine.cpu(), group=group)
                T.cur.copy_(torch.cat(parts).to(device))
            else:
                dist.all_gather_into_tensor(T.cur, mine, group=group)  # RCCL, in place: every rank's tile of cst[:, k]
            T.swap()
        spl = np.zeros(K + 1, dtype=np.int64)
        spl[K] = n + 1
        v = torch.zeros(1, dtype=torch.int64, device="cpu" if host_staged else device)
        for k in range(K, 0, -1):
            v[0] = T.ptr_at(k, int(spl[k]))             # 0 unless this rank owns row spl[k+1]
            dist.all_reduce(v, op=dist.ReduceOp.MAX, group=group)
            spl[k - 1] = int(v.item())
        return spl
    finally:
        T.close()
