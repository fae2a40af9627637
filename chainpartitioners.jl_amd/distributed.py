"""Row-tiled K-part DP across the GPUs of one node: one process per GPU, `torch.distributed` (backend "nccl" = RCCL
over xGMI) completes every layer's cost vector with ONE all_gather of the per-rank row tiles; nothing else is
exchanged on the data path.

Reference semantics: all cst[j', k] of a layer depend only on layer k-1 (DynamicSplitter.jl:33-46), so the rows
j' = 1..n+1 are tiled contiguously over the ranks.  Each rank holds a replica of the CSR
arrays and link arrays; `ptr` stays sharded; unravel_splits (DynamicSplitter.jl:89-99) is K single-integer
MAX all_reduces that ask the owner of each row.

Methods: Dynamic{Total,Bottleneck}{Splitter,Chunker}(f) -- one static tile per rank -- and
DynamicTotal{Splitter,Chunker}(ConstrainedCost(f, VertexCount(), w_max)) (DynamicSplitter.jl:206-314): every layer k lives
on its own window j'_lo[k] .. j'_hi[k] (column_constraints, :144-172), so each layer's window is tiled over the ranks
afresh (cp_dp_set_rows) and the gathered row is masked outside the window before it feeds the next layer.

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


def width_windows(n, K, w):
    """column_constraints (DynamicSplitter.jl:144-172) for the width weight: j'_lo[k], j'_hi[k], 1-based, k = 1..K"""
    lo, hi = [0] * (K + 1), [0] * (K + 1)
    jp = n + 1
    for k in range(K, 0, -1):
        lo[k] = jp
        jp = max(1, jp - w)
    j = 1
    for k in range(1, K + 1):
        hi[k] = min(n + 1, j + w)
        j = hi[k]
    return lo, hi


def width_of_weight(weight, w_max, n):
    """The number of columns a WIDTH weight allows: VertexCount() -> w_max; AffineWorkModel(alpha, c, 0), c > 0 -> the largest nv with
    alpha + nv * c <= w_max in the weight's own arithmetic (the C library's width_of_weight, csrc/capi.hip; the reference's tests
    constrain with AffineWorkModel(0, 1, 0), test/test_Partitioners.jl:178-183).  None: not a function of the width."""
    if isinstance(weight, M.VertexCount):
        return int(w_max)
    if not isinstance(weight, M.AffineWorkModel) or getattr(weight, "alpha_k", None) is not None:
        return None
    if weight.beta_pin != 0 or not weight.beta_vertex > 0:
        return None
    if weight.dtype == M.CP_I64:
        fits = lambda nv: int(weight.alpha) + nv * int(weight.beta_vertex) <= int(w_max)
    else:
        fits = lambda nv: (np.float64(weight.alpha) + np.float64(nv) * np.float64(weight.beta_vertex)) + np.float64(0) * np.float64(weight.beta_pin) <= np.float64(w_max)
    if not fits(0):
        return -1
    lo, hi = 0, n + 1
    if fits(hi):
        return hi
    while hi - lo > 1:
        mid = (lo + hi) // 2
        if fits(mid):
            lo = mid
        else:
            hi = mid
    return lo


class TiledDP:
    """One rank's share of the DP.  Per layer k: `begin_layer(k)`, `step_layer(k)` (computes this rank's tile of layer k into
    `cur`), then the caller completes `cur` (all_gather of `slice_of(k)` from every rank), `complete_layer(k)`, `swap()`."""

    def __init__(self, hip, handle, n, K, method, rank, world, device):
        mdl, weight, w_max = M.split_constraint(method.f)
        self.hip, self.handle, self.n, self.K, self.rank, self.world = hip, handle, n, K, rank, world
        self.mm = mdl.marshal(w_table=n + 1)
        self.windowed = weight is not None
        wv = width_of_weight(weight, w_max, n) if self.windowed else None
        if self.windowed and (wv is None or wv < 1):
            raise NotImplementedError("the tiled constrained DP takes width weights only: VertexCount() or AffineWorkModel(alpha, c, 0) with room for a column")
        self.dtype = torch.int64 if mdl.dtype == M.CP_I64 else torch.float64
        self.big = (1 << 61) if mdl.dtype == M.CP_I64 else float(1 << 60)
        self.tiles, self.tile = tile_bounds(n, world)
        self.feasible = True
        if self.windowed:
            self.w = int(wv)
            self.lo, self.hi = width_windows(n, K, self.w)
            self.feasible = self.hi[K] >= n + 1
            self.tile = max(1, max(-(-(self.hi[k] - self.lo[k] + 1) // world) for k in range(1, K + 1)))
            lo, hi = 1, n + 2
        else:
            lo, hi = self.tiles[rank]
        self.dp = hip.dp_begin(handle, K, method.combine, method.order, self.mm, lo, hi)
        if self.windowed:
            hip.dp_set_window(self.dp, max(1, min(self.w, max(n, 1))))
        padded = max(self.tile * world, n + 1)           # equal slices for all_gather_into_tensor
        self.prev = torch.zeros(padded, dtype=self.dtype, device=device)
        self.cur = torch.zeros(padded, dtype=self.dtype, device=device)
        self.stage = torch.zeros(self.tile * world, dtype=self.dtype, device=device) if self.windowed else None

    # ---- geometry of layer k: (first row, rows per rank) of the gathered range, 0-based offsets into the layer row
    def layer_range(self, k):
        if not self.windowed or k == 1:
            return 0, self.tile
        L = self.hi[k] - self.lo[k] + 1
        return self.lo[k] - 1, max(1, -(-L // self.world))

    def begin_layer(self, k):
        if self.windowed and k >= 2:
            off, t = self.layer_range(k)
            a = min(off + self.rank * t, self.hi[k])              # 0-based first row of my slice (may be empty)
            b = min(off + (self.rank + 1) * t, self.hi[k])
            if b > a:
                self.hip.dp_set_rows(self.dp, a + 1, b + 1)
            self._mine = (a, b)

    def step_layer(self, k):
        if self.windowed and k >= 2 and self._mine[1] <= self._mine[0]:
            return                                                # an empty slice of this layer's window
        self.hip.dp_layer(self.dp, k, self.prev.data_ptr(), self.cur.data_ptr())

    def slice_of(self, k, buf, rank=None):
        """this rank's contribution to the gathered layer row: a view of `buf` (length = rows per rank)"""
        off, t = self.layer_range(k)
        g = self.rank if rank is None else rank
        return buf[off + g * t: off + (g + 1) * t]

    def complete_layer(self, k):
        """`cur` holds every rank's tile: what the next layer reads outside this layer's window is a value no total reaches"""
        if self.windowed:
            a, b = self.lo[k] - 1, self.hi[k]
            self.cur[:a] = self.big
            self.cur[b:] = self.big

    def swap(self):
        self.prev, self.cur = self.cur, self.prev

    def ptr_at(self, k, jp):
        return self.hip.dp_ptr_at(self.dp, k, jp)

    def close(self):
        self.hip.dp_destroy(self.dp)


def partition_stripe_tiled(hip, handle, n, K, method, *, device, group=None):
    """partition_stripe(A, K, method) with the DP rows tiled over the ranks of `group` (default: the world).  Every rank
    returns the same (K+1) split vector (1-based numpy int64).

    Ordering: the handle's kernels are enqueued on torch's CURRENT stream (cp_set_stream), the stream the collective is
    ordered against, so layer k+1 cannot read `prev` before the gather of layer k has landed -- whatever stream context the
    caller runs under.  The gather is in place: every rank's tile of `cur` is the send buffer (no clone, no device-wide sync).
    The binding lasts for this call only: on return the handle is back on a stream of its own (cp_reset_stream), so it never
    outlives a torch stream it borrowed."""
    import torch.distributed as dist
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    host_staged = dist.get_backend(group) == "gloo"      # CPU rehearsal of the same exchange (tests); RCCL works on HBM
    borrowed = torch.device(device).type == "cuda"
    if borrowed:
        hip.set_stream(handle, torch.cuda.current_stream(device).cuda_stream)
    T = None
    try:
        T = TiledDP(hip, handle, n, K, method, rank, world, device)
        if not T.feasible:                              # DynamicSplitter.jl:217-222: a degenerate partition, no exception
            spl = np.ones(K + 1, dtype=np.int64)
            spl[K] = n + 1
            return spl
        T.step_layer(1)                                 # whole layer on every rank, no exchange
        T.complete_layer(1)
        T.swap()
        for k in range(2, K + 1):
            T.begin_layer(k)
            T.step_layer(k)                             # (returns with the tile written: cp_dp_layer waits for its stream)
            off, t = T.layer_range(k)
            mine = T.slice_of(k, T.cur)
            if mine.numel() < t:                        # (the padded tail of the last rank's slice lies beyond the buffer: stage it)
                pad = torch.zeros(t, dtype=mine.dtype, device=device); pad[:mine.numel()] = mine; mine = pad
            if host_staged:
                parts = [torch.empty(t, dtype=mine.dtype) for _ in range(world)]
                dist.all_gather(parts, mine.cpu(), group=group)
                full = torch.cat(parts).to(device)
            elif T.windowed or off + t * world > T.cur.numel():
                assert T.stage is not None, "a gathered range beyond the layer buffer needs the staging buffer"
                full = T.stage[:t * world]
                dist.all_gather_into_tensor(full, mine.contiguous(), group=group)
            else:
                full = None
                dist.all_gather_into_tensor(T.cur[off:off + t * world], mine, group=group)   # RCCL, in place
            if full is not None:
                m = min(t * world, T.cur.numel() - off)
                T.cur[off:off + m] = full[:m]
            T.complete_layer(k)
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
        if T is not None:
            T.close()
        if borrowed:
            hip.reset_stream(handle)
