"""partition_stripe / pack_stripe / oracle_stripe / bound_stripe / total_value /
bottleneck_value -- the reference's dispatch API for the hot path, host side.

Each function marshals its Julia-shaped arguments into the flat C structs of
include/chainpart_types.h and calls a backend.  The product backend is the HIP C-ABI
library (`_lib.HipBackend`, libchainpart.so); there is no CPU fallback in this package --
if the library or a GPU is missing the call raises.  Tests inject the CPU oracle through
the same `backend=` hook to compare the two on identical marshalled inputs.

Reference entry points mirrored (under /root/reference/src):
  partition_stripe  EquiPartitioner.jl:5, DynamicSplitter.jl:15,52,206,260,
                    BisectCostBottleneckSplitter.jl:6,70, ConvexTotalChunker.jl:26,170
  pack_stripe       EquiPartitioner.jl:15, DynamicChunker.jl:15,20, ConvexTotalChunker.jl:9,141
  oracle_stripe / bound_stripe / total_value / bottleneck_value   Costs.jl:3-66
"""
from __future__ import annotations

import numpy as np

from . import models as M
from .types import (SparseMatrixCSC, SplitPartition, MapPartition, DomainPartition, StepHint, NoHint, to_map)

_default_backend = None


def set_default_backend(b):
    global _default_backend
    _default_backend = b


def get_backend(backend=None):
    if backend is not None:
        return backend
    global _default_backend
    if _default_backend is None:
        from ._lib import HipBackend
        _default_backend = HipBackend()          # raises loudly if the HIP library / GPU is absent
    return _default_backend


class CPError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__(f"chainpart status {code}: {msg}")
        self.code = code


def _check(rc, what, backend):
    if rc == M.CP_OK or rc == M.CP_INFEASIBLE:
        return rc
    msg = backend.last_error() if hasattr(backend, "last_error") else ""
    if rc == M.CP_EINVAL:
        raise AssertionError(f"{what}: violated precondition ({msg})")   # Julia: AssertionError
    if rc == M.CP_EUNSUPPORTED:
        raise NotImplementedError(f"{what}: no method for this (method, model) pair ({msg})")
    raise CPError(rc, f"{what}: {msg}")


def _rowpart(Pi, need_spl=False):
    """Marshal an optional row partition -> (cp_rowpart_t | None, keepalive)."""
    if Pi is None:
        return None, []
    rp = M.cp_rowpart_t()
    keep = []
    rp.K = Pi.K
    if isinstance(Pi, SplitPartition):
        spl = np.ascontiguousarray(Pi.spl, dtype=np.int64)
        asg = to_map(Pi).asg
        keep += [spl, asg]
        rp.spl = spl.ctypes.data
        rp.asg = asg.ctypes.data
    elif isinstance(Pi, MapPartition):
        asg = np.ascontiguousarray(Pi.asg, dtype=np.int64)
        keep += [asg]
        rp.asg = asg.ctypes.data
        rp.spl = None
    else:
        raise TypeError("row partition must be a SplitPartition or MapPartition")
    return rp, keep


def _w_table_for(A, f, weight, w_max, stack_method=False):
    """Tabulation range (lo, hi) of closures over the part width.  DP methods only evaluate feasible pairs: 0 .. w_max under a
    VertexCount constraint, else 0 .. n.  The Convex/Concave chunkers (stack_method) also evaluate pairs outside the
    constraint and -- when the cost is not convex and their candidate stack goes stale -- pairs with j > j' inside one
    feasible window (ConvexTotalChunker.jl:76,99): widths -(w_max+1) .. 2(w_max+1) under VertexCount, -n .. n otherwise."""
    if weight is not None and isinstance(weight, M.VertexCount):
        w = int(w_max)
        return (-(2 * w + 2), 2 * w + 3) if stack_method else (0, w + 1)
    return (-(A.n + 1), A.n + 1) if stack_method else (0, A.n + 1)


def _marshal(A, f, Pi, stack_method=False):
    mdl, weight, w_max = M.split_constraint(f)
    wlo, wt = _w_table_for(A, mdl, weight, w_max, stack_method)
    if not isinstance(mdl, (M.BlockComponentCostModel, M.ColumnBlockComponentCostModel)):
        wlo = 0
    if isinstance(mdl, M.BlockComponentCostModel):
        ut = A.m + 1
        mm = mdl.marshal(w_table=wt, w_lo=wlo) if mdl.u_table is not None else _marshal_block(mdl, wt, ut, wlo)
    else:
        mm = mdl.marshal(w_table=wt, w_lo=wlo)
    wm = weight.marshal() if weight is not None else None
    wmax_i = int(w_max) if weight is not None and (weight.dtype == M.CP_I64) else 0
    wmax_f = float(w_max) if weight is not None else 0.0
    rp, keep = _rowpart(Pi)
    return mdl, mm, wm, wmax_i, wmax_f, rp, keep


def _marshal_block(mdl, wt, ut, wlo=0):
    old = (mdl.w_table, mdl.u_table)
    mdl.w_table, mdl.u_table = wt, ut
    try:
        return mdl.marshal(w_lo=wlo)
    finally:
        mdl.w_table, mdl.u_table = old


# ---------------------------------------------------------------- partition_stripe
def partition_stripe(A: SparseMatrixCSC, K, method, Pi=None, *, backend=None) -> SplitPartition:
    K = int(K)
    if isinstance(method, M.EquiSplitter):
        # closed form, O(K) host arithmetic (EquiPartitioner.jl:7)
        n = A.n
        k = np.arange(0, K + 1, dtype=np.int64)
        return SplitPartition(K, k * (n // K) + np.minimum(n % K, k) + 1)
    b = get_backend(backend)
    if isinstance(method, (M.DynamicTotalSplitter, M.DynamicBottleneckSplitter,
                           M.DynamicTotalChunker, M.DynamicBottleneckChunker)):
        mdl, mm, wm, wi, wf, rp, keep = _marshal(A, method.f, Pi)
        spl = np.zeros(K + 1, dtype=np.int64)
        rc = b.partition_dynamic(A, K, method.combine, method.order, mm, rp, wm, wi, wf, spl)
        _check(rc, "partition_stripe(Dynamic*)", b)
        return SplitPartition(K, spl)
    if isinstance(method, M.BisectCostBottleneckSplitter):
        mdl, mm, wm, wi, wf, rp, keep = _marshal(A, method.f, Pi if getattr(M.split_constraint(method.f)[0], "needs_rowpart", False) else None)
        if wm is not None:
            raise NotImplementedError("BisectCost on a ConstrainedCost errors in the reference (Costs.jl:150)")
        spl = np.zeros(K + 1, dtype=np.int64)
        rc = b.partition_bisect_cost(A, K, mm, method.eps, method.flip, spl, rp)
        _check(rc, "partition_stripe(BisectCost)", b)
        return SplitPartition(K, spl)
    if isinstance(method, M.BisectIndexBottleneckSplitter):
        mdl, mm, wm, wi, wf, rp, keep = _marshal(A, method.f, Pi if getattr(M.split_constraint(method.f)[0], "needs_rowpart", False) else None)
        if wm is not None:
            raise NotImplementedError("BisectIndex on a ConstrainedCost errors in the reference (Costs.jl:150)")
        spl = np.zeros(K + 1, dtype=np.int64)
        rc = b.partition_bisect_index(A, K, mm, method.flip, spl, rp)
        _check(rc, "partition_stripe(BisectIndex)", b)
        return SplitPartition(K, spl)
    if isinstance(method, M.LazyBisectCostBottleneckSplitter):
        mdl, mm, wm, wi, wf, rp, keep = _marshal(A, method.f, None)
        if wm is not None:
            raise NotImplementedError("LazyBisectCost on a ConstrainedCost has no method in the reference")
        spl = np.zeros(K + 1, dtype=np.int64)
        rc = b.partition_lazy_bisect_cost(A, K, mm, method.eps, spl)
        _check(rc, "partition_stripe(LazyBisectCost)", b)
        return SplitPartition(K, spl)
    if isinstance(method, M.ConvexTotalSplitter):
        mdl, mm, wm, wi, wf, rp, keep = _marshal(A, method.f, Pi, stack_method=True)
        spl = np.zeros(K + 1, dtype=np.int64)
        rc = b.partition_convex(A, K, mm, rp, wm, wi, wf, spl)
        _check(rc, "partition_stripe(ConvexTotalSplitter)", b)
        return SplitPartition(K, spl)
    if isinstance(method, M.ConcaveTotalSplitter):
        mdl, mm, wm, wi, wf, rp, keep = _marshal(A, method.f, Pi, stack_method=True)
        spl = np.zeros(K + 1, dtype=np.int64)
        rc = b.partition_concave(A, K, mm, rp, wm, wi, wf, spl)
        _check(rc, "partition_stripe(ConcaveTotalSplitter)", b)
        return SplitPartition(K, spl)
    raise NotImplementedError(f"partition_stripe: method {type(method).__name__} is outside the hot path")


def partition_stripe_batch(A: SparseMatrixCSC, requests, *, backend=None):
    """[partition_stripe(A, K, method) for (K, method) in requests] in ONE device launch, for the methods whose single partition is a
    sequential chain on one wave: BisectCostBottleneckSplitter(f, eps) with Work / Connectivity costs (a sweep over K, eps and the
    model constants; BisectCostBottleneckSplitter.jl:6-63 per request).  The reference has no batch call -- it would loop; the
    results are those of the loop."""
    b = get_backend(backend)
    if not requests:
        return []
    if not all(isinstance(m, M.BisectCostBottleneckSplitter) for _, m in requests):
        raise NotImplementedError("partition_stripe_batch takes BisectCostBottleneckSplitter requests")
    if not hasattr(b, "partition_bisect_cost_batch"):
        return [partition_stripe(A, K, m, backend=b) for K, m in requests]
    mms = []
    for K, m in requests:
        mdl, mm, wm, wi, wf, rp, keep = _marshal(A, m.f, None)
        if wm is not None:
            raise NotImplementedError("BisectCost on a ConstrainedCost errors in the reference (Costs.jl:150)")
        mms.append(mm)
    rc, spls = b.partition_bisect_cost_batch(A, [int(K) for K, _ in requests], mms, [m.eps for _, m in requests], [int(m.flip) for _, m in requests])
    _check(rc, "partition_stripe_batch(BisectCost)", b)
    return [SplitPartition(int(K), s) for (K, _), s in zip(requests, spls)]


def pack_stripe_batch(A: SparseMatrixCSC, methods, *, backend=None):
    """[pack_stripe(A, method) for method in methods] in ONE device launch for ConvexTotalChunker(ConstrainedCost(f, VertexCount(), w))
    requests with 1 <= w <= 15 and ColumnBlock / Connectivity / Work costs (ConvexTotalChunker.jl:141-168, 211-265 per request): a sweep
    over the cost constants and width limits.  The results are those of the loop."""
    b = get_backend(backend)
    if not methods:
        return []
    ok = all(isinstance(m, M.ConvexTotalChunker) and isinstance(M.split_constraint(m.f)[1], M.VertexCount) for m in methods)
    if not ok:
        raise NotImplementedError("pack_stripe_batch takes ConvexTotalChunker(ConstrainedCost(f, VertexCount(), w)) requests")
    if not hasattr(b, "pack_convex_batch"):
        return [pack_stripe(A, m, backend=b) for m in methods]
    mms, ws = [], []
    for m in methods:
        mdl, mm, wm, wi, wf, rp, keep = _marshal(A, m.f, None, stack_method=True)
        mms.append(mm); ws.append(int(wi))
    rc, spls = b.pack_convex_batch(A, mms, ws, A.n)
    _check(rc, "pack_stripe_batch(ConvexTotalChunker)", b)
    return [SplitPartition(len(s) - 1, s) for s in spls]


# ---------------------------------------------------------------- pack_stripe
def pack_stripe(A: SparseMatrixCSC, method, Pi=None, *, backend=None) -> SplitPartition:
    if isinstance(method, M.EquiChunker):
        n, w = A.n, method.w
        spl = np.concatenate([np.arange(1, n + 1, w, dtype=np.int64), [n + 1]])   # [1:w:n; n+1]
        return SplitPartition(len(spl) - 1, spl)
    b = get_backend(backend)
    if isinstance(method, M.DynamicTotalChunker):
        mdl, mm, wm, wi, wf, rp, keep = _marshal(A, method.f, Pi)
        spl = np.zeros(A.n + 1, dtype=np.int64)
        Kout = np.zeros(1, dtype=np.int64)
        rc = b.pack_dynamic(A, mm, rp, wm, wi, wf, spl, Kout)
        _check(rc, "pack_stripe(DynamicTotalChunker)", b)
        return SplitPartition(int(Kout[0]), spl[:int(Kout[0]) + 1].copy())
    if isinstance(method, M.ConvexTotalChunker):
        mdl, mm, wm, wi, wf, rp, keep = _marshal(A, method.f, Pi, stack_method=True)
        spl = np.zeros(A.n + 1, dtype=np.int64)
        Kout = np.zeros(1, dtype=np.int64)
        rc = b.pack_convex(A, mm, rp, wm, wi, wf, spl, Kout)
        _check(rc, "pack_stripe(ConvexTotalChunker)", b)
        return SplitPartition(int(Kout[0]), spl[:int(Kout[0]) + 1].copy())
    if isinstance(method, M.ConcaveTotalChunker):
        mdl, mm, wm, wi, wf, rp, keep = _marshal(A, method.f, Pi, stack_method=True)
        spl = np.zeros(A.n + 1, dtype=np.int64)
        Kout = np.zeros(1, dtype=np.int64)
        rc = b.pack_concave(A, mm, rp, wm, wi, wf, spl, Kout)
        _check(rc, "pack_stripe(ConcaveTotalChunker)", b)
        return SplitPartition(int(Kout[0]), spl[:int(Kout[0]) + 1].copy())
    raise NotImplementedError(f"pack_stripe: method {type(method).__name__} is outside the hot path")


# ---------------------------------------------------------------- oracles / scoring
class Oracle:
    """Callable cost oracle ocl(j, j', k...) (vectorised over arrays of queries)."""

    def __init__(self, hint, mdl, A, Pi, backend):
        self.hint, self.mdl, self.A, self.Pi, self.backend = hint, mdl, A, Pi, backend

    def __call__(self, j, jp, k=None):
        scalar = np.isscalar(j)
        jj = np.atleast_1d(np.asarray(j, dtype=np.int64))
        jjp = np.atleast_1d(np.asarray(jp, dtype=np.int64))
        kk = None if k is None else np.broadcast_to(np.asarray(k, dtype=np.int64), jj.shape).copy()
        _, mm, _, _, _, rp, keep = _marshal(self.A, self.mdl, self.Pi)
        out = np.zeros(jj.shape, dtype=self.mdl.cost_dtype())
        rc = self.backend.oracle_eval(self.A, mm, rp, self.hint.code, jj, jjp, kk, out)
        _check(rc, "oracle", self.backend)
        return out[0].item() if scalar else out


class _Move:
    code = None

    def __init__(self, arg):
        self.arg = int(arg)


class Same(_Move):
    code = 0


class Next(_Move):
    code = 1


class Prev(_Move):
    code = 2


class Jump(_Move):
    code = 3


class Step:
    """Step(ocl)(Same(j) | Next(j) | Prev(j) | Jump(j), <same for j'>, [Same(k)])  (Costs.jl:174-195).  Single calls keep the
    walk's previous position (the moves are promises about it, checked by the backend); `walk` takes a whole sequence."""

    def __init__(self, ocl):
        self.ocl = ocl
        self._last = None

    def __call__(self, mj, mjp, mk=None):
        k = None if mk is None else (mk.arg if isinstance(mk, _Move) else int(mk))
        if self._last is None:
            vals = self.walk([(mj, mjp, k)])
        else:                                       # replay the previous position so that the promise is checked
            vals = self.walk([(Jump(self._last[0]), Jump(self._last[1]), self._last[2]), (mj, mjp, k)])
        self._last = (mj.arg, mjp.arg, k)
        return vals[-1].item()

    def walk(self, moves):
        o = self.ocl
        mj = np.array([m[0].code for m in moves], dtype=np.int32); j = np.array([m[0].arg for m in moves], dtype=np.int64)
        mjp = np.array([m[1].code for m in moves], dtype=np.int32); jp = np.array([m[1].arg for m in moves], dtype=np.int64)
        ks = [m[2] if len(m) > 2 else None for m in moves]
        kk = None if all(x is None for x in ks) else np.array([0 if x is None else (x.arg if isinstance(x, _Move) else int(x)) for x in ks], dtype=np.int64)
        _, mm, _, _, _, rp, keep = _marshal(o.A, o.mdl, o.Pi)
        out = np.zeros(j.shape, dtype=o.mdl.cost_dtype())
        rc = o.backend.oracle_step(o.A, mm, rp, mj, j, mjp, jp, kk, out)
        _check(rc, "Step(oracle)", o.backend)
        return out


def oracle_stripe(hint, mdl, A, Pi=None, *, backend=None) -> Oracle:
    return Oracle(hint, mdl, A, Pi, get_backend(backend))


def bound_stripe(A, K, mdl, Pi=None, *, backend=None):
    """bound_stripe(A, K, [Pi], mdl): Pi only matters to the secondary connectivity model (Costs.jl:17-19)."""
    b = get_backend(backend)
    mm = mdl.marshal(w_table=A.n + 1)
    if Pi is not None and isinstance(mdl, M.AffineSecondaryConnectivityModel):
        rp, keep = _rowpart(Pi)
        rc, lo, hi = b.bound_stripe_pi(A, int(K), rp, mm)
    else:
        rc, lo, hi = b.bound_stripe(A, int(K), mm)
    _check(rc, "bound_stripe", b)
    return lo, hi


def _objective(g, A, Phi, mdl, Pi, backend):
    b = get_backend(backend)
    if not isinstance(Phi, SplitPartition):
        raise NotImplementedError("scoring of non-contiguous partitions is outside the hot path")
    _, mm, _, _, _, rp, keep = _marshal(A, mdl, Pi)
    rc, v = b.objective(A, Phi.K, np.ascontiguousarray(Phi.spl, dtype=np.int64), mm, rp, g)
    _check(rc, "objective", b)
    return v


def total_value(A, Phi, mdl, Pi=None, *, backend=None):
    """total_value(A, [Pi], Phi, mdl)  Costs.jl:28-29"""
    return _objective(M.CP_COMBINE_SUM, A, Phi, M.split_constraint(mdl)[0], Pi, backend)


def bottleneck_value(A, Phi, mdl, Pi=None, *, backend=None):
    """bottleneck_value(A, [Pi], Phi, mdl)  Costs.jl:26-27"""
    return _objective(M.CP_COMBINE_MAX, A, Phi, M.split_constraint(mdl)[0], Pi, backend)


def partition_plaid(A: SparseMatrixCSC, K, method, *, adj_A=None, backend=None):
    """partition_plaid(A, K, method) -> (Pi, Phi)  (AlternatingPartitioner.jl:6-87): the 2-D callers of partition_stripe.
    A and its adjoint both stay resident on the device between the sweeps."""
    K = int(K)
    if isinstance(method, M.DisjointPartitioner):
        Phi = partition_stripe(A, K, method.mtd, backend=backend)
        T = adj_A if adj_A is not None else adjointpattern(A, backend=backend)
        Pi = partition_stripe(T, K, method.mtd2, Phi, backend=backend)
        return Pi, Phi
    if isinstance(method, M.AlternatingPartitioner):            # also AlternatingNetPartitioner
        T = adj_A if adj_A is not None else adjointpattern(A, backend=backend)
        Phi = partition_stripe(A, K, method.mtds[0], backend=backend)
        Pi = partition_stripe(T, K, method.mtds[1], Phi, backend=backend)
        for i, mtd in enumerate(method.mtds[2:], start=1):
            if i % 2 == 1:
                Phi = partition_stripe(A, K, mtd, Pi, backend=backend)
            else:
                Pi = partition_stripe(T, K, mtd, Phi, backend=backend)
        return Pi, Phi
    if isinstance(method, M.SymmetricPartitioner):
        if len(method.mtds) > 1:
            T = adj_A if adj_A is not None else adjointpattern(A, backend=backend)
            Pi = partition_stripe(A, K, method.mtds[0], backend=backend)
            for i, mtd in enumerate(method.mtds[1:], start=1):
                Pi = partition_stripe(A if i % 2 == 1 else T, K, mtd, Pi, backend=backend)
        else:
            Pi = partition_stripe(A, K, method.mtds[0], backend=backend)
        return Pi, Pi
    raise NotImplementedError(f"partition_plaid: method {type(method).__name__} is outside the hot path")


def adjointpattern(A: SparseMatrixCSC, *, backend=None) -> SparseMatrixCSC:
    """adjointpattern(A) (util.jl:67-95): transposed pattern, computed on the device (cp_adjoint); the returned matrix
    keeps its device handle, so partitioning it needs no upload."""
    return get_backend(backend).adjoint(A)


# ---------------------------------------------------------------- counting structures
class CountMatrix:
    """netcount / selfnetcount / dominancecount object: obj[j, j'] (or obj(i, j)), vectorised."""

    def __init__(self, kind, A, hint, backend):
        self.kind, self.A, self.hint, self.backend = kind, A, hint, backend
        self.handle = backend.count_build(kind, A, hint.code)

    def __call__(self, a, b):
        scalar = np.isscalar(a)
        aa = np.atleast_1d(np.asarray(a, dtype=np.int64))
        bb = np.atleast_1d(np.asarray(b, dtype=np.int64))
        out = np.zeros(aa.shape, dtype=np.int64)
        rc = self.backend.count_query(self.kind, self.handle, aa, bb, out)
        _check(rc, "count query", self.backend)
        return int(out[0]) if scalar else out

    def __getitem__(self, ij):
        return self(*ij)

    def __del__(self):
        try:
            self.backend.count_free(self.kind, self.handle)
        except Exception:
            pass


def netcount(A, hint=None, *, backend=None):
    """netcount(hint, A)  SparseColorArrays.jl:57-58"""
    return CountMatrix("net", A, hint or NoHint(), get_backend(backend))


def selfnetcount(A, hint=None, *, backend=None):
    """selfnetcount(hint, A)  SparseColorArrays.jl:165-166"""
    return CountMatrix("selfnet", A, hint or NoHint(), get_backend(backend))


def dominancecount(A, hint=None, *, backend=None):
    """dominancecount(hint, A)  SparsePrefixMatrices.jl:438-446 : C[i, j] = #{nonzeros row < i, col < j}"""
    return CountMatrix("dom", A, hint or NoHint(), get_backend(backend))
