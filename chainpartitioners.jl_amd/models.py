"""Cost models, weights and partitioning methods of the hot path, plus their marshalling
into the flat C structs of include/chainpart_types.h.

Mirrors (file:line under /root/reference/src):
  AffineWorkModel                WorkCosts.jl:5-17
  AffineConnectivityModel        ConnectivityCosts.jl:7-20
  AffineHyperedgeCutModel        HyperedgeCutCosts.jl:7-21
  ColumnBlockComponentCostModel  BlockCosts.jl:1-17
  BlockComponentCostModel        BlockCosts.jl:19-44
  VertexCount                    SparseColorArrays.jl:1-6
  ConstrainedCost                Costs.jl:105-116
  EquiSplitter / EquiChunker     EquiPartitioner.jl:3-22
  Dynamic{Total,Bottleneck}{Splitter,Chunker}   DynamicSplitter.jl:1-13, DynamicChunker.jl:1-13
  Reference*                     ReferenceSplitter.jl:1-21
  [Flip]BisectCostBottleneckSplitter            BisectCostBottleneckSplitter.jl:1-4, 65-68
  ConvexTotalChunker / ConvexTotalSplitter      ConvexTotalChunker.jl:1-7

Julia's constructors `promote` all parameters to one element type (ConnectivityCosts.jl:14-16):
all-int arguments give an Int64 model, any float gives Float64.  Closures (e.g.
`w -> 1 + w`) cannot cross a C ABI, so callables are tabulated host-side for w = 0..w_table.
"""
from __future__ import annotations

import ctypes as C
import numpy as np

CP_I64, CP_F64 = 0, 1
(CP_MODEL_FEASIBLE, CP_MODEL_WORK, CP_MODEL_CONNECTIVITY, CP_MODEL_HYPEREDGE_CUT,
 CP_MODEL_COLBLOCK, CP_MODEL_BLOCK, CP_MODEL_VERTEX_COUNT, CP_MODEL_POWER_WORK, CP_MODEL_PRIMARY,
 CP_MODEL_SECONDARY) = range(10)
CP_COMBINE_SUM, CP_COMBINE_MAX = 0, 1
CP_ORDER_SPLITTER, CP_ORDER_CHUNKER = 0, 1
CP_MAX_R = 4
CP_OK, CP_EINVAL, CP_INFEASIBLE, CP_EHIP, CP_EUNSUPPORTED, CP_EINTERNAL = range(6)


class cp_component_t(C.Structure):
    _fields_ = [("is_const", C.c_int32), ("_pad", C.c_int32), ("c_i64", C.c_int64), ("c_f64", C.c_double),
                ("table", C.c_void_p), ("len", C.c_int64), ("lo", C.c_int64)]


class cp_model_t(C.Structure):
    _fields_ = [("kind", C.c_int32), ("dtype", C.c_int32), ("p_i64", C.c_int64 * 5), ("p_f64", C.c_double * 5),
                ("alpha_k", C.c_void_p), ("n_alpha_k", C.c_int64), ("R", C.c_int32), ("_pad", C.c_int32),
                ("alpha_row", cp_component_t), ("alpha_col", cp_component_t),
                ("beta_row", cp_component_t * CP_MAX_R), ("beta_col", cp_component_t * CP_MAX_R)]


class cp_rowpart_t(C.Structure):
    _fields_ = [("K", C.c_int64), ("asg", C.c_void_p), ("spl", C.c_void_p)]


def _is_int(x):
    return isinstance(x, (int, np.integer)) and not isinstance(x, bool) or isinstance(x, bool)


def _promote(*xs):
    """Julia promote(): Int64 iff every argument is an integer/bool."""
    flat = []
    for x in xs:
        if isinstance(x, (list, tuple, np.ndarray)):
            flat.extend(np.asarray(x).ravel().tolist())
        else:
            flat.append(x)
    return CP_I64 if all(_is_int(v) for v in flat) else CP_F64


class Marshalled:
    """A cp_model_t plus the numpy buffers it points into (kept alive together)."""

    def __init__(self, struct, keep):
        self.struct = struct
        self.keep = keep

    @property
    def ptr(self):
        return C.byref(self.struct)


class _Model:
    kind = None
    dtype = CP_I64
    w_table = None          # tabulation length for callables; set via with_table()

    def cost_dtype(self):
        return np.int64 if self.dtype == CP_I64 else np.float64

    def _params(self):
        return []

    def _alpha_k(self):
        return None

    def marshal(self, w_table=None, w_lo=0):
        s = cp_model_t()
        self._w_lo = w_lo
        keep = []
        s.kind = self.kind
        s.dtype = self.dtype
        for i, v in enumerate(self._params()):
            if self.dtype == CP_I64:
                s.p_i64[i] = int(v)
            else:
                s.p_f64[i] = float(v)
        ak = self._alpha_k()
        if ak is not None:
            arr = np.ascontiguousarray(ak, dtype=self.cost_dtype())
            keep.append(arr)
            s.alpha_k = arr.ctypes.data
            s.n_alpha_k = arr.size
        self._marshal_extra(s, keep, w_table)
        return Marshalled(s, keep)

    def _marshal_extra(self, s, keep, w_table):
        pass


class AffineWorkModel(_Model):
    kind = CP_MODEL_WORK

    def __init__(self, alpha=0, beta_vertex=0, beta_pin=0, *, alpha_k=None):
        self.alpha, self.beta_vertex, self.beta_pin = alpha, beta_vertex, beta_pin
        self.alpha_k = alpha_k
        self.dtype = _promote(alpha, beta_vertex, beta_pin, *( [alpha_k] if alpha_k is not None else []))

    def _params(self):
        return [self.alpha, self.beta_vertex, self.beta_pin]

    def _alpha_k(self):
        return self.alpha_k

    def __call__(self, n_vertices, n_pins, k=None):
        a = self.alpha if self.alpha_k is None or k is None else self.alpha_k[k - 1]
        return a + n_vertices * self.beta_vertex + n_pins * self.beta_pin


class PowerWorkModel(_Model):
    """alpha + (n_vertices*beta_vertex + n_pins*beta_pin)^gamma, Float64: the ConvexWorkModel (gamma 0.8) and
    ConcaveWorkModel (gamma 2) of the reference's tests (test/test_Partitioners.jl:54-74)."""
    kind = CP_MODEL_POWER_WORK

    def __init__(self, alpha, beta_vertex, beta_pin, gamma):
        self.alpha, self.beta_vertex, self.beta_pin, self.gamma = float(alpha), float(beta_vertex), float(beta_pin), float(gamma)
        self.alpha_k = None
        self.dtype = CP_F64

    def _params(self):
        return [self.alpha, self.beta_vertex, self.beta_pin, self.gamma]

    def _alpha_k(self):
        return None

    def __call__(self, n_vertices, n_pins, k=None):
        x = n_vertices * self.beta_vertex + n_pins * self.beta_pin
        return self.alpha + (x * x if self.gamma == 2.0 else x ** self.gamma)


def ConvexWorkModel(alpha, beta_vertex, beta_pin):
    return PowerWorkModel(alpha, beta_vertex, beta_pin, 0.8)


def ConcaveWorkModel(alpha, beta_vertex, beta_pin):
    return PowerWorkModel(alpha, beta_vertex, beta_pin, 2.0)


class AffineConnectivityModel(_Model):
    """alpha + nv*beta_vertex + np*beta_pin + nets*beta_net.  `alpha_k` gives the per-part
    alpha[k] of the reference tests' FunkyConnectivityModel (test_Partitioners.jl:1-8)."""
    kind = CP_MODEL_CONNECTIVITY

    def __init__(self, alpha=0, beta_vertex=0, beta_pin=0, beta_net=0, *, alpha_k=None):
        self.alpha, self.beta_vertex, self.beta_pin, self.beta_net = alpha, beta_vertex, beta_pin, beta_net
        self.alpha_k = alpha_k
        self.dtype = _promote(alpha, beta_vertex, beta_pin, beta_net, *([alpha_k] if alpha_k is not None else []))

    def _params(self):
        return [self.alpha, self.beta_vertex, self.beta_pin, self.beta_net]

    def _alpha_k(self):
        return self.alpha_k

    def __call__(self, n_vertices, n_pins, n_nets, k=None):
        a = self.alpha if self.alpha_k is None or k is None else self.alpha_k[k - 1]
        return a + n_vertices * self.beta_vertex + n_pins * self.beta_pin + n_nets * self.beta_net


class AffineHyperedgeCutModel(_Model):
    kind = CP_MODEL_HYPEREDGE_CUT

    def __init__(self, alpha=0, beta_vertex=0, beta_pin=0, beta_self_net=0, beta_cut_net=0, *, alpha_k=None):
        self.alpha, self.beta_vertex, self.beta_pin = alpha, beta_vertex, beta_pin
        self.beta_self_net, self.beta_cut_net = beta_self_net, beta_cut_net
        self.alpha_k = alpha_k
        self.dtype = _promote(alpha, beta_vertex, beta_pin, beta_self_net, beta_cut_net,
                              *([alpha_k] if alpha_k is not None else []))

    def _params(self):
        return [self.alpha, self.beta_vertex, self.beta_pin, self.beta_self_net, self.beta_cut_net]

    def _alpha_k(self):
        return self.alpha_k

    def __call__(self, n_vertices, n_pins, n_self, n_cut, k=None):
        a = self.alpha if self.alpha_k is None or k is None else self.alpha_k[k - 1]
        return (a + n_vertices * self.beta_vertex + n_pins * self.beta_pin
                + n_self * self.beta_self_net + n_cut * self.beta_cut_net)


class AffinePrimaryConnectivityModel(_Model):
    """PrimaryConnectivityCosts.jl:5-20: nets of a column part split by the row partition Pi into local (owned by the same
    part number) and remote ones.  Needs Pi (any partition of the rows)."""
    kind = CP_MODEL_PRIMARY
    needs_rowpart = True

    def __init__(self, alpha=0, beta_vertex=0, beta_pin=0, beta_local_net=0, beta_remote_net=0, *, alpha_k=None):
        self.alpha, self.beta_vertex, self.beta_pin = alpha, beta_vertex, beta_pin
        self.beta_local_net, self.beta_remote_net = beta_local_net, beta_remote_net
        self.alpha_k = alpha_k
        self.dtype = _promote(alpha, beta_vertex, beta_pin, beta_local_net, beta_remote_net, *([alpha_k] if alpha_k is not None else []))

    def _params(self):
        return [self.alpha, self.beta_vertex, self.beta_pin, self.beta_local_net, self.beta_remote_net]

    def _alpha_k(self):
        return self.alpha_k

    def __call__(self, n_vertices, n_pins, n_local, n_remote, k=None):
        a = self.alpha if self.alpha_k is None or k is None else self.alpha_k[k - 1]
        return a + n_vertices * self.beta_vertex + n_pins * self.beta_pin + n_local * self.beta_local_net + n_remote * self.beta_remote_net


class AffineSecondaryConnectivityModel(AffinePrimaryConnectivityModel):
    """SecondaryConnectivityCosts.jl:5-20: the cost of giving a range of columns to part k of the SplitPartition Pi of the
    rows -- the alternating partitioners call it on the adjoint, with Pi the column split found in the previous sweep."""
    kind = CP_MODEL_SECONDARY


def _tabulate(f, lo, hi, npdt):
    """f(w) for w = lo .. hi; vectorised when the callable allows it, checked against three scalar calls."""
    ws = np.arange(lo, hi + 1, dtype=np.int64)
    try:
        tab = np.asarray(f(ws))
        if tab.shape == ws.shape and all(tab[i] == f(int(ws[i])) for i in (0, len(ws) // 2, len(ws) - 1)):
            return tab.astype(npdt)
    except Exception:
        pass
    return np.array([f(int(w)) for w in ws], dtype=npdt)


def _component(f, dtype, w_table, keep, w_lo=0):
    """block_component(f, w) (BlockCosts.jl:41-44): number | callable | tuple/array (1-based).
    Callables are tabulated for w = w_lo .. w_table (w_lo < 0: see cp_component_t in include/chainpart_types.h)."""
    c = cp_component_t()
    npdt = np.int64 if dtype == CP_I64 else np.float64
    if callable(f):
        if w_table is None:
            raise ValueError("a callable block component needs w_table (tabulation length)")
        tab = _tabulate(f, w_lo, w_table, npdt)
        c.lo = int(w_lo)
    elif isinstance(f, (list, tuple, np.ndarray)):
        arr = np.asarray(f)
        tab = np.concatenate([[0], arr]).astype(npdt)   # Julia f[w], w >= 1
    else:
        c.is_const = 1
        if dtype == CP_I64:
            c.c_i64 = int(f)
        else:
            c.c_f64 = float(f)
        return c
    tab = np.ascontiguousarray(tab)
    keep.append(tab)
    c.is_const = 0
    c.table = tab.ctypes.data
    c.len = tab.size
    return c


class ColumnBlockComponentCostModel(_Model):
    """1-D VBR cost alpha_col(w) + nets * beta_col(w); an AbstractConnectivityModel (BlockCosts.jl:1-17)."""
    kind = CP_MODEL_COLBLOCK

    def __init__(self, alpha_col=0, beta_col=0, dtype=int, w_table=None):
        self.alpha_col, self.beta_col = alpha_col, beta_col
        self.dtype = CP_I64 if dtype in (int, np.int64, "int", CP_I64) else CP_F64
        self.w_table = w_table

    def _marshal_extra(self, s, keep, w_table):
        wt = w_table if w_table is not None else self.w_table
        s.R = 1
        s.alpha_col = _component(self.alpha_col, self.dtype, wt, keep, getattr(self, "_w_lo", 0))
        s.beta_col[0] = _component(self.beta_col, self.dtype, wt, keep, getattr(self, "_w_lo", 0))

    def __call__(self, n_vertices, n_pins, n_nets, k=None):
        bc = lambda f, w: f(w) if callable(f) else (f[w - 1] if isinstance(f, (list, tuple, np.ndarray)) else f)
        return bc(self.alpha_col, n_vertices) + n_nets * bc(self.beta_col, n_vertices)


class BlockComponentCostModel(_Model):
    """Rank-R separable 2-D VBR cost (BlockCosts.jl:19-44); needs a row partition Pi."""
    kind = CP_MODEL_BLOCK

    def __init__(self, alpha_row=0, alpha_col=0, beta_row=(), beta_col=(), dtype=int, w_table=None, u_table=None):
        assert len(beta_row) == len(beta_col) <= CP_MAX_R
        self.alpha_row, self.alpha_col = alpha_row, alpha_col
        self.beta_row, self.beta_col = tuple(beta_row), tuple(beta_col)
        self.dtype = CP_I64 if dtype in (int, np.int64, "int", CP_I64) else CP_F64
        self.w_table, self.u_table = w_table, u_table

    def _marshal_extra(self, s, keep, w_table):
        wt = w_table if w_table is not None else self.w_table
        ut = self.u_table if self.u_table is not None else wt
        s.R = len(self.beta_row)
        s.alpha_row = _component(self.alpha_row, self.dtype, ut, keep)
        lo = getattr(self, "_w_lo", 0)
        s.alpha_col = _component(self.alpha_col, self.dtype, wt, keep, lo)
        for r in range(s.R):
            s.beta_row[r] = _component(self.beta_row[r], self.dtype, ut, keep)
            s.beta_col[r] = _component(self.beta_col[r], self.dtype, wt, keep, lo)


class VertexCount(_Model):
    """w(j, j') = j' - j, always Int (SparseColorArrays.jl:1-6)."""
    kind = CP_MODEL_VERTEX_COUNT
    dtype = CP_I64


class FeasibleCost(_Model):
    kind = CP_MODEL_FEASIBLE
    dtype = CP_I64


class ConstrainedCost:
    """ConstrainedCost(f, w, w_max): f where w(j, j') <= w_max, infinity elsewhere (Costs.jl:105-147)."""

    def __init__(self, f, w, w_max):
        self.f, self.w, self.w_max = f, w, w_max


def split_constraint(f):
    """-> (model, weight | None, w_max)"""
    if isinstance(f, ConstrainedCost):
        if isinstance(f.w, FeasibleCost):
            return f.f, None, 0
        return f.f, f.w, f.w_max
    return f, None, 0


# ---------------------------------------------------------------- methods
class EquiSplitter:
    pass


class EquiChunker:
    def __init__(self, w):
        self.w = int(w)


class _FMethod:
    def __init__(self, f):
        self.f = f


class DynamicTotalSplitter(_FMethod):
    combine, order = CP_COMBINE_SUM, CP_ORDER_SPLITTER


class DynamicBottleneckSplitter(_FMethod):
    combine, order = CP_COMBINE_MAX, CP_ORDER_SPLITTER


class DynamicTotalChunker(_FMethod):
    combine, order = CP_COMBINE_SUM, CP_ORDER_CHUNKER


class DynamicBottleneckChunker(_FMethod):
    combine, order = CP_COMBINE_MAX, CP_ORDER_CHUNKER


class ReferenceTotalSplitter(DynamicTotalSplitter):
    """ReferenceSplitter.jl:1-6: invokes the generic DynamicTotalSplitter method."""


class ReferenceBottleneckSplitter(DynamicBottleneckSplitter):
    """ReferenceSplitter.jl:8-13"""


class ReferenceTotalChunker(DynamicTotalChunker):
    """ReferenceSplitter.jl:16-21"""


class BisectCostBottleneckSplitter:
    flip = 0

    def __init__(self, f, eps):
        self.f, self.eps = f, float(eps)


class FlipBisectCostBottleneckSplitter(BisectCostBottleneckSplitter):
    flip = 1


class BisectIndexBottleneckSplitter:
    """BisectIndexBottleneckSplitter.jl:1-83 (exact bottleneck by bisection over split indices)"""
    flip = 0

    def __init__(self, f):
        self.f = f


class FlipBisectIndexBottleneckSplitter(BisectIndexBottleneckSplitter):
    """BisectIndexBottleneckSplitter.jl:85-166"""
    flip = 1


class LazyBisectCostBottleneckSplitter:
    """LazyBisectCostBottleneckSplitter.jl:1-4, connectivity specialisation :140-258"""

    def __init__(self, f, eps):
        self.f, self.eps = f, float(eps)


class DisjointPartitioner:
    """AlternatingPartitioner.jl:1-10: columns first, then the rows of the adjoint given the column split."""

    def __init__(self, mtd, mtd2):
        self.mtd, self.mtd2 = mtd, mtd2


class AlternatingPartitioner:
    """AlternatingPartitioner.jl:12-32: Phi on A, Pi on the adjoint given Phi, then alternately again."""

    def __init__(self, *mtds):
        assert len(mtds) >= 2
        self.mtds = tuple(mtds)


class AlternatingNetPartitioner(AlternatingPartitioner):
    """AlternatingPartitioner.jl:34-58: same sweeps, sharing one net counter of A (the device keeps A's link arrays anyway)."""

    def __init__(self, *mtds, hint=None):
        super().__init__(*mtds)
        self.hint = hint


class SymmetricPartitioner:
    """AlternatingPartitioner.jl:60-87: one partition for rows and columns of a square matrix."""

    def __init__(self, *mtds):
        assert len(mtds) >= 1
        self.mtds = tuple(mtds)


class ConvexTotalChunker(_FMethod):
    pass


class ConvexTotalSplitter(_FMethod):
    pass


class ConcaveTotalChunker(_FMethod):
    """ConcaveTotalChunker.jl:1-24"""


class ConcaveTotalSplitter(_FMethod):
    """ConcaveTotalChunker.jl:5-55, :140-180"""
