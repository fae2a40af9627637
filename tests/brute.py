"""Brute-force statements of one DP layer, written from the recurrence (DynamicSplitter.jl:33-46, :232-247) and from the
DEFINITIONS of the counts (test/test_SparseColorArrays.jl:1-11), sharing no code with the oracle, with tests/dc_model.py
or with the device kernels.  0-based boundary positions throughout: p = j - 1, r = j' - 1, a part is the columns [p, r).

  net_table(A)[p, r]      = number of distinct rows in the columns [p, r)
  selfnet_table(A)[p, r]  = number of rows whose whole support lies in the columns [p, r)
  cost_table(A, mdl, k)   = f(p, r, k) for all p <= r (object dtype for Int64 models would be slow: int64 / float64 arrays,
                            evaluated left to right like the reference's models)
  layer(W, F)             = (cst[r], ptr[r]) : min over p <= r of W[p] + F[p, r], the LARGEST p on ties
  layer_windowed(...)     = the same over p in [lo[r], hi[r]]
  block_argmins(W, F)     = the rightmost arg-min of every row over each of its Fenwick blocks [r_b - 2^b, r_b)
"""
import numpy as np


def _cols(A):
    return np.repeat(np.arange(A.n, dtype=np.int64), np.diff(A.colptr))


def net_table(A):
    n = A.n
    pos = (A.colptr - 1).astype(np.int64)
    rows = (A.rowval - 1).astype(np.int64)
    cols = _cols(A)
    # previous column holding the same row (stable sort by row keeps the column order)
    order = np.argsort(rows, kind="stable")
    prev = np.full(A.nnz, -1, dtype=np.int64)
    same = rows[order][1:] == rows[order][:-1]
    prev[order[1:][same]] = cols[order[:-1][same]]
    T = np.zeros((n + 1, n + 1), dtype=np.int64)
    for p in range(n + 1):
        fl = prev[pos[p]:] < p                      # first occurrence of its row at or after column p
        cs = np.concatenate([[0], np.cumsum(fl)])
        T[p, p:] = cs[pos[p:] - pos[p]]
    return T


def selfnet_table(A):
    n, m = A.n, A.m
    rows = (A.rowval - 1).astype(np.int64)
    cols = _cols(A)
    first = np.full(m, n, dtype=np.int64); last = np.full(m, -1, dtype=np.int64)
    np.minimum.at(first, rows, cols); np.maximum.at(last, rows, cols)
    ne = last >= 0
    T = np.zeros((n + 1, n + 1), dtype=np.int64)
    for p in range(n + 1):
        sel = ne & (first >= p)
        h = np.bincount(last[sel], minlength=n + 1)[:n + 1]       # rows by last column
        cs = np.concatenate([[0], np.cumsum(h)])                     # cs[r] = #rows with last < r
        T[p, p:] = cs[p:n + 1]
    return T


def cost_table(A, mdl, k=None, NT=None, ST=None):
    """F[p, r] for p <= r (upper triangle; the rest is left at 0)."""
    from util import cp
    n = A.n
    pos = (A.colptr - 1).astype(np.int64)
    P, R = np.meshgrid(np.arange(n + 1), np.arange(n + 1), indexing="ij")
    nv = (R - P).astype(np.int64)
    npins = (pos[R] - pos[P]).astype(np.int64)
    dt = np.int64 if mdl.dtype == cp.models.CP_I64 else np.float64
    if mdl.kind == cp.models.CP_MODEL_POWER_WORK:              # alpha + (nv*b_vertex + np*b_pin)^gamma  (test_Partitioners.jl:54-74)
        x = nv.astype(np.float64) * mdl.beta_vertex + npins.astype(np.float64) * mdl.beta_pin
        return np.triu(mdl.alpha + (x * x if mdl.gamma == 2.0 else np.power(np.maximum(x, 0.0), mdl.gamma)))
    a = mdl.alpha if (getattr(mdl, "alpha_k", None) is None or k is None) else mdl.alpha_k[k - 1]
    F = np.full((n + 1, n + 1), a, dtype=dt)
    F = F + nv.astype(dt) * dt(mdl.beta_vertex)
    F = F + npins.astype(dt) * dt(mdl.beta_pin)
    if mdl.kind == cp.models.CP_MODEL_CONNECTIVITY:
        NT = net_table(A) if NT is None else NT
        F = F + NT.astype(dt) * dt(mdl.beta_net)
    elif mdl.kind == cp.models.CP_MODEL_HYPEREDGE_CUT:
        NT = net_table(A) if NT is None else NT
        ST = selfnet_table(A) if ST is None else ST
        F = F + ST.astype(dt) * dt(mdl.beta_self_net)
        F = F + (NT - ST).astype(dt) * dt(mdl.beta_cut_net)
    return np.triu(F)


def _rightmost_argmin(v):
    """index of the LAST minimum of a 1-d array"""
    return v.size - 1 - int(np.argmin(v[::-1]))


def layer(W, F, lo=None, hi=None):
    """cst[r] = min_{lo[r] <= p <= hi[r]} W[p] + F[p, r] (default 0 <= p <= r), ptr[r] = the largest minimiser; rows with an
    empty range get (None, -1)."""
    n1 = F.shape[0]
    W = np.asarray(W)
    cst = np.zeros(n1, dtype=np.result_type(W.dtype, F.dtype)); ptr = np.full(n1, -1, dtype=np.int64)
    for r in range(n1):
        a = 0 if lo is None else int(lo[r])
        b = r if hi is None else int(hi[r])
        if b < a:
            continue
        v = W[a:b + 1] + F[a:b + 1, r]
        i = _rightmost_argmin(v)
        cst[r] = v[i]; ptr[r] = a + i
    return cst, ptr


def block_argmins(W, F, nbits):
    """opt[b, r] = rightmost arg-min of W[p] + F[p, r] over p in [r_b - 2^b, r_b) for every set bit b of r (-1 elsewhere)."""
    n1 = F.shape[0]
    W = np.asarray(W)
    opt = np.full((nbits, n1), -1, dtype=np.int64)
    for r in range(1, n1):
        for b in range(nbits):
            if not (r >> b) & 1:
                continue
            rb = (r >> b) << b
            v = W[rb - (1 << b):rb] + F[rb - (1 << b):rb, r]
            opt[b, r] = rb - (1 << b) + _rightmost_argmin(v)
    return opt


def lws_optimum(F, wmax=None):
    """min total cost of covering 0 .. n with parts [p, r) of width <= wmax (any number of parts): the least-weight
    subsequence problem of pack_stripe(A, *TotalChunker(f))"""
    n1 = F.shape[0]
    best = np.full(n1, np.inf); best[0] = 0.0
    for r in range(1, n1):
        a = 0 if wmax is None else max(0, r - wmax)
        best[r] = np.min(best[a:r] + F[a:r, r])
    return best[n1 - 1]


def kpart_optimum(Fk, K, wmax=None):
    """min total cost of K consecutive (possibly empty) parts, part k costed by Fk(k) (1-based), widths <= wmax"""
    n1 = Fk(1).shape[0]
    cst = Fk(1)[0, :].astype(np.float64).copy()
    if wmax is not None:
        cst[wmax + 1:] = np.inf
    for k in range(2, K + 1):
        F = Fk(k)
        new = np.full(n1, np.inf)
        for r in range(n1):
            a = 0 if wmax is None else max(0, r - wmax)
            new[r] = np.min(cst[a:r + 1] + F[a:r + 1, r])
        cst = new
    return cst[n1 - 1]


def partition_value(F_of_k, spl):
    """total cost of a split vector (1-based) from the definition tables"""
    return sum(F_of_k(k + 1)[spl[k] - 1, spl[k + 1] - 1] for k in range(len(spl) - 1))
