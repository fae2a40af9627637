"""Executable specification (numpy, small n) of the device algorithm for one layer of the
total-cost splitter DP -- the "Fenwick rectangles + monotone divide-and-conquer + staircase
path prefix sums" scheme of DESIGN.md section 4.  Used by the CPU tests to prove that the scheme
reproduces the literal DP tables of the oracle (values AND argmins, ties -> largest j), and by
the GPU tests as a readable statement of what the kernels in csrc/dp_total.hip compute.

All indices here are 0-based boundary positions: p = j-1, r = j'-1, part = columns [p, r).
"""
import numpy as np


def link_arrays(A):
    """prev[q] / next[q]: previous / next column holding the same row (0-based; -1 / n if none)."""
    n, N = A.n, A.nnz
    cols = np.repeat(np.arange(n), np.diff(A.colptr))
    rows = A.rowval - 1
    prev = np.full(N, -1, dtype=np.int64)
    nxt = np.full(N, n, dtype=np.int64)
    last = {}
    for q in range(N):
        i = rows[q]
        if i in last:
            prev[q] = cols[last[i]]
            nxt[last[i]] = cols[q]
        last[i] = q
    return prev, nxt


def first_last(A):
    """first / last column (0-based) of every row; -1 for empty rows."""
    n = A.n
    first = np.full(A.m, -1, dtype=np.int64); last = np.full(A.m, -1, dtype=np.int64)
    cols = np.repeat(np.arange(n), np.diff(A.colptr))
    for q in range(A.nnz):
        i = A.rowval[q] - 1
        if first[i] < 0:
            first[i] = cols[q]
        last[i] = cols[q]
    return first, last


def layer_total(A, Wprev, fcost, prev, nxt, first=None, last=None, blocks=False):
    """Return (cst[r], ptr[r]) for r in 0..n:  min over p<=r of Wprev[p] + fcost(p, r, nets(p,r)[, selfnets(p,r)]),
    ties -> largest p.  With first/last given the count is the pair (nets, selfnets) (hyperedge-cut costs):
    selfnets(p, r) = #rows with first >= p and last < r; a column c joining on the right of a part starting at B
    adds the rows with last == c and first >= B; a column p joining on the left of a part ending before r adds
    the rows with first == p and last < r."""
    n = A.n
    pos = A.colptr - 1
    nb = max(1, int(n).bit_length())
    opt = np.full((n + 1, nb), -1, dtype=np.int64)
    nnopt = np.zeros((n + 1, nb), dtype=object)
    val = np.full((n + 1, nb), np.inf)
    hyper = first is not None
    zero = np.array([0, 0]) if hyper else 0

    def right_delta(c, thr):      # add column c on the right of a part starting at thr
        d = int(np.sum(prev[pos[c]:pos[c + 1]] < thr))
        if hyper:
            return np.array([d, int(np.sum((last == c) & (first >= thr)))])
        return d

    def left_delta(p, r):         # add column p on the left of a part ending before r
        d = int(np.sum(nxt[pos[p]:pos[p + 1]] >= r))
        if hyper:
            return np.array([d, int(np.sum((first == p) & (last < r)))])
        return d

    def run_task(r, b, B, a, S0, cols_right, virtual):
        nn = S0 + 0                # copy: the hyperedge count is a numpy pair
        best = None
        for c in cols_right:
            nn = nn + right_delta(c, B)
        if not virtual:
            best = (Wprev[B] + fcost(B, r, nn), B, nn)
        for p in range(B - 1, a - 1, -1):
            nn = nn + left_delta(p, r)
            v = Wprev[p] + fcost(p, r, nn)
            if best is None or v < best[0]:        # strict: ties keep the larger p
                best = (v, p, nn)
        val[r, b], opt[r, b], nnopt[r, b] = best

    # round A: rho == 0 rows of every rectangle
    for r in range(1, n + 1):
        b = (r & -r).bit_length() - 1
        run_task(r, b, r, r - (1 << b), zero, [], True)
    # rounds tau = high .. 0
    for tau in range(nb - 1, -1, -1):
        for r in range(1 << tau, n + 1, 1 << (tau + 1)):      # ctz(r) == tau
            for b in range(tau + 1, nb):
                if not (r >> b) & 1:
                    continue
                rb = (r >> b) << b
                B0 = rb - (1 << b)
                rL = r - (1 << tau)
                rR = r + (1 << tau)
                B = opt[rL, b]
                S0 = nnopt[rL, b]
                a = opt[rR, b] if (rR - rb) < (1 << b) and rR <= n else B0
                run_task(r, b, B, a, S0, range(rL, r), False)
    cst = np.zeros(n + 1, dtype=object)
    ptr = np.zeros(n + 1, dtype=np.int64)
    for r in range(n + 1):
        bv, bp = Wprev[r] + fcost(r, r, zero), r
        for b in range(nb):
            if (r >> b) & 1 and val[r, b] < bv:
                bv, bp = val[r, b], opt[r, b]
        cst[r], ptr[r] = bv, bp
    if blocks:
        return cst, ptr, opt.T.copy()          # opt[b, r]: the per-block winners the combine step merged
    return cst, ptr


# ---------------------------------------------------------------------------------------------------------------------
# Width-windowed layer (the ConstrainedCost DP, DynamicSplitter.jl:206-258 with w = VertexCount()):
#     cst[r] = min over max(0, r - w) <= p <= r of Wprev[p] + f(p, r),   the LARGEST p on ties
# (the windows [j'_lo[k-1], j'_hi[k-1]] of the previous layer enter through Wprev: a huge value outside them).
#
# Geometry.  s = floor(log2 w), S = 2^s <= w < 2S.  The candidates of row r = i S + t below the diagonal split into
#   standard planes  b < s, bit b of r SET  : the Fenwick block [r_b - 2^b, r_b)            (all inside [i S, r))
#   the common plane b = s                    : [rho - w + S - 1, rho),            rho = i S   (the same for the whole row block)
#   mirrored planes  b < s, bit b of r CLEAR : [rho + 2^b - 1 - w, rho + 2^(b+1) - 1 - w), rho = r with the bits <= b cleared
# -- the mirrored blocks of a row tile [r - w, i S - w + S - 1) by the CLEAR bits of t, largest bit rightmost, exactly as the
# standard ones tile [i S, r) by its set bits.  In every plane b the rows sharing a block are the aligned block of 2^b rows
# containing r, so each plane is again a family of FULL rectangles (rows x columns) on which W[p] + f(p, r) is
# inverse-Monge: the same divide and conquer by tau = ctz(r) applies, with the same count path (anchor at the left tree
# neighbour, right steps, left steps).  A row is the head of its rectangle in every plane b <= min(ctz(r), s) (round A).
def window_block(r, b, s, w):
    """(cs, ce, kind) of row r in plane b <= s: candidates [cs, ce), clamped at column 0"""
    S = 1 << s
    lo = (r >> b) << b
    if b == s:
        cs, ce, kind = lo - w + S - 1, lo, "common"
    elif (r >> b) & 1:
        cs, ce, kind = lo - (1 << b), lo, "standard"
    else:
        cs, ce, kind = lo + (1 << b) - 1 - w, lo + (2 << b) - 1 - w, "mirrored"
    return max(cs, 0), max(ce, 0), kind


def layer_windowed(A, Wprev, fcost, prev, nxt, w, first=None, last=None, blocks=False):
    n = A.n
    pos = A.colptr - 1
    assert w >= 1
    s = int(w).bit_length() - 1
    npl = s + 1
    hyper = first is not None
    zero = np.array([0, 0]) if hyper else 0
    opt = np.full((n + 1, npl), -1, dtype=np.int64)
    nnopt = np.zeros((n + 1, npl), dtype=object)
    val = np.full((n + 1, npl), np.inf, dtype=object)

    def right_delta(c, thr):
        d = int(np.sum(prev[pos[c]:pos[c + 1]] < thr))
        if hyper:
            return np.array([d, int(np.sum((last == c) & (first >= thr)))])
        return d

    def left_delta(p, r):
        d = int(np.sum(nxt[pos[p]:pos[p + 1]] >= r))
        if hyper:
            return np.array([d, int(np.sum((first == p) & (last < r)))])
        return d

    def nets_direct(p, r):          # anchors of the mirrored head tasks (layer-independent; the device caches them)
        d = sum(int(np.sum(prev[pos[c]:pos[c + 1]] < p)) for c in range(p, r))
        if hyper:
            return np.array([d, int(np.sum((first >= p) & (last < r) & (first >= 0)))])
        return d

    def run_task(r, b, B, a, S0, cols_right, virtual):
        nn = S0 + 0
        best = None
        for c in cols_right:
            nn = nn + right_delta(c, B)
        if not virtual:
            best = (Wprev[B] + fcost(B, r, nn), B, nn)
        for p in range(B - 1, a - 1, -1):
            nn = nn + left_delta(p, r)
            v = Wprev[p] + fcost(p, r, nn)
            if best is None or v < best[0]:
                best = (v, p, nn)
        val[r, b], opt[r, b], nnopt[r, b] = best

    # round A: row r heads its rectangle in the planes b <= min(ctz(r), s)
    for r in range(1, n + 1):
        c = (r & -r).bit_length() - 1
        for b in range(min(c, s) + 1):
            cs, ce, kind = window_block(r, b, s, w)
            if ce <= cs:
                continue
            run_task(r, b, ce, cs, zero if ce == r else nets_direct(ce, r), [], True)
    for tau in range(s - 1, -1, -1):
        for r in range(1 << tau, n + 1, 1 << (tau + 1)):
            for b in range(tau + 1, npl):
                cs, ce, kind = window_block(r, b, s, w)
                if ce <= cs:
                    continue
                rL, rR = r - (1 << tau), r + (1 << tau)
                rect_hi = ((r >> b) + 1) << b
                B = opt[rL, b]
                a = opt[rR, b] if (rR < rect_hi and rR <= n) else cs
                run_task(r, b, B, a, nnopt[rL, b], range(rL, r), False)
    cst = np.zeros(n + 1, dtype=object)
    ptr = np.zeros(n + 1, dtype=np.int64)
    for r in range(n + 1):
        bv, bp = Wprev[r] + fcost(r, r, zero), r
        # candidates from the right: standard planes by ascending bit, the common plane, mirrored planes by descending bit
        order = [b for b in range(s) if (r >> b) & 1] + [s] + [b for b in range(s - 1, -1, -1) if not (r >> b) & 1]
        for b in order:
            cs, ce, _ = window_block(r, b, s, w)
            if r >= 1 and ce > cs and val[r, b] < bv:
                bv, bp = val[r, b], opt[r, b]
        cst[r], ptr[r] = bv, bp
    if blocks:
        return cst, ptr, opt.T.copy()
    return cst, ptr
