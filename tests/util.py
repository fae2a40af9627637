"""Shared test helpers: seeded CSC generators, the reference's embedded matrices
(tests/golden/matrices.json, extracted from test/matrices.jl by
tools/extract_reference_matrices.py), and brute-force definitions re-expressed from the
reference's own test files (test_SparsePrefixMatrices.jl:14, test_SparseColorArrays.jl:1-11)."""
import json
import os

import numpy as np

import cpamd

cp = cpamd.load()
HERE = os.path.dirname(os.path.abspath(__file__))


def golden_matrices():
    d = json.load(open(os.path.join(HERE, "golden", "matrices.json")))
    return {k: cp.SparseMatrixCSC(v["m"], v["n"], v["colptr"], v["rowval"]) for k, v in d.items()}


def sprand(m, n, p, rng):
    """Pattern of sprand(m, n, p): each entry present independently w.p. p."""
    mask = rng.random((n, m)) < p               # [col, row]
    cols, rows = np.nonzero(mask)
    colptr = np.concatenate([[1], 1 + np.cumsum(np.bincount(cols, minlength=n))]).astype(np.int64)
    return cp.SparseMatrixCSC(m, n, colptr, rows.astype(np.int64) + 1)


def suitesparse_shaped(n, mean_deg, seed, m=None):
    """SURVEY.md 8(d) `suitesparse_shaped` family at test sizes: lognormal column degrees,
    80 % of rows drawn from N(j, (n/100)^2) (banded locality), 20 % uniform; rows sorted, deduplicated."""
    rng = np.random.default_rng(seed)
    m = m or n
    deg = np.clip(np.round(rng.lognormal(np.log(mean_deg) - 0.5, 1.0, n)), 1, max(1, min(m, 10000))).astype(np.int64)
    cols = np.repeat(np.arange(n, dtype=np.int64), deg)
    local = rng.random(cols.size) < 0.8
    sigma = max(1.0, n / 100.0)
    rows = np.where(local, np.round(cols * (m / n) + rng.normal(0, sigma, cols.size)), rng.integers(0, m, cols.size))
    rows = np.clip(rows, 0, m - 1).astype(np.int64)
    key = np.unique(cols * m + rows)
    cols, rows = key // m, key % m
    colptr = np.concatenate([[1], 1 + np.cumsum(np.bincount(cols, minlength=n))]).astype(np.int64)
    return cp.SparseMatrixCSC(m, n, colptr, rows + 1)


def banded(n, half_bw, fill, seed):
    """SURVEY.md 8(d) `banded`: half-bandwidth, fill inside band, full diagonal."""
    rng = np.random.default_rng(seed)
    cols, rows = [], []
    for d in range(-half_bw, half_bw + 1):
        j = np.arange(max(0, -d), min(n, n - d))
        keep = np.ones(j.size, bool) if d == 0 else rng.random(j.size) < fill
        cols.append(j[keep]); rows.append(j[keep] + d)
    cols = np.concatenate(cols); rows = np.concatenate(rows)
    key = np.unique(cols * n + rows)
    cols, rows = key // n, key % n
    colptr = np.concatenate([[1], 1 + np.cumsum(np.bincount(cols, minlength=n))]).astype(np.int64)
    return cp.SparseMatrixCSC(n, n, colptr, rows + 1)


def dense_mask(A):
    D = np.zeros((A.m, A.n), dtype=bool)
    for j in range(A.n):
        D[A.rowval[A.colptr[j] - 1:A.colptr[j + 1] - 1] - 1, j] = True
    return D


# ---- brute-force definitions (1-based arguments) ----
def ref_dominancecount(D, i, j):                     # sum(A[1:i-1, 1:j-1] .!= 0)
    return int(D[:i - 1, :j - 1].sum())


def ref_netcount(D, j, jp):                          # distinct rows in columns j : j'-1
    return int(D[:, j - 1:jp - 1].any(axis=1).sum())


def ref_selfnetcount(D, j, jp):                      # rows whose support lies inside j : j'-1
    inside = D[:, j - 1:jp - 1].any(axis=1)
    outside = D[:, :j - 1].any(axis=1) | D[:, jp - 1:].any(axis=1)
    return int((inside & ~outside).sum())


def net_table(D):
    """nets[j, j'] for all 1 <= j <= j' <= n+1 (index 0 unused)."""
    n = D.shape[1]
    T = np.zeros((n + 2, n + 2), dtype=np.int64)
    for j in range(1, n + 2):
        seen = np.zeros(D.shape[0], bool)
        for jp in range(j + 1, n + 2):
            seen |= D[:, jp - 2]
            T[j, jp] = seen.sum()
    return T


def selfnet_table(D):
    n = D.shape[1]
    T = np.zeros((n + 2, n + 2), dtype=np.int64)
    first = np.where(D.any(axis=1), D.argmax(axis=1) + 1, 0)
    last = np.where(D.any(axis=1), n - D[:, ::-1].argmax(axis=1), 0)
    for j in range(1, n + 2):
        for jp in range(j, n + 2):
            T[j, jp] = int(((first >= j) & (last < jp) & (first > 0)).sum())
    return T
