"""Value types of the partition_stripe / pack_stripe path.

Host-side mirror of the reference's Julia types (all index values stay 1-based, exactly
as Julia stores them, so split vectors compare element-for-element):

  SparseMatrixCSC (pattern only; nzval is never read on the path)
  SplitPartition / DomainPartition / MapPartition      /root/reference/src/Partitions.jl:1-84
  NoHint / RandomHint / SparseHint / StepHint          /root/reference/src/ChainPartitioners.jl:172-176
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "SparseMatrixCSC", "SplitPartition", "DomainPartition", "MapPartition",
    "NoHint", "RandomHint", "SparseHint", "StepHint",
]


class _Hint:
    code = 0

    def __repr__(self):
        return type(self).__name__ + "()"


class NoHint(_Hint):
    code = 0


class RandomHint(_Hint):
    code = 1


class SparseHint(_Hint):
    code = 2


class StepHint(_Hint):
    code = 3


class SparseMatrixCSC:
    """Sparsity pattern in compressed-column form, 1-based like Julia's SparseMatrixCSC.

    "Partition the rows of a CSR matrix" is the same computation on the same two arrays
    (CSR row pointer == colptr, CSR column index == rowval of the transpose).
    """

    __slots__ = ("m", "n", "colptr", "rowval", "__weakref__")

    def __init__(self, m, n, colptr, rowval):
        self.m = int(m)
        self.n = int(n)
        self.colptr = np.ascontiguousarray(colptr, dtype=np.int64)
        self.rowval = np.ascontiguousarray(rowval, dtype=np.int64)
        if self.colptr.shape != (self.n + 1,):
            raise ValueError("colptr must have n+1 entries")
        if self.n >= 0 and (self.colptr[0] != 1 or self.colptr[-1] != self.rowval.size + 1):
            raise ValueError("colptr must be 1-based with colptr[end] == nnz+1")

    @property
    def nnz(self):
        return int(self.rowval.size)

    @property
    def shape(self):
        return (self.m, self.n)

    @classmethod
    def from_scipy(cls, S):
        S = S.tocsc()
        S.sort_indices()
        S.sum_duplicates()
        return cls(S.shape[0], S.shape[1], S.indptr.astype(np.int64) + 1, S.indices.astype(np.int64) + 1)

    def to_scipy(self):
        import scipy.sparse as sp
        return sp.csc_matrix((np.ones(self.nnz, dtype=np.int8), self.rowval - 1, self.colptr - 1), shape=self.shape)

    def __repr__(self):
        return f"SparseMatrixCSC({self.m}x{self.n}, nnz={self.nnz})"


class SplitPartition:
    """K contiguous parts; part k = columns spl[k] : spl[k+1]-1 (1-based, len K+1)."""

    __slots__ = ("K", "spl")

    def __init__(self, K, spl):
        self.K = int(K)
        self.spl = np.ascontiguousarray(spl, dtype=np.int64)

    def __len__(self):
        return self.K

    def __eq__(self, other):  # Partitions.jl:25-27: equality is on spl only
        return isinstance(other, SplitPartition) and np.array_equal(self.spl, other.spl)

    def __repr__(self):
        return f"SplitPartition({self.K}, {self.spl.tolist()})"


class MapPartition:
    __slots__ = ("K", "asg")

    def __init__(self, K, asg):
        self.K = int(K)
        self.asg = np.ascontiguousarray(asg, dtype=np.int64)

    def __len__(self):
        return self.K

    def __eq__(self, other):
        return isinstance(other, MapPartition) and self.K == other.K and np.array_equal(self.asg, other.asg)


class DomainPartition:
    __slots__ = ("K", "prm", "spl")

    def __init__(self, K, prm, spl):
        self.K = int(K)
        self.prm = np.ascontiguousarray(prm, dtype=np.int64)
        self.spl = np.ascontiguousarray(spl, dtype=np.int64)

    def __len__(self):
        return self.K

    def __eq__(self, other):
        return (isinstance(other, DomainPartition) and np.array_equal(self.prm, other.prm)
                and np.array_equal(self.spl, other.spl))


def to_map(P) -> MapPartition:
    """convert(MapPartition, P)  Partitions.jl:62-84"""
    if isinstance(P, MapPartition):
        return P
    if isinstance(P, SplitPartition):
        asg = np.repeat(np.arange(1, P.K + 1, dtype=np.int64), np.diff(P.spl))
        return MapPartition(P.K, asg)
    if isinstance(P, DomainPartition):
        asg = np.empty(P.spl[-1] - 1, dtype=np.int64)
        for k in range(1, P.K + 1):
            asg[P.prm[P.spl[k - 1] - 1:P.spl[k] - 1] - 1] = k
        return MapPartition(P.K, asg)
    raise TypeError(type(P))


def to_domain(P) -> DomainPartition:
    """convert(DomainPartition, P)  Partitions.jl:37-60"""
    if isinstance(P, DomainPartition):
        return P
    if isinstance(P, SplitPartition):
        return DomainPartition(P.K, np.arange(1, P.spl[-1], dtype=np.int64), P.spl)
    if isinstance(P, MapPartition):
        cnt = np.bincount(P.asg - 1, minlength=P.K)
        spl = np.concatenate([[1], 1 + np.cumsum(cnt)]).astype(np.int64)
        prm = np.argsort(P.asg, kind="stable").astype(np.int64) + 1
        return DomainPartition(P.K, prm, spl)
    raise TypeError(type(P))
