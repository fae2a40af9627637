"""Seeded synthetic sparsity patterns from ONE counter-based generator, so that bench.py (torch, on the
device) and the tests (numpy, on the host) build the SAME matrix for the same (family, size, seed) on any torch build:

  * randomness = SplitMix64 of (seed, stream, index) -- pure 64-bit integer arithmetic, written once for numpy (uint64) and once
    for torch (int64 with the same bit patterns);
  * the two non-uniform laws (lognormal column degrees, normal row offsets) come from 2^16-entry quantile tables computed on the
    host in float64 and applied with integer interpolation, so no transcendental function runs on the device;
  * base seed 0xDEADBEEF (the reference's test seed, test/runtests.jl:13) + the config index is the caller's convention.

Families: `suitesparse_shaped` (lognormal degrees, sigma = 1, clipped to [1, 10^4]; 80 % of a column's rows ~ N(j, (n/100)^2),
20 % uniform; rows sorted + deduplicated; optionally trimmed to exactly N nonzeros) and `banded` (half bandwidth, fill inside
the band, full diagonal).  All results are 1-based int64 colptr / rowval, as Julia stores a SparseMatrixCSC pattern.
"""
import numpy as np

QBITS = 16
QN = 1 << QBITS
_M64 = (1 << 64) - 1


# ---------------------------------------------------------------- SplitMix64, two spellings of the same function
def _sm64_np(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    z = x
    z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return z ^ (z >> np.uint64(31))


def _i64(v):
    """python int (any 64-bit pattern) -> the signed value with the same bits"""
    v &= _M64
    return v - (1 << 64) if v >> 63 else v


def _lsr_t(x, s):
    return (x >> s) & ((1 << (64 - s)) - 1)


def _sm64_t(x):
    x = x + _i64(0x9E3779B97F4A7C15)
    z = x
    z = (z ^ _lsr_t(z, 30)) * _i64(0xBF58476D1CE4E5B9)
    z = (z ^ _lsr_t(z, 27)) * _i64(0x94D049BB133111EB)
    return z ^ _lsr_t(z, 31)


def _stream_key(seed, stream):
    """scalar: SplitMix64(seed + stream * golden) as a python int"""
    x = (seed + stream * 0xD1342543DE82EF95) & _M64
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def hash_np(seed, stream, idx):
    """uint64 hash of every index (numpy int64/uint64 array)"""
    with np.errstate(over="ignore"):
        return _sm64_np(np.uint64(_stream_key(seed, stream)) + idx.astype(np.uint64) * np.uint64(0x2545F4914F6CDD1D))


def hash_t(seed, stream, idx):
    """the same bits as int64 (torch tensor)"""
    return _sm64_t(idx * _i64(0x2545F4914F6CDD1D) + _i64(_stream_key(seed, stream)))


# ---------------------------------------------------------------- quantile tables (host, float64)
def _norm_quantiles():
    """z_i = Phi^-1((i + 1/2) / QN), i = 0 .. QN (one extra entry for interpolation): Acklam's rational approximation"""
    p = (np.arange(QN + 1, dtype=np.float64) + 0.5) / (QN + 1)
    a = [-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02, 1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00]
    b = [-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01, -1.328068155288572e+01]
    c = [-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00, -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00]
    d = [7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00]
    z = np.zeros_like(p)
    lo, hi = p < 0.02425, p > 1 - 0.02425
    mid = ~(lo | hi)
    q = np.sqrt(-2 * np.log(p[lo]))
    z[lo] = (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1)
    q = np.sqrt(-2 * np.log(1 - p[hi]))
    z[hi] = -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1)
    q = p[mid] - 0.5
    r = q * q
    z[mid] = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q / (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1)
    return z


_Z = None


def tables(n, mean_deg, m):
    """(degree table int64[QN], offset table int64[QN + 1]) for lognormal(mu, 1) degrees with the given mean and N(0, (n/100)^2)
    row offsets"""
    global _Z
    if _Z is None:
        _Z = _norm_quantiles()
    mu = np.log(mean_deg) - 0.5                        # lognormal mean = exp(mu + 1/2)
    deg = np.clip(np.rint(np.exp(mu + _Z[:QN])), 1, max(1, min(m, 10000))).astype(np.int64)
    sigma = max(1.0, n / 100.0)
    off = np.rint(_Z * sigma).astype(np.int64)
    return deg, off


# ---------------------------------------------------------------- suitesparse_shaped
def suitesparse_shaped_np(n, mean_deg, seed, m=None, nnz=None):
    """numpy (host) build; nnz: trim to exactly this many nonzeros (the degree law then gets 8 % head-room)"""
    m = m or n
    degtab, offtab = tables(n, mean_deg * (1.08 if nnz else 1.0), m)
    j = np.arange(n, dtype=np.int64)
    deg = degtab[(hash_np(seed, 1, j) >> np.uint64(64 - QBITS)).astype(np.int64)]
    cols = np.repeat(j, deg)
    e = np.arange(cols.size, dtype=np.int64)
    local = (hash_np(seed, 2, e) >> np.uint64(48)).astype(np.int64) < 52429          # 0.8 * 2^16
    h3 = hash_np(seed, 3, e)
    qi = (h3 >> np.uint64(64 - QBITS)).astype(np.int64)
    fr = ((h3 >> np.uint64(32)) & np.uint64(0xFFFF)).astype(np.int64)
    off = offtab[qi] + (((offtab[qi + 1] - offtab[qi]) * fr) >> 16)
    centre = cols if m == n else (cols * m) // n
    uni = ((hash_np(seed, 4, e) >> np.uint64(1)) % np.uint64(m)).astype(np.int64)
    rows = np.clip(np.where(local, centre + off, uni), 0, m - 1)
    key = np.unique(cols * m + rows)
    if nnz is not None and key.size > nnz:
        hk = (hash_np(seed, 5, key) >> np.uint64(1)).astype(np.int64)
        thr = np.partition(hk, key.size - nnz - 1)[key.size - nnz - 1]              # drop the (size - nnz) smallest hashes
        key = key[hk > thr]
        assert key.size == nnz
    cols, rows = key // m, key % m
    colptr = np.concatenate([[1], 1 + np.cumsum(np.bincount(cols, minlength=n))]).astype(np.int64)
    return m, n, colptr, (rows + 1).astype(np.int64)


def suitesparse_shaped_t(n, mean_deg, seed, device, m=None, nnz=None):
    """torch build (any device): the same matrix as suitesparse_shaped_np"""
    import torch
    m = m or n
    degtab, offtab = tables(n, mean_deg * (1.08 if nnz else 1.0), m)
    degtab = torch.from_numpy(degtab).to(device); offtab = torch.from_numpy(offtab).to(device)
    j = torch.arange(n, dtype=torch.int64, device=device)
    deg = degtab[_lsr_t(hash_t(seed, 1, j), 64 - QBITS)]
    cols = torch.repeat_interleave(j, deg)
    del j, deg
    e = torch.arange(cols.numel(), dtype=torch.int64, device=device)
    local = _lsr_t(hash_t(seed, 2, e), 48) < 52429
    h3 = hash_t(seed, 3, e)
    qi = _lsr_t(h3, 64 - QBITS)
    fr = _lsr_t(h3, 32) & 0xFFFF
    lo = offtab[qi]
    off = lo + (((offtab[qi + 1] - lo) * fr) >> 16)
    del h3, qi, fr, lo
    centre = cols if m == n else torch.div(cols * m, n, rounding_mode="floor")
    uni = _lsr_t(hash_t(seed, 4, e), 1) % m
    del e
    rows = torch.where(local, centre + off, uni).clamp_(0, m - 1)
    del local, off, uni, centre
    key = torch.unique(cols * m + rows)
    del cols, rows
    if nnz is not None and key.numel() > nnz:
        hk = _lsr_t(hash_t(seed, 5, key), 1)
        k = key.numel() - nnz
        thr = torch.kthvalue(hk, k).values if device == "cpu" or str(device) == "cpu" else torch.sort(hk).values[k - 1]
        key = key[hk > thr]
        assert key.numel() == nnz
        del hk
    cols = torch.div(key, m, rounding_mode="floor")
    rowval = (key - cols * m) + 1
    cnt = torch.bincount(cols, minlength=n)
    colptr = torch.cat([torch.ones(1, dtype=torch.int64, device=device), 1 + torch.cumsum(cnt, 0)])
    return m, n, colptr.contiguous(), rowval.contiguous()


# ---------------------------------------------------------------- banded
def banded_np(n, half_bw, fill, seed):
    thr = int(fill * 65536)
    cols, rows = [], []
    for d in range(-half_bw, half_bw + 1):
        j = np.arange(max(0, -d), min(n, n - d), dtype=np.int64)
        keep = np.ones(j.size, bool) if d == 0 else (hash_np(seed, 100 + d + half_bw, j) >> np.uint64(48)).astype(np.int64) < thr
        cols.append(j[keep]); rows.append(j[keep] + d)
    cols = np.concatenate(cols); rows = np.concatenate(rows)
    key = np.unique(cols * n + rows)
    cols, rows = key // n, key % n
    colptr = np.concatenate([[1], 1 + np.cumsum(np.bincount(cols, minlength=n))]).astype(np.int64)
    return n, n, colptr, (rows + 1).astype(np.int64)


def banded_t(n, half_bw, fill, seed, device):
    import torch
    thr = int(fill * 65536)
    keys = []
    for d in range(-half_bw, half_bw + 1):
        j = torch.arange(max(0, -d), min(n, n - d), dtype=torch.int64, device=device)
        if d != 0:
            j = j[_lsr_t(hash_t(seed, 100 + d + half_bw, j), 48) < thr]
        keys.append(j * n + (j + d))
    key = torch.unique(torch.cat(keys))
    cols = torch.div(key, n, rounding_mode="floor")
    colptr = torch.cat([torch.ones(1, dtype=torch.int64, device=device), 1 + torch.cumsum(torch.bincount(cols, minlength=n), 0)])
    return n, n, colptr.contiguous(), ((key - cols * n) + 1).contiguous()
