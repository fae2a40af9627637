"""ctypes binding of the product C-ABI library libchainpart.so (HIP, gfx950).

No fallback of any kind lives here: if the shared object is missing, or no HIP device is
visible, construction raises.  The symbols bound are exactly those of include/chainpart.h.
"""
from __future__ import annotations

import ctypes as C
import os
import weakref

import numpy as np

from . import models as M

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CP_LIB_PATH") or os.path.join(_HERE, "libchainpart.so")   # CP_LIB_PATH: A/B builds only

SYMBOLS = [
    "cp_last_error", "cp_version", "cp_device_count", "cp_csr_create", "cp_csr_create_device", "cp_csr_destroy",
    "cp_csr_reset_cache", "cp_count_build", "cp_count_query", "cp_count_destroy", "cp_link_array", "cp_partwise", "cp_domsum_build", "cp_rook_build", "cp_wsum_query", "cp_wsum_destroy",
    "cp_oracle_eval", "cp_oracle_step", "cp_bound_stripe", "cp_objective", "cp_partition_dynamic", "cp_pack_dynamic",
    "cp_partition_bisect_cost", "cp_partition_bisect_cost_batch", "cp_pack_convex", "cp_pack_convex_batch", "cp_partition_convex", "cp_partition_equi", "cp_pack_equi",
    "cp_dynamic_tables", "cp_dynamic_tables_constrained", "cp_dynamic_tables_constrained_combine", "cp_set_stream", "cp_reset_stream", "cp_get_stat", "cp_set_option", "cp_prof_enable", "cp_prof_reset", "cp_prof_get",
    "cp_dp_begin", "cp_dp_layer", "cp_dp_ptr_at", "cp_dp_destroy", "cp_dp_ptr_row", "cp_dp_block_tables", "cp_dp_set_window", "cp_dp_set_rows",
    "cp_partition_bisect_index", "cp_partition_lazy_bisect_cost", "cp_pack_concave", "cp_partition_concave",
    "cp_adjoint", "cp_csr_download", "cp_bound_stripe_pi", "cp_partition_bisect_cost_pi", "cp_partition_bisect_index_pi",
]

_lib = None


def load_library():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: build it with __graft_entry__.build() "
                               "(make -C chainpartitioners.jl_amd/csrc); there is no CPU fallback")
        _lib = C.CDLL(LIB_PATH)
        _lib.cp_last_error.restype = C.c_char_p
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i64(x):
    return C.c_int64(int(x))


_COUNT_KIND = {"dom": 0, "net": 1, "selfnet": 2}


class HipBackend:
    """backend interface of api.py over libchainpart.so (one HIP device per process)."""
    name = "hip"

    def __init__(self, device=0):
        self.lib = load_library()
        if self.lib.cp_device_count() <= 0:
            raise RuntimeError("libchainpart: no HIP device visible (the product path has no CPU fallback)")
        self.device = int(device)
        self._handles = {}

    def last_error(self):
        return (self.lib.cp_last_error() or b"").decode()

    # ---- CSR residency: one device handle per host matrix object, dropped with it
    def csr(self, A):
        key = id(A)
        ent = self._handles.get(key)
        if ent is not None and ent[0]() is A:
            return ent[1]
        h = C.c_void_p()
        rc = self.lib.cp_csr_create(_i64(A.m), _i64(A.n), _i64(A.nnz), _p(A.colptr), _p(A.rowval), C.c_int32(self.device), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"cp_csr_create failed ({rc}): {self.last_error()}")
        lib = self.lib
        handles = self._handles

        def _drop(_ref, key=key, h=h):
            handles.pop(key, None)
            lib.cp_csr_destroy(h)
        self._handles[key] = (weakref.ref(A, _drop), h)
        return h

    def csr_from_device(self, m, n, N, colptr_dev_ptr, rowval_dev_ptr):
        """Handle over colptr/rowval already resident in HBM (1-based int64 device arrays)."""
        h = C.c_void_p()
        rc = self.lib.cp_csr_create_device(_i64(m), _i64(n), _i64(N), C.c_void_p(colptr_dev_ptr), C.c_void_p(rowval_dev_ptr),
                                           C.c_int32(self.device), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"cp_csr_create_device failed ({rc}): {self.last_error()}")
        return h

    def csr_destroy(self, h):
        self.lib.cp_csr_destroy(h)

    def adjoint(self, A):
        """adjointpattern(A) on the device; the result's device handle stays registered, so a following
        partition_stripe(adj_A, ...) needs no upload."""
        from .types import SparseMatrixCSC
        t = C.c_void_p()
        rc = self.lib.cp_adjoint(self._h(A), C.byref(t))
        if rc != 0:
            raise RuntimeError(f"cp_adjoint failed ({rc}): {self.last_error()}")
        dims = np.zeros(3, dtype=np.int64)
        self.lib.cp_csr_download(t, _p(dims), None, None)
        m, n, N = (int(x) for x in dims)
        colptr = np.zeros(n + 1, dtype=np.int64); rowval = np.zeros(max(N, 1), dtype=np.int64)
        rc = self.lib.cp_csr_download(t, _p(dims), _p(colptr), _p(rowval))
        if rc != 0:
            self.lib.cp_csr_destroy(t)
            raise RuntimeError(f"cp_csr_download failed ({rc}): {self.last_error()}")
        T = SparseMatrixCSC(m, n, colptr, rowval[:N])
        lib, handles, key = self.lib, self._handles, id(T)

        def _drop(_ref, key=key, t=t):
            handles.pop(key, None)
            lib.cp_csr_destroy(t)
        self._handles[key] = (weakref.ref(T, _drop), t)
        return T

    def reset_cache(self, A_or_handle):
        h = A_or_handle if isinstance(A_or_handle, C.c_void_p) else self.csr(A_or_handle)
        return self.lib.cp_csr_reset_cache(h)

    def set_stream(self, A_or_handle, stream_ptr):
        h = A_or_handle if isinstance(A_or_handle, C.c_void_p) else self.csr(A_or_handle)
        return self.lib.cp_set_stream(h, C.c_void_p(stream_ptr))

    def reset_stream(self, A_or_handle):
        h = A_or_handle if isinstance(A_or_handle, C.c_void_p) else self.csr(A_or_handle)
        return self.lib.cp_reset_stream(h)

    def get_stat(self, name):
        out = C.c_int64()
        rc = self.lib.cp_get_stat(name.encode(), C.byref(out))
        if rc != 0:
            raise KeyError(name)
        return out.value

    def set_option(self, name, value):
        return self.lib.cp_set_option(name.encode(), _i64(value))

    def _h(self, A):
        return A if isinstance(A, C.c_void_p) else self.csr(A)

    # ---- partitioners
    def partition_dynamic(self, A, K, combine, order, mm, rp, wm, wi, wf, spl):
        return self.lib.cp_partition_dynamic(self._h(A), _i64(K), C.c_int32(combine), C.c_int32(order), mm.ptr,
                                             C.byref(rp) if rp is not None else None,
                                             wm.ptr if wm is not None else None, _i64(wi), C.c_double(wf), _p(spl))

    def pack_dynamic(self, A, mm, rp, wm, wi, wf, spl, Kout):
        return self.lib.cp_pack_dynamic(self._h(A), mm.ptr, C.byref(rp) if rp is not None else None,
                                        wm.ptr if wm is not None else None, _i64(wi), C.c_double(wf), _p(spl), _p(Kout))

    def partition_bisect_cost(self, A, K, mm, eps, flip, spl, rp=None):
        return self.lib.cp_partition_bisect_cost_pi(self._h(A), _i64(K), mm.ptr, C.byref(rp) if rp is not None else None,
                                                    C.c_double(eps), C.c_int32(flip), _p(spl))

    def partition_bisect_cost_batch(self, A, Ks, mms, epss, flips):
        """B requests on one pattern in one launch -> (rc, [split vector of request b])"""
        B = len(Ks)
        arr = (M.cp_model_t * B)(*[m.struct for m in mms])          # (struct copies; the buffers they point into live in `mms`)
        Kv = np.ascontiguousarray(Ks, dtype=np.int64)
        ev = np.ascontiguousarray(epss, dtype=np.float64)
        fv = np.ascontiguousarray(flips, dtype=np.int32)
        ld = int(Kv.max()) + 1
        out = np.zeros((B, ld), dtype=np.int64)
        rc = self.lib.cp_partition_bisect_cost_batch(self._h(A), _i64(B), _p(Kv), arr, _p(ev), _p(fv), _i64(ld), _p(out))
        return rc, [out[b, :int(Kv[b]) + 1].copy() for b in range(B)]

    def partition_bisect_index(self, A, K, mm, flip, spl, rp=None):
        return self.lib.cp_partition_bisect_index_pi(self._h(A), _i64(K), mm.ptr, C.byref(rp) if rp is not None else None,
                                                     C.c_int32(flip), _p(spl))

    def partition_lazy_bisect_cost(self, A, K, mm, eps, spl):
        return self.lib.cp_partition_lazy_bisect_cost(self._h(A), _i64(K), mm.ptr, C.c_double(eps), _p(spl))

    def pack_convex(self, A, mm, rp, wm, wi, wf, spl, Kout):
        return self.lib.cp_pack_convex(self._h(A), mm.ptr, C.byref(rp) if rp is not None else None,
                                       wm.ptr if wm is not None else None, _i64(wi), C.c_double(wf), _p(spl), _p(Kout))

    def pack_convex_batch(self, A, mms, wmaxs, n):
        """B requests (model, w_max) on one pattern in one launch -> (rc, [chunk boundaries of request b])"""
        B = len(mms)
        arr = (M.cp_model_t * B)(*[m.struct for m in mms])
        wv = np.ascontiguousarray(wmaxs, dtype=np.int64)
        ld = int(n) + 1
        out = np.zeros((B, ld), dtype=np.int64)
        Kout = np.zeros(B, dtype=np.int64)
        rc = self.lib.cp_pack_convex_batch(self._h(A), _i64(B), arr, _p(wv), _i64(ld), _p(out), _p(Kout))
        return rc, [out[b, :int(Kout[b]) + 1].copy() for b in range(B)]

    def partition_convex(self, A, K, mm, rp, wm, wi, wf, spl):
        return self.lib.cp_partition_convex(self._h(A), _i64(K), mm.ptr, C.byref(rp) if rp is not None else None,
                                            wm.ptr if wm is not None else None, _i64(wi), C.c_double(wf), _p(spl))

    def pack_concave(self, A, mm, rp, wm, wi, wf, spl, Kout):
        return self.lib.cp_pack_concave(self._h(A), mm.ptr, C.byref(rp) if rp is not None else None,
                                        wm.ptr if wm is not None else None, _i64(wi), C.c_double(wf), _p(spl), _p(Kout))

    def partition_concave(self, A, K, mm, rp, wm, wi, wf, spl):
        return self.lib.cp_partition_concave(self._h(A), _i64(K), mm.ptr, C.byref(rp) if rp is not None else None,
                                             wm.ptr if wm is not None else None, _i64(wi), C.c_double(wf), _p(spl))

    # ---- oracles / scoring
    def oracle_eval(self, A, mm, rp, hint, j, jp, k, out):
        oi = out if out.dtype == np.int64 else None
        of = out if out.dtype == np.float64 else None
        return self.lib.cp_oracle_eval(self._h(A), mm.ptr, C.byref(rp) if rp is not None else None, C.c_int32(hint),
                                       _i64(j.size), _p(j), _p(jp), _p(k), _p(oi), _p(of))

    def oracle_step(self, A, mm, rp, mj, j, mjp, jp, k, out):
        oi = out if out.dtype == np.int64 else None
        of = out if out.dtype == np.float64 else None
        return self.lib.cp_oracle_step(self._h(A), mm.ptr, C.byref(rp) if rp is not None else None, _i64(j.size),
                                       _p(mj), _p(j), _p(mjp), _p(jp), _p(k), _p(oi), _p(of))

    def bound_stripe(self, A, K, mm):
        li, hi, lf, hf = C.c_int64(), C.c_int64(), C.c_double(), C.c_double()
        rc = self.lib.cp_bound_stripe(self._h(A), _i64(K), mm.ptr, C.byref(li), C.byref(hi), C.byref(lf), C.byref(hf))
        if mm.struct.dtype == M.CP_I64:
            return rc, li.value, hi.value
        return rc, lf.value, hf.value

    def bound_stripe_pi(self, A, K, rp, mm):
        li, hi, lf, hf = C.c_int64(), C.c_int64(), C.c_double(), C.c_double()
        rc = self.lib.cp_bound_stripe_pi(self._h(A), _i64(K), C.byref(rp), mm.ptr, C.byref(li), C.byref(hi), C.byref(lf), C.byref(hf))
        if mm.struct.dtype == M.CP_I64:
            return rc, li.value, hi.value
        return rc, lf.value, hf.value

    def objective(self, A, K, spl, mm, rp, g):
        oi, of = C.c_int64(), C.c_double()
        rc = self.lib.cp_objective(self._h(A), _i64(K), _p(spl), mm.ptr, C.byref(rp) if rp is not None else None,
                                   C.c_int32(g), C.byref(oi), C.byref(of))
        return rc, (oi.value if mm.struct.dtype == M.CP_I64 else of.value)

    def dynamic_tables(self, A, K, combine, mm, rp):
        ptr = np.zeros((K, A.n + 1), dtype=np.int64)
        cst = np.zeros((K, A.n + 1), dtype=np.int64 if mm.struct.dtype == M.CP_I64 else np.float64)
        rc = self.lib.cp_dynamic_tables(self._h(A), _i64(K), C.c_int32(combine), mm.ptr,
                                        C.byref(rp) if rp is not None else None, _p(ptr),
                                        _p(cst) if mm.struct.dtype == M.CP_I64 else None,
                                        _p(cst) if mm.struct.dtype == M.CP_F64 else None)
        return rc, ptr.T, cst.T

    def dynamic_tables_constrained(self, A, K, mm, wmax, combine=0, wm=None):
        """(rc, j'_lo[K], j'_hi[K], ptr[j', k], cst[j', k]) of Dynamic{Total,Bottleneck}Splitter(ConstrainedCost(f, w, wmax))
        (combine 0 = total, 1 = bottleneck; wm: the marshalled weight, None = VertexCount())"""
        ptr = np.zeros((K, A.n + 1), dtype=np.int64)
        cst = np.zeros((K, A.n + 1), dtype=np.int64 if mm.struct.dtype == M.CP_I64 else np.float64)
        lo = np.zeros(K, dtype=np.int64); hi = np.zeros(K, dtype=np.int64)
        rc = self.lib.cp_dynamic_tables_constrained_combine(self._h(A), _i64(K), C.c_int32(combine), mm.ptr, wm.ptr if wm is not None else None,
                                                            _i64(int(wmax)), C.c_double(float(wmax)), _p(lo), _p(hi), _p(ptr),
                                                            _p(cst) if mm.struct.dtype == M.CP_I64 else None,
                                                            _p(cst) if mm.struct.dtype == M.CP_F64 else None)
        return rc, lo, hi, ptr.T, cst.T

    # ---- counting structures
    def count_build(self, kind, A, hint):
        h = C.c_void_p()
        rc = self.lib.cp_count_build(self._h(A), C.c_int32(_COUNT_KIND[kind]), C.c_int32(hint), C.byref(h))
        if rc != 0:
            raise NotImplementedError(f"cp_count_build({kind}) -> {rc}: {self.last_error()}")
        return h

    def count_query(self, kind, h, a, b, out):
        return self.lib.cp_count_query(h, _i64(a.size), _p(a), _p(b), _p(out))

    def count_free(self, kind, h):
        self.lib.cp_count_destroy(h)

    # ---- weighted dominance (a13)
    def domsum_build(self, A, val):
        val = np.ascontiguousarray(val)
        dt = M.CP_F64 if val.dtype == np.float64 else M.CP_I64
        h = C.c_void_p()
        rc = self.lib.cp_domsum_build(self._h(A), C.c_int32(dt), _p(val), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"cp_domsum_build -> {rc}: {self.last_error()}")
        return h, dt

    def rook_build(self, N, idx, val=None):
        idx = np.ascontiguousarray(idx, dtype=np.int64)
        dt = M.CP_I64
        if val is not None:
            val = np.ascontiguousarray(val)
            dt = M.CP_F64 if val.dtype == np.float64 else M.CP_I64
        h = C.c_void_p()
        rc = self.lib.cp_rook_build(_i64(N), _p(idx), C.c_int32(dt), _p(val), C.c_int32(self.device), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"cp_rook_build -> {rc}: {self.last_error()}")
        return h, dt

    def wsum_query(self, h, dt, i, j, unsigned=False):
        i = np.ascontiguousarray(i, dtype=np.int64); j = np.ascontiguousarray(j, dtype=np.int64)
        cnt = np.zeros(i.shape, dtype=np.int64)
        sm = np.zeros(i.shape, dtype=np.float64 if dt == M.CP_F64 else (np.uint64 if unsigned else np.int64))
        rc = self.lib.cp_wsum_query(h, _i64(i.size), _p(i), _p(j), _p(cnt), _p(sm) if dt == M.CP_I64 else None, _p(sm) if dt == M.CP_F64 else None)
        if rc != 0:
            raise RuntimeError(f"cp_wsum_query -> {rc}: {self.last_error()}")
        return cnt, sm

    def wsum_free(self, h):
        self.lib.cp_wsum_destroy(h)

    def link_array(self, A):
        out = np.zeros(max(A.nnz, 1), dtype=np.int64)
        rc = self.lib.cp_link_array(self._h(A), _p(out))
        if rc != 0:
            raise RuntimeError(self.last_error())
        return out[:A.nnz]

    def partwise(self, A, K, asg):
        asg = np.ascontiguousarray(asg, dtype=np.int64)
        npr = C.c_int64()
        pios = np.zeros(K + 1, dtype=np.int64)
        prm = np.zeros(max(A.nnz, 1), dtype=np.int64)
        pos = np.zeros(A.nnz + 1, dtype=np.int64)
        idx = np.zeros(max(A.nnz, 1), dtype=np.int64)
        rc = self.lib.cp_partwise(self._h(A), _i64(K), _p(asg), C.byref(npr), _p(pios), _p(prm), _p(pos), _p(idx))
        if rc != 0:
            raise NotImplementedError(f"cp_partwise -> {rc}: {self.last_error()}")
        n = npr.value
        return n, pios, prm[:n].copy(), pos[:n + 1].copy(), idx[:A.nnz].copy()

    # ---- row-tiled DP (multi-GPU): see distributed.py
    def dp_begin(self, A, K, combine, order, mm, row_lo, row_hi):
        h = C.c_void_p()
        rc = self.lib.cp_dp_begin(self._h(A), _i64(K), C.c_int32(combine), C.c_int32(order), mm.ptr, _i64(row_lo), _i64(row_hi), C.byref(h))
        if rc != 0:
            raise RuntimeError(f"cp_dp_begin -> {rc}: {self.last_error()}")
        return h

    def dp_layer(self, dp, k, prev_ptr, cur_ptr):
        rc = self.lib.cp_dp_layer(dp, _i64(k), C.c_void_p(prev_ptr), C.c_void_p(cur_ptr))
        if rc != 0:
            raise RuntimeError(f"cp_dp_layer -> {rc}: {self.last_error()}")

    def dp_ptr_at(self, dp, k, jp):
        out = C.c_int64()
        rc = self.lib.cp_dp_ptr_at(dp, _i64(k), _i64(jp), C.byref(out))
        if rc != 0:
            raise RuntimeError(f"cp_dp_ptr_at -> {rc}: {self.last_error()}")
        return out.value

    def dp_ptr_row(self, dp, k, n):
        out = np.zeros(n + 1, dtype=np.int64)
        rc = self.lib.cp_dp_ptr_row(dp, _i64(k), _p(out))
        if rc != 0:
            raise RuntimeError(f"cp_dp_ptr_row -> {rc}: {self.last_error()}")
        return out

    def dp_block_tables(self, dp, n, hyper=False):
        """(nplanes, opt[b, r], nets[b, r], selfnets[b, r] | None) of the last layer computed through `dp`."""
        nb = C.c_int32()
        opt = np.zeros((31, n + 1), dtype=np.int64); nn = np.zeros((31, n + 1), dtype=np.int64)
        nl = np.zeros((31, n + 1), dtype=np.int64) if hyper else None
        rc = self.lib.cp_dp_block_tables(dp, C.byref(nb), _p(opt), _p(nn), _p(nl))
        if rc != 0:
            raise RuntimeError(f"cp_dp_block_tables -> {rc}: {self.last_error()}")
        # the library lays the planes out with stride n+1
        k = nb.value
        return k, opt.reshape(-1)[:k * (n + 1)].reshape(k, n + 1), nn.reshape(-1)[:k * (n + 1)].reshape(k, n + 1), \
            (nl.reshape(-1)[:k * (n + 1)].reshape(k, n + 1) if hyper else None)

    def dp_set_window(self, dp, wmax):
        rc = self.lib.cp_dp_set_window(dp, _i64(wmax))
        if rc != 0:
            raise RuntimeError(f"cp_dp_set_window -> {rc}")

    def dp_set_rows(self, dp, lo, hi):
        rc = self.lib.cp_dp_set_rows(dp, _i64(lo), _i64(hi))
        if rc != 0:
            raise RuntimeError(f"cp_dp_set_rows({lo}, {hi}) -> {rc}")

    def windowed_layer(self, A, mm, W, wmax, lo=None, hi=None):
        """one DP layer over injected previous costs W (numpy, n+1) with the width window wmax: (cst[r], ptr[r]) 0-based,
        rows [lo, hi) 1-based like cp_dp_begin (default: all)"""
        import torch
        n = A.n
        dev = torch.device("cuda", self.device)
        dp = self.dp_begin(A, 3, 0, 0, mm, lo or 1, hi or n + 2)
        try:
            rc = self.lib.cp_dp_set_window(dp, _i64(wmax))
            if rc != 0:
                raise RuntimeError(f"cp_dp_set_window -> {rc}")
            prev = torch.from_numpy(np.ascontiguousarray(W)).to(dev)
            cur = torch.zeros(n + 1, dtype=prev.dtype, device=dev)
            self.dp_layer(dp, 2, prev.data_ptr(), cur.data_ptr())
            return cur.cpu().numpy(), self.dp_ptr_row(dp, 2, n) - 1
        finally:
            self.dp_destroy(dp)

    def dp_destroy(self, dp):
        self.lib.cp_dp_destroy(dp)

    # ---- measurement
    def prof_enable(self, on=True):
        self.lib.cp_prof_enable(C.c_int32(1 if on else 0))

    def prof_reset(self):
        self.lib.cp_prof_reset()

    def prof_get(self):
        out = {}
        name = C.c_char_p()
        n, ms, by = C.c_int64(), C.c_double(), C.c_double()
        nslots = self.lib.cp_prof_get(C.c_int32(0), C.byref(name), C.byref(n), C.byref(ms), C.byref(by))
        for s in range(nslots):
            self.lib.cp_prof_get(C.c_int32(s), C.byref(name), C.byref(n), C.byref(ms), C.byref(by))
            out[name.value.decode()] = {"launches": n.value, "ms": ms.value, "units": by.value}
        return out
