"""ctypes binding of the TEST ORACLE (oracle/liborc.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never by the product package.  It implements the same
`backend` interface as chainpartitioners.jl_amd._lib.HipBackend so the host-side API
(api.partition_stripe etc.) can drive either on identical marshalled inputs.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force=False):
    so = os.path.join(_HERE, "liborc.so")
    srcs = [os.path.join(_HERE, f) for f in ("orc_counts.c", "orc_i64.c", "orc_f64.c", "orc_api.c",
                                             "orc_algos.inc", "orc.h")]
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "liborc.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        for name in ("orc_dom_query", "orc_dom_step", "orc_net_query", "orc_net_step", "orc_partwise",
                     "orc_pack_equi"):
            getattr(_LIB, name).restype = C.c_int64
        for name in ("orc_dom_build", "orc_netcount_build", "orc_selfnetcount_build"):
            getattr(_LIB, name).restype = C.c_void_p
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _i64(x):
    return C.c_int64(int(x))


class OracleBackend:
    name = "oracle"

    def last_error(self):
        return ""

    def _A(self, A):
        return (_i64(A.m), _i64(A.n), _i64(A.nnz), _p(A.colptr), _p(A.rowval))

    def partition_dynamic(self, A, K, combine, order, mm, rp, wm, wi, wf, spl):
        return lib().orc_partition_dynamic(*self._A(A), _i64(K), C.c_int32(combine), C.c_int32(order), mm.ptr,
                                           C.byref(rp) if rp is not None else None,
                                           wm.ptr if wm is not None else None, _i64(wi), C.c_double(wf), _p(spl))

    def pack_dynamic(self, A, mm, rp, wm, wi, wf, spl, Kout):
        return lib().orc_pack_dynamic(*self._A(A), mm.ptr, C.byref(rp) if rp is not None else None,
                                      wm.ptr if wm is not None else None, _i64(wi), C.c_double(wf), _p(spl), _p(Kout))

    def adjoint(self, A):
        """adjointpattern(A): CSC transpose of the pattern by a stable counting sort (util.jl:67-95)."""
        m, n = A.shape
        cols = np.repeat(np.arange(1, n + 1, dtype=np.int64), np.diff(A.colptr))
        order = np.argsort(A.rowval, kind="stable")
        cnt = np.bincount(A.rowval - 1, minlength=m)
        pos = np.concatenate([[1], 1 + np.cumsum(cnt)]).astype(np.int64)
        return type(A)(n, m, pos, cols[order])

    def partition_bisect_index(self, A, K, mm, flip, spl, rp=None):
        pr = np.zeros(1, dtype=np.int64)
        rc = lib().orc_partition_bisect_index(*self._A(A), _i64(K), mm.ptr, C.byref(rp) if rp is not None else None, C.c_int32(flip),
                                              _p(spl), _p(pr))
        self.last_probes = int(pr[0])
        return rc

    def partition_lazy_bisect_cost(self, A, K, mm, eps, spl):
        pr = np.zeros(1, dtype=np.int64)
        rc = lib().orc_partition_lazy_bisect_cost(*self._A(A), _i64(K), mm.ptr, C.c_double(eps), _p(spl), _p(pr))
        self.last_probes = int(pr[0])
        return rc

    def partition_bisect_cost(self, A, K, mm, eps, flip, spl, rp=None):
        pr = np.zeros(1, dtype=np.int64)
        rc = lib().orc_partition_bisect_cost(*self._A(A), _i64(K), mm.ptr, C.byref(rp) if rp is not None else None, C.c_double(eps),
                                             C.c_int32(flip), _p(spl), _p(pr))
        self.last_probes = int(pr[0])
        return rc

    def pack_convex(self, A, mm, rp, wm, wi, wf, spl, Kout):
        return lib().orc_pack_convex(*self._A(A), mm.ptr, C.byref(rp) if rp is not None else None,
                                     wm.ptr if wm is not None else None, _i64(wi), C.c_double(wf), _p(spl), _p(Kout))

    def partition_convex(self, A, K, mm, rp, wm, wi, wf, spl):
        return lib().orc_partition_convex(*self._A(A), _i64(K), mm.ptr, C.byref(rp) if rp is not None else None,
                                          wm.ptr if wm is not None else None, _i64(wi), C.c_double(wf), _p(spl))

    def pack_concave(self, A, mm, rp, wm, wi, wf, spl, Kout):
        return lib().orc_pack_concave(*self._A(A), mm.ptr, C.byref(rp) if rp is not None else None,
                                     wm.ptr if wm is not None else None, _i64(wi), C.c_double(wf), _p(spl), _p(Kout))

    def partition_concave(self, A, K, mm, rp, wm, wi, wf, spl):
        return lib().orc_partition_concave(*self._A(A), _i64(K), mm.ptr, C.byref(rp) if rp is not None else None,
                                          wm.ptr if wm is not None else None, _i64(wi), C.c_double(wf), _p(spl))

    def oracle_eval(self, A, mm, rp, hint, j, jp, k, out):
        oi = out if out.dtype == np.int64 else None
        of = out if out.dtype == np.float64 else None
        return lib().orc_oracle_eval(*self._A(A), mm.ptr, C.byref(rp) if rp is not None else None, C.c_int32(hint),
                                     _i64(j.size), _p(j), _p(jp), _p(k), _p(oi), _p(of))

    def oracle_step(self, A, mm, rp, mj, j, mjp, jp, k, out):
        oi = out if out.dtype == np.int64 else None
        of = out if out.dtype == np.float64 else None
        return lib().orc_oracle_step(*self._A(A), mm.ptr, C.byref(rp) if rp is not None else None, _i64(j.size),
                                     _p(mj), _p(j), _p(mjp), _p(jp), _p(k), _p(oi), _p(of))

    def bound_stripe(self, A, K, mm):
        li, hi, lf, hf = C.c_int64(), C.c_int64(), C.c_double(), C.c_double()
        rc = lib().orc_bound_stripe(*self._A(A), _i64(K), mm.ptr, C.byref(li), C.byref(hi), C.byref(lf), C.byref(hf))
        if mm.struct.dtype == 0:
            return rc, li.value, hi.value
        return rc, lf.value, hf.value

    def bound_stripe_pi(self, A, K, rp, mm):
        li, hi, lf, hf = C.c_int64(), C.c_int64(), C.c_double(), C.c_double()
        rc = lib().orc_bound_stripe_pi(*self._A(A), _i64(K), C.byref(rp), mm.ptr, C.byref(li), C.byref(hi), C.byref(lf), C.byref(hf))
        if mm.struct.dtype == 0:
            return rc, li.value, hi.value
        return rc, lf.value, hf.value

    def objective(self, A, K, spl, mm, rp, g):
        oi, of = C.c_int64(), C.c_double()
        rc = lib().orc_objective(*self._A(A), _i64(K), _p(spl), mm.ptr, C.byref(rp) if rp is not None else None,
                                 C.c_int32(g), C.byref(oi), C.byref(of))
        return rc, (oi.value if mm.struct.dtype == 0 else of.value)

    def dynamic_tables(self, A, K, combine, mm, rp):
        ptr = np.zeros((K, A.n + 1), dtype=np.int64)       # column-major (n+1) x K
        cst = np.zeros((K, A.n + 1), dtype=np.int64 if mm.struct.dtype == 0 else np.float64)
        rc = lib().orc_dynamic_tables(*self._A(A), _i64(K), C.c_int32(combine), mm.ptr,
                                      C.byref(rp) if rp is not None else None, _p(ptr),
                                      _p(cst) if mm.struct.dtype == 0 else None,
                                      _p(cst) if mm.struct.dtype == 1 else None)
        return rc, ptr.T, cst.T                               # [j', k] views

    def dynamic_tables_constrained(self, A, K, combine, mm, rp, wm, wi, wf):
        """(rc, j'_lo[K], j'_hi[K], ptr[j', k], cst[j', k]) of the ConstrainedCost splitter, tables densified"""
        ptr = np.zeros((K, A.n + 1), dtype=np.int64)
        cst = np.zeros((K, A.n + 1), dtype=np.int64 if mm.struct.dtype == 0 else np.float64)
        lo = np.zeros(K, dtype=np.int64); hi = np.zeros(K, dtype=np.int64)
        rc = lib().orc_dynamic_tables_constrained(*self._A(A), _i64(K), C.c_int32(combine), mm.ptr,
                                                  C.byref(rp) if rp is not None else None, wm.ptr, _i64(wi), C.c_double(wf),
                                                  _p(lo), _p(hi), _p(ptr),
                                                  _p(cst) if mm.struct.dtype == 0 else None, _p(cst) if mm.struct.dtype == 1 else None)
        return rc, lo, hi, ptr.T, cst.T

    # counting structures
    def count_build(self, kind, A, hint, b=0, H=0, bp=0):
        L = lib()
        if kind == "net":
            return C.c_void_p(L.orc_netcount_build(C.c_int32(hint), *self._A(A)))
        if kind == "selfnet":
            return C.c_void_p(L.orc_selfnetcount_build(C.c_int32(hint), *self._A(A)))
        return C.c_void_p(L.orc_dom_build(C.c_int32(hint), *self._A(A), _i64(b), _i64(H), _i64(bp)))

    def count_query(self, kind, h, a, b, out):
        L = lib()
        f = L.orc_dom_query if kind == "dom" else L.orc_net_query
        for t in range(a.size):
            out[t] = f(h, _i64(a[t]), _i64(b[t]))
        return 0

    def count_step(self, kind, h, ma, a, mb, b):
        L = lib()
        f = L.orc_dom_step if kind == "dom" else L.orc_net_step
        return f(h, C.c_int32(ma), _i64(a), C.c_int32(mb), _i64(b))

    def count_free(self, kind, h):
        L = lib()
        (L.orc_dom_free if kind == "dom" else L.orc_net_free)(h)

    def link_array(self, A):
        out = np.zeros(A.nnz, dtype=np.int64)
        lib().orc_net_link_array(*self._A(A), _p(out))
        return out

    def partwise(self, A, K, asg):
        asg = np.ascontiguousarray(asg, dtype=np.int64)
        pios = np.zeros(K + 1, dtype=np.int64)
        prm = np.zeros(max(A.nnz, 1), dtype=np.int64)
        pos = np.zeros(A.nnz + 1, dtype=np.int64)
        idx = np.zeros(max(A.nnz, 1), dtype=np.int64)
        npr = lib().orc_partwise(*self._A(A), _i64(K), _p(asg), _p(pios), _p(prm), _p(pos), _p(idx))
        return int(npr), pios, prm[:npr].copy(), pos[:npr + 1].copy(), idx[:A.nnz].copy()
