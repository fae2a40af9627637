"""CPU-only checks: the C-ABI library loads and exports every symbol include/chainpart.h declares
(no compute calls without a GPU), the product refuses to run without a device, host-side closed
forms, and the struct layouts the Python marshalling assumes."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from util import cp, sprand

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    txt = open(os.path.join(ROOT, "include", "chainpart.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(cp_[a-z_0-9]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    from chainpartitioners_jl_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "run __graft_entry__.build() first"
    lib = C.CDLL(_lib.LIB_PATH)
    decl = declared_symbols()
    assert len(decl) >= 25
    for name in decl:
        assert hasattr(lib, name), name
    assert sorted(_lib.SYMBOLS) == decl          # the binding covers exactly the header


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from chainpartitioners_jl_amd import _lib
    lib = C.CDLL(_lib.LIB_PATH)
    assert lib.cp_device_count() == 0
    with pytest.raises(RuntimeError):
        _lib.HipBackend()
    A = sprand(4, 6, 0.5, np.random.default_rng(0))
    h = C.c_void_p()
    rc = lib.cp_csr_create(C.c_int64(A.m), C.c_int64(A.n), C.c_int64(A.nnz), A.colptr.ctypes.data_as(C.c_void_p),
                           A.rowval.ctypes.data_as(C.c_void_p), C.c_int32(0), C.byref(h))
    assert rc == 3 and not h.value                # CP_EHIP: refuses loudly, no handle
    lib.cp_last_error.restype = C.c_char_p
    assert b"no HIP device" in lib.cp_last_error()


def test_equi_closed_forms_match_library_and_oracle(orc):
    from chainpartitioners_jl_amd import _lib
    lib = C.CDLL(_lib.LIB_PATH)
    import orc_binding
    olib = orc_binding.lib()
    for n in (0, 1, 5, 17, 100):
        for K in (1, 2, 3, 8, 23):
            a = np.zeros(K + 1, dtype=np.int64); b = np.zeros(K + 1, dtype=np.int64)
            assert lib.cp_partition_equi(C.c_int64(n), C.c_int64(K), a.ctypes.data_as(C.c_void_p)) == 0
            olib.orc_partition_equi(C.c_int64(n), C.c_int64(K), b.ctypes.data_as(C.c_void_p))
            A = cp.SparseMatrixCSC(1, n, np.ones(n + 1, dtype=np.int64), np.zeros(0, dtype=np.int64))
            assert a.tolist() == b.tolist() == cp.partition_stripe(A, K, cp.EquiSplitter()).spl.tolist()
        for w in (1, 2, 5):
            a = np.zeros(n + 2, dtype=np.int64); b = np.zeros(n + 2, dtype=np.int64); Ka = C.c_int64()
            assert lib.cp_pack_equi(C.c_int64(n), C.c_int64(w), a.ctypes.data_as(C.c_void_p), C.byref(Ka)) == 0
            Kb = olib.orc_pack_equi(C.c_int64(n), C.c_int64(w), b.ctypes.data_as(C.c_void_p))
            assert Ka.value == Kb and a[:Kb + 1].tolist() == b[:Kb + 1].tolist()


def test_struct_layout_matches_header(tmp_path):
    """ctypes mirrors vs the C header itself: sizes and field offsets printed by a program compiled from
    include/chainpart_types.h."""
    import subprocess
    from chainpartitioners_jl_amd import models as M
    src = tmp_path / "layout.c"
    src.write_text('''#include <stdio.h>
#include <stddef.h>
#include "chainpart_types.h"
int main(void) {
    printf("%zu %zu %zu ", sizeof(cp_component_t), sizeof(cp_model_t), sizeof(cp_rowpart_t));
    printf("%zu %zu %zu ", offsetof(cp_component_t, table), offsetof(cp_component_t, len), offsetof(cp_component_t, lo));
    printf("%zu %zu %zu %zu %zu\\n", offsetof(cp_model_t, p_f64), offsetof(cp_model_t, alpha_k), offsetof(cp_model_t, R),
           offsetof(cp_model_t, alpha_row), offsetof(cp_model_t, beta_col));
    return 0;
}
''')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(M.cp_component_t), C.sizeof(M.cp_model_t), C.sizeof(M.cp_rowpart_t),
            M.cp_component_t.table.offset, M.cp_component_t.len.offset, M.cp_component_t.lo.offset,
            M.cp_model_t.p_f64.offset, M.cp_model_t.alpha_k.offset, M.cp_model_t.R.offset,
            M.cp_model_t.alpha_row.offset, M.cp_model_t.beta_col.offset]
    assert got == want


def test_model_promotion_rules():
    assert cp.AffineConnectivityModel(0, 10, 1, 100).dtype == 0          # all Int -> Int64
    assert cp.AffineConnectivityModel(0.0, 0, 0, 1).dtype == 1           # promote -> Float64
    assert cp.AffineConnectivityModel(0, 3, 1, 3, alpha_k=[1, 2]).dtype == 0
    assert cp.AffineWorkModel(0, 10, 1)(3, 7) == 37


def test_plain_c_client_builds_against_the_header():
    """examples/c_abi_demo.c uses nothing but include/chainpart.h and the shared library (no Python, no torch types)."""
    import subprocess
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "examples"), "clean", "c_abi_demo"], stdout=subprocess.DEVNULL)
    assert os.path.exists(os.path.join(ROOT, "examples", "c_abi_demo"))
