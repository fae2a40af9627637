#!/usr/bin/env python3
"""Timing experiment for the leaf pass: k_leaf's time per layer with parts of it compiled out (library builds with -DCP_LEAF_SKIP=mask,
selected through CP_LIB_PATH; the results of such a build are WRONG -- only the `dp_leaf` profile slot is read).
usage: CP_LIB_PATH=build/ab/libchainpart_skipN.so python tools/leaf_parts.py [constrained|plain] [profile slots, comma separated]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cpamd, synth
cp = cpamd.load()
from chainpartitioners_jl_amd import _lib
dev = torch.device("cuda", 0)
hip = _lib.HipBackend(device=0)
n, N, K = 10_000_000, 100_000_000, 6
mm_, nn_, colptr, rowval = synth.suitesparse_shaped_t(n, 10, 0xDEADBEEF + 2, dev, None, N)
h = hip.csr_from_device(n, n, int(rowval.numel()), colptr.data_ptr(), rowval.data_ptr())
mdl = cp.AffineConnectivityModel(0, 0, 0, 1).marshal()
spl = np.zeros(K + 1, dtype=np.int64)
con = len(sys.argv) > 1 and sys.argv[1] == "constrained"
wm = cp.VertexCount().marshal() if con else None
w = -(-3 * n // (2 * 64)) if con else 0
for rep in range(2):
    hip.reset_cache(h); hip.prof_reset(); hip.prof_enable(True)
    hip.partition_dynamic(h, K, 0, 0, mdl, None, wm, w, float(w), spl)
    torch.cuda.synchronize()
    p = hip.prof_get()
slots = sys.argv[2].split(",") if len(sys.argv) > 2 else ["dp_leaf"]
print(os.environ.get("CP_LIB_PATH", "base"), "  ".join("%s ms per launch: %.3f (%d)" % (k, p[k]["ms"] / max(p[k]["launches"], 1), p[k]["launches"]) for k in slots))
