#!/usr/bin/env python3
"""Where does a ~1 s host stall every 12th bottleneck partition come from?  Per-step wall times of (a) the bottleneck DP with K = 4,
(b) the net counter build alone (cp_count_build + destroy), (c) the link build alone (reset_cache + a Work-model total DP with K = 1)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cpamd, synth
cp = cpamd.load()
from chainpartitioners_jl_amd import _lib
dev = torch.device("cuda", 0)
hip = _lib.HipBackend(device=0)
n, N = 10_000_000, 100_000_000
mm_, nn_, colptr, rowval = synth.suitesparse_shaped_t(n, 10, 0xDEADBEEF + 2, dev, None, N)
h = hip.csr_from_device(n, n, int(rowval.numel()), colptr.data_ptr(), rowval.data_ptr())
mdl = cp.AffineConnectivityModel(0, 10, 1, 100).marshal()
which = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
spl = np.zeros(5, dtype=np.int64)
out = []
for rep in range(nsteps):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if which == "dp4":
        hip.reset_cache(h); hip.partition_dynamic(h, 4, 1, 0, mdl, None, None, 0, 0.0, spl)
    elif which == "count":
        hip.reset_cache(h); c = hip.count_build("net", h, 0); hip.count_free("net", c)
    elif which == "links":
        hip.reset_cache(h); hip.partition_dynamic(h, 1, 0, 0, cp.AffineWorkModel(0, 1, 1).marshal(), None, None, 0, 0.0, spl[:2])
    torch.cuda.synchronize(); out.append((time.perf_counter() - t0) * 1e3)
print(which, " ".join("%.0f" % t for t in out))
