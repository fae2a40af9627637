import os, sys, time
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import cpamd, synth
cp = cpamd.load()
from chainpartitioners_jl_amd import _lib
dev = torch.device("cuda", 0)
hip = _lib.HipBackend(device=0)
n, N, K = 10_000_000, 100_000_000, 64
mm_, nn_, colptr, rowval = synth.suitesparse_shaped_t(n, 10, 0xDEADBEEF + 2, dev, None, N)
h = hip.csr_from_device(n, n, int(rowval.numel()), colptr.data_ptr(), rowval.data_ptr())
mode = sys.argv[1] if len(sys.argv) > 1 else "bottleneck"
mdl = (cp.AffineConnectivityModel(0, 10, 1, 100) if mode == "bottleneck" else cp.AffineConnectivityModel(0, 0, 0, 1)).marshal()
comb = 1 if mode == "bottleneck" else 0
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 24
spl = np.zeros(K + 1, dtype=np.int64)
for rep in range(nsteps):
    hip.prof_reset(); hip.prof_enable(True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hip.reset_cache(h)
    t1 = time.perf_counter()
    hip.partition_dynamic(h, K, comb, 0, mdl, None, None, 0, 0.0, spl)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    p = hip.prof_get()
    print("step %d: %.1f ms (reset_cache %.1f ms; kernel sum %.1f)" % (rep, (t2 - t0) * 1e3, (t1 - t0) * 1e3, sum(v["ms"] for v in p.values())), flush=True)
