"""Full-size consistency check on the GPU box: a model whose layers all differ (alpha > 0), K = 16, n = 1e7.  The split vector must be
the same for every setting of the layer-driver options (they only choose between code paths).
Usage: python tools/check_options_fullsize.py [n] [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
import torch
import cpamd
cp = cpamd.load()
from chainpartitioners_jl_amd import _lib
from bench import gen_suitesparse_shaped
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 16
dev = torch.device("cuda", 0)
hip = _lib.HipBackend(device=0)
colptr, rowval = gen_suitesparse_shaped(n, 10 * n, 0xDEADBEEF + 2, dev)
h = hip.csr_from_device(n, n, int(rowval.numel()), colptr.data_ptr(), rowval.data_ptr())
defaults = {"nospec": 0, "gap_tau": 6, "ra_cache": 1, "fixed_point": 0}
bad = 0
for mdl in (cp.AffineConnectivityModel(20000, 10, 1, 100), cp.AffineHyperedgeCutModel(3000, 0, 0, 1, 3)):
    mm = mdl.marshal()
    ref = None
    for opts in ({}, {"nospec": 1}, {"gap_tau": -1}, {"ra_cache": 0}, {"fixed_point": 1}):
        for k, v in {**defaults, **opts}.items():
            hip.set_option(k, v)
        spl = np.zeros(K + 1, dtype=np.int64)
        hip.reset_cache(h)
        torch.cuda.synchronize(); t0 = time.time()
        rc = hip.partition_dynamic(h, K, 0, 0, mm, None, None, 0, 0.0, spl)
        torch.cuda.synchronize(); dt = time.time() - t0
        assert rc == 0, hip.last_error()
        rc, obj = hip.objective(h, K, spl, mm, None, 0)
        same = ref is None or np.array_equal(ref, spl)
        bad += not same
        if ref is None: ref = spl.copy()
        print(type(mdl).__name__, opts, "%.3f s" % dt, "objective", obj, "parts", int((np.diff(spl) > 0).sum()), "OK" if same else "MISMATCH", flush=True)
for k, v in defaults.items():
    hip.set_option(k, v)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
