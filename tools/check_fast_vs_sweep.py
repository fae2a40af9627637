"""GPU self-check at sizes the CPU oracle cannot reach: the O(n log^2 n) scheme (gap passes, cached round A, two-phase long
gaps) against the library's own literal O(n^2) device sweep (cp_set_option("force_brute")) -- full DP tables, bit for bit.
Usage: python tools/check_fast_vs_sweep.py [n] [K]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
import numpy as np
from util import cp, suitesparse_shaped, banded
hip = cp.get_backend()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
K = int(sys.argv[2]) if len(sys.argv) > 2 else 3
bad = 0
for name, A in (("suitesparse_shaped", suitesparse_shaped(n, 10, 5)), ("banded", banded(n, 16, 0.5, 2))):
    for mdl in (cp.AffineConnectivityModel(0, 0, 0, 1), cp.AffineConnectivityModel(0, 10, 1, 100), cp.AffineHyperedgeCutModel(0, 0, 0, 0, 1),
                cp.AffineHyperedgeCutModel(0, 2, 1, 1, 3)):
        mm = mdl.marshal()
        t0 = time.time(); rc1, p1, c1 = hip.dynamic_tables(A, K, 0, mm, None); t1 = time.time()
        hip.set_option("force_brute", 1)
        try:
            rc2, p2, c2 = hip.dynamic_tables(A, K, 0, mm, None)
        finally:
            hip.set_option("force_brute", 0)
        t2 = time.time()
        ok = rc1 == 0 and rc2 == 0 and np.array_equal(p1, p2) and np.array_equal(c1, c2)
        bad += not ok
        print(name, n, K, type(mdl).__name__, "OK" if ok else "MISMATCH", "fast %.2fs sweep %.2fs" % (t1 - t0, t2 - t1), flush=True)
print("mismatches:", bad)
sys.exit(1 if bad else 0)
