import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
from util import cp, suitesparse_shaped, banded
import orc_binding
hip = cp.get_backend(); orc = orc_binding.OracleBackend()
hip.set_option("gap_tau", 8); hip.set_option("gap_min", 8)
if len(sys.argv) > 1: hip.set_option("dbg", int(sys.argv[1]))
MODELS = [cp.AffineConnectivityModel(0, 0, 0, 1), cp.AffineConnectivityModel(0, 10, 1, 100), cp.AffineHyperedgeCutModel(0, 2, 1, 1, 3)]
for A in [suitesparse_shaped(3000, 8, 1), banded(2500, 6, 0.5, 3), suitesparse_shaped(1025, 5, 7)]:
    for mi, mdl in enumerate(MODELS):
        for (K, w) in [(4, -(-3 * A.n // 8)), (7, A.n // 4), (16, A.n // 8), (3, A.n // 3 + 97), (3, 700)]:
            mm = mdl.marshal()
            rc1, lo1, hi1, p1, c1 = hip.dynamic_tables_constrained(A, K, mm, w)
            print(A.n, mi, K, w, rc1, hip.last_error() if rc1 not in (0, 2) else "", flush=True)
