"""Debug helper: DP tables of the HIP path vs the oracle on the test matrices, for a given option set."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + "/oracle", ROOT + "/tests"]
import numpy as np
from util import cp, golden_matrices, suitesparse_shaped, banded
import cpamd
import orc_binding
hip = cp.get_backend()
orc = orc_binding.OracleBackend()
for kv in sys.argv[1:]:
    k, v = kv.split("=")
    hip.set_option(k, int(v))
mats = list(golden_matrices().values()) + [suitesparse_shaped(1000, 6, 3), banded(777, 4, 0.5, 9), suitesparse_shaped(5000, 8, 1)]
models = [cp.AffineConnectivityModel(0, 0, 0, 1), cp.AffineHyperedgeCutModel(0, 0, 0, 0, 1), cp.AffineHyperedgeCutModel(0, 2, 1, 1, 3)]
nbad = 0
for A in mats:
    for mi, mdl in enumerate(models):
        for K in (2, 5):
            mm = mdl.marshal()
            rc1, p1, c1 = hip.dynamic_tables(A, K, 0, mm, None)
            rc2, p2, c2 = orc.dynamic_tables(A, K, 0, mm, None)
            if not (np.array_equal(p1, p2) and np.array_equal(c1, c2)):
                nbad += 1
                bad = np.argwhere((p1 != p2) | (c1 != c2))
                r, k = bad[0]
                print("MISMATCH", A, "model", mi, "K", K, "n bad", len(bad), "first (row, layer)", r, k, "ptr hip/orc", p1[r, k], p2[r, k], "cst", c1[r, k], c2[r, k])
                print("   bad rows (first 20):", [tuple(x) for x in bad[:20]])
print("mismatches:", nbad)
