import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
from util import cp, suitesparse_shaped, banded
hip = cp.get_backend()
n, K = 200000, 4
for A in (suitesparse_shaped(n, 10, 5), banded(n, 16, 0.5, 2)):
    for mdl in (cp.AffineConnectivityModel(0, 10, 1, 100), cp.AffineHyperedgeCutModel(0, 2, 1, 1, 3)):
        rc1, p1, c1 = hip.dynamic_tables(A, K, 0, mdl.marshal(), None)
        print("RC", rc1, hip.last_error() if rc1 else "", flush=True)
