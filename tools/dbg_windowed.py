"""debug driver: one constrained-table call (n, K, w from argv)"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cpamd
cp = cpamd.load()
from util import suitesparse_shaped
n, K, w = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
opts = sys.argv[4:]
hip = cp.get_backend()
for o in opts:
    k, v = o.split("=")
    hip.set_option(k, int(v))
A = suitesparse_shaped(n, 8, 1)
mdl = cp.AffineConnectivityModel(0, 0, 0, 1)
print("calling", n, K, w, opts, flush=True)
rc, lo, hi, p, c = hip.dynamic_tables_constrained(A, K, mdl.marshal(), w)
print("rc", rc, hip.last_error(), lo, hi, flush=True)
import orc_binding
orc = orc_binding.OracleBackend()
rc2, lo2, hi2, p2, c2 = orc.dynamic_tables_constrained(A, K, 0, mdl.marshal(), None, cp.VertexCount().marshal(), w, float(w))
print("match", np.array_equal(p, p2), np.array_equal(c, c2), flush=True)
if not np.array_equal(p, p2):
    bad = np.argwhere(p != p2)
    print("first mismatches (row, layer):", bad[:10].tolist(), p[tuple(bad[0])], p2[tuple(bad[0])])
