import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "tests"), os.path.join(os.getcwd(), "oracle")]
import numpy as np
from util import cp
from test_gpu_dynamic import mats, MODELS
from chainpartitioners_jl_amd import _lib
import orc_binding
hip = _lib.HipBackend(); orc = orc_binding.OracleBackend()
mdl = MODELS[6]
for A in mats():
    for K in (1, 2, 5):
        mm = mdl.marshal()
        rc1, p1, c1 = hip.dynamic_tables(A, K, 0, mm, None)
        rc2, p2, c2 = orc.dynamic_tables(A, K, 0, mm, None)
        if not np.array_equal(p1, p2):
            print("MISMATCH", A, K, np.argwhere(p1 != p2)[:4].tolist())
            for name, (st, se, dbg) in (("again", (4, 64, 0)), ("no-interior", (4, 64, 32)), ("again", (4, 64, 0)), ("t2", (2, 64, 0)), ("t3", (3, 64, 0)), ("t4e8", (4, 8, 0)), ("again", (4, 64, 0))):
                hip.set_option("short_t", st); hip.set_option("short_e", se); hip.set_option("dbg", dbg)
                rc1, p1, c1 = hip.dynamic_tables(A, K, 0, mm, None)
                print("   ", name, "mismatches:", int(np.sum(p1 != p2)))
            hip.set_option("short_t", 4); hip.set_option("short_e", 64)
print("done")
