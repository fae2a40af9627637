#!/usr/bin/env python3
"""Generate tests/golden/splits.json: split vectors of the hot-path methods on the six matrices the
reference embeds in test/matrices.jl and on seeded synthetic matrices, computed by the literal CPU
restatement (oracle/).  The reference's own tests hold no expected split vectors (SURVEY.md 8c:
"split indices parity unpinned"), so these goldens pin the RESTATEMENT's output -- they guard against
regressions of the oracle and are the fixed targets the GPU tests are also compared with.

    python tools/make_golden_splits.py      (CPU only)
"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np
import cpamd
cp = cpamd.load()
import orc_binding
from util import golden_matrices, suitesparse_shaped, banded


def cases():
    """(name, callable(A, backend) -> SplitPartition) ; constants follow test/runbenchmarks.jl:16-33"""
    work = cp.AffineWorkModel(0, 10, 1)
    net = cp.AffineConnectivityModel(0, 10, 1, 100)
    lam = cp.AffineConnectivityModel(0, 0, 0, 1)
    hyp = cp.AffineHyperedgeCutModel(0, 0, 0, 0, 1)
    colb = cp.ColumnBlockComponentCostModel(3, lambda w: 1 + w)
    out = []
    for K in (2, 8):
        out += [(f"DynamicTotalSplitter(net),K={K}", lambda A, b, K=K: cp.partition_stripe(A, K, cp.DynamicTotalSplitter(net), backend=b)),
                (f"DynamicTotalSplitter(lambda-1),K={K}", lambda A, b, K=K: cp.partition_stripe(A, K, cp.DynamicTotalSplitter(lam), backend=b)),
                (f"DynamicTotalSplitter(hyperedge),K={K}", lambda A, b, K=K: cp.partition_stripe(A, K, cp.DynamicTotalSplitter(hyp), backend=b)),
                (f"DynamicBottleneckSplitter(net),K={K}", lambda A, b, K=K: cp.partition_stripe(A, K, cp.DynamicBottleneckSplitter(net), backend=b)),
                (f"DynamicBottleneckSplitter(work),K={K}", lambda A, b, K=K: cp.partition_stripe(A, K, cp.DynamicBottleneckSplitter(work), backend=b)),
                (f"BisectCost(work,0.01),K={K}", lambda A, b, K=K: cp.partition_stripe(A, K, cp.BisectCostBottleneckSplitter(work, 0.01), backend=b)),
                (f"BisectCost(net,0.01),K={K}", lambda A, b, K=K: cp.partition_stripe(A, K, cp.BisectCostBottleneckSplitter(net, 0.01), backend=b)),
                (f"ConvexTotalSplitter(lambda-1),K={K}", lambda A, b, K=K: cp.partition_stripe(A, K, cp.ConvexTotalSplitter(lam), backend=b)),
                (f"DynamicTotalSplitter(Constrained(net,work-width,12)),K={K}",
                 lambda A, b, K=K: cp.partition_stripe(A, K, cp.DynamicTotalSplitter(cp.ConstrainedCost(net, cp.AffineWorkModel(0, 1, 0), -(-A.n // K) + 4)), backend=b)),
                # the reference's own script: DynamicTotalSplitter(ConstrainedCost(lambda-1, VertexCount(), ceil(1.5 n / K)))
                # (bin/test_table_constrained_splits.jl:28) -- on the GPU the windowed O(K n log^2 n) path -- and the hyperedge / chunker-order twins
                (f"DynamicTotalSplitter(Constrained(lambda-1,VertexCount,1.5n/K)),K={K}",
                 lambda A, b, K=K: cp.partition_stripe(A, K, cp.DynamicTotalSplitter(cp.ConstrainedCost(lam, cp.VertexCount(), -(-3 * A.n // (2 * K)))), backend=b)),
                (f"DynamicTotalSplitter(Constrained(hyperedge,VertexCount,1.5n/K)),K={K}",
                 lambda A, b, K=K: cp.partition_stripe(A, K, cp.DynamicTotalSplitter(cp.ConstrainedCost(hyp, cp.VertexCount(), -(-3 * A.n // (2 * K)))), backend=b)),
                (f"DynamicTotalChunker(Constrained(net,VertexCount,1.5n/K)),K={K}",
                 lambda A, b, K=K: cp.partition_stripe(A, K, cp.DynamicTotalChunker(cp.ConstrainedCost(net, cp.VertexCount(), -(-3 * A.n // (2 * K)))), backend=b))]
    out += [("DynamicTotalChunker(net,8)", lambda A, b: cp.pack_stripe(A, cp.DynamicTotalChunker(cp.ConstrainedCost(net, cp.VertexCount(), 8)), backend=b)),
            ("DynamicTotalChunker(col_block,8)", lambda A, b: cp.pack_stripe(A, cp.DynamicTotalChunker(cp.ConstrainedCost(colb, cp.VertexCount(), 8)), backend=b)),
            ("ConvexTotalChunker(col_block,8)", lambda A, b: cp.pack_stripe(A, cp.ConvexTotalChunker(cp.ConstrainedCost(colb, cp.VertexCount(), 8)), backend=b)),
            ("ConvexTotalChunker(lambda-1)", lambda A, b: cp.pack_stripe(A, cp.ConvexTotalChunker(lam), backend=b))]
    return out


def matrices():
    m = dict(golden_matrices())
    m["synthetic/suitesparse_shaped(300,5,seed=1)"] = suitesparse_shaped(300, 5, 1)
    m["synthetic/banded(200,4,0.5,seed=2)"] = banded(200, 4, 0.5, 2)
    m["synthetic/suitesparse_shaped(1500,6,seed=3)"] = suitesparse_shaped(1500, 6, 3)      # (wide enough for windows that span the 64-row leaf groups)
    return m


def main():
    orc = orc_binding.OracleBackend()
    out = {}
    for mname, A in matrices().items():
        for cname, fn in cases():
            out[f"{mname} :: {cname}"] = fn(A, orc).spl.tolist()
    json.dump(out, open(os.path.join(ROOT, "tests", "golden", "splits.json"), "w"), indent=0)
    print(len(out), "golden split vectors")


if __name__ == "__main__":
    main()
