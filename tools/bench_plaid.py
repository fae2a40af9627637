#!/usr/bin/env python3
"""Timing of the 2-D path: partition_plaid(A, K, AlternatingPartitioner(net, local, comm, local)) with the
reference benchmark's models (runbenchmarks.jl:16-20), device vs one host core of the oracle."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import numpy as np
import torch
torch.cuda.init()           # torch's HIP runtime first, then the library (as bench.py does)
from util import cp, suitesparse_shaped
from chainpartitioners_jl_amd import _lib
import orc_binding
hip = _lib.HipBackend(); orc = orc_binding.OracleBackend()
net = cp.AffineConnectivityModel(0, 10, 1, 100); comm = cp.AffinePrimaryConnectivityModel(0, 10, 1, 0, 100); local = cp.AffineSecondaryConnectivityModel(0, 10, 1, 0, 100)
out = {}
for n, K, do_cpu in ((5000, 8, True), (20000, 8, True), (60000, 8, False)):
    A = suitesparse_shaped(n, 8, 11)
    meth = cp.AlternatingPartitioner(cp.DynamicBottleneckSplitter(net), cp.DynamicBottleneckSplitter(local), cp.DynamicBottleneckSplitter(comm), cp.DynamicBottleneckSplitter(local))
    cp.partition_plaid(A, K, meth, backend=hip)
    t0 = time.perf_counter(); Pi, Phi = cp.partition_plaid(A, K, meth, backend=hip); tg = time.perf_counter() - t0
    rec = {"n": n, "nnz": A.nnz, "K": K, "gpu_s": tg, "comm_bottleneck": cp.bottleneck_value(A, Phi, comm, Pi, backend=hip)}
    if do_cpu:
        t0 = time.perf_counter(); Pi2, Phi2 = cp.partition_plaid(A, K, meth, backend=orc); rec["cpu_s"] = time.perf_counter() - t0
        rec["same"] = bool(Pi == Pi2 and Phi == Phi2)
    out["n%d" % n] = rec
    print(json.dumps(rec), flush=True)

# the scalable methods of the reference's 2-D benchmark (runbenchmarks.jl:65-78) at config-2 scale, device only
from bench import gen_suitesparse_shaped
dev = torch.device("cuda", 0)
n, K = 1_000_000, 32
colptr, rowval = gen_suitesparse_shaped(n, 13 * n, 0xDEADBEEF + 1, dev)
A = cp.SparseMatrixCSC(n, n, colptr.cpu().numpy(), rowval.cpu().numpy())
t0 = time.perf_counter(); adjA = cp.adjointpattern(A, backend=hip); t_adj = time.perf_counter() - t0
Pi = cp.partition_stripe(adjA, K, cp.EquiSplitter())
rec = {"n": n, "nnz": A.nnz, "K": K, "adjoint_s": t_adj}
for name, meth in (("bisect_cost_comm", cp.BisectCostBottleneckSplitter(comm, 0.01)), ("bisect_index_comm", cp.BisectIndexBottleneckSplitter(comm))):
    cp.partition_stripe(A, K, meth, Pi, backend=hip)
    t0 = time.perf_counter(); Phi = cp.partition_stripe(A, K, meth, Pi, backend=hip); rec[name + "_s"] = time.perf_counter() - t0
    rec[name + "_bottleneck"] = cp.bottleneck_value(A, Phi, comm, Pi, backend=hip)
Phi = cp.partition_stripe(A, K, cp.BisectCostBottleneckSplitter(net, 0.01), backend=hip)
for name, meth in (("flip_bisect_cost_local", cp.FlipBisectCostBottleneckSplitter(local, 0.01)),):
    cp.partition_stripe(adjA, K, meth, Phi, backend=hip)
    t0 = time.perf_counter(); P2 = cp.partition_stripe(adjA, K, meth, Phi, backend=hip); rec[name + "_s"] = time.perf_counter() - t0
print(json.dumps(rec), flush=True)
