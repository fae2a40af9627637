#!/usr/bin/env python3
"""Extract the six SuiteSparse matrices the reference embeds as literal data in
test/matrices.jl (test/matrices.jl:4-9) into a small CSC pattern fixture,
tests/golden/matrices.json.

This copies DATA (I, J index lists and dimensions), not source: the output holds only
the sparsity patterns as 1-based CSC (colptr, rowval).  `sparse(I, J, V, m, n)` sums
duplicate (i, j) pairs, so the pattern is the set of distinct pairs; `Symmetric(S, :L)`
(HB/can_292) mirrors the lower triangle.

Run in the build container only (needs /root/reference):  python tools/extract_reference_matrices.py
"""
import json, re, sys, os
import numpy as np

SRC = "/root/reference/test/matrices.jl"
OUT = os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "matrices.json")

def csc_from_pairs(I, J, m, n):
    pairs = sorted(set(zip(J, I)))            # column-major, rows ascending
    colptr = np.zeros(n + 1, dtype=np.int64)
    for (j, i) in pairs:
        colptr[j] += 1
    colptr = np.concatenate([[1], 1 + np.cumsum(colptr[1:])]).astype(np.int64)
    rowval = np.array([i for (_, i) in pairs], dtype=np.int64)
    return colptr.tolist(), rowval.tolist()

def main():
    out = {}
    for line in open(SRC):
        mt = re.match(r'^"([^"]+)"\s*=>\s*(Symmetric\()?sparse\(\[([^\]]*)\],\s*\[([^\]]*)\],\s*(?:Bool)?\[[^\]]*\],\s*(\d+),\s*(\d+)\)', line)
        if not mt:
            continue
        name, sym, Is, Js, m, n = mt.groups()
        I = [int(x) for x in Is.split(",")]
        J = [int(x) for x in Js.split(",")]
        m, n = int(m), int(n)
        if sym:
            tail = line[mt.end():]
            assert ":L" in tail or 'Symbol("L")' in tail
            I2, J2 = [], []
            for i, j in zip(I, J):
                if i >= j:                     # lower triangle is the source of truth
                    I2 += [i, j]; J2 += [j, i]
            I, J = I2, J2
        colptr, rowval = csc_from_pairs(I, J, m, n)
        out[name] = {"m": m, "n": n, "colptr": colptr, "rowval": rowval}
        print(name, m, n, len(rowval), file=sys.stderr)
    json.dump(out, open(OUT, "w"))

if __name__ == "__main__":
    main()
