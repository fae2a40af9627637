#!/bin/bash
# Hardware counters of the dispatches of one kernel (run on the GPU box), one rocprofv3 --pmc pass (kernel trace only):
#   bash tools/pmc_kernel.sh <tag> <kernel-name-substring> "<COUNTER ...>" [bench.py arguments]
# -> gpurun_out/pmc_<tag>/table.txt : one line per dispatch of the LAST layer (in dispatch order), the counters side by side
set -e
tag=$1; pat=$2; ctrs=$3; shift 3
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/pmc_$tag
rm -rf "$out" && mkdir -p "$out"
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out/raw" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --parts 4 "$@" > "$out/bench.log" 2>&1
python3 - "$out" "$pat" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
f = glob.glob(f"{out}/raw/*/*_counter_collection.csv")[0]
rows = [r for r in csv.DictReader(open(f)) if pat in r["Kernel_Name"]]
by = collections.OrderedDict()
for r in rows:
    by.setdefault(int(r["Dispatch_Id"]), {})[r["Counter_Name"]] = float(r["Counter_Value"])
names = sorted({k for v in by.values() for k in v})
ids = sorted(by)
with open(f"{out}/table.txt", "w") as g:
    g.write("dispatch " + " ".join(f"{n:>22}" for n in names) + "\n")
    for i in ids[-40:]:
        g.write(f"{i:8d} " + " ".join(f"{by[i].get(n, 0):22.0f}" for n in names) + "\n")
print(open(f"{out}/table.txt").read())
PY
rm -rf "$out/raw"
