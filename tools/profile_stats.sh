#!/bin/bash
# rocprofv3 kernel-trace statistics of one bench configuration (run on the GPU box):
#   bash tools/profile_stats.sh <tag> [bench.py arguments]     -> gpurun_out/prof_<tag>/{kernel_stats.csv, domain_stats.csv, bench.json}
# (the program after `--` is python3 itself: no env / bash -c hop between rocprofv3 and the process that opens the GPU)
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/prof_$tag
rm -rf "$out" && mkdir -p "$out"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/raw" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > "$out/bench.json" 2> "$out/bench.err"
python3 - "$out" <<'PY'
import csv, glob, sys
out = sys.argv[1]
for name in ("kernel_stats", "domain_stats"):
    f = glob.glob(f"{out}/raw/*/*_{name}.csv")
    if not f:
        continue
    rows = list(csv.reader(open(f[0])))
    with open(f"{out}/{name}.csv", "w") as g:
        w = csv.writer(g)
        for r in rows:
            r[0] = r[0][:110]              # kernel names truncated to 110 characters
            w.writerow(r)
print(open(f"{out}/kernel_stats.csv").read()[:3000])
PY
rm -rf "$out/raw"
