#!/bin/bash
# per-round kernel table of one DP layer (run on the GPU box): bash tools/trace_round_table.sh <tag> [bench.py arguments]
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/trace_$tag
rm -rf "$out" && mkdir -p "$out"
rocprofv3 --kernel-trace --output-format csv -d "$out/raw" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --parts ${PARTS:-4} "$@" > "$out/bench.json" 2> "$out/bench.err"
f=$(ls $out/raw/*/*_kernel_trace.csv | head -1)
python3 tools/round_table.py "$f" ${LAYER:-0} > "$out/round_table.txt"
rm -rf "$out/raw"
cat "$out/round_table.txt"
