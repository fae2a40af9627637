#!/bin/bash
# Idle time between dependent kernels of one bench step (run on the GPU box):
#   bash tools/gap_analysis.sh <tag> [bench.py arguments]   -> gpurun_out/gaps_<tag>/{summary.txt, bench.json}
# rocprofv3 --kernel-trace (no counters); the LAST step's dispatches are taken (after the largest pause = the host-side check
# between steps is excluded by looking at the final `frac` of the trace), busy = sum of durations, idle = gaps between one
# kernel's end and the next one's start, attributed to the kernel that follows.
set -e
tag=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out/gaps_$tag
rm -rf "$out" && mkdir -p "$out"
rocprofv3 --kernel-trace --output-format csv -d "$out/raw" -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras "$@" > "$out/bench.json" 2> "$out/bench.err"
python3 - "$out" <<'PY'
import csv, glob, sys, collections, re
out = sys.argv[1]
f = glob.glob(f"{out}/raw/*/*_kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
# the timed step = the last third of the DP dispatches (steps: untimed profiling step, timed step; layers are alike): take the
# dispatches after the last k_links-type build launch
idx = [i for i, e in enumerate(ev) if "k_links" in e[2] or "k_link" in e[2]]
start = idx[-1] if idx else 0
# (the build of the last step begins a little before k_links: back up to the previous pause > 1 ms)
i0 = start
while i0 > 0 and ev[i0][0] - ev[i0 - 1][1] < 1_000_000:
    i0 -= 1
seg = ev[i0:]
busy = sum(e[1] - e[0] for e in seg)
span = seg[-1][1] - seg[0][0]
gaps = collections.Counter(); cnt = collections.Counter(); dur = collections.Counter()
def short(n):
    n = re.sub(r"^void ", "", n); n = re.sub(r"\(.*", "", n); return n[:70]
prev_end = seg[0][1]
for s, e, n in seg[1:]:
    g = max(0, s - prev_end)
    gaps[short(n)] += g; cnt[short(n)] += 1; dur[short(n)] += e - s
    prev_end = max(prev_end, e)
with open(f"{out}/summary.txt", "w") as g:
    g.write("dispatches %d  span %.1f ms  busy %.1f ms  idle %.1f ms (%.1f %%)\n" % (len(seg), span / 1e6, busy / 1e6, (span - busy) / 1e6, 100.0 * (span - busy) / span))
    g.write("%-72s %7s %10s %10s %8s\n" % ("kernel (idle time BEFORE it starts)", "calls", "idle ms", "run ms", "idle us/call"))
    for k, v in gaps.most_common(25):
        g.write("%-72s %7d %10.2f %10.2f %8.2f\n" % (k, cnt[k], v / 1e6, dur[k] / 1e6, v / 1e3 / cnt[k]))
print(open(f"{out}/summary.txt").read())
PY
rm -rf "$out/raw"
