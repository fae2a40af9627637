#!/bin/bash
# Collect HBM traffic counters for the dominant kernel (k_lpass_own) with rocprofv3, as MI355X_MICROARCH.md
# "HBM" prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slots), kernel-trace only.
# Run on the GPU box:  bash tools/pmc_lpass.sh   -> gpurun_out/pmc_r01d/{fetch,write}/...
set -e
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_r01d
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_r01d/fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --parts 4 > gpurun_out/pmc_r01d/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_r01d/write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --parts 4 > gpurun_out/pmc_r01d/write.log 2>&1
python3 - <<'PY'
import csv, glob, json
out = {}
for name in ("fetch", "write"):
    f = glob.glob(f"gpurun_out/pmc_r01d/{name}/*/*_counter_collection.csv")
    rows = list(csv.DictReader(open(f[0])))
    cname = "FETCH_SIZE" if name == "fetch" else "WRITE_SIZE"
    vals = [float(r["Counter_Value"]) for r in rows if "k_lpass_own" in r["Kernel_Name"] and r["Counter_Name"] == cname]
    out[cname] = {"launches": len(vals), "sum": sum(vals), "mean_per_launch": sum(vals) / max(len(vals), 1)}
# bench line of the fetch run gives the algorithmic bytes per launch at the same configuration
for line in open("gpurun_out/pmc_r01d/fetch.log"):
    if line.startswith("{"):
        d = json.loads(line)
        out["alg_bytes_per_launch"] = d["roofline"]["alg_bytes_per_launch"]
        out["avg_launch_ms_under_pmc"] = d["roofline"]["avg_launch_ms"]
json.dump(out, open("gpurun_out/pmc_r01d/summary.json", "w"), indent=1)
print(json.dumps(out))
PY
