#!/bin/bash
# Collect HBM traffic counters for the dominant kernel (k_lpass_own) with rocprofv3, as MI355X_MICROARCH.md
# "HBM" prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE --pmc passes (TCC slots), kernel-trace only.
# Run on the GPU box:  bash tools/pmc_lpass.sh [tag] [commit]   -> gpurun_out/pmc_<tag>/summary.json (tag defaults to r03)
set -e
TAG=${1:-r03}; COMMIT=${2:-unknown}
export TAG COMMIT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_$TAG
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/pmc_$TAG/fetch -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --parts 4 > gpurun_out/pmc_$TAG/fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/pmc_$TAG/write -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --parts 4 > gpurun_out/pmc_$TAG/write.log 2>&1
python3 - <<'PY'
import csv, glob, json, os
TAG = os.environ['TAG']
out = {}
for name in ("fetch", "write"):
    f = glob.glob(f"gpurun_out/pmc_{TAG}/{name}/*/*_counter_collection.csv")
    rows = list(csv.DictReader(open(f[0])))
    cname = "FETCH_SIZE" if name == "fetch" else "WRITE_SIZE"
    # the plain streaming kernel k_lpass_own<TC, HYP, GAP = false> (the gap-pass variant is a different kernel and profile slot)
    def plain(nm):
        return "k_lpass_own<" in nm and nm.split("k_lpass_own<")[1].split(">")[0].replace(" ", "").endswith("false")
    vals = [float(r["Counter_Value"]) for r in rows if plain(r["Kernel_Name"]) and r["Counter_Name"] == cname]
    out[cname] = {"launches": len(vals), "sum": sum(vals), "mean_per_launch": sum(vals) / max(len(vals), 1)}
# bench line of the fetch run gives the algorithmic bytes per launch at the same configuration
for line in open(f"gpurun_out/pmc_{TAG}/fetch.log"):
    if line.startswith("{"):
        d = json.loads(line)
        out["alg_bytes_per_launch"] = d["roofline"]["alg_bytes_per_launch"]
        out["avg_launch_ms_under_pmc"] = d["roofline"]["avg_launch_ms"]
f, w = out["FETCH_SIZE"]["mean_per_launch"], out["WRITE_SIZE"]["mean_per_launch"]
out["traffic_bytes_per_launch_corrected"] = (2 * f + w) * 1024
out["traffic_bytes_per_launch_raw"] = (f + w) * 1024
out["note"] = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE in two separate passes (tools/pmc_lpass.sh), kernel "
               "k_lpass_own<long,false,false>, config-3 matrix, K=4 (the per-launch mean equals the K=64 run: every layer runs the same rounds). "
               "Counter unit is KiB. MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reports half the bytes of wide (16 B/lane) streaming reads; the "
               "link-array stream is 16 B/lane, the remaining reads (colptr, previous-layer cost, tile records) are 4-16 B/lane and uncalibrated; "
               "the stores are the tile records only (16 B + 4 B per 256-step tile -- round 1 also wrote 96 B of private scratch per lane here): traffic_bytes_per_launch_corrected = (2*FETCH_SIZE + WRITE_SIZE) * 1024 is an upper "
               "estimate, traffic_bytes_per_launch_raw = (FETCH_SIZE + WRITE_SIZE) * 1024 a lower one.")
out["measured_at"] = os.environ.get("COMMIT", "unknown")
json.dump(out, open(f"gpurun_out/pmc_{TAG}/summary.json", "w"), indent=1)
print(json.dumps(out))
PY
rm -rf gpurun_out/pmc_$TAG/fetch gpurun_out/pmc_$TAG/write
