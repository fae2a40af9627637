#!/bin/bash
# A/B of library builds on the GPU box: bash tools/ab_libs.sh "<bench args>" base build/ab/libX.so ...   (each variant twice, interleaved)
args=$1; shift
mkdir -p gpurun_out/ab
for rep in 1 2; do
for v in "$@"; do
  if [ "$v" = base ]; then unset CP_LIB_PATH; else export CP_LIB_PATH=$PWD/$v; fi
  timeout -k 10 150 python bench.py --no-extras --no-cpu-baseline $args > gpurun_out/ab/s.json 2> gpurun_out/ab/s.err || { echo "FAIL $v"; tail -n 3 gpurun_out/ab/s.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/ab/s.json')); k=d['kernels_ms_per_step']
print('%-40s %7.1f  ' % ('$v', d['ms_per_step']) + ' '.join('%s %.1f' % (a[3:], b) for a, b in sorted(k.items(), key=lambda x: -x[1])[:9]))" | tee -a gpurun_out/ab/ab.txt
done
done
