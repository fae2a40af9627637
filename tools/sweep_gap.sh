mkdir -p gpurun_out/r07
for o in "" "gap_min=32" "gap_min=128" "gap_min=256" "gap_min=1024" "gap_tau=7" "gap_tau=7 gap_min=256" "gap_tau=7 gap_min=1024" "gap_tau=-1" "own_min=128" "own_min=256"; do
  args=""; for kv in $o; do args="$args --opt $kv"; done
  timeout -k 10 120 python bench.py --steps 3 --no-extras --no-cpu-baseline $args > gpurun_out/r07/s.json 2> gpurun_out/r07/s.err || { echo "FAIL $o"; tail -n 3 gpurun_out/r07/s.err; continue; }
  python -c "
import json; d=json.load(open('gpurun_out/r07/s.json')); k=d['kernels_ms_per_step']
print('%-28s %7.1f  own %6.1f gapown %5.1f gapfin %5.1f leaf %5.1f span %5.1f setup %5.1f' % ('$o', d['ms_per_step'], k.get('dp_lpass_own',0), k.get('dp_lpass_gap',0), k.get('dp_gap_finish',0), k.get('dp_leaf',0), k.get('dp_span_fix',0), k.get('dp_task_setup',0)))" | tee -a gpurun_out/r07/sweep.txt
done
