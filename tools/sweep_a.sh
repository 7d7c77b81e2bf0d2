#!/bin/bash
# sweep stage A window load (GP_STAGE_A_LOAD_DIV) on both bench workloads
for d in 1 3 6 10; do
  for wl in config2 goldenpath; do
    GP_STAGE_A_LOAD_DIV=$d python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aux --workload $wl 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['roofline']['kernel_ms_per_run']; print('div=$d $wl', d['value'], d['ms_per_step'], 'fp', k['cbf_first_pass'], 'rounds', k['cbf_round'], d['counters']['a_rounds'])
"
  done
done
