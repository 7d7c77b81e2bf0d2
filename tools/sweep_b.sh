#!/bin/bash
for cfg in "1 6" "0 6" "0 4" "1 8" "0 8"; do
  set -- $cfg
  for wl in config2 goldenpath; do
    GP_STAGE_A_NIBBLE=$1 GP_STAGE_A_LOAD_DIV=$2 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-aux --workload $wl 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['roofline']['kernel_ms_per_run']; print('nib=$1 div=$2 $wl', d['value'], d['ms_per_step'], 'fp', k['cbf_first_pass'], 'rounds', k['cbf_round'], d['counters']['a_rounds'])
"
  done
done
