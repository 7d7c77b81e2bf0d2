#!/bin/bash
# sweep stage A knobs (window load, tail hand-over length) on the headline workload; each
# workload is generated once per process, so this takes a while
for d in ${DIVS:-3 4 6 8}; do
  for t in ${TAILS:-4096 16384}; do
    for wl in ${WLS:-config2}; do
    GP_STAGE_A_LOAD_DIV=$d GP_STAGE_A_TAIL=$t python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-aux --workload $wl 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['roofline']['kernel_ms_per_run']; print('div=$d tail=$t $wl', d['value'], d['ms_per_step'], 'fp', k['cbf_first_pass'], 'rounds', k['cbf_round'], d['counters']['a_rounds'], flush=True)
"
    done
  done
done
