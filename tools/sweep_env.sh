#!/bin/bash
# run bench.py once per environment setting given on the command line ("A=1 B=2" strings)
WL=${WL:-goldenpath}
for e in "$@"; do
  env $e python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-aux --workload $WL 2>/dev/null | python3 -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); k=d['roofline']['kernel_ms_per_run']; print('[$e] $WL', d['value'], d['ms_per_step'], 'fp', k['cbf_first_pass'], 'rounds', k['cbf_round'], d['counters']['a_rounds'], flush=True)
"
done
