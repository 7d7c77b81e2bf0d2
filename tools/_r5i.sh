D=gpurun_out/r5i; mkdir -p $D; export PYTHONUNBUFFERED=1
timeout -k 10 400 python -m pytest tests/test_gpu_stage_a.py tests/test_gpu_fuzz.py tests/test_gpu_fullsize.py -m gpu -x -q < /dev/null > $D/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $D/pytest.log | cut -c1-200
[ $rc -eq 0 ] || exit 1
show() { python3 -c "
import json,sys
d=json.load(open('$1'))
print('$1', d['value'], d['ms_per_step'], {k:round(v,1) for k,v in d['roofline']['kernel_ms_per_run'].items() if v}, d['roofline']['frac'], d.get('output_checksum'))
"; }
timeout -k 10 500 python bench.py --steps 4 --warmup 2 --no-aux --no-cpu-baseline --resident-steps 0 --cache /tmp/gpc < /dev/null > $D/bench1.json 2> $D/bench1.err || { echo bench1 failed; tail -5 $D/bench1.err; exit 1; }
show $D/bench1.json
GP_STAGE_A_CENSUS_OVERLAP=0 timeout -k 10 300 python bench.py --steps 4 --warmup 2 --no-aux --no-cpu-baseline --resident-steps 0 --cache /tmp/gpc < /dev/null > $D/bench0.json 2> $D/bench0.err || { echo bench0 failed; tail -5 $D/bench0.err; exit 1; }
show $D/bench0.json
