#!/bin/bash
# Runs on the GPU box (via gpurun): the round's closing run -- every GPU test, smoke(), the profile set
# (tools/make_profiles.sh) and the driver's bench command; everything into gpurun_out/$1/.
D=gpurun_out/${1:-final}; mkdir -p $D; export PYTHONUNBUFFERED=1
timeout -k 10 600 python -m pytest tests -m gpu -x -q < /dev/null > $D/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $D/pytest.log | cut -c1-200
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" < /dev/null > $D/smoke.log 2>&1; echo "smoke rc=$?"
bash tools/make_profiles.sh < /dev/null > $D/make_profiles.log 2>&1; echo "profiles rc=$?"
timeout -k 10 500 python bench.py --steps 20 --warmup 5 --cache /tmp/gp_bench_cache < /dev/null > $D/bench.json 2> $D/bench.err; echo "bench rc=$?"
python3 - $D/bench.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print("value", d["value"], "ms", d["ms_per_step"], "frac", d["roofline"]["frac"], d["roofline"]["kernel_ms_per_run"])
print("resident", d.get("hbm_resident", {}).get("value"), "cpu", d["cpu_baseline"]["value"], "GiB", d["config"].get("device_GiB_in_use"),
      "aux", d.get("aux", {}).get("value"), "upload ms", d["stream"]["upload_ms"], d.get("output_checksum"))
PY
