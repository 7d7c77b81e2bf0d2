#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace/stats and PMC passes of bench.py.
# Outputs under gpurun_out/profiles_raw/; tools/collect_profiles.py condenses them into profiles/.
# Counters are collected in runs of their own (FETCH_SIZE and WRITE_SIZE do not fit one pass, and
# --pmc is never combined with a trace domain); every profiled command is the python program itself.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_raw
rm -rf $OUT && mkdir -p $OUT
# The workloads are generated first, by an unprofiled run that only writes them to a cache: the
# generator forks worker processes, which a process rocprofv3 has already GPU-initialised must not
# do (round 2's profiles were full of "Aborted ... SIGTERM" from the torn-down pool).
CACHE=/tmp/gp_bench_cache   # (not under gpurun_out/: 4 GB of generated reads must not travel back)
python3 bench.py --prepare-only --cache $CACHE || exit 1
python3 bench.py --prepare-only --cache $CACHE --workload config2 || exit 1
# the headline workload (BASELINE.json configs[2]); two steps keep the traces small
CMD="python3 bench.py --steps 2 --warmup 1 --no-aux --no-cpu-baseline --resident-steps 0 --cache $CACHE"
timeout -k 10 500 $CMD > $OUT/config3_bench.json 2> $OUT/config3_bench.err || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/config3_trace -- $CMD > $OUT/config3_trace.log 2>&1 || exit 1
timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/config3_fetch -- $CMD > $OUT/config3_fetch.log 2>&1 || exit 1
timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/config3_write -- $CMD > $OUT/config3_write.log 2>&1 || exit 1
# configs[1] (aux in the bench line)
C2="python3 bench.py --steps 5 --warmup 1 --no-aux --no-cpu-baseline --resident-steps 0 --workload config2 --cache $CACHE"
timeout -k 10 300 $C2 > $OUT/config2_bench.json 2> $OUT/config2_bench.err || exit 1
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/config2_trace -- $C2 > $OUT/config2_trace.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/config2_fetch -- $C2 > $OUT/config2_fetch.log 2>&1 || exit 1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/config2_write -- $C2 > $OUT/config2_write.log 2>&1 || exit 1
# SURVEY 8d config 3's own shape (aux_survey3 in the bench line): kernel stats only; the pipeline holds four such
# sets, so three warm-up collects before the two timed steps (steady state)
python3 bench.py --prepare-only --cache $CACHE --workload survey3 || exit 1
S3="python3 bench.py --steps 2 --warmup 3 --no-aux --no-cpu-baseline --resident-steps 0 --workload survey3 --cache $CACHE"
timeout -k 10 500 $S3 > $OUT/survey3_bench.json 2> $OUT/survey3_bench.err || exit 1
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/survey3_trace -- $S3 > $OUT/survey3_trace.log 2>&1 || exit 1
# what FETCH_SIZE / WRITE_SIZE report for THIS access pattern (random 1-byte loads / load+store
# pairs over 16 GiB, 2^28 of them per launch): the calibration MI355X_MICROARCH.md asks for before
# an absolute is trusted
if [ -x tools/probes/bin/random_rmw_probe ]; then
  timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/probe_fetch -- tools/probes/bin/random_rmw_probe 34 > $OUT/probe_fetch.log 2>&1
  timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/probe_write -- tools/probes/bin/random_rmw_probe 34 > $OUT/probe_write.log 2>&1
fi
# big traces stay on the box; keep the stats and counter tables
find $OUT -name "*kernel_trace.csv" -size +20M -delete
ls $OUT
