#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel-trace/stats and PMC passes of bench.py.
# Outputs under gpurun_out/profiles_raw/; tools/collect_profiles.py condenses them into profiles/.
set -e -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/profiles_raw
rm -rf $OUT && mkdir -p $OUT
CMD="python3 bench.py --steps 5 --warmup 1 --no-aux --no-cpu-baseline"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/config2_trace -- $CMD > $OUT/config2_trace.log 2>&1
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/config2_fetch -- $CMD > $OUT/config2_fetch.log 2>&1
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/config2_write -- $CMD > $OUT/config2_write.log 2>&1
GCMD="python3 bench.py --steps 2 --warmup 1 --no-aux --no-cpu-baseline --workload goldenpath"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/goldenpath_trace -- $GCMD > $OUT/goldenpath_trace.log 2>&1
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/goldenpath_fetch -- $GCMD > $OUT/goldenpath_fetch.log 2>&1
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/goldenpath_write -- $GCMD > $OUT/goldenpath_write.log 2>&1
# the bench line of the same command without the profiler, for the agreement check
timeout -k 10 300 $CMD > $OUT/config2_bench.json 2> $OUT/config2_bench.err
timeout -k 10 300 $GCMD > $OUT/goldenpath_bench.json 2> $OUT/goldenpath_bench.err
# big traces stay on the box; keep the stats and counter tables
find $OUT -name "*kernel_trace.csv" -size +20M -delete
ls -la $OUT/*/*/ | head -40
