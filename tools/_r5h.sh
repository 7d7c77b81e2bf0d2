D=gpurun_out/r5h; mkdir -p $D; export PYTHONUNBUFFERED=1
python tools/kernel_split.py --make /tmp/w1000.npz 1000 25000 40 < /dev/null || exit 1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
export GP_STAGE_A_CENSUS_OVERLAP=0
for v in 1 4; do
  GP_STAGE_A_SCATTER_ROUNDS=$v timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$D/p$v -- python3 $R/tools/kernel_split.py /tmp/w1000.npz 3 < /dev/null > $R/$D/run$v.log 2>&1 || { echo "variant $v failed"; tail -5 $R/$D/run$v.log; exit 1; }
  grep "ms {" $R/$D/run$v.log | cut -c1-100
  f=$(ls $R/$D/p$v/*/*kernel_stats.csv 2>/dev/null | tail -1)
  [ -n "$f" ] || { echo "no stats file"; exit 1; }
  cp "$f" $R/$D/stats$v.csv; rm -rf $R/$D/p$v
  grep census $R/$D/stats$v.csv | cut -d, -f1-4 | cut -c1-120
done
