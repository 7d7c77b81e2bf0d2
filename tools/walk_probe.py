#!/usr/bin/env python3
"""development aid: a few mid-size contigs (the long-contig regime's slow ones) through the path,
with the walker's per-request clocks (GP_WALK_PROFILE=1) and the stage timers"""
import os, sys, time, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from goldpolish_amd import synth
if len(sys.argv) > 1 and sys.argv[1] == "survey3":  # SURVEY 8d config 3's shape, as bench.py's aux_survey3
    import bench
    lens = bench.survey_lengths(100_000_000, 2_000_000, 3)
    w = synth.make_workload_chunked(lens, 40, seed=2, chunk=1)
else:
    lens = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1000000,1300000,600000,2000000").split(",")]
    w = synth.make_workload_chunked(lens, 40, seed=3, chunk=1)
from goldpolish_amd import core
p = core.default_params()
with core.Context(0) as ctx, core.Job.from_workload(ctx, p, w) as job:
    for rep in range(2):
        t = time.time()
        job.polish()
        print("polish", round(time.time() - t, 3), "s", {n: round(job.kernel_time(n)[0], 1) for n in
              ("cbf_first_pass", "cbf_round", "ntedit_scan", "ntedit_walk", "sealer_bfs", "mask")}, flush=True)
    print(job.counters())
