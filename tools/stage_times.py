#!/usr/bin/env python3
"""development aid: stage timers of one job under a list of environment settings
usage: stage_times.py N_CONTIGS LEN COV "VAR=a,b,c" ["VAR2=x,y"]  (settings are tried one variable at a time)"""
import os, sys, time, pathlib, hashlib
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from goldpolish_amd import synth
n, L, cov = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
w = synth.make_workload_chunked([L] * n, cov, seed=3, chunk=50)
from goldpolish_amd import core
p = core.default_params()
names = ("cbf_census", "cbf_ckc", "cbf_local", "cbf_first_pass", "cbf_round", "ntedit_scan", "ntedit_walk", "sealer_bfs", "mask")
def run(tag):
    with core.Context(0) as ctx, core.Job.from_workload(ctx, p, w) as job:
        best = None
        for rep in range(3):
            t = time.time()
            job.polish()
            dt = time.time() - t
            kt = {k: round(job.kernel_time(k)[0], 1) for k in names}
            if best is None or dt < best[0]:
                best = (dt, kt)
        h = hashlib.sha256(b"".join(x or b"-" for x in job.sequences())).hexdigest()[:12]
        print(f"{tag:32s} {best[0]*1e3:8.1f} ms  {best[1]}  out {h}", flush=True)
run("default")
for spec in sys.argv[4:]:
    var, vals = spec.split("=")
    for v in vals.split(","):
        os.environ[var] = v
        run(f"{var}={v}")
    del os.environ[var]
