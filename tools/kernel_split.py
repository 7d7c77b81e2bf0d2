#!/usr/bin/env python3
"""development aid: polish a cached workload a few times (for a run under rocprofv3 --kernel-trace --stats,
which must not fork: make the cache first, without the profiler)
usage: kernel_split.py --make PATH N_CONTIGS LEN COV   |   kernel_split.py PATH [REPS]"""
import sys, time, pathlib, hashlib
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from goldpolish_amd import synth
if sys.argv[1] == "--make":
    path, n, L, cov = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
    synth.save_workload(synth.make_workload_chunked([L] * n, cov, seed=3, chunk=50), path)
    sys.exit(0)
w = synth.load_workload(sys.argv[1])
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
from goldpolish_amd import core
names = ("cbf_census", "cbf_local", "ntedit_scan", "ntedit_walk", "sealer_bfs", "mask")
with core.Context(0) as ctx, core.Job.from_workload(ctx, core.default_params(), w) as job:
    for rep in range(reps):
        t = time.time()
        job.polish()
        dt = time.time() - t
        print(f"{dt*1e3:8.1f} ms", {k: round(job.kernel_time(k)[0], 1) for k in names}, flush=True)
    print("out", hashlib.sha256(b"".join(x or b"-" for x in job.sequences())).hexdigest()[:12])
