#!/usr/bin/env python3
"""development aid: does the census of one job overlap the candidate replay of another?
Two jobs (halves of N contigs) build their filters serially, together, and with the second
started when the first has finished its census.  usage: overlap_a_probe.py N LEN COV"""
import os, sys, time, threading, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from goldpolish_amd import synth
n, L, cov = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
w = synth.make_workload_chunked([L] * n, cov, seed=3, chunk=50)
from goldpolish_amd import core
p = core.default_params()
h = n // 2
def measure(tag):
    ctxs = [core.Context(0), core.Context(0)]
    jobs = [core.Job(ctxs[0], p, **core.slice_workload(w, 0, h)), core.Job(ctxs[1], p, **core.slice_workload(w, h, n))]
    try:
        for j in jobs:            # warm-up: buffers, pool
            j.build_filters()
        t = time.time(); jobs[0].build_filters(); jobs[1].build_filters(); serial = time.time() - t
        cen = jobs[0].kernel_time("cbf_census")[0] / 1e3
        loc = jobs[0].kernel_time("cbf_local")[0] / 1e3
        res = {}
        for name, delay in (("together", 0.0), ("offset by the census", cen), ("offset by half", (cen + loc) / 2)):
            best = None
            for rep in range(2):
                def second():
                    time.sleep(delay)
                    jobs[1].build_filters()
                th = threading.Thread(target=second)
                t = time.time(); th.start(); jobs[0].build_filters(); th.join(); dt = time.time() - t
                best = dt if best is None else min(best, dt)
            res[name] = round(best * 1e3, 1)
        print(f"{tag:28s} one job: census {cen*1e3:.1f} + replay {loc*1e3:.1f} ms; serial {serial*1e3:.1f} ms; two threads: {res}", flush=True)
    finally:
        for j in jobs: j.close()
        for c in ctxs: c.close()
measure("default")
for spec in sys.argv[4:]:
    var, vals = spec.split("=")
    for v in vals.split(","):
        os.environ[var] = v
        measure(f"{var}={v}")
    del os.environ[var]
