#!/usr/bin/env python3
"""development aid (round 4, review item 4): the filter build of job i + 1 beside stages B/C of job i
(gp_jobs_polish) with the two streams on DISJOINT sets of compute units (gp_ctx_create_masked ->
hipExtStreamCreateWithCUMask), against the same jobs with every stream on the whole chip and against
one job after the other.  usage: overlap_cu_probe.py N_CONTIGS_PER_JOB N_JOBS"""
import sys, time, pathlib, hashlib
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from goldpolish_amd import synth
n, n_jobs = int(sys.argv[1]), int(sys.argv[2])
w = synth.make_workload_chunked([25000] * (n * n_jobs), 40, seed=3, chunk=50)
from goldpolish_amd import core
p = core.default_params()
names = ("cbf_census", "cbf_local", "ntedit_scan", "ntedit_walk", "sealer_bfs", "mask")
N_CU = 256


def run(tag, front, back):
    """front / back: CU lists for the filter-build stream (job 0's context) and the back-half stream (job 1's)"""
    ctxs = [core.Context(0, cu_mask=front if i == 0 else back if i == 1 else None) for i in range(n_jobs)]
    jobs = [core.Job(ctxs[i], p, **core.slice_workload(w, n * i, n * (i + 1))) for i in range(n_jobs)]
    best = None
    for rep in range(3):
        t = time.time()
        core.polish_jobs(jobs)
        dt = time.time() - t
        best = dt if best is None or dt < best else best
    kt = {k: round(sum(j.kernel_time(k)[0] for j in jobs), 1) for k in names}
    h = hashlib.sha256(b"".join(x or b"-" for j in jobs for x in j.sequences())).hexdigest()[:12]
    print(f"{tag:46s} {best*1e3:8.1f} ms  {kt}  out {h}", flush=True)
    for j in jobs:
        j.close()
    for c in ctxs:
        c.close()


def one_after_the_other():
    ctx = core.Context(0)
    jobs = [core.Job(ctx, p, **core.slice_workload(w, n * i, n * (i + 1))) for i in range(n_jobs)]
    best = None
    for rep in range(3):
        t = time.time()
        for j in jobs:
            j.polish()
        dt = time.time() - t
        best = dt if best is None or dt < best else best
    print(f"{'one job after the other, whole chip':46s} {best*1e3:8.1f} ms", flush=True)
    for j in jobs:
        j.close()
    ctx.close()


one_after_the_other()
run("two streams, whole chip both", None, None)
allc = list(range(N_CU))
run("front 192 CUs [0,192) | back 64 [192,256)", allc[:192], allc[192:])
run("front CU%4!=3 | back CU%4==3", [c for c in allc if c % 4 != 3], [c for c in allc if c % 4 == 3])
run("front 224 | back 32 (CU%8==7)", [c for c in allc if c % 8 != 7], [c for c in allc if c % 8 == 7])
run("front 128 (even) | back 128 (odd)", allc[0::2], allc[1::2])
run("front all | back 64 [192,256)", None, allc[192:])
run("front 192 [0,192) | back all", allc[:192], None)
