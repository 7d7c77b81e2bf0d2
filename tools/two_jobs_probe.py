"""How many jobs in flight per GPU?  The same batches as one job and as 2..N jobs polished side by
side (gp_jobs_polish: one host thread and one stream each): the memory-bound filter build of one
overlaps the latency-bound ntEdit walk / Sealer search of another.  Dev probe.
usage: two_jobs_probe.py [n_contigs] [max_parts] [coverage]"""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from goldpolish_amd import synth

n_contigs = int(sys.argv[1]) if len(sys.argv) > 1 else 400
max_parts = int(sys.argv[2]) if len(sys.argv) > 2 else 4
cov = float(sys.argv[3]) if len(sys.argv) > 3 else 30
w = synth.make_workload_chunked([25_000] * n_contigs, cov, seed=2, chunk=50)
from goldpolish_amd import core
p = core.default_params()
ref = None
for parts in [x for x in (1, 2, 3, 4, 6, 8) if x <= max_parts]:
    ctxs = [core.Context(0) for _ in range(parts)]
    cuts = [n_contigs * i // parts for i in range(parts + 1)]
    jobs = [core.Job(ctxs[i], p, **core.slice_workload(w, cuts[i], cuts[i + 1])) for i in range(parts)]
    best = 1e9
    for rep in range(3):
        t = time.perf_counter(); core.polish_jobs(jobs); best = min(best, time.perf_counter() - t)
    out = [s for j in jobs for s in j.sequences()]
    if ref is None:
        ref = out
    print(f"{parts} job(s) in flight: {best*1e3:.0f} ms = {w.total_bp/best/1e6:.1f} Mbp/s; same output: {out == ref}; "
          f"device GiB {core.device_bytes_in_use()/2**30:.1f}", flush=True)
    for j in jobs: j.close()
    for c in ctxs: c.close()
