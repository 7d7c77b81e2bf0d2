"""Dev probe for the walker's pass-count trap seen with three jobs in flight: which ingredient
matters?  usage: jobs_race_probe.py <parts> <poison 0|1> [reps]   (env picks modes)"""
import pathlib, sys, time
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
from goldpolish_amd import synth
parts = int(sys.argv[1]); poison = int(sys.argv[2]); reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
n_contigs = 1600
w = synth.make_workload_chunked([25_000] * n_contigs, 40, seed=2, chunk=50)
from goldpolish_amd import core
import ctypes as C
if poison:
    # recycled device memory full of ones: whatever is read before it is written shows
    import torch
    t = [torch.full((1 << 30,), 0xFF, dtype=torch.uint8, device="cuda") for _ in range(150)]
    torch.cuda.synchronize(); del t; torch.cuda.empty_cache()
p = core.default_params()
ctxs = [core.Context(0) for _ in range(parts)]
cuts = [n_contigs * i // parts for i in range(parts + 1)]
jobs = [core.Job(ctxs[i], p, **core.slice_workload(w, cuts[i], cuts[i + 1])) for i in range(parts)]
outs = []
for rep in range(reps):
    try:
        t = time.perf_counter(); core.polish_jobs(jobs); dt = time.perf_counter() - t
        out = [s for j in jobs for s in j.sequences()]
        outs.append(out)
        print(f"parts {parts} poison {poison} rep {rep}: {dt*1e3:.0f} ms, same as rep 0: {out == outs[0]}", flush=True)
    except core.GpError as e:
        print(f"parts {parts} poison {poison} rep {rep}: ERROR {e}", flush=True)
import hashlib
if outs:
    print("digest", hashlib.sha256(b"|".join(x or b"-" for x in outs[0])).hexdigest()[:16])
