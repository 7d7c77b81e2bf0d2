"""Ad-hoc probe: time stage A on a synthetic workload (not the official bench)."""
import argparse, sys, time, pathlib
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import numpy as np
from goldpolish_amd import core, synth

ap = argparse.ArgumentParser()
ap.add_argument("--contigs", type=int, default=1)
ap.add_argument("--len", type=int, default=10_000_000)
ap.add_argument("--cov", type=float, default=30)
ap.add_argument("--k", type=str, default="32")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
t = time.time()
w = synth.make_workload([a.len] * a.contigs, a.cov, seed=2)
print(f"gen {time.time()-t:.1f}s contigs={w.n_contigs} bp={w.total_bp} reads={len(w.read_off)-1} "
      f"read_bases={int(w.read_off[-1])} thr={w.kmer_threshold[:4]}", flush=True)
p = core.default_params(k_values=[int(x) for x in a.k.split(",")])
with core.Context(0) as ctx:
    t = time.time()
    job = core.Job.from_workload(ctx, p, w)
    print(f"upload {time.time()-t:.2f}s", flush=True)
    for r in range(a.reps):
        t = time.time()
        job.build_filters()
        dt = time.time() - t
        c = job.counters()
        fp = job.kernel_time("cbf_first_pass")
        rd = job.kernel_time("cbf_round")
        print(f"rep{r}: {dt*1e3:.1f} ms  -> {w.total_bp/dt/1e6:.1f} Mbp/s (stage A only) "
              f"first_pass {fp[0]:.2f} ms/{fp[1]} launches, rounds {rd[0]:.2f} ms, counters {c}", flush=True)
    f = job.get_filter(0, 0)
    print("bf fill", np.unpackbits(f).mean())
    job.close()
