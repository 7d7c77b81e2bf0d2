#!/usr/bin/env python3
"""development aid: long contigs walked whole (GP_WALK_SEG=0) and as segments must give the same
bytes and the same counters"""
import os, sys, time, pathlib, hashlib
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from goldpolish_amd import synth
lens = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1000000,1300000,600000,2000000,200000,30000").split(",")]
w = synth.make_workload_chunked(lens, 40, seed=3, chunk=1)
from goldpolish_amd import core
p = core.default_params()
ref = None
for env in ({"GP_WALK_SEG": "0"}, {}, {"GP_WALK_SEG": "16384"}, {"GP_WALK_SEG": "16384", "GP_WALK_SEG_FORCE": "1"},
            {"GP_WALK_SEG": "40000"}):
    for k in ("GP_WALK_SEG", "GP_WALK_SEG_FORCE"):
        os.environ.pop(k, None)
    os.environ.update(env)
    with core.Context(0) as ctx, core.Job.from_workload(ctx, p, w) as job:
        best = 1e9
        for rep in range(2):
            t = time.time()
            job.polish()
            best = min(best, time.time() - t)
        h = hashlib.sha256(b"".join(x or b"-" for x in job.sequences())).hexdigest()[:16]
        c = job.counters()
        key = (h, c["b_sites"], c["b_probes"], c["b_edits"], c["c_closed"])   # (b_scan_kmers counts look-ahead k-mers of scans too: an implementation counter, it moves with the scan lengths)
        if ref is None:
            ref = key
        print(f"{str(env):56s} {best*1e3:8.1f} ms walk {job.kernel_time('ntedit_walk')[0]:8.1f} ms  {key}  {'SAME' if key == ref else 'DIFFERENT'}", flush=True)
