#!/usr/bin/env python3
"""development aid: the filter build (stage A) of SURVEY 8d config 3's long-contig shape under a list of
environment settings, one job per setting -- wall time, the stage-A timers, reservation atomics, and a
hash of every filter (all settings must give the same bits).
usage: long_probe.py TOTAL_MBP|config3 "VAR=a,b" ["VAR2=x VAR3=y" ...]   (a spec with spaces sets several variables at once)"""
import hashlib
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import bench
from goldpolish_amd import synth

t = time.time()
if sys.argv[1] == "config3":   # BASELINE configs[2] (bench.py's headline workload) instead of the long contigs
    w = bench.make_workload("config3", seed=2)[0]
else:
    total = int(float(sys.argv[1]) * 1e6)
    lens = bench.survey_lengths(total, 2_000_000, 3)
    w = synth.make_workload_chunked(lens, 40, seed=2, chunk=1)
print(f"workload: {w.n_contigs} contigs, {w.total_bp} bp, generated in {time.time() - t:.0f} s", flush=True)
from goldpolish_amd import core
p = core.default_params()
names = ("cbf_census", "cbf_ckc", "cbf_local", "cbf_first_pass", "cbf_round")


def run(tag):
    with core.Context(0) as ctx, core.Job.from_workload(ctx, p, w) as job:
        best = None
        for rep in range(2):
            t0 = time.time()
            job.build_filters()
            dt = time.time() - t0
            if best is None or dt < best:
                best = dt
        kt = {k: round(job.kernel_time(k)[0], 1) for k in names}
        c = job.counters()
        h = hashlib.sha256()
        for b in range(0, w.n_contigs, max(1, w.n_contigs // 6)):
            for ki in range(4):
                h.update(job.get_filter(b, ki).tobytes())
        print(f"{tag:44s} {best*1e3:9.1f} ms  {kt}  ordered {c['a_ordered_ops']/max(c['a_ops'],1):.3f} "
              f"resv {c['a_reservations']/1e9:.2f} G rounds {c['a_rounds']}  filters {h.hexdigest()[:12]}", flush=True)


run("default")
for spec in sys.argv[2:]:
    if " " in spec:
        kv = dict(x.split("=") for x in spec.split())
        os.environ.update(kv)
        run(spec)
        for k in kv:
            del os.environ[k]
        continue
    var, vals = spec.split("=")
    for v in vals.split(","):
        os.environ[var] = v
        run(f"{var}={v}")
    del os.environ[var]
