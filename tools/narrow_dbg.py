#!/usr/bin/env python3
"""development aid: counters of the census path under the two record geometries"""
import os, sys, pathlib
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from goldpolish_amd import synth, core
import numpy as np
os.environ.update(GP_STAGE_A_LOCAL="1", GP_STAGE_A_CENSUS="1", GP_STAGE_A_CENSUS_MIN="1")
w = synth.make_workload([30000, 12000, 21000, 800], 30, seed=23)
p = core.default_params()
res = {}
for ckc in ("0", "1"):
    for wide in ("1", "0"):
        os.environ["GP_STAGE_A_CENSUS_WIDE"] = wide
        os.environ["GP_STAGE_A_CKC"] = ckc
        with core.Context(0) as ctx, core.Job.from_workload(ctx, p, w) as job:
            job.build_filters()
            c = job.counters()
            f = [job.get_filter(b, ki).copy() for b in range(4) for ki in range(4)]
        res[(ckc, wide)] = f
        print("ckc", ckc, "wide", wide, {k: c[k] for k in ("a_ops", "a_ordered_ops", "a_replayed", "a_settled_hits", "a_counted", "a_rounds")},
              [int(np.unpackbits(x).sum()) for x in f], flush=True)
for ckc in ("0", "1"):
    a, b = res[(ckc, "1")], res[(ckc, "0")]
    for i, (x, y) in enumerate(zip(a, b)):
        d = np.flatnonzero(x != y)
        if d.size:
            print("ckc", ckc, "filter", i // 4, i % 4, "differs at bytes", d[:10], "wide", x[d[:10]], "narrow", y[d[:10]])
