#!/usr/bin/env python3
"""Regenerate tests/golden/pipeline_small.json -- SELF-GENERATED vectors.

These pin the oracle (and through it the HIP path) against silent drift between
rounds.  They are NOT reference outputs: the reference's stage A / B / C need
btllib and boost, which are absent here, so the semantics they encode are the
restated ones documented in oracle/*.c (counting filter returns the count after
insertion; common edges visited in insertion order; ...).
"""
import hashlib
import json
import pathlib
import sys

import numpy as np

root = pathlib.Path(__file__).resolve().parents[2]
sys.path.insert(0, str(root))
sys.path.insert(0, str(root / "tests"))
import oracle_lib  # noqa: E402
from goldpolish_amd import synth  # noqa: E402

O = oracle_lib.Oracle(str(root / "oracle" / "libgpo.so"))


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


cases = []
for name, lens, cov, seed, ks, extra in [
    ("three_contigs_full_cascade", [6000, 3000, 4500], 25, 101, (32, 28, 24, 20), {}),
    ("k32_only", [8000], 30, 102, (32,), {}),
    ("noisy_draft", [5000, 5000], 35, 103, (32, 28, 24, 20), {"draft_err": 0.04}),
]:
    w = synth.make_workload(lens, cov, seed=seed, **extra)
    p = O.batch_params(k_values=ks)
    rec = {"name": name, "lens": lens, "coverage": cov, "seed": seed, "k": list(ks), "extra": extra,
           "input_sha": sha(w.contig_buf.tobytes() + w.read_buf.tobytes()), "contigs": []}
    for c in range(w.n_contigs):
        reads = w.reads_of(c)
        thr = int(w.kmer_threshold[c])
        _, bfs = O.fill_bfs(reads, list(ks), thr, p.cbf_bytes, p.bf_bytes, p.hash_num)
        cur = w.contig(c)
        stages = {"filters": [sha(b.tobytes()) for b in bfs],
                  "filter_bits": [int(np.unpackbits(b).sum()) for b in bfs], "ntedit": []}
        for ki, k in enumerate(ks):
            cur, _ = O.ntedit(cur, bfs[ki], k)
            stages["ntedit"].append({"sha": sha(cur), "len": len(cur)})
        m = O.mask(cur, ks[0])
        stages["mask"] = {"sha": sha(m), "len": len(m)}
        s, st = O.sealer(m, list(ks), bfs)
        stages["sealer"] = {"sha": sha(s), "len": len(s), "gaps_found": st.gaps_found,
                            "gaps_closed": st.gaps_closed, "vertex_tests": st.vertex_tests}
        outs, bst, _ = O.polish_batch([w.names[c]], [w.contig(c)], [reads], [thr], p)
        assert outs[0] == s.upper()
        stages["final"] = {"sha": sha(outs[0]), "len": len(outs[0])}
        stages["kmer_threshold"] = thr
        rec["contigs"].append(stages)
    cases.append(rec)
(root / "tests" / "golden" / "pipeline_small.json").write_text(json.dumps(cases, indent=1))
print("wrote", len(cases), "cases")
