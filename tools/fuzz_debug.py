#!/usr/bin/env python3
"""Stage-by-stage comparison of one fuzz seed (dev tool): fuzz_debug.py <seed> [local_mode]"""
import os, pathlib, sys
ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
import numpy as np
import fuzz_parity
from goldpolish_amd import core
import oracle_lib

seed = int(sys.argv[1]); mode = sys.argv[2] if len(sys.argv) > 2 else "0"
os.environ["GP_STAGE_A_LOCAL"] = mode
O = oracle_lib.Oracle(str(ROOT / "oracle" / "libgpo.so"))
w, ks, (cbf, bf), bcb, extra = fuzz_parity.random_case(seed)
print("lens", [int(w.contig_off[i + 1] - w.contig_off[i]) for i in range(w.n_contigs)], "ks", ks, "cbf/bf", cbf, bf, "bcb", bcb, extra, "thr", w.kmer_threshold)
p = core.default_params(k_values=ks, cbf_bytes=cbf, bf_bytes=bf, **extra)


def diff(a, b):
    if a is None or b is None:
        return (a is None, b is None)
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return i, a[max(0, i - 30):i + 30], b[max(0, i - 30):i + 30], len(a), len(b)
    return n, len(a), len(b)


contig = lambda c: bytes(w.contig_buf[int(w.contig_off[c]):int(w.contig_off[c + 1])])
with core.Context(0) as ctx, core.Job(ctx, p, contig_buf=w.contig_buf, contig_off=w.contig_off,
                                      contig_read_begin=w.contig_read_begin, kmer_threshold=w.kmer_threshold,
                                      read_buf=w.read_buf, read_off=w.read_off, batch_contig_begin=bcb,
                                      name_len=[len(x) for x in w.names]) as job:
    job.build_filters()
    filt = [[job.get_filter(b, ki).copy() for ki in range(len(ks))] for b in range(len(bcb) - 1)]
    batch_of = {c: b for b in range(len(bcb) - 1) for c in range(bcb[b], bcb[b + 1])}
    cur = [contig(c) for c in range(w.n_contigs)]
    for ki, k in enumerate(ks):
        job.ntedit_pass(ki)
        got = job.sequences()
        for c in range(w.n_contigs):
            if cur[c] is None or len(cur[c]) < p.min_contig_len:
                cur[c] = None
                continue
            q = O.ntedit_params(k, p.hash_num, max_insertions=p.max_insertions, max_deletions=p.max_deletions,
                                missing_ratio=p.missing_ratio, edit_ratio=p.edit_ratio, jump=p.jump, mask=p.soft_mask)
            want, _ = O.ntedit(cur[c], filt[batch_of[c]][ki], k, p.hash_num, q)
            if got[c] != want:
                print("NTEDIT MISMATCH k", k, "contig", c, diff(got[c], want)); sys.exit(1)
            cur[c] = want
    print("ntedit cascade ok")
    job.ntedit_guard(); g = job.sequences()
    print("after guard lens", [None if x is None else len(x) for x in g])
    job.mask(); m = job.sequences()
    for c in range(w.n_contigs):
        if g[c] is not None and m[c] != O.mask(g[c], ks[0]):
            print("MASK MISMATCH contig", c, diff(m[c], O.mask(g[c], ks[0]))); sys.exit(1)
    print("mask ok")
    job.sealer(); s = job.sequences()
    for c in range(w.n_contigs):
        if m[c] is None:
            continue
        want, st = O.sealer(m[c], ks, filt[batch_of[c]], p.hash_num, max_paths=p.sealer_max_paths, max_branches=p.sealer_max_branches)
        if s[c] != want:
            print("SEALER MISMATCH contig", c, diff(s[c], want), "stats", st.gaps_found if hasattr(st, "gaps_found") else ""); sys.exit(1)
    print("sealer ok")
