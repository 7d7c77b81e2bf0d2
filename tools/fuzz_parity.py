#!/usr/bin/env python3
"""Randomised whole-path parity: many small random jobs (contig shapes, error rates, k lists,
filter sizes, thresholds, batch grouping, ntEdit / Sealer parameters, soft-masked and N-holding
drafts) through libgpcore on the GPU against the C oracle, stage by stage.  Dev tool; a few of
its seeds run in tests/test_gpu_fuzz.py.  Usage: fuzz_parity.py [n_cases] [first_seed]"""
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import numpy as np


def random_case(seed):
    from goldpolish_amd import synth
    rng = np.random.default_rng(seed)
    n = int(rng.integers(1, 7))
    lens = [int(rng.choice([90, 150, 400, 1500, 4000, 9000, 20000], p=[.05, .05, .1, .2, .3, .2, .1])) for _ in range(n)]
    cov = float(rng.choice([3, 8, 15, 25, 40]))
    w = synth.make_workload(lens, cov, seed=seed, draft_err=float(rng.choice([0.0, 0.003, 0.01, 0.03, 0.08])),
                            read_err=float(rng.choice([0.0, 0.02, 0.06, 0.12])),
                            read_mean=float(rng.choice([300, 2000, 15000])), read_min=int(rng.choice([25, 200, 1000])))
    # sprinkle N runs, lower case and IUPAC letters into some drafts
    buf = w.contig_buf.copy()
    for c in range(w.n_contigs):
        a, b = int(w.contig_off[c]), int(w.contig_off[c + 1])
        mode = rng.integers(0, 4)
        if mode == 1 and b - a > 200:
            for _ in range(int(rng.integers(1, 4))):
                s = int(rng.integers(a, b - 60)); L = int(rng.integers(1, 50))
                buf[s:s + L] = ord("N")
        elif mode == 2 and b - a > 200:
            s = int(rng.integers(a, b - 100)); L = int(rng.integers(5, 90))
            buf[s:s + L] = np.frombuffer(bytes(buf[s:s + L]).lower(), dtype=np.uint8)
        elif mode == 3 and b - a > 50:
            for _ in range(3):
                buf[int(rng.integers(a, b))] = ord(rng.choice(list("RYSWKMBDHVn")))
    w.contig_buf = buf
    if rng.random() < 0.35 and w.read_buf.size > 400:
        # low-complexity stretches (homopolymers, short tandem repeats) in reads and drafts: the same
        # k-mer many times in a row, i.e. long chains of operations on the same counters
        rb = w.read_buf.copy()
        for target in (rb, buf):
            for _ in range(int(rng.integers(1, 6))):
                L = int(rng.integers(20, 300))
                if target.size <= L + 1:
                    continue
                s0 = int(rng.integers(0, target.size - L))
                unit = [b"A", b"T", b"AC", b"GGC", b"ACGT", b"AAAAAAAAAC"][int(rng.integers(0, 6))]
                target[s0:s0 + L] = np.frombuffer((unit * (L // len(unit) + 1))[:L], dtype=np.uint8)
        w.read_buf = rb
        w.contig_buf = buf
    ks = [[32, 28, 24, 20], [32], [40, 25], [31, 21, 11], [63, 33], [24, 20, 16, 12], [20]][int(rng.integers(0, 7))]
    sizes = [(10 * 1024 * 1024, 512 * 1024), (65536, 8192), (4099, 1024), (1 << 20, 1 << 16)][int(rng.integers(0, 4))]
    # batch grouping: singletons, pairs, or everything in one batch
    g = int(rng.integers(0, 3))
    if g == 0:
        bcb = list(range(n + 1))
    elif g == 1:
        bcb = sorted(set([0, n] + [int(x) for x in rng.integers(0, n + 1, size=2)]))
    else:
        bcb = [0, n]
    if rng.random() < 0.4 and w.read_buf.size > 200:
        # reads are not clean either: N runs, lower-case stretches, IUPAC letters
        rb = w.read_buf.copy()
        for _ in range(int(rng.integers(1, 12))):
            s0 = int(rng.integers(0, rb.size - 60))
            L = int(rng.integers(1, 60))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                rb[s0:s0 + L] = ord("N")
            elif kind == 1:
                rb[s0:s0 + L] = np.frombuffer(bytes(rb[s0:s0 + L]).lower(), dtype=np.uint8)
            else:
                rb[s0] = ord(rng.choice(list("RYKMSWnacgt")))
        w.read_buf = rb
    hash_num = int(rng.choice([1, 2, 3, 4, 4, 4, 5, 8]))
    extra = dict(hash_num=hash_num, missing_ratio=float(rng.choice([0.3, 0.5, 0.8])), edit_ratio=float(rng.choice([0.3, 0.5, 0.7])),
                 sealer_max_paths=int(rng.choice([1, 2, 10])), sealer_max_branches=int(rng.choice([10, 1000])))
    if rng.random() < 0.3:
        w.kmer_threshold[:] = int(rng.integers(4, 14))
    if seed >= 400000:
        # (seeds below keep the cases they always had)  Contigs of one batch with DIFFERENT
        # thresholds, ascending or interleaved, on small counting filters so that their counters
        # collide: the settle shortcut of stage A may only use the batch's largest threshold
        r2 = np.random.default_rng(seed ^ 0x5EED)
        if r2.random() < 0.7:
            t = r2.integers(4, 14, size=n)
            w.kmer_threshold[:] = np.sort(t) if r2.random() < 0.5 else t
            bcb = [0, n] if r2.random() < 0.6 else sorted(set([0, n, int(r2.integers(0, n + 1))]))
            sizes = [(4099, 1024), (65536, 8192), (20011, 4096)][int(r2.integers(0, 3))]
    return w, ks, sizes, bcb, extra


def set_schedule(mode):
    """"0": grid-wide schedule; "1": cbf_local replaying every operation; "2": cbf_local behind the
    census of first touches (forced for batches of any length); "3": the census level by level"""
    os.environ["GP_STAGE_A_LOCAL"] = "0" if mode in ("0", "3") else "1"
    os.environ["GP_STAGE_A_CENSUS"] = "1" if mode == "2" else "0"
    os.environ["GP_STAGE_A_CENSUS_MIN"] = "1"
    # "3": the census level by level for every batch that qualifies (one threshold, hash_num <= 4), whatever its
    # load; the others of the job take the grid-wide schedule
    os.environ["GP_STAGE_A_LEVELS"] = "1" if mode == "3" else "0"
    os.environ["GP_STAGE_A_LEVELS_LOAD"] = "1000000"


def run_case(core, ctx, O, seed, local_mode):
    w, ks, (cbf, bf), bcb, extra = random_case(seed)
    set_schedule(local_mode)
    p = core.default_params(k_values=ks, cbf_bytes=cbf, bf_bytes=bf, **extra)
    names = w.names
    with core.Job(ctx, p, contig_buf=w.contig_buf, contig_off=w.contig_off, contig_read_begin=w.contig_read_begin,
                  kmer_threshold=w.kmer_threshold, read_buf=w.read_buf, read_off=w.read_off,
                  batch_contig_begin=bcb, name_len=[len(x) for x in names]) as job:
        job.polish()
        got = job.sequences()
        filt = [[job.get_filter(b, ki).copy() for ki in range(len(ks))] for b in range(len(bcb) - 1)]
    # the oracle's batch driver takes the same knobs under its own field names
    op = O.batch_params(k_values=tuple(ks), cbf_bytes=cbf, bf_bytes=bf, ntedit_x=extra["missing_ratio"],
                        ntedit_y=extra["edit_ratio"], sealer_P=extra["sealer_max_paths"],
                        sealer_B=extra["sealer_max_branches"], hash_num=extra["hash_num"])
    for b in range(len(bcb) - 1):
        cs = list(range(bcb[b], bcb[b + 1]))
        if not cs:
            continue
        outs, st, bfs = O.polish_batch([names[c] for c in cs], [bytes(w.contig_buf[int(w.contig_off[c]):int(w.contig_off[c + 1])]) for c in cs],
                                       [w.reads_of(c) for c in cs], [int(w.kmer_threshold[c]) for c in cs], op, want_bfs=True)
        for ki in range(len(ks)):
            if not np.array_equal(filt[b][ki], bfs[ki]):
                return f"filter mismatch batch {b} k index {ki}"
        for c, o in zip(cs, outs):
            if got[c] != o:
                return f"sequence mismatch contig {c} (batch {b})"
    return None


def run_staged(core, ctx, O, seed, local_mode):
    """step-wise API with random ntEdit / Sealer parameters, every stage against the oracle"""
    w, ks, (cbf, bf), bcb, extra = random_case(seed)
    rng = np.random.default_rng(seed + 77)
    set_schedule(local_mode)
    nt = dict(max_insertions=int(rng.choice([0, 1, 3, 5])), max_deletions=int(rng.choice([0, 2, 5, 10])),
              jump=int(rng.choice([1, 2, 3, 5])), missing_ratio=extra["missing_ratio"], edit_ratio=extra["edit_ratio"],
              soft_mask=int(rng.integers(0, 2)), min_contig_len=int(rng.choice([100, 50, 500])))
    se = dict(sealer_max_paths=extra["sealer_max_paths"], sealer_max_branches=extra["sealer_max_branches"],
              sealer_max_gap=int(rng.choice([30, 800])), sealer_max_cost=int(rng.choice([300, 100000])))
    p = core.default_params(k_values=ks, cbf_bytes=cbf, bf_bytes=bf, hash_num=extra["hash_num"], **nt, **se)
    contig = lambda c: bytes(w.contig_buf[int(w.contig_off[c]):int(w.contig_off[c + 1])])
    batch_of = {c: b for b in range(len(bcb) - 1) for c in range(bcb[b], bcb[b + 1])}
    with core.Job(ctx, p, contig_buf=w.contig_buf, contig_off=w.contig_off, contig_read_begin=w.contig_read_begin,
                  kmer_threshold=w.kmer_threshold, read_buf=w.read_buf, read_off=w.read_off,
                  batch_contig_begin=bcb, name_len=[len(x) for x in w.names]) as job:
        try:
            job.build_filters()
            filt = [[job.get_filter(b, ki).copy() for ki in range(len(ks))] for b in range(len(bcb) - 1)]
            cur = [contig(c) for c in range(w.n_contigs)]
            for ki, k in enumerate(ks):
                job.ntedit_pass(ki)
                got = job.sequences()
                for c in range(w.n_contigs):
                    if cur[c] is None or len(cur[c]) < p.min_contig_len:
                        cur[c] = None
                        if got[c] is not None:
                            return f"contig {c} should have been dropped at k index {ki}"
                        continue
                    q = O.ntedit_params(k, p.hash_num, max_insertions=p.max_insertions, max_deletions=p.max_deletions,
                                        missing_ratio=p.missing_ratio, edit_ratio=p.edit_ratio, jump=p.jump,
                                        mask=p.soft_mask, min_contig_len=p.min_contig_len)
                    want, _ = O.ntedit(cur[c], filt[batch_of[c]][ki], k, p.hash_num, q)
                    if got[c] != want:
                        return f"ntedit mismatch contig {c} k {k}"
                    cur[c] = want
            job.mask()
            m = job.sequences()
            for c in range(w.n_contigs):
                if cur[c] is not None and m[c] != O.mask(cur[c], ks[0]):
                    return f"mask mismatch contig {c}"
            job.sealer()
            sq = job.sequences()
            for c in range(w.n_contigs):
                if m[c] is None:
                    continue
                want, _ = O.sealer(m[c], ks, filt[batch_of[c]], p.hash_num, max_paths=p.sealer_max_paths,
                                   max_branches=p.sealer_max_branches, max_gap_len=p.sealer_max_gap,
                                   max_cost=p.sealer_max_cost)
                if sq[c] != want:
                    return f"sealer mismatch contig {c}"
        except core.GpError as e:
            if e.code == -9:
                return None  # a contig outgrew its slot: the step-wise calls report it (gp_job_polish retries)
            raise
    return None


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    from goldpolish_amd import core
    import oracle_lib
    O = oracle_lib.Oracle(str(ROOT / "oracle" / "libgpo.so"))
    bad = 0
    t0 = time.time()
    with core.Context(0) as ctx:
        for seed in range(first, first + n):
            for mode in ("0", "1", "2", "3"):
                try:
                    err = run_case(core, ctx, O, seed, mode) or run_staged(core, ctx, O, seed, mode)
                except Exception as e:  # noqa: BLE001 -- a fuzz loop reports and goes on
                    err = f"exception {type(e).__name__}: {e}"
                if err:
                    bad += 1
                    print(f"seed {seed} local={mode}: {err}", flush=True)
            if (seed - first) % 10 == 9:
                print(f"... {seed - first + 1} cases, {bad} failures, {time.time() - t0:.0f} s", flush=True)
    print(f"fuzz: {n} cases x 4 schedules, {bad} failures")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
