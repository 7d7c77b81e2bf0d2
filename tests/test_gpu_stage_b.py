"""Stage B parity on a real MI355X: the ntEdit sweep (one k at a time and as the
full cascade), the 0.75 guard, soft mask and upper-casing, all through the C ABI,
against the oracle (reference: subprojects/ntedit/ntedit.cpp:1415-1771,
scripts/goldpolish-ntedit, scripts/goldpolish-mask, scripts/goldpolish-to-upper)."""
import numpy as np
import pytest

from goldpolish_amd import synth
from test_gpu_stage_a import _custom_workload, oracle_filters

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["sites ahead of the walk", "every site on the spot", "128-thread workgroups",
                        "128-thread workgroups, every site on the spot"], autouse=True)
def walk_mode(request, monkeypatch):
    """both ways the walk gets a site's decision: from ntedit_site_eval's records (default) and
    worked out on the spot at every site (GP_NTEDIT_SPECULATE=0); and both workgroup shapes: the
    256-thread build of small jobs and the four-waves-per-SIMD build with 128-thread workgroups that
    jobs of many contigs get (forced here: GP_WALK_WAVES=4)"""
    if "every site on the spot" in request.param:
        monkeypatch.setenv("GP_NTEDIT_SPECULATE", "0")
    if "128-thread" in request.param:
        monkeypatch.setenv("GP_WALK_WAVES", "4")
        monkeypatch.setenv("GP_WALK_THREADS", "128")
    else:
        monkeypatch.setenv("GP_EVAL_THREADS", "256")   # (ntedit_site_eval: 128-thread workgroups are its default)
    return request.param


@pytest.fixture(scope="module")
def gp():
    from goldpolish_amd import core
    ctx = core.Context(0)
    yield core, ctx
    ctx.close()


def _filters_blob(filters):
    return np.concatenate([f for per_batch in filters for f in per_batch])


def _cascade_check(core, ctx, oracle, w, params, filters, batch_contig_begin=None):
    ks = list(params.k[:params.n_k])
    n = w.n_contigs
    bcb = list(batch_contig_begin) if batch_contig_begin is not None else list(range(n + 1))
    batch_of = {}
    for b in range(len(bcb) - 1):
        for c in range(bcb[b], bcb[b + 1]):
            batch_of[c] = b
    job = core.Job(ctx, params, contig_buf=w.contig_buf, contig_off=w.contig_off,
                   contig_read_begin=w.contig_read_begin, kmer_threshold=w.kmer_threshold,
                   read_buf=w.read_buf, read_off=w.read_off, batch_contig_begin=batch_contig_begin,
                   name_len=[len(x) for x in w.names])
    try:
        job.set_filters(_filters_blob(filters))
        cur = [w.contig(c) for c in range(n)]
        for ki, k in enumerate(ks):
            job.ntedit_pass(ki)
            got = job.sequences()
            for c in range(n):
                if cur[c] is None or len(cur[c]) < params.min_contig_len:
                    cur[c] = None
                    assert got[c] is None, (ki, c)
                    continue
                p = oracle.ntedit_params(k, params.hash_num, max_insertions=params.max_insertions,
                                         max_deletions=params.max_deletions,
                                         missing_ratio=params.missing_ratio,
                                         edit_ratio=params.edit_ratio, jump=params.jump,
                                         mask=params.soft_mask)
                want, _ = oracle.ntedit(cur[c], filters[batch_of[c]][ki], k, params.hash_num, p)
                assert got[c] == want, (f"k={k} contig={c}", _first_diff(got[c], want))
                cur[c] = want
        return job.counters(), cur
    finally:
        job.close()


def _first_diff(a, b):
    n = min(len(a), len(b))
    for i in range(n):
        if a[i] != b[i]:
            return i, a[max(0, i - 20):i + 20], b[max(0, i - 20):i + 20], len(a), len(b)
    return n, len(a), len(b)


def test_cascade_matches_oracle_on_polishable_contigs(gp, oracle):
    core, ctx = gp
    w = synth.make_workload([25000, 9000, 14000], 30, seed=21)
    p = core.default_params()
    filters = oracle_filters(oracle, w, p)
    counters, _ = _cascade_check(core, ctx, oracle, w, p, filters)
    assert counters["b_sites"] > 0 and counters["b_edits"] > 0


def test_high_error_draft_many_indels(gp, oracle):
    """5% draft error: dense sites, insertions next to deletions, repeat guards"""
    core, ctx = gp
    w = synth.make_workload([12000, 12000, 8000], 40, seed=22, draft_err=0.05, read_err=0.03)
    p = core.default_params()
    filters = oracle_filters(oracle, w, p)
    _cascade_check(core, ctx, oracle, w, p, filters)


def test_other_parameters(gp, oracle):
    core, ctx = gp
    w = synth.make_workload([10000, 7000], 30, seed=23, draft_err=0.02)
    for kw in (dict(max_insertions=2, max_deletions=3), dict(max_insertions=0, max_deletions=0),
               dict(jump=2, missing_ratio=0.3, edit_ratio=0.7), dict(soft_mask=0),
               dict(max_insertions=5, max_deletions=10, edit_ratio=0.3)):
        p = core.default_params(k_values=[30, 21], **kw)
        filters = oracle_filters(oracle, w, p)
        _cascade_check(core, ctx, oracle, w, p, filters)


def test_edge_case_contigs(gp, oracle):
    core, ctx = gp
    rng = np.random.default_rng(7)
    g = synth.random_genome(9000, rng)
    reads = [synth.mutate(g[s:s + 3000], 0.02, rng).tobytes()
             for s in range(0, 6000, 150) for _ in range(2)]
    truth = g.tobytes()
    d = bytearray(synth.mutate(g, 0.015, rng).tobytes())
    with_n = bytearray(d)
    with_n[500:520] = b"N" * 20
    with_n[2000] = ord("N")
    with_n[2040:2043] = b"nnn"
    with_n[4000] = ord("R")
    with_n[4100] = ord("y")
    with_n[4200] = ord("X")
    lower = bytes(d[:3000]).lower() + bytes(d[3000:])
    contigs = [
        bytes(with_n), lower, truth[:99], truth[:100], truth[:131], truth[100:164],
        b"N" * 300, b"ACGT" * 40, bytes(d[:5000]) + b"N" * 40, b"N" * 35 + bytes(d[:4000]),
        bytes(d[:2000]) + bytes(d[1000:3000]),
    ]
    w = _custom_workload(contigs, [reads] + [[]] * (len(contigs) - 1), [6] + [5] * (len(contigs) - 1))
    p = core.default_params()
    # every contig is polished against the filters built from the same reads
    f0 = oracle_filters(oracle, w, p)[0]
    filters = [f0 for _ in contigs]
    _cascade_check(core, ctx, oracle, w, p, filters)
    # an empty filter (every k-mer absent) and a full one (nothing absent)
    empty = [[np.zeros(p.bf_bytes, np.uint8) for _ in range(p.n_k)] for _ in contigs]
    _cascade_check(core, ctx, oracle, w, p, empty)
    full = [[np.full(p.bf_bytes, 255, np.uint8) for _ in range(p.n_k)] for _ in contigs]
    _cascade_check(core, ctx, oracle, w, p, full)


def test_noisy_filters_stress_the_edit_list(gp, oracle):
    """random half-full filters make ntEdit chain insertions, deletions and substitutions at
    neighbouring positions: long runs of inserted bases, the insertion cap and the
    repeat-insertion guard (ntedit.cpp:1030-1102), edits on inserted bases, node splitting"""
    core, ctx = gp
    rng = np.random.default_rng(11)
    for dens in (0.5, 0.6, 0.7, 0.85, 0.9):
        contigs = [synth.random_genome(n, rng).tobytes() for n in (3000, 1500, 2200, 900)]
        w = _custom_workload(contigs, [[] for _ in contigs], [5] * len(contigs))
        p = core.default_params(k_values=[32, 24, 20], bf_bytes=4096)
        filters = [[np.packbits(rng.random(4096 * 8) < dens, bitorder="little")
                    for _ in range(p.n_k)] for _ in contigs]
        counters, final = _cascade_check(core, ctx, oracle, w, p, filters)
        assert counters["b_edits"] > 50
        assert any(len(f) > len(c) + 50 for f, c in zip(final, contigs) if f is not None)


MASK_CASES = [
    # hand-derived from scripts/goldpolish-mask:44-72 with k=4 unless stated
    (b"ACGTACGTACGT", 4, b"ACGTACGTACGT"),
    (b"ACGTacgtACGT", 4, b"ACGTacgtACGT"),             # lower run of 4 == k stays as is
    (b"ACGTAacgACGTA", 4, b"ACGTAacgACGTA"),
    (b"ACGTAAaCGaAAAAA", 4, b"ACGTAAacgaAAAAA"),       # CG between two lower bases: run of 2 < k
    (b"NNNNACGTACGTNN", 4, b"ACGTACGT"),               # ends stripped
    (b"ACGTACGTNNNNNNACGTACGT", 4, b"ACGTACGTNNNNNNACGTACGT"),
    (b"ACGTACGTnnnACGTACGT", 4, b"ACGTACGTnnnACGTACGT"),   # 'n' run shorter than k: lower() no-op
    (b"ACGTAXACGTA", 4, b"ACGTAACGTA"),                # X matches no class: dropped, splits runs
    (b"ACGTACGTAXAXACGTACGT", 4, b"ACGTACGTAaACGTACGT"),  # ... and a split-off run is short
    (b"acgt", 4, b"ACGT"),                             # len < 2k: all upper-cased first
    (b"nnnn", 4, b"N"),
    (b"", 4, b"N"),
    (b"ACGTACGaNNtACGTACGT", 4, b"ACGTACGaNNtACGTACGT"),   # third class swallows N: run aNNt of 4
    (b"ACGTACGaNtACGTACGT", 4, b"ACGTACGantACGTACGT"),     # run aNt of 3 < k lowered incl. N
    (b"ACGTRRACGTACGT", 4, b"ACGTrrACGTACGT"),         # upper-case IUPAC is in the third class
]


def test_mask_and_upper(gp, oracle):
    core, ctx = gp
    for seq, k, want in MASK_CASES:
        assert oracle.mask(seq, k) == want, seq
    rng = np.random.default_rng(9)
    seqs = [b"".join(rng.choice([b"A", b"C", b"G", b"T", b"a", b"c", b"g", b"t", b"N", b"n", b"R",
                                 b"x", b"ACGTACGTACGTACGTACGTACGTACGTACGTACGT",
                                 b"acgtacgtacgtacgtacgtacgtacgtacgtacgtacg"],
                                size=int(n)).tolist())
            for n in (1, 5, 40, 300, 3000, 20000)]
    seqs += [s for s, _, _ in MASK_CASES if len(s) > 0]
    seqs += [b"N" * 1000, b"n" * 70 + b"A" * 100 + b"N" * 5, b"a" * 31 + b"C" * 1000 + b"g" * 33]
    w = _custom_workload(seqs, [[] for _ in seqs], [5] * len(seqs))
    for k in (32, 4):
        p = core.default_params(k_values=[k], min_contig_len=0)
        job = core.Job(ctx, p, contig_buf=w.contig_buf, contig_off=w.contig_off,
                       contig_read_begin=w.contig_read_begin, kmer_threshold=w.kmer_threshold,
                       read_buf=w.read_buf, read_off=w.read_off)
        try:
            job.set_filters(np.full(len(seqs) * p.bf_bytes, 255, np.uint8))
            job.ntedit_pass(0)   # full filter: sequences pass through unchanged
            assert job.sequences() == seqs
            job.mask()
            got = job.sequences()
            for i, s in enumerate(seqs):
                assert got[i] == oracle.mask(s, k), (k, i, s[:80])
            job.upper()
            assert job.sequences() == [oracle.mask(s, k).upper() for s in seqs]
        finally:
            job.close()


def test_pipeline_without_sealer_and_guard(gp, oracle):
    """filters -> cascade -> guard -> mask -> upper, per batch, vs the oracle's batch driver;
    the second batch is built so that the 0.75 byte-ratio guard fires."""
    core, ctx = gp
    w0 = synth.make_workload([9000, 7000], 30, seed=31)
    rng = np.random.default_rng(3)
    shorts = [synth.random_genome(n, rng).tobytes() for n in (99, 98, 97, 150)]
    contigs = [w0.contig(0), w0.contig(1)] + shorts
    reads = [w0.reads_of(0), w0.reads_of(1), [], [], [], []]
    thr = [int(w0.kmer_threshold[0]), int(w0.kmer_threshold[1]), 5, 5, 5, 5]
    w = _custom_workload(contigs, reads, thr)
    bcb = [0, 2, 6]
    p = core.default_params(run_sealer=0)
    job = core.Job(ctx, p, contig_buf=w.contig_buf, contig_off=w.contig_off,
                   contig_read_begin=w.contig_read_begin, kmer_threshold=w.kmer_threshold,
                   read_buf=w.read_buf, read_off=w.read_off, batch_contig_begin=bcb,
                   name_len=[len(x) for x in w.names])
    try:
        job.polish()
        got = job.sequences()
    finally:
        job.close()
    op = oracle.batch_params(run_sealer=0)
    skipped = []
    for b in range(2):
        cs = list(range(bcb[b], bcb[b + 1]))
        outs, st, _ = oracle.polish_batch([w.names[c] for c in cs], [contigs[c] for c in cs],
                                          [reads[c] for c in cs], [thr[c] for c in cs], op)
        skipped.append(st.ntedit_skipped)
        for c, o in zip(cs, outs):
            assert got[c] == o, (b, c)
    assert skipped == [0, 1]
