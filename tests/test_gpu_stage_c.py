"""Stage C parity on a real MI355X: Sealer gap filling (gap discovery, flank
bookkeeping, bidirectional BFS, path enumeration, consensus, splicing) through
the C ABI against the oracle, stage by stage and as the whole per-batch path."""
import numpy as np
import pytest

from goldpolish_amd import synth
from test_gpu_stage_a import _custom_workload, oracle_filters
from test_gpu_stage_b import _filters_blob

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gp():
    from goldpolish_amd import core
    ctx = core.Context(0)
    yield core, ctx
    ctx.close()


def _sealer_check(core, ctx, oracle, seqs, filters_per_contig, params):
    """run mask-free: feed `seqs` straight into stage C and compare with the oracle"""
    ks = list(params.k[:params.n_k])
    w = _custom_workload(seqs, [[] for _ in seqs], [5] * len(seqs))
    p = params
    job = core.Job(ctx, p, contig_buf=w.contig_buf, contig_off=w.contig_off,
                   contig_read_begin=w.contig_read_begin, kmer_threshold=w.kmer_threshold,
                   read_buf=w.read_buf, read_off=w.read_off)
    try:
        full = np.full(len(seqs) * p.n_k * p.bf_bytes, 255, np.uint8)
        job.set_filters(full)
        job.ntedit_pass(0)          # full filter: nothing changes, state initialised
        assert job.sequences() == list(seqs)
        job.set_filters(_filters_blob(filters_per_contig))
        job.sealer()
        got = job.sequences()
        stats = []
        for i, s in enumerate(seqs):
            want, st = oracle.sealer(s, ks, filters_per_contig[i], p.hash_num,
                                     max_paths=p.sealer_max_paths,
                                     max_branches=p.sealer_max_branches,
                                     max_cost=p.sealer_max_cost, max_gap_len=p.sealer_max_gap)
            assert got[i] == want, (i, len(got[i]), len(want))
            stats.append(st)
        # Sealer's log lines (sealer.cpp:656-683) as counters, summed over contigs and k runs
        c = job.counters()
        for dev, orc in (("c_gaps", "gaps_found"), ("c_closed", "gaps_closed"),
                         ("c_no_start_goal", "no_start_goal"), ("c_no_path", "no_path"),
                         ("c_unique_path", "unique_path"), ("c_multiple_paths", "multiple_paths"),
                         ("c_too_many_paths", "too_many_paths"),
                         ("c_too_many_branches", "too_many_branches"),
                         ("c_max_cost", "max_cost_exceeded"), ("c_vertex_tests", "vertex_tests")):
            assert c[dev] == sum(getattr(st, orc) for st in stats), (dev, c[dev])
        return c, stats
    finally:
        job.close()


def _filters_from(oracle, reads, params, thr=5):
    ks = list(params.k[:params.n_k])
    _, bfs = oracle.fill_bfs(reads, ks, thr, params.cbf_bytes, params.bf_bytes, params.hash_num)
    return bfs


def test_whole_path_matches_oracle(gp, oracle):
    core, ctx = gp
    w = synth.make_workload([25000, 9000, 14000, 6000], 30, seed=41)
    p = core.default_params()
    with core.Job.from_workload(ctx, p, w) as job:
        job.polish()
        got = job.sequences()
        counters = job.counters()
    closed = 0
    for c in range(w.n_contigs):
        outs, st, _ = oracle.polish_batch([w.names[c]], [w.contig(c)], [w.reads_of(c)],
                                          [int(w.kmer_threshold[c])])
        assert got[c] == outs[0], c
        closed += st.gaps_closed
    assert counters["c_gaps"] > 0 and counters["c_closed"] == closed > 0


def test_sealer_stage_on_masked_contigs(gp, oracle):
    """stage C alone on realistic soft-masked input (what ntEdit + mask leave behind)"""
    core, ctx = gp
    w = synth.make_workload([20000, 12000], 30, seed=42, draft_err=0.03)
    p = core.default_params()
    filters = oracle_filters(oracle, w, p)
    seqs = []
    for c in range(w.n_contigs):
        cur = w.contig(c)
        for ki, k in enumerate(p.k[:p.n_k]):
            cur, _ = oracle.ntedit(cur, filters[c][ki], k)
        seqs.append(oracle.mask(cur, p.k[0]))
    counters, stats = _sealer_check(core, ctx, oracle, seqs, filters, p)
    assert sum(s.gaps_closed for s in stats) > 0


def test_crafted_gaps(gp, oracle):
    core, ctx = gp
    rng = np.random.default_rng(17)
    g = synth.random_genome(4000, rng).tobytes()
    p = core.default_params()
    hap_b = bytearray(g)
    hap_b[1000] = ord("A") if g[1000] != ord("A") else ord("C")     # SNP -> two paths
    hap_b = bytes(hap_b)
    reads = [g] * 8 + [hap_b] * 8
    bfs = _filters_from(oracle, reads, p)

    def lower(s, a, b):
        return s[:a] + s[a:b].lower() + s[b:]

    seqs = [
        lower(g, 990, 1010),                       # bubble inside the gap: consensus path
        lower(g, 2000, 2030),                      # plain gap, unique path
        lower(lower(g, 500, 520), 540, 560),       # neighbouring gaps sharing a short flank
        g[:1500] + b"N" * 25 + g[1525:],           # N gap of the right size
        g[:1500] + b"N" * 10 + g[1525:],           # N gap shorter than the missing sequence
        g[:1500] + b"n" * 40 + g[1525:],           # ... longer
        lower(g, 3990, 4000),                      # sequence ends with a gap: left alone
        lower(g, 0, 12),                           # starts with a gap: empty left flank
        lower(g, 10, 20),                          # left flank shorter than k
        g[:700] + g[600:700] + g[600:],            # no gap at all
        lower(g[:300] + g[100:300] + g[100:], 290, 310),  # repeat: same flanks twice
        lower(g, 100, 110) + lower(g, 100, 110),   # identical (left, right) flank pairs
        b"acgt" * 30,                              # nothing but gap characters
    ]
    counters, stats = _sealer_check(core, ctx, oracle, seqs, [bfs] * len(seqs), p)
    assert stats[0].multiple_paths >= 1 and stats[1].unique_path >= 1
    assert counters["c_closed"] == sum(s.gaps_closed for s in stats)


def test_limits_abort_like_the_reference(gp, oracle):
    """tiny cost / branch / path limits: the search gives up exactly where the oracle does"""
    core, ctx = gp
    w = synth.make_workload([15000], 30, seed=43, draft_err=0.03)
    filters = oracle_filters(oracle, w, core.default_params())
    cur = w.contig(0)
    base = core.default_params()
    for ki, k in enumerate(base.k[:base.n_k]):
        cur, _ = oracle.ntedit(cur, filters[0][ki], k)
    seq = oracle.mask(cur, 32)
    for kw in (dict(sealer_max_cost=40), dict(sealer_max_branches=1), dict(sealer_max_paths=1),
               dict(sealer_max_gap=40)):
        p = core.default_params(**kw)
        _sealer_check(core, ctx, oracle, [seq], filters, p)
