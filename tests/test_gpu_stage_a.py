"""Stage A parity on a real MI355X: the Bloom filters built by the HIP path
(through the C ABI) must equal, bit for bit, the filters the oracle builds by
replaying the reads serially (reference: src/utils.cpp:96-123)."""
import os

import numpy as np
import pytest

from goldpolish_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(params=["grid", "levels", "local", "census", "census_all"], autouse=True)
def stage_a_path(request, monkeypatch):
    """every case runs through the stage-A schedules: grid-wide reservation rounds
    (GP_STAGE_A_LOCAL=0), the census level by level (long batches with one threshold per filter: batches with
    several take the grid-wide schedule there too), one workgroup per (batch, k) pair replaying every operation
    (GP_STAGE_A_LOCAL=1, census off), the same behind the census of first touches (forced
    for batches of any length; pairs it does not apply to or whose buckets overflow fall back) with
    the candidates of private k-mers settled by their count (ckc_*: the default), and the census with
    every candidate replayed (GP_STAGE_A_CKC=0: rounds 2-3's path, which pairs the count does not apply
    to still take)"""
    monkeypatch.setenv("GP_STAGE_A_LOCAL", "0" if request.param in ("grid", "levels") else "1")
    monkeypatch.setenv("GP_STAGE_A_LEVELS", "1" if request.param == "levels" else "0")
    monkeypatch.setenv("GP_STAGE_A_LEVELS_LOAD", "1000000")   # (filters of any load: the tiny ones of these tests saturate)
    monkeypatch.setenv("GP_STAGE_A_CENSUS", "1" if request.param.startswith("census") else "0")
    monkeypatch.setenv("GP_STAGE_A_CENSUS_MIN", "1")
    monkeypatch.setenv("GP_STAGE_A_CKC", "0" if request.param == "census_all" else "1")
    return "census" if request.param == "census_all" else request.param


@pytest.fixture(scope="module")
def gp():
    from goldpolish_amd import core
    ctx = core.Context(0)
    yield core, ctx
    ctx.close()


def oracle_filters(oracle, w, params, batch_contig_begin=None):
    """filters[batch][ki] as numpy arrays, from the serial oracle"""
    ks = list(params.k[:params.n_k])
    n = w.n_contigs
    bcb = list(batch_contig_begin) if batch_contig_begin is not None else list(range(n + 1))
    out = []
    for b in range(len(bcb) - 1):
        cbfs = [np.zeros(params.cbf_bytes, dtype=np.uint8) for _ in ks]
        bfs = [np.zeros(params.bf_bytes, dtype=np.uint8) for _ in ks]
        for c in range(bcb[b], bcb[b + 1]):
            reads = w.reads_of(c)
            if reads:
                oracle.fill_bfs(reads, ks, int(w.kmer_threshold[c]), params.cbf_bytes,
                                params.bf_bytes, params.hash_num, cbfs, bfs)
        out.append(bfs)
    return out


def run_and_compare(core, ctx, oracle, w, params, batch_contig_begin=None):
    job = core.Job(ctx, params, contig_buf=w.contig_buf, contig_off=w.contig_off,
                   contig_read_begin=w.contig_read_begin, kmer_threshold=w.kmer_threshold,
                   read_buf=w.read_buf, read_off=w.read_off,
                   batch_contig_begin=batch_contig_begin)
    try:
        job.build_filters()
        want = oracle_filters(oracle, w, params, batch_contig_begin)
        for b in range(job.n_batches):
            for ki in range(params.n_k):
                got = job.get_filter(b, ki)
                assert np.array_equal(got, want[b][ki]), (b, ki, int((got != want[b][ki]).sum()))
        return job.counters()
    finally:
        job.close()


@pytest.mark.parametrize("knobs", [
    {"GP_STAGE_A_CENSUS_WIDE": "0"},                                   # ranges of 8192 counters, 16-byte record groups
    {"GP_STAGE_A_SCATTER_ROUNDS": "4"},                                # four rounds per staged tile
    {"GP_STAGE_A_CENSUS_OVERLAP": "3"},                                # three buffer sets, the pack on its own stream
    {"GP_STAGE_A_CENSUS_OVERLAP": "0"},                                # one set, one stream, no cap on the scatter's grid
    {"GP_STAGE_A_SCATTER_PER_CU": "0", "GP_STAGE_A_SCATTER_SPAN": "1"},  # a workgroup per stretch
    {"GP_STAGE_A_CENSUS_MB": "64"},                                    # many small cohorts
])
def test_census_geometries_and_schedules_give_the_same_filters(gp, oracle, stage_a_path, knobs, monkeypatch):
    """the census kernels' compile-time geometries and the host's ways of scheduling them (default:
    wide ranges, one round, two sets, two scatter workgroups per CU) against the serial oracle, at
    GoldPolish's filter sizes (the 5 * 2^21 index arithmetic) and at a size that takes the general one"""
    if stage_a_path != "census":
        pytest.skip("census knobs")
    core, ctx = gp
    for k, v in knobs.items():
        monkeypatch.setenv(k, v)
    w = synth.make_workload([30000, 12000, 21000, 800], 30, seed=23)
    run_and_compare(core, ctx, oracle, w, core.default_params())
    run_and_compare(core, ctx, oracle, w, core.default_params(cbf_bytes=3000017, bf_bytes=262144))


def test_filters_match_oracle_default_sizes(gp, oracle, stage_a_path):
    core, ctx = gp
    w = synth.make_workload([25000, 9000, 14000], 30, seed=11)
    c = run_and_compare(core, ctx, oracle, w, core.default_params())
    assert c["a_ops"] > 0 and (c["a_ordered_ops"] > 0 or stage_a_path == "levels")   # (the levels order nothing)
    # the schedule the fixture asked for is the one that ran (a silent fall-back would still match the oracle)
    if stage_a_path == "grid":
        assert c["a_reservations"] > 0 and c["a_replayed"] == 0
    elif stage_a_path == "levels":
        assert c["a_reservations"] == 0 and c["a_replayed"] == 0 and c["a_ordered_ops"] == 0
    elif stage_a_path == "local":
        assert c["a_reservations"] == 0 and c["a_replayed"] == c["a_ops"]
    else:
        assert c["a_reservations"] == 0 and 0 < c["a_replayed"] < c["a_ops"] // 2
        if os.environ.get("GP_STAGE_A_CKC") == "1":
            # nine candidates in ten belong to k-mers that share no counter with another: counted, not replayed
            assert c["a_counted"] > 5 * c["a_replayed"]
        else:
            assert c["a_counted"] == 0


def test_small_filters_heavy_collisions_and_saturation(gp, oracle):
    """tiny counting filter: long dependency chains, saturation, settled shortcut"""
    core, ctx = gp
    w = synth.make_workload([20000, 20000], 25, seed=12)
    for cbf, bf in [(4096, 2048), (65536 + 4, 4096), (10 * 1024, 1024 * 12)]:
        p = core.default_params(k_values=[32, 20], cbf_bytes=cbf, bf_bytes=bf)
        run_and_compare(core, ctx, oracle, w, p)


def test_byte_counter_fallback_and_wide_thresholds(gp, oracle, monkeypatch):
    """optional 4-bit packed counters (GP_STAGE_A_NIBBLE=1); thresholds > 15 always use bytes"""
    core, ctx = gp
    w = synth.make_workload([12000, 7000], 25, seed=15)
    monkeypatch.setenv("GP_STAGE_A_NIBBLE", "1")
    run_and_compare(core, ctx, oracle, w, core.default_params(k_values=[32, 24]))
    # 8 k values and kmer_threshold 13: thresholds 11..18 do not fit a nibble
    w.kmer_threshold[:] = 13
    p = core.default_params(k_values=[32, 30, 28, 26, 24, 22, 20, 18], cbf_bytes=200_000,
                            bf_bytes=1 << 16)
    run_and_compare(core, ctx, oracle, w, p)
    # many repeats so that counters really climb to 14 in nibble mode
    rng = np.random.default_rng(3)
    g = synth.random_genome(3000, rng).tobytes()
    w2 = _custom_workload([g], [[g] * 40], [13])
    run_and_compare(core, ctx, oracle, w2, core.default_params())
    monkeypatch.delenv("GP_STAGE_A_NIBBLE")
    run_and_compare(core, ctx, oracle, w2, core.default_params())


def test_many_windows(gp, oracle, monkeypatch):
    core, ctx = gp
    monkeypatch.setenv("GP_STAGE_A_WINDOW", "4096")
    w = synth.make_workload([15000, 8000], 12, seed=13)
    run_and_compare(core, ctx, oracle, w, core.default_params(k_values=[28, 24]))


def _custom_workload(contigs, reads_per_contig, thresholds):
    names = [f"c{i}" for i in range(len(contigs))]
    coff = np.zeros(len(contigs) + 1, dtype=np.uint64)
    coff[1:] = np.cumsum([len(c) for c in contigs])
    flat = [r for rs in reads_per_contig for r in rs]
    roff = np.zeros(len(flat) + 1, dtype=np.uint64)
    if flat:
        roff[1:] = np.cumsum([len(r) for r in flat])
    crb = np.zeros(len(contigs) + 1, dtype=np.uint32)
    crb[1:] = np.cumsum([len(rs) for rs in reads_per_contig])
    return synth.Workload(
        names=names, contig_buf=np.frombuffer(b"".join(contigs), dtype=np.uint8),
        contig_off=coff, read_buf=np.frombuffer(b"".join(flat) or b"", dtype=np.uint8),
        read_off=roff, contig_read_begin=crb,
        kmer_threshold=np.asarray(thresholds, dtype=np.uint32))


def test_edge_cases_ragged_N_lowercase_repeats_and_shared_batches(gp, oracle):
    core, ctx = gp
    rng = np.random.default_rng(5)
    g = synth.random_genome(6000, rng).tobytes()
    low = g[:3000].lower()
    with_n = bytearray(g[1000:4000])
    for i in (10, 500, 501, 502, 1700, 2999):
        with_n[i] = ord("N")
    with_n[40] = ord("R")
    homopolymer = b"A" * 700 + g[:100] + b"ACAC" * 300 + b"T" * 400
    short = g[:19]          # shorter than every k
    exactly_k = g[100:120]  # one 20-mer only
    reads0 = [g, g[::1], low, bytes(with_n), homopolymer, homopolymer, short, exactly_k, b""]
    reads1 = [g[2000:5000]] * 9 + [homopolymer] * 3
    reads3 = [synth.revcomp(np.frombuffer(g, dtype=np.uint8)).tobytes()] * 7
    contigs = [g, g[:500], b"ACGT" * 50, g[:4000]]
    # contig 2 has no reads at all; contigs 0+1 share a batch with different thresholds
    w = _custom_workload(contigs, [reads0, reads1, [], reads3], [5, 7, 5, 13])
    p = core.default_params()
    run_and_compare(core, ctx, oracle, w, p, batch_contig_begin=[0, 2, 3, 4])
    # same data, one contig per batch, tiny filters
    p2 = core.default_params(k_values=[32, 28, 24, 20], cbf_bytes=5000, bf_bytes=4096)
    run_and_compare(core, ctx, oracle, w, p2)


def test_mixed_thresholds_in_one_batch_with_colliding_counters(gp, oracle):
    """Contigs of one batch share the counting filters but not the threshold
    (goldpolish_targeted_bfs.cpp:126-131).  With a filter this small their counters collide, so a
    counter one contig's operation leaves at its thr-1 instead of thr is seen by a later contig
    with a larger threshold: settling may only use the batch's largest threshold."""
    core, ctx = gp
    rng = np.random.default_rng(21)
    contigs, reads = [], []
    for i in range(6):
        g = synth.random_genome(2500, rng)
        contigs.append(g.tobytes())
        rs = []
        for _ in range(24):
            s0 = int(rng.integers(0, 1500))
            rs.append(synth.mutate(g[s0:s0 + 1000], 0.04, rng).tobytes())
        reads.append(rs)
    for thr in ([4, 5, 7, 9, 11, 13], [13, 4, 9, 5, 12, 6], [4, 13, 4, 13, 4, 13], [6, 6, 6, 7, 6, 6]):
        w = _custom_workload(contigs, reads, thr)
        for cbf, bf, ks in [(4099, 1024, [32, 28, 24, 20]), (20011, 4096, [32, 20]), (70001, 8192, [25])]:
            p = core.default_params(k_values=ks, cbf_bytes=cbf, bf_bytes=bf)
            run_and_compare(core, ctx, oracle, w, p, batch_contig_begin=[0, 6])
            run_and_compare(core, ctx, oracle, w, p, batch_contig_begin=[0, 2, 6])


def test_more_pairs_than_arena_slots(gp, oracle, monkeypatch):
    """the counting filters are scratch in an arena of slots: with 8 slots for 22 batches x 4 k the
    grid-wide schedule works through groups of two batches, and cbf_local's workgroups replay
    several pairs each in the slot they own, clearing it in between"""
    core, ctx = gp
    monkeypatch.setenv("GP_STAGE_A_SLOTS", "8")
    lens = [3000, 1200, 5000, 800, 2500, 4000, 600, 3500, 1500, 2000, 4500,
            1000, 700, 2600, 3100, 900, 5200, 1100, 2300, 1700, 2900, 400]
    w = synth.make_workload(lens, 20, seed=23)
    run_and_compare(core, ctx, oracle, w, core.default_params())
    run_and_compare(core, ctx, oracle, w, core.default_params(k_values=[32, 20], cbf_bytes=20011, bf_bytes=4096))
    # short pairs are unwound (zeros stored at the counters their k-mers hash to) instead of the
    # slot being cleared: windows of a few hundred bases between longer contigs, every workgroup
    # replaying several of them in its one slot
    tiny = [300, 5000, 450, 800, 260, 2500, 1200, 350, 700, 9000, 500, 640, 1500, 420, 980, 310, 3300, 560, 770, 290,
            1100, 480, 6100, 330, 900, 610, 2700, 370, 1300, 520]
    w2 = synth.make_workload(tiny, 30, seed=24, read_mean=600.0, read_min=60, s_per_10kbp=400.0)
    run_and_compare(core, ctx, oracle, w2, core.default_params())
    run_and_compare(core, ctx, oracle, w2, core.default_params(k_values=[32, 24], cbf_bytes=1 << 20, bf_bytes=1 << 16))
    # shared batches too: three contigs per batch
    bcb = list(range(0, 22, 3)) + [22]
    run_and_compare(core, ctx, oracle, w, core.default_params(k_values=[28, 24], cbf_bytes=70001, bf_bytes=8192),
                    batch_contig_begin=bcb)


def test_census_cohorts_and_bucket_overflow(gp, oracle, monkeypatch):
    """the census in several cohorts (a tiny record budget), and pairs whose buckets overflow (long
    runs of one k-mer) falling back to the plain replay beside pairs that do not"""
    core, ctx = gp
    monkeypatch.setenv("GP_STAGE_A_CENSUS_MB", "2")   # 256 k records: a cohort holds a pair or two
    w = synth.make_workload([30000, 12000, 20000, 8000, 25000], 30, seed=26)
    run_and_compare(core, ctx, oracle, w, core.default_params())
    rng = np.random.default_rng(27)
    g = synth.random_genome(20000, rng).tobytes()
    reads = [g[i:i + 5000] for i in range(0, 15000, 700)] * 3
    flood = [b"A" * 9000 + g[:500], b"ACACACAC" * 1200]           # thousands of touches on four counters
    w2 = _custom_workload([g, g[:9000], g[5000:]], [reads + flood, reads, flood + reads], [6, 5, 7])
    run_and_compare(core, ctx, oracle, w2, core.default_params())
    run_and_compare(core, ctx, oracle, w2, core.default_params(k_values=[32, 20]), batch_contig_begin=[0, 2, 3])


def test_census_pair_with_more_runs_than_a_wave_keeps(gp, oracle, stage_a_path):
    """a pair of more than 128 stretches (1 Mi read positions): census_mark keeps the records of
    128 runs per range in registers and fetches the rest again in both its passes"""
    if stage_a_path != "census":
        pytest.skip("the census schedule only")
    core, ctx = gp
    w = synth.make_workload([32000, 9000], 45, seed=31, s_per_10kbp=60.0)
    assert int(w.read_off[int(w.contig_read_begin[1])]) > 129 * 8192
    run_and_compare(core, ctx, oracle, w, core.default_params(k_values=[32, 20]))


def test_threshold_below_4_is_refused(gp):
    core, ctx = gp
    w = _custom_workload([b"ACGT" * 100], [[b"ACGT" * 100]], [3])
    with pytest.raises(core.GpError) as e:
        core.Job(ctx, core.default_params(), contig_buf=w.contig_buf, contig_off=w.contig_off,
                 contig_read_begin=w.contig_read_begin, kmer_threshold=w.kmer_threshold,
                 read_buf=w.read_buf, read_off=w.read_off)
    assert e.value.code == -5


def test_idempotent_rebuild(gp, oracle):
    """building twice gives the same filters (buffers are reset, no stale reservations)"""
    core, ctx = gp
    w = synth.make_workload([12000], 20, seed=14)
    p = core.default_params(k_values=[32])
    with core.Job.from_workload(ctx, p, w) as job:
        job.build_filters()
        a = job.get_filter(0, 0).copy()
        job.build_filters()
        assert np.array_equal(a, job.get_filter(0, 0))
