"""Full-size checks on a real MI355X (BASELINE.json configs[1]: one 10 Mbp contig + 30x
reads, k=32) and the frozen self-generated vectors, through the C ABI.

At this size the oracle is only run for the filter (about 20 s); the rest uses properties
that do not depend on size: the result must not depend on how the work is scheduled
(window size), on which other batches share the job, or on being repeated."""
import hashlib
import json
import pathlib

import numpy as np
import pytest

from goldpolish_amd import synth

pytestmark = pytest.mark.gpu
GOLD = pathlib.Path(__file__).parent / "golden"


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


@pytest.fixture(scope="module")
def gp():
    from goldpolish_amd import core
    ctx = core.Context(0)
    yield core, ctx
    ctx.close()


def test_frozen_vectors_through_the_abi(gp):
    core, ctx = gp
    for case in json.loads((GOLD / "pipeline_small.json").read_text()):
        w = synth.make_workload(case["lens"], case["coverage"], seed=case["seed"], **case["extra"])
        p = core.default_params(k_values=case["k"])
        with core.Job.from_workload(ctx, p, w) as job:
            job.build_filters()
            for c, want in enumerate(case["contigs"]):
                got = [sha(job.get_filter(c, ki).tobytes()) for ki in range(p.n_k)]
                assert got == want["filters"], (case["name"], c)
            for ki in range(p.n_k):
                job.ntedit_pass(ki)
                seqs = job.sequences()
                for c, want in enumerate(case["contigs"]):
                    assert sha(seqs[c]) == want["ntedit"][ki]["sha"], (case["name"], c, ki)
            job.ntedit_guard()
            job.mask()
            for c, s in enumerate(job.sequences()):
                assert sha(s) == case["contigs"][c]["mask"]["sha"]
            job.sealer()
            for c, s in enumerate(job.sequences()):
                assert sha(s) == case["contigs"][c]["sealer"]["sha"]
            job.upper()
            for c, s in enumerate(job.sequences()):
                assert sha(s) == case["contigs"][c]["final"]["sha"]
            # and in one call
            job.polish()
            for c, s in enumerate(job.sequences()):
                assert sha(s) == case["contigs"][c]["final"]["sha"]


def test_config2_full_size(gp, oracle, monkeypatch):
    core, ctx = gp
    w = synth.make_workload([10_000_000], 30, seed=2)
    p = core.default_params(k_values=[32])
    with core.Job.from_workload(ctx, p, w) as job:
        job.polish()
        out = job.sequences()[0]
        f_a = job.get_filter(0, 0).copy()
        c_a = job.counters()
        # the whole filter against the serial oracle (296 M insert_thresh_contains calls)
        _, bfs = oracle.fill_bfs(w.reads_of(0), [32], int(w.kmer_threshold[0]), p.cbf_bytes,
                                 p.bf_bytes)
        assert np.array_equal(f_a, bfs[0])
        # idempotence
        job.polish()
        assert job.sequences()[0] == out and np.array_equal(job.get_filter(0, 0), f_a)
    # schedule independence: a very different window size gives the same bits
    monkeypatch.setenv("GP_STAGE_A_WINDOW", str(3 << 20))
    with core.Job.from_workload(ctx, p, w) as job:
        job.build_filters()
        assert np.array_equal(job.get_filter(0, 0), f_a)
        assert job.counters()["a_ops"] == c_a["a_ops"]
    # the polished contig: same length class, upper case, and what the oracle's ntEdit,
    # mask and Sealer make of it given that filter
    assert out == out.upper()
    cur, _ = oracle.ntedit(w.contig(0), f_a, 32)
    m = oracle.mask(cur, 32)
    s, _ = oracle.sealer(m, [32], [f_a])
    assert out == s.upper()


def test_batches_do_not_influence_each_other(gp):
    core, ctx = gp
    w = synth.make_workload([30000, 18000, 25000, 9000, 40000], 30, seed=61)
    p = core.default_params()
    with core.Job.from_workload(ctx, p, w) as job:
        job.polish()
        together = job.sequences()
    for c in (0, 3, 4):
        one = synth.Workload(
            names=[w.names[c]], contig_buf=np.frombuffer(w.contig(c), dtype=np.uint8),
            contig_off=np.array([0, len(w.contig(c))], dtype=np.uint64),
            read_buf=np.frombuffer(b"".join(w.reads_of(c)), dtype=np.uint8),
            read_off=np.concatenate([[0], np.cumsum([len(r) for r in w.reads_of(c)])]).astype(np.uint64),
            contig_read_begin=np.array([0, len(w.reads_of(c))], dtype=np.uint32),
            kmer_threshold=w.kmer_threshold[c:c + 1])
        with core.Job.from_workload(ctx, p, one) as job:
            job.polish()
            assert job.sequences()[0] == together[c], c


def test_device_export_matches_host_export(gp):
    """gp_job_get_sequences_device packs the same bytes into an HBM buffer (the RCCL send
    buffer of the multi-GPU gather) as gp_job_get_sequences returns on the host."""
    import torch
    core, ctx = gp
    w = synth.make_workload([30_000, 90, 12_000, 150, 45_000], 25, seed=77)   # 90 bp is dropped
    p = core.default_params(k_values=[32, 24])
    # one batch of five records, so the 90 bp one is dropped without tripping the guard
    with core.Job(ctx, p, contig_buf=w.contig_buf, contig_off=w.contig_off,
                  contig_read_begin=w.contig_read_begin, kmer_threshold=w.kmer_threshold,
                  read_buf=w.read_buf, read_off=w.read_off, batch_contig_begin=[0, 5],
                  name_len=[len(n) for n in w.names]) as job:
        job.polish()
        host = job.sequences()
        lens, dropped = job.lengths()
        assert dropped.tolist() == [0, 1, 0, 0, 0] and lens[1] == 0
        total = int(lens.sum())
        buf = torch.zeros(total + 64, dtype=torch.uint8, device="cuda")
        job.sequences_into_device(buf.data_ptr(), total)
        got = buf.cpu().numpy()
        assert got[:total].tobytes() == b"".join(s for s in host if s is not None)
        assert not got[total:].any()
        with pytest.raises(core.GpError):
            job.sequences_into_device(buf.data_ptr(), total - 1)


def test_mixed_job_long_and_many_short_batches(gp, oracle, monkeypatch):
    """One job holding a long contig's batch and a few hundred short ones.  By itself the filter build
    gives the long batch (filters that stay unsaturated) the census level by level and the short ones
    one workgroup per filter pair; with GP_STAGE_A_LEVELS=0 the long batch takes the grid-wide reservation
    rounds instead; with GP_STAGE_A_LOCAL=0 every batch does.  Every filter is the same in all three, and a
    sample of them equals the serial oracle's."""
    core, ctx = gp
    lens = [2_500_000] + [6000] * 150 + [9000] * 60
    w = synth.make_workload(lens, 20, seed=91)
    p = core.default_params(k_values=[32, 24])

    def filters(**env):
        for name in ("GP_STAGE_A_LOCAL", "GP_STAGE_A_LEVELS"):
            monkeypatch.delenv(name, raising=False)
        for name, value in env.items():
            monkeypatch.setenv(name, value)
        with core.Job.from_workload(ctx, p, w) as job:
            job.build_filters()
            launches = {name: job.kernel_time(name)[1] for name in ("cbf_local", "cbf_first_pass", "cbf_census")}
            rounds = job.counters()["a_rounds"]
            return ([[sha(job.get_filter(b, ki).tobytes()) for ki in range(p.n_k)]
                     for b in range(job.n_batches)],
                    [job.get_filter(b, 1) for b in (0, 1, 77, 210)], launches, rounds)

    auto, sample, n, _ = filters()
    assert n["cbf_local"] == 1 and n["cbf_first_pass"] == 0 and n["cbf_census"] >= 1   # levels + workgroup-local
    mixed, _, n, _ = filters(GP_STAGE_A_LEVELS="0")
    assert n["cbf_local"] == 1 and n["cbf_first_pass"] > 0 and mixed == auto            # grid-wide + workgroup-local
    grid, _, n, _ = filters(GP_STAGE_A_LOCAL="0", GP_STAGE_A_LEVELS="0")
    assert n["cbf_local"] == 0 and n["cbf_first_pass"] > 0 and grid == auto             # grid-wide alone
    for i, b in enumerate((0, 1, 77, 210)):
        _, bfs = oracle.fill_bfs(w.reads_of(b), [32, 24], int(w.kmer_threshold[b]), p.cbf_bytes,
                                 p.bf_bytes)
        assert np.array_equal(sample[i], bfs[1]), b


def test_a_thousand_short_contigs_whole_path(gp, oracle):
    """More contigs than the chip holds workgroups: stage A's workgroup-local schedule and the
    register-tight build of the ntEdit walk (chosen above 768 contigs), whole path against the
    oracle's batch driver, one contig per batch."""
    core, ctx = gp
    w = synth.make_workload([1500] * 700 + [3000] * 200 + [800] * 100, 20, seed=123)
    p = core.default_params()
    with core.Job.from_workload(ctx, p, w) as job:
        job.polish()
        got = job.sequences()
        c = job.counters()
        assert job.kernel_time("cbf_local")[1] == 1
    assert c["b_edits"] > 100 and c["c_gaps"] > 0
    op = oracle.batch_params()
    for i in range(w.n_contigs):
        outs, _, _ = oracle.polish_batch([w.names[i]], [w.contig(i)], [w.reads_of(i)],
                                         [int(w.kmer_threshold[i])], op)
        assert got[i] == outs[0], i


def test_device_memory_is_accounted_and_released(gp):
    """gp_device_bytes_in_use follows job creation (inputs only), the first polish (filters + work
    buffers), the capacity retry and destruction (no leak; freed blocks wait in the pool)"""
    core, ctx = gp
    base = core.device_bytes_in_use()
    w = synth.make_workload([20_000, 8_000], 20, seed=5)
    p = core.default_params()
    with core.Job.from_workload(ctx, p, w) as job:
        inputs = core.device_bytes_in_use() - base
        assert 0 < inputs < (16 << 20)                   # reads + contigs + small tables, nothing else yet
        job.polish()
        held = core.device_bytes_in_use() - base
        filters = 2 * p.n_k * (p.cbf_bytes + p.bf_bytes)
        assert held > filters + 1 * (1 << 30)            # filters + the Sealer's search arenas for a small job
        assert held < filters + 6 * (1 << 30)
        job.polish()
        assert core.device_bytes_in_use() - base >= held
    assert core.device_bytes_in_use() == base


def test_sixty_contigs_of_25_kbp_whole_path(gp, oracle):
    """The aux benchmark's regime at a size the oracle handles in seconds: 25 kbp contigs, 30x
    reads, full cascade -- hundreds of edits and dozens of gaps per contig; every polished contig
    and every filter against the oracle's batch driver."""
    core, ctx = gp
    w = synth.make_workload([25_000] * 60, 30, seed=404)
    p = core.default_params()
    with core.Job.from_workload(ctx, p, w) as job:
        job.polish()
        got = job.sequences()
        c = job.counters()
        filt = [job.get_filter(b, 3).copy() for b in (0, 17, 59)]
    assert c["b_edits"] > 5000 and c["c_closed"] > 500
    op = oracle.batch_params()
    for i in range(w.n_contigs):
        outs, _, bfs = oracle.polish_batch([w.names[i]], [w.contig(i)], [w.reads_of(i)],
                                           [int(w.kmer_threshold[i])], op, want_bfs=True)
        assert got[i] == outs[0], i
        if i in (0, 17, 59):
            assert np.array_equal(filt[(0, 17, 59).index(i)], bfs[3]), i
