"""gp_jobs_polish: several jobs of one GPU pipelined (filter build of job i + 1 beside the rest of
job i) give exactly what one job over all the batches gives."""
import numpy as np
import pytest

from goldpolish_amd import synth

pytestmark = pytest.mark.gpu


def test_jobs_in_flight_equal_one_job(oracle):
    from goldpolish_amd import core
    w = synth.make_workload([9000, 4000, 12000, 300, 7000, 5000, 90, 6000], 25, seed=31)
    p = core.default_params()
    with core.Context(0) as ctx, core.Job.from_workload(ctx, p, w) as one:
        one.polish()
        want = one.sequences()
        want_counters = one.counters()
    assert sum(x is not None and x != w.contig(i).upper() for i, x in enumerate(want)) >= 4   # real edits
    for parts in (2, 3, 5):
        cuts = [w.n_contigs * i // parts for i in range(parts + 1)]
        ctxs = [core.Context(0) for _ in range(parts)]
        jobs = [core.Job(ctxs[i], p, **core.slice_workload(w, cuts[i], cuts[i + 1])) for i in range(parts)]
        try:
            for _ in range(2):   # twice: jobs restart from their uploaded inputs
                core.polish_jobs(jobs)
                assert [s for j in jobs for s in j.sequences()] == want
            got = {}
            for j in jobs:
                for k, v in j.counters().items():
                    got[k] = got.get(k, 0) + v
            for k in ("a_ops", "b_sites", "b_edits", "c_gaps", "c_closed", "c_vertex_tests"):
                assert got[k] == want_counters[k], k
        finally:
            for j in jobs:
                j.close()
            for c in ctxs:
                c.close()
    # and against the oracle, batch by batch
    for c in range(w.n_contigs):
        outs, _, _ = oracle.polish_batch([w.names[c]], [w.contig(c)], [w.reads_of(c)], [int(w.kmer_threshold[c])])
        assert want[c] == outs[0]


def test_jobs_must_have_contexts_of_their_own_and_errors_name_the_job():
    from goldpolish_amd import core
    w = synth.make_workload([3000, 2000], 10, seed=32)
    p = core.default_params(k_values=[32])
    with core.Context(0) as ctx:
        a = core.Job(ctx, p, **core.slice_workload(w, 0, 1))
        b = core.Job(ctx, p, **core.slice_workload(w, 1, 2))
        with pytest.raises(core.GpError) as e:
            core.polish_jobs([a, b])
        assert e.value.code == -1 and "context of its own" in str(e.value)
        with pytest.raises(core.GpError):
            core.polish_jobs([])
        core.polish_jobs([a])
        a.close()
        b.close()


def test_a_job_without_reads_among_pipelined_jobs(oracle):
    """gp_jobs_polish with a legal job that has no reads at all (its filters are empty: nothing to
    edit, every soft-masked gap stays): the filter build of such a job returns at once, and the back
    half on the other stream must still find its filters cleared (round 2 left the clearing in flight)"""
    from goldpolish_amd import core
    w = synth.make_workload([9000, 5000, 7000], 25, seed=33)
    p = core.default_params()
    parts = [core.slice_workload(w, 0, 1), core.slice_workload(w, 1, 2), core.slice_workload(w, 2, 3)]
    # the middle job loses its reads
    parts[1]["read_buf"] = np.zeros(0, np.uint8)
    parts[1]["read_off"] = np.zeros(1, np.uint64)
    parts[1]["contig_read_begin"] = np.zeros(2, np.uint32)
    ctxs = [core.Context(0) for _ in parts]
    jobs = [core.Job(c, p, **kw) for c, kw in zip(ctxs, parts)]
    try:
        for _ in range(3):
            core.polish_jobs(jobs)
            got = [j.sequences()[0] for j in jobs]
            for i in (0, 2):
                outs, _, _ = oracle.polish_batch([w.names[i]], [w.contig(i)], [w.reads_of(i)], [int(w.kmer_threshold[i])])
                assert got[i] == outs[0], i
            outs, _, _ = oracle.polish_batch([w.names[1]], [w.contig(1)], [[]], [int(w.kmer_threshold[1])])
            assert got[1] == outs[0]
            assert not np.any(jobs[1].get_filter(0, 0))
    finally:
        for j in jobs:
            j.close()
        for c in ctxs:
            c.close()
