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
