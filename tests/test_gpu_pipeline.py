"""gp_pipeline_*: batch sets from host memory to polished bytes in host memory, the upload of one
set beside the polishing of the one before (SURVEY 8d's timed region).  Every set must come out
exactly as a job of its own gives it -- and as the oracle does -- in submission order."""
import numpy as np
import pytest

from goldpolish_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("overlap", ["3", "0", "1", "2"])
def test_stream_of_sets_equals_single_jobs_and_the_oracle(oracle, overlap, monkeypatch):
    """overlap 3 (the default): sets with a long contig are split -- their filter build on 6 CUs in 8 beside
    the back half of the set before on the other two -- the rest runs whole (here: the set with the 25 kbp
    contig is split, GP_PIPELINE_SPLIT_FROM lowered); 0: never; 1: every set, 7 | 1; 2: every set, no masks"""
    monkeypatch.setenv("GP_PIPELINE_SPLIT_FROM", "20000")
    from goldpolish_amd import core
    monkeypatch.setenv("GP_PIPELINE_OVERLAP", overlap)
    p = core.default_params()
    shapes = [[9000, 4000, 12000], [300, 7000], [25000, 90, 6000, 5000], [15000], [8000, 8000]]
    ws = [synth.make_workload(l, 25, seed=70 + i) for i, l in enumerate(shapes)]
    want = []
    with core.Context(0) as ctx:
        for w in ws:
            with core.Job.from_workload(ctx, p, w) as job:
                job.polish()
                want.append((job.sequences(), job.counters()))
    sets = [core.batch_set_of_workload(w) for w in ws]
    with core.Pipeline(0, p, depth=2) as pipe:
        # more than `depth` in flight is refused, not queued behind the caller's back
        pipe.submit(*sets[0], tag=100)
        pipe.submit(*sets[1], tag=101)
        with pytest.raises(core.GpError) as e:
            pipe.submit(*sets[2], tag=102)
        assert e.value.code == -6
        got = []
        nxt = 2
        for i in range(len(ws)):
            r = pipe.collect()
            assert r["tag"] == 100 + i
            got.append(r)
            if nxt < len(ws):
                pipe.submit(*sets[nxt], tag=100 + nxt)
                nxt += 1
        # twice through the same pipeline: recycled device blocks, same bytes
        pipe.submit(*sets[3], tag=7)
        again = pipe.collect()
        with pytest.raises(core.GpError):
            pipe.collect()          # nothing in flight
    for i, (r, (seqs, cnt)) in enumerate(zip(got, want)):
        assert core.split_polished(r) == seqs, i
        for k in ("a_ops", "b_sites", "b_edits", "c_gaps", "c_closed"):
            assert r["counters"][k] == cnt[k], (i, k)
        assert r["upload_ms"] > 0 and r["polish_ms"] > 0
    assert core.split_polished(again) == want[3][0]
    # and the oracle, batch by batch, for one of the sets
    w = ws[2]
    for c in range(w.n_contigs):
        outs, _, _ = oracle.polish_batch([w.names[c]], [w.contig(c)], [w.reads_of(c)], [int(w.kmer_threshold[c])])
        assert core.split_polished(got[2])[c] == outs[0]


@pytest.mark.parametrize("threads,mask_from", [("2", "0"), ("1", "0"), ("2", "100"), ("2", "3")])
def test_two_back_halves_at_once_come_back_in_submission_order(threads, mask_from, monkeypatch):
    """split sets (every set here: GP_PIPELINE_SPLIT_FROM lowered) are taken by two polishing threads when they
    run on CU-masked contexts (mask_from 0: every set; 3: the sets of three contigs and more); a short set behind
    a long one finishes first and must still be collected second.  mask_from 100: every set split WITHOUT masks
    (filter build on a plain context of its own, back half on the whole-path context, one polishing thread)."""
    monkeypatch.setenv("GP_PIPELINE_SPLIT_FROM", "1000")
    monkeypatch.setenv("GP_PIPELINE_MASK_FROM", mask_from)
    monkeypatch.setenv("GP_PIPELINE_BACK_THREADS", threads)
    from goldpolish_amd import core
    p = core.default_params()
    shapes = [[60000, 30000, 9000], [2000], [40000, 5000], [1500, 1200], [20000], [3000, 2500, 2000, 1800]]
    ws = [synth.make_workload(l, 25, seed=170 + i) for i, l in enumerate(shapes)]
    want = []
    with core.Context(0) as ctx:
        for w in ws:
            with core.Job.from_workload(ctx, p, w) as job:
                job.polish()
                want.append(job.sequences())
    sets = [core.batch_set_of_workload(w) for w in ws]
    with core.Pipeline(0, p, depth=4) as pipe:
        for rep in range(2):
            nxt = 0
            for nxt in range(4):
                pipe.submit(*sets[nxt], tag=10 * rep + nxt)
            nxt = 4
            for i in range(len(ws)):
                r = pipe.collect()
                assert r["tag"] == 10 * rep + i
                assert core.split_polished(r) == want[i], (rep, i)
                if nxt < len(ws):
                    pipe.submit(*sets[nxt], tag=10 * rep + nxt)
                    nxt += 1


def test_a_failing_set_is_reported_at_collect_and_the_stream_goes_on():
    from goldpolish_amd import core
    p = core.default_params(k_values=[32])
    good = synth.make_workload([6000, 3000], 20, seed=81)
    bad = synth.make_workload([5000], 20, seed=82)
    bad.kmer_threshold[:] = 3          # fill_bfs refuses thresholds below 4 (src/utils.cpp:105-107)
    with core.Context(0) as ctx, core.Job.from_workload(ctx, p, good) as job:
        job.polish()
        want = job.sequences()
    with core.Pipeline(0, p) as pipe:
        pipe.submit(*core.batch_set_of_workload(bad), tag=1)
        pipe.submit(*core.batch_set_of_workload(good), tag=2)
        with pytest.raises(core.GpError) as e:
            pipe.collect()
        assert e.value.code == -5 and "set 1" in str(e.value)
        r = pipe.collect()
        assert r["tag"] == 2 and core.split_polished(r) == want


def test_pool_recycles_device_blocks():
    """a destroyed job's blocks are what the next job of the same shape gets: no growth"""
    from goldpolish_amd import core
    p = core.default_params()
    w = synth.make_workload([20000, 8000], 20, seed=5)
    core.device_pool_trim()             # what earlier tests left behind
    with core.Context(0) as ctx:
        with core.Job.from_workload(ctx, p, w) as job:
            job.polish()
            first = job.sequences()
        pooled = core.device_pool_bytes()
        assert pooled > 3 * (1 << 30)
        with core.Job.from_workload(ctx, p, w) as job:
            job.polish()
            assert job.sequences() == first
            assert core.device_pool_bytes() < pooled // 8      # nearly everything was taken from the pool
        assert abs(core.device_pool_bytes() - pooled) < (64 << 20)
        core.device_pool_trim()
        assert core.device_pool_bytes() == 0


def test_device_copy_of_a_result_is_what_rccl_would_send():
    """bench.py's multi-GPU step sends the pipeline's HBM copy of the polished bytes (wrapped as a
    torch tensor without copying): it must be the bytes the host copy holds"""
    import torch
    import bench
    from goldpolish_amd import core
    p = core.default_params(k_values=[32, 24])
    w = synth.make_workload([12000, 90, 7000], 25, seed=91)
    with core.Pipeline(0, p) as pipe:
        pipe.submit(*core.batch_set_of_workload(w), tag=5)
        r = pipe.collect(copy=False)
        n = r["total_bytes"]
        assert n > 15000 and r["device_ptr"]
        t = torch.as_tensor(bench._DevView(r["device_ptr"], n), device="cuda")
        assert t.is_cuda and t.numel() == n and t.dtype == torch.uint8
        assert np.array_equal(t.cpu().numpy(), r["bytes"])
