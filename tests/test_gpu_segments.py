"""Long contigs walked as segments (stage_b.hip, SegDesc): whatever the segment length, and with
cuts forced where the records would not have put them (so that failed segments, their repair and
the whole-contig fallback all run), the polished bytes and the oracle-comparable counters must be
those of the one-workgroup walk -- which tests/test_gpu_configs.py compares with the oracle on
contigs of 0.6-3.6 Mbp (segmented by default)."""
import hashlib

import pytest

from goldpolish_amd import synth

pytestmark = pytest.mark.gpu


def run(core, p, w):
    with core.Context(0) as ctx, core.Job.from_workload(ctx, p, w) as job:
        job.polish()
        c = job.counters()
        return job.sequences(), tuple(c[k] for k in ("b_sites", "b_probes", "b_edits", "c_gaps", "c_closed"))


def test_segmented_walk_equals_the_whole_contig_walk(monkeypatch, oracle):
    from goldpolish_amd import core
    w = synth.make_workload_chunked([420_000, 160_000, 70_000, 9_000], 40, seed=23, chunk=1)
    p = core.default_params()
    monkeypatch.setenv("GP_WALK_SEG_MIN", "0")     # cut everything of two segment lengths and more
    monkeypatch.setenv("GP_WALK_SEG", "0")
    want = run(core, p, w)
    assert want[1][2] > 1000     # real edits
    for seg, force in (("65536", None), ("16384", None), ("8192", "1"), ("20000", "1")):
        monkeypatch.setenv("GP_WALK_SEG", seg)
        if force:
            monkeypatch.setenv("GP_WALK_SEG_FORCE", force)
        else:
            monkeypatch.delenv("GP_WALK_SEG_FORCE", raising=False)
        got = run(core, p, w)
        assert got[1] == want[1], (seg, force)
        assert got[0] == want[0], (seg, force)
    # and the oracle, for the contig that is cut into the most pieces
    outs, _, _ = oracle.polish_batch([w.names[1]], [w.contig(1)], [w.reads_of(1)], [int(w.kmer_threshold[1])])
    assert want[0][1] == outs[0]
