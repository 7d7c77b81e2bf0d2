"""BASELINE.json's configurations that no other test runs, each through the C ABI on a
real MI355X against the CPU oracle:

  configs[0]  the reference's own fixture contigs (tests/golden/configs0/, a data file of the
              reference's test-suite: 152 ragged contigs, 2-43 kbp) + reads synthesised from them as
              SURVEY 8d prescribes (seed 1, 30x, log-normal 15 kb, 6 % errors, phred U[7,20]), k = 32,
              bsize = 1 -- and the full cascade on a draft with 1 % errors;
  configs[2]  in SURVEY 8d's own shape: a few contigs of 0.5-4 Mbp, 40x, full cascade, every filter
              at full size against the oracle's serial fill_bfs (the long-contig regime: grid-wide
              schedule, saturating filters);
  configs[3]  a rank's share: ONE job with more than 2^32 read positions, checked by a property
              that does not depend on size (replicas of one assembly must come out alike, wherever
              their reads lie) on top of a full oracle comparison of the first replica;
  configs[4]  GoldPolish-Target with --bed: thousands of BED intervals -> windows -> polished ->
              re-inserted, against the same chain with the oracle polishing."""
import concurrent.futures
import pathlib

import numpy as np
import pytest

import feeder_restatement as R
from goldpolish_amd import synth

pytestmark = pytest.mark.gpu
FIXTURE = pathlib.Path(__file__).parent / "golden" / "configs0" / "goldrush_test_golden_path.fa"


@pytest.fixture(scope="module")
def gp():
    from goldpolish_amd import core
    ctx = core.Context(0)
    yield core, ctx
    ctx.close()


def oracle_all(oracle, w, params, threads=0):
    """every contig its own batch, batches in parallel (the oracle's batch driver)"""
    outs, _ = oracle.polish_batches(w.names, [w.contig(c) for c in range(w.n_contigs)],
                                    [w.reads_of(c) for c in range(w.n_contigs)],
                                    [int(t) for t in w.kmer_threshold], list(range(w.n_contigs + 1)),
                                    params=oracle.batch_params(list(params.k[:params.n_k])), n_threads=threads)
    return outs


def test_configs0_reference_fixture_contigs(gp, oracle):
    core, ctx = gp
    names, contigs = synth.read_fasta(FIXTURE)
    assert len(contigs) == 152 and sum(len(c) for c in contigs) == 1_141_569
    # (a) BASELINE configs[0] as SURVEY 8d words it: the fixture as the draft, k = 32, bsize = 1
    w = synth.make_workload(None, 30, seed=1, draft_err=0.0, truths=contigs, names_in=names)
    assert [w.contig(c) for c in range(3)] == [contigs[c].tobytes() for c in range(3)]
    p = core.default_params(k_values=[32])
    with core.Job.from_workload(ctx, p, w) as job:
        job.polish()
        got = job.sequences()
        for c in (0, 17, 80, 151):   # filters too, for a few batches
            _, bfs = oracle.fill_bfs(w.reads_of(c), [32], int(w.kmer_threshold[c]), p.cbf_bytes, p.bf_bytes)
            assert np.array_equal(job.get_filter(c, 0), bfs[0]), c
    want = oracle_all(oracle, w, p)
    assert got == want
    # (b) the same contigs as the truth of a draft with 1 % errors, full cascade: edits and gap
    # fills really happen on real sequence (repeats, low complexity)
    w = synth.make_workload(None, 30, seed=1, draft_err=0.01, truths=contigs, names_in=names)
    p = core.default_params()
    with core.Job.from_workload(ctx, p, w) as job:
        job.polish()
        got = job.sequences()
        c = job.counters()
    assert c["b_edits"] > 1000 and c["c_closed"] > 0
    assert got == oracle_all(oracle, w, p)
    assert sum(g != w.contig(i) for i, g in enumerate(got)) > 100   # nearly every contig was edited


def test_configs2_survey_shape_long_contigs(gp, oracle):
    """contigs ~ log-normal around 2 Mbp, 40x, k = 32,28,24,20 (SURVEY 8d config 3's shape, a sample
    of it): the filters of such batches fill up, which is what the reference's fixed 10 MiB / 512 KiB
    sizes do with them; every filter byte is compared, and the polished contigs with what the
    oracle's ntEdit cascade, mask and Sealer make of the drafts given those filters"""
    core, ctx = gp
    lens = [3_600_000, 1_900_000, 1_100_000, 600_000]
    w = synth.make_workload_chunked(lens, 40, seed=3, chunk=1)
    p = core.default_params()
    ks = list(p.k[:p.n_k])
    with core.Job.from_workload(ctx, p, w) as job:
        job.polish()
        got = job.sequences()
        filt = [[job.get_filter(c, ki).copy() for ki in range(p.n_k)] for c in range(w.n_contigs)]
        cnt = job.counters()
        job.polish()                                   # idempotence
        assert job.sequences() == got
    assert cnt["a_reservations"] > 0                   # the long batches took the grid-wide schedule

    def fill(c):
        _, bfs = oracle.fill_bfs(w.reads_of(c), ks, int(w.kmer_threshold[c]), p.cbf_bytes, p.bf_bytes)
        return bfs

    with concurrent.futures.ThreadPoolExecutor(len(lens)) as ex:   # ctypes calls release the GIL
        want_f = list(ex.map(fill, range(w.n_contigs)))
    for c in range(w.n_contigs):
        for ki in range(p.n_k):
            assert np.array_equal(filt[c][ki], want_f[c][ki]), (c, ki)

    def back(c):
        cur = w.contig(c)
        for ki, k in enumerate(ks):
            cur, _ = oracle.ntedit(cur, want_f[c][ki], k)
        m = oracle.mask(cur, ks[0])
        s, _ = oracle.sealer(m, ks, want_f[c])
        return s.upper()

    with concurrent.futures.ThreadPoolExecutor(len(lens)) as ex:
        want = list(ex.map(back, range(w.n_contigs)))
    assert got == want


def test_configs3_share_past_2_to_32_read_positions(gp, oracle):
    """One job whose read buffer is longer than 2^32 bytes (a rank's share of the 3 Gbp / 8 GPU
    configuration holds 45 G positions): ten replicas of one 11 Mbp assembly in one job.  Replica r
    must come out exactly as replica 0 -- the same contigs and reads, only at other positions, the
    last ones past 2^32 -- and replica 0 is compared with the oracle in full."""
    core, ctx = gp
    base = synth.make_workload_chunked([25_000] * 440, 40, seed=4, chunk=40)
    reps = 10
    w = synth.concat_workloads([base] * reps)
    assert int(w.read_off[-1]) + len(w.read_off) - 1 > 2**32   # positions = bases + one separator per read
    p = core.default_params()
    with core.Job.from_workload(ctx, p, w) as job:
        job.polish()
        got = job.sequences()
        cnt = job.counters()
        n = base.n_contigs
        last = [[job.get_filter((reps - 1) * n + c, ki).copy() for ki in range(p.n_k)] for c in (0, n - 1)]
        first = [[job.get_filter(c, ki).copy() for ki in range(p.n_k)] for c in (0, n - 1)]
    for r in range(1, reps):
        assert got[r * n:(r + 1) * n] == got[:n], r
    for a, b in zip(first, last):
        for ki in range(p.n_k):
            assert np.array_equal(a[ki], b[ki])
    ks = list(p.k[:p.n_k])
    for j, c in enumerate((0, n - 1)):
        _, bfs = oracle.fill_bfs(base.reads_of(c), ks, int(base.kmer_threshold[c]), p.cbf_bytes, p.bf_bytes)
        for ki in range(p.n_k):
            assert np.array_equal(last[j][ki], bfs[ki]), (c, ki)
    assert got[:n] == oracle_all(oracle, base, p)


def test_configs4_target_mode_with_bed(gp, oracle, tmp_path):
    """GoldPolish-Target --bed (BASELINE configs[4] in SURVEY 8d's shape, a sample of it): BED
    intervals of 50-2000 bases placed uniformly on an assembly, windows = interval +- 64 with the
    reference's merge rule (gp_target_extract), the PAF moved onto the windows, every window a
    batch of its own through the whole path, polished windows re-inserted."""
    core, ctx = gp
    rng = np.random.default_rng(5)
    asm, fq, paf, bed = tmp_path / "asm.fa", tmp_path / "reads.fq", tmp_path / "asm.paf", tmp_path / "t.bed"
    rows, n_read, n_iv = [], 0, 0
    with open(asm, "wb") as fa, open(fq, "wb") as fr, open(bed, "wb") as fb:
        for ci, n in enumerate([4_800_000, 2_600_000, 1_400_000, 180_000]):
            truth = synth.random_genome(n, rng)
            draft = synth.mutate(truth, 0.01, rng)
            name = b"scaf%d" % ci
            fa.write(b">" + name + b"\n" + draft.tobytes() + b"\n")
            # ~3000 intervals over the assembly, sorted by start as a BED file is (the reference
            # merges neighbours in file order); some overlap or lie closer than two flanks
            for a in sorted(int(x) for x in rng.integers(0, draft.size - 2100, size=int(n / 3000))):
                fb.write(name + b"\t%d\t%d\n" % (a, a + int(rng.integers(50, 2001))))
                n_iv += 1
            got = 0
            while got < 25 * n:
                L = int(np.clip(rng.lognormal(np.log(8000.0), 0.5), 500, n))
                s0 = int(rng.integers(0, n - L + 1))
                frag = truth[s0:s0 + L]
                strand = b"+"
                if rng.random() < 0.5:
                    frag, strand = synth.revcomp(frag), b"-"
                rd = synth.mutate(frag, 0.05, rng)
                rid = b"read%06d" % n_read
                n_read += 1
                q = bytes(rng.integers(33 + 5, 33 + 30, size=rd.size, dtype=np.uint8))
                fr.write(b"@" + rid + b"\n" + rd.tobytes() + b"\n+\n" + q + b"\n")
                rows.append(b"\t".join([rid, b"%d" % rd.size, b"0", b"%d" % rd.size, strand, name, b"%d" % draft.size,
                                        b"%d" % s0, b"%d" % min(s0 + L, draft.size), b"%d" % L, b"%d" % L, b"60"]))
                got += L
    rng.shuffle(rows)
    paf.write_bytes(b"\n".join(rows) + b"\n")

    gaps, gpaf = tmp_path / "gaps.fa", tmp_path / "gaps.paf"
    core.target_extract(asm, gaps, flank=64, bed_path=bed)
    core.target_update_paf(gaps, paf, gpaf)
    recs = gaps.read_bytes().split(b"\n")[:-1]
    heads, wins = recs[0::2], recs[1::2]
    assert 1500 <= len(wins) <= n_iv            # close intervals are merged
    ids = [h[1:].split(b" ")[0] for h in heads]
    s = 400.0
    p = core.default_params()

    with core.Context(0) as c2, core.SeqIndex(gaps) as ti, core.SeqIndex(fq) as ri, core.Mappings(gpaf, ti) as m:
        with core.Job.from_files(c2, p, ti, ri, m, [[i] for i in ids], s) as job:
            job.polish()
            job.write_fasta(tmp_path / "gaps.polished.fa")
            cnt = job.counters()
    core.target_reinsert(asm, tmp_path / "gaps.polished.fa", tmp_path / "out.fa")
    assert cnt["b_edits"] > 1000

    tix, _ = R.build_index(gaps)
    rix, _ = R.build_index(fq)
    maps = R.load_mappings(gpaf, tix)
    reads_bytes = fq.read_bytes()
    sel = [R.select_reads(len(win), maps.get(wid, []), rix, s) for wid, win in zip(ids, wins)]
    outs, _ = oracle.polish_batches([h[1:] for h in heads], wins,
                                    [[reads_bytes[rix[r][0]:rix[r][0] + rix[r][1]] for r in chosen] for chosen, _ in sel],
                                    [thr for _, thr in sel], list(range(len(wins) + 1)),
                                    params=oracle.batch_params(list(p.k[:p.n_k])))
    with open(tmp_path / "gaps.expected.fa", "wb") as f:
        for h, o in zip(heads, outs):
            if o is not None:
                f.write(h + b"\n" + o + b"\n")
    assert (tmp_path / "gaps.polished.fa").read_bytes() == (tmp_path / "gaps.expected.fa").read_bytes()
    core.target_reinsert(asm, tmp_path / "gaps.expected.fa", tmp_path / "expected.fa")
    assert (tmp_path / "out.fa").read_bytes() == (tmp_path / "expected.fa").read_bytes()
    before = asm.read_bytes().split(b"\n")[1::2]
    after = (tmp_path / "out.fa").read_bytes().split(b"\n")[1::2]
    assert sum(a != b for a, b in zip(before, after)) >= 3
