"""GoldPolish-Target end to end on the GPU (BASELINE.json configs[4] in miniature):
soft-masked stretches of an assembly -> windows with flanks (gp_target_extract) -> the mapper's
PAF moved onto the windows (gp_target_update_paf) -> the windows polished as ordinary targets
(index, mappings, read selection, the whole path on the device) -> put back (gp_target_reinsert).
Expected: the same chain with the polishing done by the CPU oracle on the reads the Python
restatement of serve_batch selects."""
import numpy as np
import pytest

import feeder_restatement as R
from goldpolish_amd import synth

pytestmark = pytest.mark.gpu


def test_target_mode_chain_equals_the_oracle_chain(tmp_path, oracle):
    from goldpolish_amd import core
    rng = np.random.default_rng(41)
    asm, fq, paf = tmp_path / "asm.fa", tmp_path / "reads.fq", tmp_path / "asm.paf"
    rows, n_read = [], 0
    with open(asm, "wb") as fa, open(fq, "wb") as fr:
        for ci, n in enumerate([30000, 18000, 7000]):
            truth = synth.random_genome(n, rng)
            draft = synth.mutate(truth, 0.01, rng).copy()
            # soft-mask a few stretches: those are what target mode polishes
            for _ in range(4 + ci):
                a = int(rng.integers(200, draft.size - 1200))
                b = a + int(rng.integers(50, 900))
                draft[a:b] = np.frombuffer(bytes(draft[a:b]).lower(), dtype=np.uint8)
            name = b"scaf%d" % ci
            fa.write(b">" + name + (b" len=%d" % n if ci == 1 else b"") + b"\n" + draft.tobytes() + b"\n")
            got = 0
            while got < 25 * n:
                L = int(np.clip(rng.lognormal(np.log(4000.0), 0.5), 500, n))
                s0 = int(rng.integers(0, n - L + 1))
                frag = truth[s0:s0 + L]
                strand = b"+"
                if rng.random() < 0.5:
                    frag, strand = synth.revcomp(frag), b"-"
                rd = synth.mutate(frag, 0.05, rng)
                rid = b"read%05d" % n_read
                n_read += 1
                q = bytes(rng.integers(33 + 5, 33 + 30, size=rd.size, dtype=np.uint8))
                fr.write(b"@" + rid + b"\n" + rd.tobytes() + b"\n+\n" + q + b"\n")
                rows.append(b"\t".join([rid, b"%d" % rd.size, b"0", b"%d" % rd.size, strand, name, b"%d" % draft.size,
                                        b"%d" % s0, b"%d" % min(s0 + L, draft.size), b"%d" % L, b"%d" % L, b"60"]))
                got += L
    rng.shuffle(rows)
    paf.write_bytes(b"\n".join(rows) + b"\n")

    gaps, gpaf = tmp_path / "gaps.fa", tmp_path / "gaps.paf"
    core.target_extract(asm, gaps, flank=64)
    core.target_update_paf(gaps, paf, gpaf)
    recs = gaps.read_bytes().split(b"\n")[:-1]
    heads, wins = recs[0::2], recs[1::2]
    assert len(wins) >= 10 and all(w == w.upper() for w in wins)
    ids = [h[1:].split(b" ")[0] for h in heads]
    s = 400.0      # enough reads for windows of a few hundred bases (size_t(len * s / 10000))
    p = core.default_params()

    # ---- device chain
    with core.Context(0) as ctx, core.SeqIndex(gaps) as ti, core.SeqIndex(fq) as ri, core.Mappings(gpaf, ti) as m:
        with core.Job.from_files(ctx, p, ti, ri, m, [[i] for i in ids], s) as job:
            job.polish()
            job.write_fasta(tmp_path / "gaps.polished.fa")
            c = job.counters()
    core.target_reinsert(asm, tmp_path / "gaps.polished.fa", tmp_path / "out.fa")
    assert c["b_edits"] > 0

    # ---- oracle chain
    tix, _ = R.build_index(gaps)
    rix, _ = R.build_index(fq)
    maps = R.load_mappings(gpaf, tix)
    reads_bytes = fq.read_bytes()
    with open(tmp_path / "gaps.expected.fa", "wb") as f:
        for h, wid, win in zip(heads, ids, wins):
            chosen, thr = R.select_reads(len(win), maps.get(wid, []), rix, s)
            rs = [reads_bytes[rix[r][0]:rix[r][0] + rix[r][1]] for r in chosen]
            outs, _, _ = oracle.polish_batch([h[1:]], [win], [rs], [thr])
            if outs[0] is not None:
                f.write(h + b"\n" + outs[0] + b"\n")
    assert (tmp_path / "gaps.polished.fa").read_bytes() == (tmp_path / "gaps.expected.fa").read_bytes()
    core.target_reinsert(asm, tmp_path / "gaps.expected.fa", tmp_path / "expected.fa")
    assert (tmp_path / "out.fa").read_bytes() == (tmp_path / "expected.fa").read_bytes()
    # the polished windows really changed the assembly, and only inside the windows' reach
    before = asm.read_bytes().split(b"\n")[1::2]
    after = (tmp_path / "out.fa").read_bytes().split(b"\n")[1::2]
    assert any(a != b for a, b in zip(before, after))
