"""GoldPolish-Target's window extraction, PAF rewrite and re-insertion in libgpcore.

Extraction and re-insertion are PINNED: the inputs and expected outputs are the reference's own
test data (tests/golden/target/README.md).  The PAF rewrite has no reference fixture (its input
comes from a mapper run at test time); it is checked against a restatement written from
scripts/goldpolish-target-update-mapping.py:70-155."""
import pathlib

import numpy as np
import pytest

G = pathlib.Path(__file__).parent / "golden" / "target"


def _core():
    from goldpolish_amd import core
    return core


def test_extraction_matches_the_reference_fixture(tmp_path):
    core = _core()
    out = tmp_path / "gaps.fa"
    core.target_extract(G / "input.fa", out, flank=64)
    assert out.read_bytes() == (G / "gaps.expected.fa").read_bytes()


@pytest.mark.parametrize("mapper", ["minimap2", "ntlink"])
def test_reinsertion_matches_the_reference_fixture(tmp_path, mapper):
    core = _core()
    out = tmp_path / "polished.fa"
    core.target_reinsert(G / "input.fa", G / f"{mapper}.gaps.goldpolished.expected.fa", out)
    assert out.read_bytes() == (G / f"{mapper}.polished.expected.fa").read_bytes()


def _fasta(path, recs):
    with open(path, "wb") as f:
        for h, s in recs:
            f.write(b">" + h + b"\n" + s + b"\n")


def test_extraction_rules_on_constructed_cases(tmp_path):
    core = _core()
    l = 5
    up, lo = lambda n: b"ACGTTGCAAC"[:n] if n <= 10 else (b"ACGTTGCAAC" * (n // 10 + 1))[:n], None
    seq = (up(30) + b"acgtacgt" + up(4) + b"ggg" + up(10) + b"t" + up(12) + b"cc" + up(3))
    #        upper30   lower8     upper4<2l: absorbed   upper10>=2l  single 't': no window  upper12  'cc'  tail
    _fasta(tmp_path / "a.fa", [(b"c1 some comment", seq), (b"c2", b"ACGT" * 10), (b"c3", b"acgtacgtac")])
    core.target_extract(tmp_path / "a.fa", tmp_path / "o.fa", flank=l)
    got = (tmp_path / "o.fa").read_bytes().split(b"\n")[:-1]
    # window 1: the lower run 8 + absorbed upper 4 + lower 3 = 15 bases from 30, flanks of 5
    assert got[0] == b">c1.1 25-50" and got[1] == seq[25:50].upper()
    # the single soft-masked base is skipped; 'cc' (length 2 > 1) is a window
    cc = 30 + 8 + 4 + 3 + 10 + 1 + 12
    assert got[2] == b">c1.2 %d-%d" % (cc - l, min(len(seq), cc + 2 + l)) and got[3] == seq[cc - l:cc + 2 + l].upper()
    # an all-lower-case record is one window clipped to the record
    assert got[4] == b">c3.1 0-10" and got[5] == b"ACGTACGTAC" and len(got) == 6
    # BED mode: intervals closer than 2 l merge; flanks clip at the ends
    (tmp_path / "r.bed").write_bytes(b"c1\t3\t6\nc1\t12\t15\nc1\t40\t44\nc9\t1\t2\nc2\t38\t40\n")
    core.target_extract(tmp_path / "a.fa", tmp_path / "b.fa", flank=l, bed_path=tmp_path / "r.bed")
    got = (tmp_path / "b.fa").read_bytes().split(b"\n")[:-1]
    assert got[0] == b">c1.1 0-20" and got[1] == seq[0:20].upper()
    assert got[2] == b">c1.2 35-49" and got[4] == b">c2.1 33-40" and len(got) == 6


def py_update_paf(gaps, rows):
    """restatement of update_paf_file; several overlapping windows: the leftmost one"""
    win = {}
    for name, comment in gaps:
        a, b = comment.split("-")[:2]
        win.setdefault(name.rsplit(".", 1)[0], []).append((int(a), int(b) + 1, name))
    out = []
    for r in rows:
        q_len, q_s, q_e, t_len, t_s, t_e, m, _, qual = map(int, (r[1], r[2], r[3], r[6], r[7], r[8], r[9], r[10], r[11]))
        hits = sorted(w for w in win.get(r[5], []) if w[0] < t_e and w[1] > t_s)
        if not hits:
            continue
        g_s, g_e, name = hits[0]
        nq_s, nq_e = q_s, q_e
        if g_s < t_s and g_e > t_e:
            nt_s, nt_e = t_s - g_s, t_e - g_s
        elif g_s < t_s and g_e <= t_e:
            nt_s, nt_e, nq_e = t_s - g_s, g_e - g_s, q_e - (t_e - g_e)
        elif g_s >= t_s and g_e > t_e:
            nt_s, nt_e, nq_s = 0, t_e - g_s, q_s + (g_s - t_s)
        else:
            nt_s, nt_e, nq_s, nq_e = 0, g_e - g_s, q_s + (g_s - t_s), q_e - (t_e - g_e)
        out.append("\t".join(map(str, [r[0], q_len, nq_s, nq_e, r[4], name, t_len, nt_s, nt_e, m, nt_e - nt_s, qual])) + "\r\n")
    return "".join(out)


def test_paf_rewrite_against_restatement(tmp_path):
    core = _core()
    rng = np.random.default_rng(4)
    gaps = [("ctgA.1", "100-400"), ("ctgA.2", "900-1500"), ("ctg.B.1", "0-250"), ("ctgC.1", "50-60")]
    with open(tmp_path / "gaps.fa", "w") as f:
        for n, c in gaps:
            f.write(f">{n} {c}\nACGT\n")
    rows = []
    for i in range(400):
        t = ["ctgA", "ctg.B", "ctgC", "ctgD"][int(rng.integers(0, 4))]
        s = int(rng.integers(0, 1600)); e = s + int(rng.integers(1, 700))
        qs = int(rng.integers(0, 500)); qe = qs + (e - s)
        rows.append([f"read{i}", "5000", str(qs), str(qe), "+-"[i % 2], t, "2000", str(s), str(e),
                     str(int(rng.integers(1, 500))), str(e - s), str(int(rng.integers(0, 61))), "tp:A:P", "cm:i:5"])
    (tmp_path / "in.paf").write_text("".join("\t".join(r) + "\n" for r in rows))
    core.target_update_paf(tmp_path / "gaps.fa", tmp_path / "in.paf", tmp_path / "out.paf")
    assert (tmp_path / "out.paf").read_bytes().decode() == py_update_paf(gaps, rows)
    with pytest.raises(core.GpError):
        core.target_update_paf(tmp_path / "gaps.fa", tmp_path / "missing.paf", tmp_path / "o.paf")
