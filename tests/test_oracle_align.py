"""Sealer's consensus step against known answers produced by the REFERENCE's
own Gotoh aligner and Sequence.cpp (tests/golden/make_nw_kat.py -> oracle/_ref/align_ref, which
links subprojects/sealer/Align/alignGlobal.cc + Common/Sequence.cpp from where they lie):
the oracle's restatement on the CPU, and -- `-m gpu` -- the device code stage C runs for
multi-path gaps, through the C ABI (gp_sealer_consensus)."""
import json
import pathlib

import pytest

GOLD = pathlib.Path(__file__).parent / "golden" / "nw_kat.json"


@pytest.fixture(scope="module")
def kat():
    v = json.loads(GOLD.read_text())
    assert len(v["pairs"]) >= 150 and len(v["multi"]) >= 50
    return v


def test_align_global_known_answers(oracle, kat):
    """alignGlobal (Align/alignGlobal.cc:55-198): consensus and number of matches."""
    for v in kat["pairs"]:
        cons, m = oracle.nw_consensus(v["a"], v["b"])
        assert cons.decode() == v["consensus"], (v["a"], v["b"])
        assert m == v["matches"]


def test_align_multi_known_answers(oracle, kat):
    """align() (alignGlobal.h:73-83): alignPair for two, alignMulti's fold (matches always 0) for more."""
    for v in kat["multi"]:
        cons, m = oracle.align_multi(v["seqs"])
        assert cons.decode() == v["consensus"]
        assert m == v["matches"]
        assert len(cons) == v["size"]


def test_revcomp_and_ambiguity_known_answers(oracle, kat):
    """reverseComplement (Common/Sequence.cpp:21-57) and ambiguityOr (Sequence.h:102-106)."""
    for v in kat["revcomp"]:
        assert oracle.revcomp(v["seq"]).decode() == v["rc"]
    codes = kat["codes"]
    for i, a in enumerate(codes):
        for j, b in enumerate(codes):
            assert oracle.amb_or(a, b).decode() == kat["amb_or"][i][j], (a, b)


@pytest.mark.gpu
def test_device_consensus_known_answers(kat):
    """The device's aligner (stage_c.hip nw_consensus, the code sealer_consensus runs) on the
    reference's vectors: every pair as a two-sequence set, every multi set as it is."""
    from goldpolish_amd import core

    sets = [[v["a"], v["b"]] for v in kat["pairs"]] + [v["seqs"] for v in kat["multi"]]
    want = [v["consensus"] for v in kat["pairs"]] + [v["consensus"] for v in kat["multi"]]
    with core.Context(0) as ctx:
        got = core.sealer_consensus(ctx, sets)
        assert [g.decode() for g in got] == want
        # limits are errors, not silent truncation
        with pytest.raises(core.GpError):
            core.sealer_consensus(ctx, [["A" * 1801, "ACGT"]])
        with pytest.raises(core.GpError):
            core.sealer_consensus(ctx, [["ACGT"]])
