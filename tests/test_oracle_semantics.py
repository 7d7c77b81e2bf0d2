"""CPU tests of the oracle: golden vectors, hand-built scenarios for each stage, and the
edge cases the reference handles (N / IUPAC / lower case, short contigs, the 0.75 guard)."""
import hashlib
import json
import pathlib
import re

import numpy as np
import pytest

from goldpolish_amd import synth

GOLD = pathlib.Path(__file__).parent / "golden"


def sha(b):
    return hashlib.sha256(bytes(b)).hexdigest()


# ------------------------------------------------------------------ stage A
def _bits(bf):
    return int(np.unpackbits(bf).sum())


def test_counting_threshold_semantics(oracle):
    """insert_thresh_contains returns the count AFTER insertion, so a k-mer enters the
    Bloom filter on its thr-th occurrence (thr = T-2+i for the i-th k)."""
    rng = np.random.default_rng(1)
    read = synth.random_genome(200, rng).tobytes()
    ks = [32, 28]
    T = 5  # thresholds 3 (k=32) and 4 (k=28)
    for n_copies, want32, want28 in [(2, False, False), (3, True, False), (4, True, True)]:
        _, bfs = oracle.fill_bfs([read] * n_copies, ks, T, 1 << 20, 1 << 16)
        assert (_bits(bfs[0]) > 0) == want32
        assert (_bits(bfs[1]) > 0) == want28
    # both strands count towards the same canonical k-mer
    rc = synth.revcomp(np.frombuffer(read, dtype=np.uint8)).tobytes()
    _, a = oracle.fill_bfs([read, rc, read], ks, T, 1 << 20, 1 << 16)
    _, b = oracle.fill_bfs([read, read, read], ks, T, 1 << 20, 1 << 16)
    assert np.array_equal(a[0], b[0])
    with pytest.raises(ValueError):
        oracle.fill_bfs([read], ks, 3, 1 << 20, 1 << 16)


def test_filter_is_order_dependent_and_skips_non_acgt(oracle):
    rng = np.random.default_rng(2)
    g = synth.random_genome(3000, rng).tobytes()
    reads = [g[i:i + 900] for i in range(0, 2100, 150)] * 2
    # small counting filter: collisions make the conservative update order-dependent, which
    # is why the device path must replay the reference's serial order exactly
    _, f1 = oracle.fill_bfs(reads, [20], 6, 8209, 4096)
    _, f2 = oracle.fill_bfs(reads[::-1], [20], 6, 8209, 4096)
    assert not np.array_equal(f1[0], f2[0])
    # every k-mer that holds an N (or IUPAC) is skipped; lower case counts like upper case
    r = bytearray(g[:300])
    r[150] = ord("N")
    with_n = [bytes(r)] * 4
    parts = [bytes(r[:150])] * 4 + [bytes(r[151:])] * 4
    _, a = oracle.fill_bfs(with_n, [32], 5, 1 << 20, 1 << 16)
    _, b = oracle.fill_bfs(parts, [32], 5, 1 << 20, 1 << 16)
    assert np.array_equal(a[0], b[0])
    _, c = oracle.fill_bfs([bytes(r).lower()] * 4, [32], 5, 1 << 20, 1 << 16)
    assert np.array_equal(a[0], c[0])


def test_threshold_formula_and_rank_cap(oracle):
    assert [oracle.kmer_threshold(b) for b in (0, 750_000, 1_563_000, 1_565_000, 10**8)] == \
        [5, 5, 5, 5, 13]
    assert oracle.kmer_threshold(3_920_000) == 5 and oracle.kmer_threshold(3_940_000) == 6
    # size_t(len*s/10000) truncates: a 249 bp contig at s=40 gets no reads at all
    order, bases = oracle.rank_reads(["a", "b"], [10.0, 20.0], [100, 100], 249, 40)
    assert order == [] and bases == 0
    order, bases = oracle.rank_reads(["a", "b"], [10.0, 20.0], [100, 100], 250, 40)
    assert order == [1] and bases == 100


# ------------------------------------------------------------------ stage B
def _exact_filter(oracle, seq, ks, bf_bytes=1 << 19):
    """filter holding exactly the k-mers of seq (each seen often enough)"""
    _, bfs = oracle.fill_bfs([seq] * 6, list(ks), 5, 1 << 22, bf_bytes)
    return bfs


def test_ntedit_fixes_substitution_insertion_deletion(oracle):
    rng = np.random.default_rng(3)
    truth = synth.random_genome(1200, rng).tobytes()
    bf = _exact_filter(oracle, truth, [32])[0]
    sub = bytearray(truth)
    sub[600] = ord("A") if truth[600] != ord("A") else ord("C")
    out, _ = oracle.ntedit(bytes(sub), bf, 32)
    assert out == truth
    # a base missing from the draft is inserted back; an extra base is deleted
    assert oracle.ntedit(truth[:500] + truth[501:], bf, 32)[0] == truth
    extra = b"G" if truth[500:501] != b"G" else b"T"
    assert oracle.ntedit(truth[:500] + extra + truth[500:], bf, 32)[0] == truth
    # 3 missing bases / 2 extra bases
    assert oracle.ntedit(truth[:700] + truth[703:], bf, 32)[0] == truth
    # a clean contig and a full filter are left alone
    assert oracle.ntedit(truth, bf, 32)[0] == truth
    full = np.full(bf.size, 255, np.uint8)
    assert oracle.ntedit(bytes(sub), full, 32)[0] == bytes(sub)


def test_ntedit_soft_masks_what_it_cannot_fix(oracle):
    rng = np.random.default_rng(4)
    truth = synth.random_genome(1000, rng).tobytes()
    bf = _exact_filter(oracle, truth, [32])[0]
    junk = synth.random_genome(60, rng).tobytes()
    draft = truth[:400] + junk + truth[460:]
    out, _ = oracle.ntedit(draft, bf, 32)
    assert out.upper() != truth and any(c in b"acgt" for c in out[395:470])
    p = oracle.ntedit_params(32, mask=0)
    out2, _ = oracle.ntedit(draft, bf, 32, params=p)
    assert out2 == out2.upper()
    # the last k positions can never be edited (the confirm step needs k more bases)
    tail = bytearray(truth)
    tail[-5] = ord("A") if truth[-5] != ord("A") else ord("C")
    assert oracle.ntedit(bytes(tail), bf, 32)[0] == bytes(tail)


def test_batch_drops_short_contigs_and_guard_restores(oracle):
    rng = np.random.default_rng(5)
    long_c = synth.random_genome(2000, rng).tobytes()
    shorts = [synth.random_genome(n, rng).tobytes() for n in (99, 60)]
    p = oracle.batch_params(run_sealer=0)
    # alone in its batch a 99 bp contig is dropped by ntEdit, then the 0.75 guard puts it back
    outs, st, _ = oracle.polish_batch(["s"], [shorts[0]], [[]], [5], p)
    assert st.ntedit_skipped == 1 and outs[0] == shorts[0]
    # next to a long contig the output stays above 75 % of the input bytes: it is gone
    outs, st, _ = oracle.polish_batch(["l", "s1", "s2"], [long_c] + shorts, [[], [], []],
                                      [5, 5, 5], p)
    assert st.ntedit_skipped == 0 and outs[1] is None and outs[2] is None and outs[0] == long_c


# ------------------------------------------------------------- mask / upper
def py_mask(seq: bytes, k: int) -> bytes:
    """the regular expression and the rule of scripts/goldpolish-mask:44-72, in Python"""
    s = seq.decode()
    s = s.upper() if len(s) < 2 * k else s[:k].upper() + s[k:-k] + s[-k:].upper()
    out = ""
    for t in re.findall(r"([ACTG]+|[Nn]+|[actgUNMRWSYKVHDBunmrwsykvhdb]+)", s):
        out += t if t[0] == "N" else (t.lower() if len(t) < k else t)
    out = out.strip("Nn")
    return (out or "N").encode()


def test_mask_hand_vectors_and_regex_restatement(oracle):
    for c in json.loads((GOLD / "mask_cases.json").read_text()):
        assert oracle.mask(c["seq"], c["k"]) == c["want"].encode() == py_mask(c["seq"].encode(), c["k"])
    rng = np.random.default_rng(6)
    alpha = [b"A", b"C", b"G", b"T", b"a", b"c", b"g", b"t", b"N", b"n", b"R", b"x", b"U", b"-",
             b"ACGTACGTAC", b"acgtacgtacg"]
    for _ in range(1500):
        s = b"".join(rng.choice(alpha, size=int(rng.integers(0, 60))).tolist())
        k = int(rng.integers(1, 12))
        assert oracle.mask(s, k) == py_mask(s, k)


# ------------------------------------------------------------------ stage C
def test_sealer_closes_simple_gaps(oracle):
    rng = np.random.default_rng(7)
    g = synth.random_genome(3000, rng).tobytes()
    ks = [32, 28, 24, 20]
    bfs = _exact_filter(oracle, g, ks)
    soft = g[:1000] + g[1000:1030].lower() + g[1030:]
    out, st = oracle.sealer(soft, ks, bfs)
    assert out == g and st.gaps_found == 1 and st.gaps_closed == 1 and st.unique_path == 1
    # an N run of the wrong length is replaced by the real sequence
    out, st = oracle.sealer(g[:1500] + b"N" * 7 + g[1540:], ks, bfs)
    assert out == g and st.gaps_closed == 1
    # a gap longer than -G 800 stays; a trailing gap is left alone
    far = g[:500] + b"N" * 10 + g[1400:]
    assert oracle.sealer(far, ks, bfs)[0] == far
    tail = g[:2990] + g[2990:].lower()
    assert oracle.sealer(tail, ks, bfs)[0] == tail
    # nothing in the graph: nothing changes
    empty = [np.zeros_like(b) for b in bfs]
    assert oracle.sealer(soft, ks, empty)[0] == soft


def test_sealer_consensus_of_two_paths(oracle):
    rng = np.random.default_rng(8)
    g = synth.random_genome(2500, rng).tobytes()
    h = bytearray(g)
    h[1200] = ord("A") if g[1200] != ord("A") else ord("C")
    ks = [32]
    _, bfs = oracle.fill_bfs([g] * 6 + [bytes(h)] * 6, ks, 5, 1 << 22, 1 << 19)
    soft = g[:1190] + g[1190:1210].lower() + g[1210:]
    out, st = oracle.sealer(soft, ks, bfs)
    assert st.multiple_paths == 1 and st.gaps_closed == 1 and len(out) == len(g)
    iupac = {frozenset(b"AC"): b"M", frozenset(b"AG"): b"R", frozenset(b"AT"): b"W",
             frozenset(b"CG"): b"S", frozenset(b"CT"): b"Y", frozenset(b"GT"): b"K"}
    want = iupac[frozenset([g[1200], h[1200]])]
    assert out[1200:1201] == want and out[:1200] == g[:1200] and out[1201:] == g[1201:]


# ------------------------------------------------------------------ goldens
def test_self_generated_pipeline_golden(oracle):
    """tests/golden/pipeline_small.json (made by make_pipeline_golden.py): the oracle today
    still produces what it produced when the vectors were frozen"""
    for case in json.loads((GOLD / "pipeline_small.json").read_text()):
        w = synth.make_workload(case["lens"], case["coverage"], seed=case["seed"], **case["extra"])
        assert sha(w.contig_buf.tobytes() + w.read_buf.tobytes()) == case["input_sha"]
        ks = case["k"]
        p = oracle.batch_params(k_values=tuple(ks))
        for c, want in enumerate(case["contigs"]):
            thr = int(w.kmer_threshold[c])
            assert thr == want["kmer_threshold"]
            _, bfs = oracle.fill_bfs(w.reads_of(c), ks, thr, p.cbf_bytes, p.bf_bytes)
            assert [sha(b.tobytes()) for b in bfs] == want["filters"]
            cur = w.contig(c)
            for ki, k in enumerate(ks):
                cur, _ = oracle.ntedit(cur, bfs[ki], k)
                assert sha(cur) == want["ntedit"][ki]["sha"]
            m = oracle.mask(cur, ks[0])
            assert sha(m) == want["mask"]["sha"]
            s, st = oracle.sealer(m, ks, bfs)
            assert sha(s) == want["sealer"]["sha"]
            assert (st.gaps_found, st.gaps_closed) == (want["sealer"]["gaps_found"],
                                                      want["sealer"]["gaps_closed"])
            assert sha(s.upper()) == want["final"]["sha"]


def test_indel_trial_strings_follow_the_reference_table(oracle):
    """tryIndels walks multi_possible_bases[first][0..340] (ntedit.cpp:198-343): every string of
    length 1..5 that starts with `first`, shortest first, then A < C < G < T.  The oracle computes
    the i-th string; here against that order spelled out, and -- where the reference tree is
    mounted (this container, not the GPU box) -- against the table's text itself."""
    import ctypes as C
    import itertools
    import pathlib
    import re
    L = oracle.L
    L.gpo_ntedit_indel_string.restype = C.c_uint
    L.gpo_ntedit_indel_string.argtypes = [C.c_ubyte, C.c_int, C.c_char_p]

    def ours(first):
        out = []
        for i in range(341):
            buf = C.create_string_buffer(8)
            n = L.gpo_ntedit_indel_string(ord(first), i, buf)
            out.append(buf.raw[:n].decode())
        return out

    for first in "ACGT":
        want = [first + "".join(t) for n in range(5) for t in itertools.product("ACGT", repeat=n)]
        assert ours(first) == want
    ref = pathlib.Path("/root/reference/subprojects/ntedit/ntedit.cpp")
    if ref.exists():
        src = ref.read_text()
        blk = src[src.index("multi_possible_bases = {"):]
        blk = blk[:blk.index("};")]
        tabs = {m.group(1): re.findall(r'"([ACGT]+)"', m.group(2))
                for m in re.finditer(r"\{\s*'([ACGT])',\s*\{(.*?)\}\s*\}", blk, re.S)}
        assert sorted(tabs) == list("ACGT")
        for first in "ACGT":
            assert ours(first) == tabs[first]
