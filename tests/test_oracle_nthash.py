"""The oracle's ntHash against known answers produced by the reference's own
in-tree header (tests/golden/make_nthash_kat.py)."""
import json
import pathlib

import numpy as np

GOLD = pathlib.Path(__file__).parent / "golden" / "nthash_kat.json"


def test_nthash_known_answers(oracle):
    vecs = json.loads(GOLD.read_text())
    assert len(vecs) >= 30
    L = oracle.L
    for v in vecs:
        k, seq = v["k"], v["seq"].encode()
        n = len(seq) - k + 1
        fwd = [L.gpo_ntf64(seq[p:p + k], k) for p in range(n)]
        rev = [L.gpo_ntr64(seq[p:p + k], k) for p in range(n)]
        assert fwd == v["fwd"]
        assert rev == v["rev"]
        # rolling canonical + 3 extension hashes
        h, valid = oracle.nthash_all(seq, k)
        for p in range(n):
            assert oracle.extend(h[p], k) == v["rolled"][p]
        # explicit roll primitives reproduce from-scratch values
        f, r = fwd[0], rev[0]
        for p in range(1, n):
            f = L.gpo_ntf64_roll(f, k, seq[p - 1], seq[p + k - 1])
            r = L.gpo_ntr64_roll(r, k, seq[p - 1], seq[p + k - 1])
            assert (f, r) == (fwd[p], rev[p])
        for ch, f2, r2, *hv in v["changelast"]:
            f = L.gpo_ntf64_changelast(fwd[0], k, seq[k - 1], ord(ch))
            r = L.gpo_ntr64_changelast(rev[0], k, seq[k - 1], ord(ch))
            assert (f, r) == (f2, r2)
            assert oracle.extend((f + r) & (2**64 - 1), k) == hv
        # validity = every byte of the window is one of ACGTacgt
        ok = np.array([all(c in b"ACGTacgt" for c in seq[p:p + k]) for p in range(n)])
        assert (valid == ok).all()
