"""CPU models of four device-side shortcuts (no GPU, no library): each restates, in a few lines of
Python, the straightforward procedure and the shortcut a kernel takes instead, and checks that they
agree on random inputs -- the argument the kernel's comment makes, executed.

1. `kmer_hash_from_table` (goldpolish_amd/csrc/gp_device.h): the ntHash of a k-mer as the XOR of k
   entries of a table of rotated seeds per strand, against the reference's own values
   (tests/golden/nthash_kat.json, made with /root/reference's nthash.hpp).
2. `gather_plan` / `advance` (stage_b.hip): the walker's window and head a node at a time, against
   the char-by-char get_char / increment loop (ntedit.cpp's node list semantics).
3. `census_scatter`'s ring (stage_a.hip): records staged in eight (sixteen) LDS slots per range and
   written in groups of four (eight), against storing every record at its slot.
4. `census_mark` / `census_pack`: marked records packed into the head of their run and read back up
   to the first record without the mark, against a flag per touch.
5. `mod_5_pow2` (gp_device.h): x mod (5 * 2^s) by folding 16- and 8-bit pieces (2^16 and 2^8 are 1
   modulo 5) and t / 5 = (t * 205) >> 10, against Python's %; and the extension hashes 1..3 from ONE
   multiply (census_scatter): base * (i ^ M) = base * M + d_i * base with d_i = ((M & 3) ^ i) - (M & 3).
6. `census_scatter`'s range word: a record's rank (low half) and the range's window base (high half)
   from one atomic add, against a counter and a base kept apart.
7. `census_run_index` (stage_a.hip): where the run of a (range, stretch) lies -- segments of 128 stretches,
   ranges inside a segment, stretches inside a range (a pair of thousands of stretches keeps a segment's
   runs of one range next to each other) -- is one-to-one onto [0, census_runs), and for pairs of up to 128
   stretches the same as the plain range-major order; and `census_mark_seg`'s table carried from segment to
   segment gives the same first touches as one table over all the stretches."""
import json
import pathlib

import numpy as np

M64 = (1 << 64) - 1
SEED = {"A": 0x3c8bfbb395c60474, "C": 0x3193c18562a02b4c, "G": 0x20323ed082572324, "T": 0x295549f54be24456}
COMP = {"A": "T", "C": "G", "G": "C", "T": "A"}


def sroln(x, n):
    """split rotate: the high 31 and the low 33 bits each rotate left by n (gp_device.h)"""
    hi, lo = x >> 33, x & 0x1FFFFFFFF
    a, b = n % 31, n % 33
    hi = ((hi << a) | (hi >> (31 - a))) & 0x7FFFFFFF
    lo = ((lo << b) | (lo >> (33 - b))) & 0x1FFFFFFFF
    return ((hi << 33) | lo) & M64


def test_hash_from_rotated_seeds_equals_the_reference_nthash():
    kat = json.loads((pathlib.Path(__file__).parent / "golden" / "nthash_kat.json").read_text())
    checked = 0
    for case in kat:
        k, seq = case["k"], case["seq"].upper()
        table = {c: [sroln(SEED[c], j) for j in range(k)] for c in "ACGT"}
        for p, (fwd, rev) in enumerate(zip(case["fwd"], case["rev"])):
            kmer = seq[p:p + k]
            if len(kmer) < k or any(c not in "ACGT" for c in kmer):
                continue
            f = r = 0
            for i, c in enumerate(kmer):
                f ^= table[c][k - 1 - i]
                r ^= table[COMP[c]][i]
            assert f == fwd and r == rev, (k, p)
            checked += 1
    assert checked > 100


# ---------------------------------------------------------------------------------- walker cursor
def node_get(nodes, ni):
    return nodes[ni] if ni < len(nodes) else (-1, 0, 0, 0)


def increment(nodes, pos, ni):
    t, s, e, c = node_get(nodes, ni)
    if t == 0:
        pos += 1
        if pos > e:
            ni += 1
            if ni < len(nodes) and nodes[ni][0] == 0:
                pos = nodes[ni][1]
    elif t == 1:
        ni += 1
        if ni < len(nodes) and nodes[ni][0] == 0:
            pos = nodes[ni][1]
    return pos, ni


def gather_chars(nodes, seq, pos, ni, want):
    """stage_b.hip's gather before round 2: one char, one increment at a time"""
    z = []
    while len(z) < want:
        if pos >= len(seq) or ni >= len(nodes):
            break
        t, s, e, c = nodes[ni]
        if t == -1:
            z += [0] * (want - len(z))
            break
        z.append(seq[pos] if t == 0 else c)
        pos, ni = increment(nodes, pos, ni)
    return z


def gather_nodes(nodes, seq, pos, ni, want):
    """gather_plan + gather_fill"""
    z = [None] * want
    n, segs = 0, []
    while n < want:
        if pos >= len(seq) or ni >= len(nodes):
            break
        t, s, e, c = nodes[ni]
        if t == 0:
            take = e - pos + 1 if pos <= e else 1
            take = min(take, want - n, len(seq) - pos)
            segs.append((n, pos, take))
            n += take
            pos += take
            if pos > e:
                ni += 1
                if ni < len(nodes) and nodes[ni][0] == 0:
                    pos = nodes[ni][1]
        elif t == 1:
            z[n] = c
            n += 1
            ni += 1
            if ni < len(nodes) and nodes[ni][0] == 0:
                pos = nodes[ni][1]
        else:
            while n < want:
                z[n] = 0
                n += 1
            break
    for z0, p, cnt in segs:
        for i in range(cnt):
            z[z0 + i] = seq[p + i]
    return z[:n]


def advance(nodes, pos, ni, steps):
    while steps:
        t, s, e, c = node_get(nodes, ni)
        if t == 0:
            if pos < e:
                m = min(steps, e - pos)
                pos += m
                steps -= m
                if not steps:
                    break
            pos += 1
            steps -= 1
            ni += 1
            if ni < len(nodes) and nodes[ni][0] == 0:
                pos = nodes[ni][1]
        elif t == 1:
            ni += 1
            steps -= 1
            if ni < len(nodes) and nodes[ni][0] == 0:
                pos = nodes[ni][1]
        else:
            break
    return pos, ni


def random_nodes(rng, n_seq):
    """a node list as edits leave it: position runs (type 0) with inserted bases (type 1) between
    them, runs that skip deleted stretches, now and then an unset node (-1) or a run that ends past
    the sequence"""
    nodes, pos = [], 0
    while pos < n_seq and len(nodes) < 30:
        r = rng.random()
        if r < 0.55:
            e = min(n_seq + int(rng.integers(-2, 3)), pos + int(rng.integers(0, 25)))
            e = max(e, pos)
            nodes.append((0, pos, e, 0))
            pos = e + 1 + int(rng.integers(0, 4))        # a deletion now and then
        elif r < 0.95:
            nodes.append((1, 0, 0, int(rng.integers(65, 90))))
        else:
            nodes.append((-1, 0, 0, 0))
    return nodes


def test_walker_window_and_head_a_node_at_a_time():
    rng = np.random.default_rng(7)
    for _ in range(3000):
        n_seq = int(rng.integers(5, 120))
        seq = [int(x) for x in rng.integers(97, 123, size=n_seq)]
        nodes = random_nodes(rng, n_seq)
        if not nodes:
            continue
        ni = int(rng.integers(0, len(nodes)))
        t, s, e, c = nodes[ni]
        # a cursor as the walker holds it: inside the run, at times just past its end; for an
        # inserted base the position is whatever the last run left
        pos = int(rng.integers(s, e + 2)) if t == 0 else int(rng.integers(0, n_seq + 2))
        want = int(rng.integers(1, 90))
        assert gather_nodes(nodes, seq, pos, ni, want) == gather_chars(nodes, seq, pos, ni, want)
        steps = int(rng.integers(0, 60))
        p1, n1 = pos, ni
        for _s in range(steps):
            p1, n1 = increment(nodes, p1, n1)
        assert advance(nodes, pos, ni, steps) == (p1, n1)


# ---------------------------------------------------------------------------------- scatter ring
K_RUN = 64
GEOMETRIES = [(64, 8), (128, 16)]       # (RUN, RING) of census_scatter: narrow and wide ranges


def scatter_with_ring(tiles, run=64, ring_n=8):
    """tiles: records a range gets in each tile of a stretch.  Returns the run as memory holds it.
    Groups of ring_n / 2 records leave the ring together (16 or 32 bytes)."""
    G = ring_n // 2
    mem = [None] * run
    ring = [None] * ring_n
    hist, gbase = 0, 0
    for recs in tiles:
        for r in recs:
            rank = hist
            hist += 1
            if rank >= run:
                continue                                 # the pair falls back (overflow)
            if 0 <= rank - G * gbase < ring_n:
                ring[rank & (ring_n - 1)] = r
            else:
                assert mem[rank] is None
                mem[rank] = r
        cnt = min(hist, run)
        if cnt > G * gbase + ring_n:
            ng, g_next = 2, (cnt + G - 1) // G
        else:
            ng = (cnt - G * gbase) // G if cnt > G * gbase else 0
            g_next = gbase + ng
        for g in range(gbase, gbase + ng):
            for i in range(G):
                assert mem[G * g + i] is None            # a group is written one way only
                mem[G * g + i] = ring[G * (g & 1) + i]
        gbase = g_next
    n = min(hist, run)
    for r in range(G * gbase, n):
        assert mem[r] is None
        mem[r] = ring[r & (ring_n - 1)]
    return mem[:n]


def test_scatter_ring_leaves_every_record_at_its_slot():
    rng = np.random.default_rng(11)
    for run, ring_n in GEOMETRIES:
        for _ in range(4000):
            lam = float(rng.choice([0.5, 3.2, 6.4, 12.0, 20.0]))
            tiles, nxt = [], 0
            for _t in range(8):
                n = int(rng.poisson(lam))
                tiles.append(list(range(nxt, nxt + n)))
                nxt += n
            want = list(range(min(nxt, run)))
            assert scatter_with_ring(tiles, run, ring_n) == want


# ---------------------------------------------------------------------------------- mark and pack
def test_marked_records_packed_into_the_run_head():
    rng = np.random.default_rng(13)
    for it in range(3000):
        K_RUN = GEOMETRIES[it & 1][0]
        cnt = int(rng.integers(0, K_RUN + 1))
        stale = [int(x) for x in rng.integers(0, 1 << 32, size=K_RUN)]   # an earlier cohort's leftovers
        recs = [int(x) for x in rng.integers(0, 1 << 29, size=cnt)]      # bit 31 clear: this cohort's records
        later = [bool(x) for x in rng.integers(0, 2, size=cnt)]
        run = recs + stale[cnt:]
        # census_mark: the marked ones to the front, bit 31 set; a zero behind them when all are marked
        packed = [r | (1 << 31) for r, m in zip(recs, later) if m]
        for i, r in enumerate(packed):
            run[i] = r
        if len(packed) == cnt and cnt < K_RUN:
            run[cnt] = 0
        # census_pack: read up to the first record without the bit
        got = []
        for r in run:
            if not r >> 31:
                break
            got.append(r & ~(1 << 31))
        assert got == [r for r, m in zip(recs, later) if m]


# ---------------------------------------------------------------------------------- index arithmetic
MULTISEED, MULTISHIFT = 0x90b45d39fb6da1fa, 27


def mod_5_pow2(x, s):
    q = x >> s
    lo, hi = q & 0xFFFFFFFF, q >> 32
    t2 = (lo & 0xFFFF) + (lo >> 16) + (hi & 0xFFFF) + (hi >> 16)
    assert t2 < 1 << 18
    t3 = (t2 & 0xFF) + ((t2 >> 8) & 0xFF) + (t2 >> 16)
    assert t3 < 1024                                  # where (t * 205) >> 10 is exact
    r5 = t3 - 5 * ((t3 * 205) >> 10)
    return (r5 << s) | (x & ((1 << s) - 1))


def test_fold_modulo_five_times_a_power_of_two():
    rng = np.random.default_rng(17)
    for t in range(1024):
        assert (t * 205) >> 10 == t // 5
    for s in list(range(0, 32)):
        xs = [0, 1, 4, 5, M64, M64 - 1, (1 << 63), (5 << s) - 1, 5 << s] + [int(v) for v in rng.integers(0, 1 << 63, size=400)]
        xs += [int(v) | (1 << 63) for v in rng.integers(0, 1 << 63, size=100)]
        for x in xs:
            assert mod_5_pow2(x & M64, s) == (x & M64) % (5 << s), (x, s)


def test_extension_hashes_from_one_multiply():
    rng = np.random.default_rng(19)
    for k in range(1, 64):
        m = (k * MULTISEED) & M64
        m2 = m & 3
        for base in [0, 1, M64] + [int(v) for v in rng.integers(0, 1 << 63, size=40)]:
            bm = (base * m) & M64
            for i in (1, 2, 3):
                d = (m2 ^ i) - m2
                want = (base * (i ^ m)) & M64                     # btllib's multi-hash: base * (i ^ k * seed)
                got = (bm + d * base) & M64
                assert got == want, (k, i, base)
                want_h = want ^ (want >> MULTISHIFT)
                assert (got ^ (got >> MULTISHIFT)) == want_h
        if k % 2 == 0:                                            # even k: the seed is even, so M & 3 == 0 and d = 1, 2, 3
            assert m2 == 0


def test_rank_and_window_base_share_a_word():
    """hist word = records so far | (groups flushed << 16): what the scatter's atomicAdd returns tells a
    record its rank AND whether its slot is in the ring's window, as two separate variables would"""
    rng = np.random.default_rng(23)
    for run, ring_n in GEOMETRIES:
        G = ring_n // 2
        for _ in range(500):
            word, cnt, gbase = 0, 0, 0
            for _round in range(8):
                for _ in range(int(rng.poisson(6.4))):
                    old = word
                    word += 1                                     # the atomic add
                    rank, g = old & 0xFFFF, old >> 16
                    assert (rank, g) == (cnt, gbase)
                    cnt += 1
                # the flush: only the high half changes, the count stays
                c = min(word & 0xFFFF, run)
                g0 = word >> 16
                if c > G * g0 + ring_n:
                    g_next = (c + G - 1) // G
                else:
                    g_next = g0 + ((c - G * g0) // G if c > G * g0 else 0)
                word = (word & 0xFFFF) | (g_next << 16)
                gbase = g_next
                assert word & 0xFFFF == cnt and g_next <= (run + G - 1) // G + 1


K_SEG = 128


def census_run_index(nst, n_buckets, bucket, stretch):
    stride = min(nst, K_SEG)
    return ((stretch // K_SEG) * n_buckets + bucket) * stride + stretch % K_SEG


def census_runs(nst, n_buckets):
    return ((nst + K_SEG - 1) // K_SEG) * n_buckets * min(nst, K_SEG)


def test_segmented_run_index_is_one_to_one_and_range_major_for_short_pairs():
    for nst, n_buckets in [(1, 5), (7, 640), (123, 640), (128, 3), (129, 3), (300, 17), (4900, 4)]:
        seen = set()
        for b in range(n_buckets):
            for st in range(nst):
                i = census_run_index(nst, n_buckets, b, st)
                assert 0 <= i < census_runs(nst, n_buckets) and i not in seen
                seen.add(i)
                if nst <= K_SEG:
                    assert i == b * nst + st          # what the workgroup-local census has always used
        assert len(seen) == nst * n_buckets
        # a range's runs of one segment are neighbours (what census_mark_seg's waves read together)
        for st in range(0, nst - 1):
            if st // K_SEG == (st + 1) // K_SEG:
                assert census_run_index(nst, n_buckets, 2 % n_buckets, st + 1) == census_run_index(nst, n_buckets, 2 % n_buckets, st) + 1


def test_first_touches_with_the_table_carried_from_segment_to_segment():
    """census_mark_seg: positions only grow from segment to segment, so after a segment's minima are in, the
    table is final for every counter that segment touches; a touch is its counter's first iff the table
    holds its own position when its segment is marked"""
    rng = np.random.default_rng(5)
    n_counters, per_stretch, nst, stretch_len = 257, 40, 700, 8192
    touches = []                                            # (position, counter), stretch by stretch
    for st in range(nst):
        pos = np.sort(rng.integers(0, stretch_len, size=per_stretch)) + st * stretch_len
        touches.append(list(zip(pos.tolist(), rng.integers(0, n_counters, size=per_stretch).tolist())))
    first = {}
    for run in touches:
        for p, c in run:
            first[c] = min(first.get(c, p), p)
    want = [[first[c] != p for p, c in run] for run in touches]      # marked = not the counter's first
    table = {}
    got = []
    for seg0 in range(0, nst, K_SEG):
        seg = touches[seg0:seg0 + K_SEG]
        for run in seg:                                      # the segment's atomicMin pass
            for p, c in run:
                table[c] = min(table.get(c, 1 << 62), p)
        for run in seg:                                      # ... then its marks
            got.append([table[c] != p for p, c in run])
    assert got == want
