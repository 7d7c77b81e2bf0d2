"""CPU model of stage A's third schedule, the census LEVEL BY LEVEL (stage_a.hip, stage_a_levels): for a
filter with one threshold, which operations of a conservative-update counting filter insert their k-mer
is decided by repeated censuses of first touches alone -- no counter is ever held.

The serial rule (oracle/gp_oracle.c, restating btllib's KmerCountingBloomFilter::insert_thresh_contains
as GoldPolish calls it, src/utils.cpp:96-123): an operation reads its hash_num counters, m = their
minimum; if m < thr it lifts the counters equal to m by one (so its own minimum becomes m + 1); it inserts
its k-mer into the Bloom filter iff the minimum it LEAVES is >= thr, i.e. iff it FOUND m >= thr - 1.

The level rule: L(0) = every operation; L(j + 1) = the operations of L(j) none of whose counter touches is
the earliest touch of that counter among the operations of L(j).  Claim: an operation finds m >= j iff it
is in L(j) (for j <= thr - 1, where no counter has stopped moving yet), so it inserts iff it is in
L(thr - 1): thr - 1 censuses, each over the marked records of the one before.
"""
import numpy as np
import pytest


def serial(idx, n_counters, thr):
    c = np.zeros(n_counters, dtype=np.int64)
    ins = np.zeros(len(idx), dtype=bool)
    for t, touches in enumerate(idx):
        m = int(c[touches].min())
        if m < thr:
            for x in set(int(v) for v in touches):   # (a counter hit twice by one operation still moves by one)
                if c[x] == m:
                    c[x] = m + 1
            m += 1
        ins[t] = m >= thr
    return ins


def by_levels(idx, n_counters, thr):
    member = np.ones(len(idx), dtype=bool)
    sizes = []
    for _ in range(thr - 1):
        sizes.append(int(member.sum()))
        first = np.full(n_counters, -1, dtype=np.int64)
        for t in np.nonzero(member)[0]:            # in order: the earliest touch of every counter among the members
            for x in idx[t]:
                if first[x] < 0:
                    first[x] = t
        nxt = np.zeros(len(idx), dtype=bool)
        for t in np.nonzero(member)[0]:
            nxt[t] = all(first[x] != t for x in idx[t])
        member = nxt
        if not member.any():
            break
    else:
        sizes.append(int(member.sum()))
    return member if len(sizes) == thr else np.zeros(len(idx), dtype=bool), sizes


@pytest.mark.parametrize("load,thr,seed", [(0.3, 3, 1), (3, 4, 2), (15, 13, 3), (30, 13, 4), (30, 6, 5), (8, 2, 6)])
def test_levels_decide_what_the_serial_filter_inserts(load, thr, seed):
    """random k-mers with copies (a read set: every k-mer of a 'genome' some 0..2 thr times, shuffled)"""
    rng = np.random.default_rng(seed)
    n_counters, h = 512, 3
    n_ops = int(load * n_counters / h)
    n_kmers = max(4, n_ops // max(2, thr // 2))
    table = rng.integers(0, n_counters, size=(n_kmers, h))
    table[::17, 1] = table[::17, 0]               # some k-mers hit one counter twice
    ops = rng.integers(0, n_kmers, size=n_ops)
    idx = table[ops]
    want = serial(idx, n_counters, thr)
    got, sizes = by_levels(idx, n_counters, thr)
    assert np.array_equal(got, want)
    assert all(a >= b for a, b in zip(sizes, sizes[1:]))   # the sets only shrink
    if load >= 15 and thr == 13:
        # what the passes cost: the sum of their sizes in operations (DESIGN section 4 quotes these)
        print(f"load {load} touches per counter, thr {thr}: passes hold {sum(sizes) / n_ops:.1f} x the operations")


def by_levels_on_records(idx, n_counters, thr):
    """the device's way (census_mark_seg from the second pass on): NOTHING is scattered again -- a pass's input
    is the records the pass before marked, filtered by one bit per operation (all of its touches marked)"""
    n = len(idx)
    records = [(t, int(x)) for t in range(n) for x in idx[t]]          # level 0: every touch
    member = np.ones(n, dtype=bool)
    for _ in range(thr - 1):
        first = {}
        for t, x in records:                                            # (records are in order of t)
            first.setdefault(x, t)
        marked = [(t, x) for t, x in records if first[x] != t]          # a touch that is not its counter's first
        n_marked = np.zeros(n, dtype=np.int64)
        for t, _x in marked:
            n_marked[t] += 1
        member = member & (n_marked == idx.shape[1])                    # census_pack: every touch marked
        records = [(t, x) for t, x in marked if member[t]]              # take(): the next pass's input
        if not records:
            member[:] = False
            break
    return member


@pytest.mark.parametrize("load,thr,seed", [(3, 4, 12), (15, 13, 13), (30, 6, 14), (8, 2, 15)])
def test_passes_over_the_marked_records_equal_passes_over_the_members(load, thr, seed):
    rng = np.random.default_rng(seed)
    n_counters, h = 512, 3
    n_ops = int(load * n_counters / h)
    n_kmers = max(4, n_ops // max(2, thr // 2))
    table = rng.integers(0, n_counters, size=(n_kmers, h))
    table[::13, 2] = table[::13, 0]               # k-mers that hit one counter twice
    idx = table[rng.integers(0, n_kmers, size=n_ops)]
    assert np.array_equal(by_levels_on_records(idx, n_counters, thr), serial(idx, n_counters, thr))
