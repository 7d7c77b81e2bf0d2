"""The rule stage A's parallel replay rests on, as a CPU model (no GPU, no library).

Serial reference (src/utils.cpp:96-123 over btllib's insert_thresh_contains): an operation
takes c = min of its counters; if c < thr the counters equal to c become c + 1 and c becomes
c + 1; the k-mer goes into the Bloom filter iff c >= thr.  Operations of different contigs of a
batch use different thresholds on the same counters (goldpolish_targeted_bfs.cpp:126-131).

Device shortcut ("settled", goldpolish_amd/csrc/stage_a.hip, top of the file): an operation that
sees all its counters >= its own thr, or >= S - 1 with S the largest threshold of itself and every
later operation, inserts and writes nothing.  The model replays random operation streams with that
shortcut taken whenever it applies and compares the insert decisions with the serial replay; the
round-1 rule (all counters >= own thr - 1) is shown to fail as soon as thresholds are mixed."""
import numpy as np


def replay(ops, thrs, n_counters, settled):
    """ops[i] = counter indices, thrs[i] = threshold; settled(i, c) -> take the shortcut?
    Returns the tuple of insert decisions."""
    cnt = np.zeros(n_counters, dtype=np.int64)
    out = []
    for i, (idx, thr) in enumerate(zip(ops, thrs)):
        c = cnt[idx].min()
        if settled(i, c):
            out.append(True)          # no counter written
            continue
        if c < thr:
            for j in set(idx.tolist()):
                if cnt[j] == c:
                    cnt[j] += 1
            c += 1
        out.append(bool(c >= thr))
    return tuple(out)


def random_case(rng, n_thr):
    n_counters = int(rng.integers(8, 40))
    n_kmers = int(rng.integers(4, 30))
    kmers = rng.integers(0, n_counters, size=(n_kmers, 4))
    n_ops = int(rng.integers(50, 400))
    which = rng.integers(0, n_kmers, size=n_ops)
    levels = np.sort(rng.choice(np.arange(2, 12), size=n_thr, replace=False))
    # contigs follow each other: a run of operations per threshold, in random threshold order
    order = rng.permutation(n_thr)
    cuts = np.sort(rng.integers(0, n_ops + 1, size=n_thr - 1)) if n_thr > 1 else []
    thrs = np.empty(n_ops, dtype=np.int64)
    for r, (a, b) in enumerate(zip([0, *cuts], [*cuts, n_ops])):
        thrs[a:b] = levels[order[r]]
    return [kmers[w] for w in which], thrs, n_counters


def never(i, c):
    return False


def test_the_device_rule_equals_the_serial_replay():
    rng = np.random.default_rng(1)
    taken = 0
    for _ in range(600):
        ops, thrs, n = random_case(rng, int(rng.integers(1, 5)))
        smax = np.maximum.accumulate(thrs[::-1])[::-1]   # largest threshold from here on

        def rule(i, c):
            nonlocal taken
            hit = c >= thrs[i] or c + 1 >= smax[i]
            taken += hit
            return hit
        assert replay(ops, thrs, n, rule) == replay(ops, thrs, n, never)
    assert taken > 20000   # the shortcut really is exercised


def test_the_batch_maximum_alone_is_right_but_settles_less():
    rng = np.random.default_rng(3)
    for _ in range(300):
        ops, thrs, n = random_case(rng, int(rng.integers(1, 4)))
        assert replay(ops, thrs, n, lambda i, c: c + 1 >= thrs.max()) == replay(ops, thrs, n, never)


def test_settling_on_the_own_threshold_minus_one_is_only_right_for_one_threshold():
    rng = np.random.default_rng(2)
    diverged = 0
    for _ in range(300):
        ops, thrs, n = random_case(rng, 1)
        assert replay(ops, thrs, n, lambda i, c: c + 1 >= thrs[i]) == replay(ops, thrs, n, never)
    for _ in range(300):
        ops, thrs, n = random_case(rng, 2)
        diverged += replay(ops, thrs, n, lambda i, c: c + 1 >= thrs[i]) != replay(ops, thrs, n, never)
    assert diverged > 30   # round 1's rule: wrong as soon as a batch mixes thresholds
