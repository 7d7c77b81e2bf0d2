"""The rule stage A's parallel replay rests on, as a CPU model (no GPU, no library).

Serial reference (src/utils.cpp:96-123 over btllib's insert_thresh_contains): an operation
takes c = min of its counters; if c < thr the counters equal to c become c + 1 and c becomes
c + 1; the k-mer goes into the Bloom filter iff c >= thr.  Operations of different contigs of a
batch use different thresholds on the same counters (goldpolish_targeted_bfs.cpp:126-131).

Device shortcut ("settled", goldpolish_amd/csrc/stage_a.hip, top of the file): an operation
whose counters are all >= S - 1 inserts and writes nothing.  The model shows that S has to be
the LARGEST threshold used on the filter: with S = the operation's own threshold the set of
inserted operations differs from the serial one as soon as thresholds are mixed and counters
collide; with S = the largest it never does."""
import numpy as np


def replay(ops, thrs, n_counters, settle):
    """ops[i] = counter indices, thrs[i] = threshold; settle(i) -> S or None (no shortcut).
    Returns the tuple of insert decisions."""
    cnt = np.zeros(n_counters, dtype=np.int64)
    out = []
    for i, (idx, thr) in enumerate(zip(ops, thrs)):
        c = cnt[idx].min()
        s = settle(i)
        if s is not None and c + 1 >= s:
            out.append(True)          # settled: no counter written
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


def test_settling_on_the_largest_threshold_equals_the_serial_replay():
    rng = np.random.default_rng(1)
    for _ in range(400):
        ops, thrs, n = random_case(rng, int(rng.integers(1, 4)))
        serial = replay(ops, thrs, n, lambda i: None)
        assert replay(ops, thrs, n, lambda i: int(thrs.max())) == serial


def test_settling_on_the_own_threshold_is_only_right_for_one_threshold():
    rng = np.random.default_rng(2)
    diverged = 0
    for _ in range(300):
        ops, thrs, n = random_case(rng, 1)
        assert replay(ops, thrs, n, lambda i: int(thrs[i])) == replay(ops, thrs, n, lambda i: None)
    for _ in range(300):
        ops, thrs, n = random_case(rng, 2)
        diverged += replay(ops, thrs, n, lambda i: int(thrs[i])) != replay(ops, thrs, n, lambda i: None)
    assert diverged > 30   # why the shortcut must not use the operation's own threshold
