"""Pure-Python restatement of the reference's SeqIndex / AllMappings / serve_batch read selection
(TEST INFRASTRUCTURE: the checker for goldpolish_amd/csrc/feeder.hip, never the product).

Follows /root/reference/src/seqindex.cpp:12-125, src/mappings.cpp:15-345 and
src/goldpolish_targeted_bfs.cpp:45-131 line by line.  The reference itself cannot be built here
(btllib is absent), so this is "parity unpinned": two restatements written from the same reading
agree; btllib::calc_phred_avg(qual, 0, n) is taken to be sum(qual[0:n]) / n - 33 (with n == 0
meaning the whole string) from btllib's published util.hpp."""
import math


def _until(s: bytes, sep: bytes) -> bytes:
    p = s.find(sep)
    return s if p < 0 else s[:p]


def calc_phred_avg(qual: bytes, n: int) -> float:
    if n == 0:
        n = len(qual)
    if n == 0:
        return -33.0
    return sum(qual[:n]) / n - 33.0


def build_index(path):
    """{id: (start, length, phred)} with first-record-wins, plus ids in file order"""
    data = open(path, "rb").read()
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()              # std::getline yields no line after the final newline
    fastq = data[:1] == b"@"
    out, order = {}, []
    byte = 0
    id_end = seq_start = seq_len = 0
    sid = b""
    for i, line in enumerate(lines):
        end = byte + len(line)
        if fastq:
            if i % 4 == 0:
                id_end = end
                sid = _until(_until(line, b" ")[1:], b"\t")      # seqindex.cpp:32-33
            elif i % 4 == 1:
                seq_start, seq_len = id_end + 1, end - id_end - 1
            elif i % 4 == 3:
                if sid not in out:
                    out[sid] = (seq_start, seq_len, calc_phred_avg(line, len(line) - 1 if line else 0))
                    order.append(sid)
        else:
            if i % 2 == 0:
                id_end = end
                sid = _until(line, b" ")[1:]                       # seqindex.cpp:52
            elif sid not in out:
                out[sid] = (id_end + 1, end - id_end - 1, 0.0)
                order.append(sid)
        byte = end + 1
    return out, order


def saved_phred(x: float) -> float:
    """what survives operator<< (six significant digits) and std::stod"""
    return float("%g" % x)


def load_mappings(path, targets, mx_min=1, mx_max=30, max_per_10kbp=100.0):
    """{target: [read ids]} as AllMappings holds them; targets = {id: (start, len, phred)}"""
    ids, mxs, seen = {}, {}, {}

    def load_mapping(mapped, target, mx=0):
        if target not in targets:
            return
        if target not in ids:
            ids[target], mxs[target], seen[target] = [], [], set()
        if mapped not in seen[target]:
            ids[target].append(mapped)
            seen[target].add(mapped)
            mxs[target].append(mx)

    name = str(path)
    if name.endswith(".sam") or name.endswith(".paf"):
        tcol = 3 if name.endswith(".sam") else 6
        mapped = target = b""
        for line in open(path, "rb").read().split(b"\n")[:-1]:
            if line[:1] == b"@":
                continue
            for col, tok in enumerate(line.split(), 1):
                if col == 1:
                    mapped = tok
                elif col == tcol:
                    target = tok
            load_mapping(mapped, target)
        return ids
    toks = open(path, "rb").read().split()
    for i in range(0, len(toks) - len(toks) % 3, 3):
        if int(toks[i + 2]) >= mx_min:
            load_mapping(toks[i], toks[i + 1], int(toks[i + 2]))
    for target, m in ids.items():                                  # filter, mappings.cpp:231-320
        if not m:
            continue
        mx = mxs[target]
        max_mapped = math.ceil(targets[target][1] * max_per_10kbp / 10000.0)
        count = lambda t: sum(1 for v in mx if v >= t)
        lo, hi = mx_min, mx_max
        if len(m) <= max_mapped:
            thr = lo
        elif count(hi) > max_mapped:
            thr = hi
        else:
            while hi - lo > 1:
                mid = (hi + lo) // 2
                if count(mid) > max_mapped:
                    lo = mid
                else:
                    hi = mid
            thr = hi
        ids[target] = [r for r, v in zip(m, mx) if v >= thr]
    return ids


def kmer_threshold(bases: int) -> int:
    return min(int(round(4.66943 + bases * 2.11391e-07)), 13)      # round-half-away == C round here


def select_reads(target_len, mapped_ids, reads, s):
    """serve_batch: ranked, truncated read ids of one target, and its counting threshold"""
    n_max = int(target_len * s / 10000.0)
    n = min(len(mapped_ids), n_max)
    ranked = sorted(mapped_ids, key=lambda r: (-int(reads[r][2]), r))
    chosen = ranked[:n]
    return chosen, kmer_threshold(sum(reads[r][1] for r in chosen))
