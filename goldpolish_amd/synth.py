"""Seeded synthetic assemblies + long reads for tests and bench.py.

Data only: a truth genome, a draft with errors, reads sampled from the truth with
ONT-like errors, and the "mapping" (which reads belong to which contig) taken
from the reads' true origin, so no external mapper is needed.  Everything is
numpy; no reference code involved.
"""
from __future__ import annotations

import dataclasses

import numpy as np

_ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTacgtNn", b"TGCAtgcaNn"):
    _COMP[a] = b


def random_genome(n: int, rng: np.random.Generator, gc: float = 0.41) -> np.ndarray:
    p = np.array([(1 - gc) / 2, gc / 2, gc / 2, (1 - gc) / 2])
    return _ACGT[rng.choice(4, size=n, p=p)]


def mutate(seq: np.ndarray, rate: float, rng: np.random.Generator,
           ratio=(2, 1, 2)) -> np.ndarray:
    """Apply substitutions / insertions / deletions at `rate` per base (sub:ins:del)."""
    n = seq.size
    if n == 0 or rate <= 0:
        return seq.copy()
    r = np.asarray(ratio, dtype=float)
    r = r / r.sum()
    u = rng.random(n)
    kind = np.zeros(n, dtype=np.int8)  # 0 keep, 1 sub, 2 ins, 3 del
    kind[u < rate * r[0]] = 1
    kind[(u >= rate * r[0]) & (u < rate * (r[0] + r[1]))] = 2
    kind[(u >= rate * (r[0] + r[1])) & (u < rate)] = 3
    out = seq.copy()
    sub = np.nonzero(kind == 1)[0]
    if sub.size:
        # a different base: rotate by 1..3 within ACGT
        code = np.searchsorted(_ACGT, out[sub])  # A0 C1 G2 T3 (sorted order ACGT)
        out[sub] = _ACGT[(code + rng.integers(1, 4, size=sub.size)) % 4]
    reps = np.ones(n, dtype=np.int64)
    reps[kind == 3] = 0
    reps[kind == 2] = 2
    res = np.repeat(out, reps)
    ins = np.nonzero(kind == 2)[0]
    if ins.size:
        # position of the duplicated copy in the output: cumulative offset
        pos = np.cumsum(reps)[ins] - 1
        res[pos] = _ACGT[rng.integers(0, 4, size=ins.size)]
    return res


def revcomp(seq: np.ndarray) -> np.ndarray:
    return _COMP[seq[::-1]]


@dataclasses.dataclass
class Workload:
    """Flat, C-ABI-shaped arrays for a set of single-contig batches (bsize=1)."""
    names: list                 # contig names ("id[ comment]")
    contig_buf: np.ndarray      # uint8, all draft contigs concatenated
    contig_off: np.ndarray      # uint64 [n_contigs+1]
    read_buf: np.ndarray        # uint8, reads concatenated in (contig, rank) order
    read_off: np.ndarray        # uint64 [n_reads+1]
    contig_read_begin: np.ndarray  # uint32 [n_contigs+1] into the read list
    kmer_threshold: np.ndarray  # uint32 [n_contigs]
    truth: list | None = None   # optional truth sequences (for quality stats)

    @property
    def n_contigs(self):
        return len(self.names)

    @property
    def total_bp(self):
        return int(self.contig_off[-1])

    def contig(self, i) -> bytes:
        return self.contig_buf[self.contig_off[i]:self.contig_off[i + 1]].tobytes()

    def reads_of(self, i) -> list:
        b, e = self.contig_read_begin[i], self.contig_read_begin[i + 1]
        return [self.read_buf[self.read_off[r]:self.read_off[r + 1]].tobytes()
                for r in range(b, e)]


def kmer_threshold(bases: int) -> int:
    """min(round(4.66943 + bases*2.11391e-07), 13) -- the constants GoldPolish uses
    (documented behaviour of goldpolish_targeted_bfs; C round = half away from zero)."""
    x = 4.66943 + float(bases) * 2.11391e-07
    return min(int(np.floor(x + 0.5)), 13)


def make_workload(contig_lens, coverage: float, seed: int, draft_err: float = 0.01,
                  read_err: float = 0.06, read_mean: float = 15000.0, read_sigma: float = 0.5,
                  read_min: int = 1000, read_max: int = 100000, s_per_10kbp: float = 40.0,
                  keep_truth: bool = False, truths=None, names_in=None) -> Workload:
    """One batch per contig.  Reads are sampled from the truth of their contig, so the
    mapping is exact; ranking follows the reference's rule (integer phred desc, id asc)
    and the per-contig cap size_t(len*s/10000).  `truths` (uint8 arrays, upper-case ACGT...)
    replaces the random genomes -- e.g. the contigs of a FASTA fixture (contig_lens is then
    ignored); `names_in` their names."""
    rng = np.random.default_rng(seed)
    names, contigs = [], []
    if truths is not None:
        contig_lens = [len(t) for t in truths]
    given, truths = truths, []
    reads, read_begin, thresholds = [], [0], []
    rid = 0
    mu = np.log(read_mean) - read_sigma ** 2 / 2
    for ci, n in enumerate(contig_lens):
        truth = random_genome(int(n), rng) if given is None else np.asarray(given[ci], dtype=np.uint8)
        draft = mutate(truth, draft_err, rng)
        names.append(names_in[ci] if names_in is not None else f"contig{ci:06d}")
        contigs.append(draft)
        if keep_truth:
            truths.append(truth)
        target = coverage * n
        mine, got = [], 0
        while got < target:
            L = int(np.clip(rng.lognormal(mu, read_sigma), read_min, read_max))
            L = min(L, int(n))
            s = int(rng.integers(0, n - L + 1))
            frag = truth[s:s + L]
            if rng.random() < 0.5:
                frag = revcomp(frag)
            rd = mutate(frag, read_err, rng)
            ph = int(rng.integers(7, 21))
            mine.append((ph, f"read{rid:09d}", rd))
            rid += 1
            got += L
        mine.sort(key=lambda t: (-t[0], t[1]))
        cap = int(float(draft.size) * s_per_10kbp / 10000.0)
        mine = mine[:min(len(mine), cap)]
        bases = sum(r[2].size for r in mine)
        thresholds.append(kmer_threshold(bases))
        reads.extend(r[2] for r in mine)
        read_begin.append(len(reads))
    contig_off = np.zeros(len(contigs) + 1, dtype=np.uint64)
    contig_off[1:] = np.cumsum([c.size for c in contigs])
    read_off = np.zeros(len(reads) + 1, dtype=np.uint64)
    if reads:
        read_off[1:] = np.cumsum([r.size for r in reads])
    return Workload(
        names=names,
        contig_buf=np.concatenate(contigs) if contigs else np.zeros(0, np.uint8),
        contig_off=contig_off,
        read_buf=np.concatenate(reads) if reads else np.zeros(0, np.uint8),
        read_off=read_off,
        contig_read_begin=np.asarray(read_begin, dtype=np.uint32),
        kmer_threshold=np.asarray(thresholds, dtype=np.uint32),
        truth=truths if keep_truth else None,
    )


def read_fasta(path):
    """(names, sequences as uint8 arrays) of a FASTA file; names are whole header lines without '>'."""
    names, seqs, cur = [], [], []
    with open(path, "rb") as f:
        for line in f:
            line = line.rstrip(b"\r\n")
            if line.startswith(b">"):
                if names:
                    seqs.append(np.frombuffer(b"".join(cur), dtype=np.uint8))
                names.append(line[1:].decode())
                cur = []
            elif line:
                cur.append(line)
    if names:
        seqs.append(np.frombuffer(b"".join(cur), dtype=np.uint8))
    return names, seqs


def _chunk_job(args):
    lens, coverage, seed, kw = args
    return make_workload(lens, coverage, seed, **kw)


def concat_workloads(parts) -> Workload:
    """Several workloads as one (contigs renumbered in order); every contig stays its own batch."""
    names, n_reads = [], 0
    cbuf, rbuf, coff, roff, crb, thr = [], [], [np.zeros(1, np.uint64)], [np.zeros(1, np.uint64)], \
        [np.zeros(1, np.uint32)], []
    c_at = r_at = 0
    for w in parts:
        for _ in w.names:
            names.append(f"contig{len(names):06d}")
        cbuf.append(w.contig_buf)
        rbuf.append(w.read_buf)
        coff.append(w.contig_off[1:] + np.uint64(c_at))
        roff.append(w.read_off[1:] + np.uint64(r_at))
        crb.append(w.contig_read_begin[1:] + np.uint32(n_reads))
        thr.append(w.kmer_threshold)
        c_at += int(w.contig_off[-1])
        r_at += int(w.read_off[-1])
        n_reads += len(w.read_off) - 1
    return Workload(names=names, contig_buf=np.concatenate(cbuf) if cbuf else np.zeros(0, np.uint8),
                    contig_off=np.concatenate(coff), read_buf=np.concatenate(rbuf) if rbuf else np.zeros(0, np.uint8),
                    read_off=np.concatenate(roff), contig_read_begin=np.concatenate(crb),
                    kmer_threshold=np.concatenate(thr) if thr else np.zeros(0, np.uint32))


def make_workload_chunked(contig_lens, coverage: float, seed: int, chunk: int = 100, workers: int = 0,
                          seeds=None, **kw) -> Workload:
    """make_workload for large assemblies: the contigs are cut into runs of `chunk`, each run is
    generated by make_workload with its own seed (seed * 1000003 + run, or seeds[run]), in `workers`
    processes (0 = one per core the process may use; started by fork, so call this BEFORE anything
    touches the GPU).  The result depends on (contig_lens, coverage, seed(s), chunk) only, never on
    the number of workers."""
    import multiprocessing as mp
    import os
    lens = [int(x) for x in contig_lens]
    jobs = [(lens[i:i + chunk], coverage, seeds[i // chunk] if seeds is not None else seed * 1000003 + i // chunk, kw)
            for i in range(0, len(lens), chunk)]
    if workers <= 0:
        workers = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    workers = max(1, min(workers, len(jobs), 32))   # a GPU box shows every core of its node
    if workers == 1:
        parts = [_chunk_job(j) for j in jobs]
    else:
        with mp.get_context("fork").Pool(workers) as pool:
            parts = pool.map(_chunk_job, jobs, chunksize=max(1, min(64, len(jobs) // (4 * workers))))
    return concat_workloads(parts)


def save_workload(w: Workload, path):
    """a workload as one .npz file (bench.py --cache: generate once, then load without forking --
    what a run under rocprofv3 needs, whose preloaded library has initialised the GPU already)"""
    np.savez(path, names=np.array("\n".join(w.names)), contig_buf=w.contig_buf, contig_off=w.contig_off,
             read_buf=w.read_buf, read_off=w.read_off, contig_read_begin=w.contig_read_begin,
             kmer_threshold=w.kmer_threshold)


def load_workload(path) -> Workload:
    z = np.load(path)
    return Workload(names=str(z["names"]).split("\n"), contig_buf=z["contig_buf"], contig_off=z["contig_off"],
                    read_buf=z["read_buf"], read_off=z["read_off"], contig_read_begin=z["contig_read_begin"],
                    kmer_threshold=z["kmer_threshold"])
