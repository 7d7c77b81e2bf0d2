"""ctypes binding of the CPU oracle (oracle/libgpo.so).  TEST INFRASTRUCTURE:
only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this."""
import ctypes as C

import numpy as np

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)


class NteditParams(C.Structure):
    _fields_ = [
        ("k", C.c_uint), ("h", C.c_uint), ("max_insertions", C.c_uint),
        ("max_deletions", C.c_uint), ("mode", C.c_int), ("use_ratio", C.c_int),
        ("missing_ratio", C.c_float), ("edit_ratio", C.c_float),
        ("missing_threshold", C.c_float), ("edit_threshold", C.c_float),
        ("jump", C.c_uint), ("min_contig_len", C.c_uint), ("mask", C.c_int),
        ("insertion_cap", C.c_uint),
    ]


class SealerParams(C.Structure):
    _fields_ = [
        ("flank_len", C.c_uint), ("max_gap_len", C.c_uint), ("max_paths", C.c_uint),
        ("max_branches", C.c_uint), ("max_cost", C.c_uint), ("mem_limit", C.c_uint64),
        ("lower", C.c_int),
    ]


class SealerStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "gaps_found", "gaps_closed", "no_start_goal", "no_path", "unique_path",
        "multiple_paths", "too_many_paths", "too_many_branches", "max_cost_exceeded",
        "vertex_tests")]


class BatchParams(C.Structure):
    _fields_ = [
        ("nk", C.c_uint), ("k_values", C.c_uint * 8), ("hash_num", C.c_uint),
        ("cbf_bytes", C.c_size_t), ("bf_bytes", C.c_size_t), ("ntedit_x", C.c_float),
        ("ntedit_y", C.c_float), ("sealer_P", C.c_uint), ("sealer_B", C.c_uint),
        ("run_sealer", C.c_int),
    ]


class BatchStats(C.Structure):
    _fields_ = [
        ("a_ops", C.c_uint64), ("b_probes", C.c_uint64), ("c_vertex_tests", C.c_uint64),
        ("gaps_found", C.c_uint64), ("gaps_closed", C.c_uint64), ("ntedit_skipped", C.c_int),
    ]


def _b(s):
    return s if isinstance(s, (bytes, bytearray)) else s.encode()


class Oracle:
    def __init__(self, path):
        L = self.L = C.CDLL(path)
        L.gpo_ntf64.restype = C.c_uint64
        L.gpo_ntf64.argtypes = [C.c_char_p, C.c_uint]
        L.gpo_ntr64.restype = C.c_uint64
        L.gpo_ntr64.argtypes = [C.c_char_p, C.c_uint]
        for f in (L.gpo_ntf64_roll, L.gpo_ntr64_roll, L.gpo_ntf64_changelast,
                  L.gpo_ntr64_changelast):
            f.restype = C.c_uint64
            f.argtypes = [C.c_uint64, C.c_uint, C.c_ubyte, C.c_ubyte]
        L.gpo_extend.argtypes = [C.c_uint64, C.c_uint, C.c_uint, u64p]
        L.gpo_nthash_all.restype = C.c_size_t
        L.gpo_nthash_all.argtypes = [C.c_char_p, C.c_size_t, C.c_uint, u64p, u8p]
        L.gpo_kmer_threshold.restype = C.c_int
        L.gpo_kmer_threshold.argtypes = [C.c_uint64]
        L.gpo_rank_reads.restype = C.c_size_t
        L.gpo_rank_reads.argtypes = [C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                     u64p, C.c_uint64, C.c_double, u32p, u64p]
        L.gpo_fill_bfs.restype = C.c_int
        L.gpo_fill_bfs.argtypes = [C.c_char_p, C.c_size_t, C.c_uint, C.c_uint,
                                   C.POINTER(C.c_uint), C.c_uint, C.POINTER(u8p), C.c_size_t,
                                   C.POINTER(u8p), C.c_size_t]
        L.gpo_ntedit_default_params.argtypes = [C.POINTER(NteditParams), C.c_uint, C.c_uint]
        L.gpo_ntedit_out_bound.restype = C.c_size_t
        L.gpo_ntedit_out_bound.argtypes = [C.c_size_t]
        L.gpo_ntedit_contig.restype = C.c_size_t
        L.gpo_ntedit_contig.argtypes = [C.c_char_p, C.c_size_t, u8p, C.c_size_t,
                                        C.POINTER(NteditParams), C.c_char_p, u64p]
        L.gpo_ntedit_contig_dyn.restype = C.c_size_t
        L.gpo_ntedit_contig_dyn.argtypes = [C.c_char_p, C.c_size_t, u8p, C.c_size_t,
                                            C.POINTER(NteditParams), C.POINTER(C.c_void_p), u64p]
        L.gpo_mask.restype = C.c_size_t
        L.gpo_mask.argtypes = [C.c_char_p, C.c_size_t, C.c_uint, C.c_char_p]
        L.gpo_sealer_default_params.argtypes = [C.POINTER(SealerParams), C.c_uint]
        L.gpo_sealer_out_bound.restype = C.c_size_t
        L.gpo_sealer_out_bound.argtypes = [C.c_size_t]
        L.gpo_sealer_record.restype = C.c_size_t
        L.gpo_sealer_record.argtypes = [C.c_char_p, C.c_size_t, C.c_uint, C.POINTER(C.c_uint),
                                        C.POINTER(u8p), C.c_size_t, C.c_uint,
                                        C.POINTER(SealerParams), C.c_char_p,
                                        C.POINTER(SealerStats)]
        L.gpo_batch_default_params.argtypes = [C.POINTER(BatchParams)]
        L.gpo_polish_batch.restype = C.c_int
        L.gpo_polish_batch.argtypes = [
            C.POINTER(BatchParams), C.c_uint, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), u64p,
            u32p, C.POINTER(C.c_char_p), u64p, u32p, C.POINTER(C.c_void_p), u64p,
            C.POINTER(u8p), C.POINTER(BatchStats)]
        L.gpo_free.argtypes = [C.c_void_p]
        L.gpo_nw_consensus.restype = C.c_size_t
        L.gpo_nw_consensus.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p,
                                       C.POINTER(C.c_uint)]
        L.gpo_align_multi.restype = C.c_size_t
        L.gpo_align_multi.argtypes = [C.c_uint, C.POINTER(C.c_char_p), C.POINTER(C.c_size_t),
                                      C.c_char_p, C.POINTER(C.c_uint)]
        L.gpo_revcomp.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.gpo_amb_or.restype = C.c_char
        L.gpo_amb_or.argtypes = [C.c_char, C.c_char]
        L.gpo_polish_batches.restype = C.c_int
        L.gpo_polish_batches.argtypes = [
            C.POINTER(BatchParams), C.c_uint, u32p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p),
            u64p, u32p, C.POINTER(C.c_char_p), u64p, u32p, C.POINTER(C.c_void_p), u64p, C.c_int,
            C.POINTER(C.c_int)]

    # ---- ntHash
    def nthash_all(self, seq, k):
        seq = _b(seq)
        n = max(len(seq) - k + 1, 0)
        h = np.zeros(max(n, 1), dtype=np.uint64)
        v = np.zeros(max(n, 1), dtype=np.uint8)
        got = self.L.gpo_nthash_all(seq, len(seq), k, h.ctypes.data_as(u64p),
                                    v.ctypes.data_as(u8p))
        return h[:got], v[:got].astype(bool)

    def extend(self, base, k, h=4):
        out = (C.c_uint64 * h)()
        self.L.gpo_extend(int(base), k, h, out)
        return [int(x) for x in out]

    # ---- stage A
    def kmer_threshold(self, bases):
        return self.L.gpo_kmer_threshold(int(bases))

    def rank_reads(self, ids, phred, lens, target_len, s):
        n = len(ids)
        cids = (C.c_char_p * n)(*[_b(i) for i in ids])
        ph = (C.c_double * n)(*phred)
        ln = (C.c_uint64 * n)(*lens)
        order = (C.c_uint32 * max(n, 1))()
        bases = C.c_uint64(0)
        m = self.L.gpo_rank_reads(n, cids, ph, ln, int(target_len), float(s), order,
                                  C.byref(bases))
        return list(order[:m]), bases.value

    def fill_bfs(self, reads, k_values, kmer_threshold, cbf_bytes, bf_bytes, hash_num=4,
                 cbfs=None, bfs=None):
        nk = len(k_values)
        if cbfs is None:
            cbfs = [np.zeros(cbf_bytes, dtype=np.uint8) for _ in range(nk)]
            bfs = [np.zeros(bf_bytes, dtype=np.uint8) for _ in range(nk)]
        pc = (u8p * nk)(*[a.ctypes.data_as(u8p) for a in cbfs])
        pb = (u8p * nk)(*[a.ctypes.data_as(u8p) for a in bfs])
        ks = (C.c_uint * nk)(*k_values)
        for r in reads:
            r = _b(r)
            rc = self.L.gpo_fill_bfs(r, len(r), hash_num, nk, ks, kmer_threshold, pc, cbf_bytes,
                                     pb, bf_bytes)
            if rc != 0:
                raise ValueError("gpo_fill_bfs: kmer_threshold must be >= 4")
        return cbfs, bfs

    # ---- stage B
    def ntedit_params(self, k, h=4, **kw):
        p = NteditParams()
        self.L.gpo_ntedit_default_params(C.byref(p), k, h)
        for a, b in kw.items():
            setattr(p, a, b)
        return p

    def ntedit(self, seq, bf, k, h=4, params=None):
        seq = _b(seq)
        p = params or self.ntedit_params(k, h)
        out = C.c_void_p()
        probes = C.c_uint64(0)
        n = self.L.gpo_ntedit_contig_dyn(seq, len(seq), bf.ctypes.data_as(u8p), bf.size, C.byref(p),
                                         C.byref(out), C.byref(probes))
        res = C.string_at(out, n)
        self.L.gpo_free(out)
        return res, probes.value

    # ---- mask / sealer
    def mask(self, seq, k):
        seq = _b(seq)
        out = C.create_string_buffer(max(len(seq), 1) + 2)
        n = self.L.gpo_mask(seq, len(seq), k, out)
        return out.raw[:n]

    def sealer(self, seq, k_values, bfs, hash_num=4, first_k=None, **kw):
        seq = _b(seq)
        nk = len(k_values)
        p = SealerParams()
        self.L.gpo_sealer_default_params(C.byref(p), first_k or k_values[0])
        for a, b in kw.items():
            setattr(p, a, b)
        st = SealerStats()
        out = C.create_string_buffer(self.L.gpo_sealer_out_bound(len(seq)) + 1)
        pb = (u8p * nk)(*[a.ctypes.data_as(u8p) for a in bfs])
        ks = (C.c_uint * nk)(*k_values)
        n = self.L.gpo_sealer_record(seq, len(seq), nk, ks, pb, bfs[0].size, hash_num,
                                     C.byref(p), out, C.byref(st))
        return out.raw[:n], st

    # ---- Sealer's consensus step alone
    def nw_consensus(self, a, b):
        a, b = _b(a), _b(b)
        out = C.create_string_buffer(len(a) + len(b) + 2)
        m = C.c_uint(0)
        n = self.L.gpo_nw_consensus(a, len(a), b, len(b), out, C.byref(m))
        return out.raw[:n], m.value

    def align_multi(self, seqs):
        seqs = [_b(s) for s in seqs]
        n = len(seqs)
        cs = (C.c_char_p * n)(*seqs)
        ln = (C.c_size_t * n)(*[len(s) for s in seqs])
        out = C.create_string_buffer(sum(len(s) for s in seqs) + 2)
        m = C.c_uint(0)
        k = self.L.gpo_align_multi(n, cs, ln, out, C.byref(m))
        return out.raw[:k], m.value

    def revcomp(self, s):
        s = _b(s)
        out = C.create_string_buffer(len(s) + 1)
        self.L.gpo_revcomp(s, len(s), out)
        return out.raw[:len(s)]

    def amb_or(self, a, b):
        return self.L.gpo_amb_or(_b(a), _b(b))

    # ---- whole batch
    def batch_params(self, k_values=(32, 28, 24, 20), **kw):
        p = BatchParams()
        self.L.gpo_batch_default_params(C.byref(p))
        p.nk = len(k_values)
        for i, k in enumerate(k_values):
            p.k_values[i] = k
        for a, b in kw.items():
            setattr(p, a, b)
        return p

    def polish_batch(self, names, seqs, reads_per_contig, thresholds, params=None,
                     want_bfs=False):
        """reads_per_contig: list (per contig) of lists of byte strings, already ranked."""
        p = params or self.batch_params()
        n = len(seqs)
        seqs = [_b(s) for s in seqs]
        flat = [_b(r) for rs in reads_per_contig for r in rs]
        begin = np.zeros(n + 1, dtype=np.uint32)
        for i, rs in enumerate(reads_per_contig):
            begin[i + 1] = begin[i] + len(rs)
        cn = (C.c_char_p * n)(*[_b(x) for x in names])
        cs = (C.c_char_p * n)(*seqs)
        cl = (C.c_uint64 * n)(*[len(s) for s in seqs])
        nr = len(flat)
        rs_ = (C.c_char_p * max(nr, 1))(*flat)
        rl = (C.c_uint64 * max(nr, 1))(*[len(r) for r in flat])
        th = (C.c_uint32 * n)(*thresholds)
        oseq = (C.c_void_p * n)()
        olen = (C.c_uint64 * n)()
        st = BatchStats()
        bfs = [np.zeros(p.bf_bytes, dtype=np.uint8) for _ in range(p.nk)] if want_bfs else None
        pb = (u8p * p.nk)(*[a.ctypes.data_as(u8p) for a in bfs]) if want_bfs else None
        rc = self.L.gpo_polish_batch(C.byref(p), n, cn, cs, cl, begin.ctypes.data_as(u32p), rs_,
                                     rl, th, oseq, olen, pb, C.byref(st))
        if rc != 0:
            raise RuntimeError(f"gpo_polish_batch failed: {rc}")
        outs = []
        for i in range(n):
            if oseq[i]:
                outs.append(C.string_at(oseq[i], olen[i]))
                self.L.gpo_free(oseq[i])
            else:
                outs.append(None)
        return outs, st, bfs

    def polish_batches(self, names, seqs, reads_per_contig, thresholds, batch_contig_begin,
                       params=None, n_threads=0):
        """Many batches in parallel (one OpenMP task per batch, each batch serial), as the reference
        runs them on a multi-core host.  Returns (sequences or None per contig, threads used)."""
        p = params or self.batch_params()
        n = len(seqs)
        seqs = [_b(s) for s in seqs]
        flat = [_b(r) for rs in reads_per_contig for r in rs]
        begin = np.zeros(n + 1, dtype=np.uint32)
        for i, rs in enumerate(reads_per_contig):
            begin[i + 1] = begin[i] + len(rs)
        bcb = np.ascontiguousarray(batch_contig_begin, dtype=np.uint32)
        cn = (C.c_char_p * max(n, 1))(*[_b(x) for x in names])
        cs = (C.c_char_p * max(n, 1))(*seqs)
        cl = (C.c_uint64 * max(n, 1))(*[len(s) for s in seqs])
        nr = len(flat)
        rs_ = (C.c_char_p * max(nr, 1))(*flat)
        rl = (C.c_uint64 * max(nr, 1))(*[len(r) for r in flat])
        th = (C.c_uint32 * max(n, 1))(*thresholds)
        oseq = (C.c_void_p * max(n, 1))()
        olen = (C.c_uint64 * max(n, 1))()
        used = C.c_int(0)
        rc = self.L.gpo_polish_batches(C.byref(p), len(bcb) - 1, bcb.ctypes.data_as(u32p), cn, cs, cl,
                                       begin.ctypes.data_as(u32p), rs_, rl, th, oseq, olen,
                                       int(n_threads), C.byref(used))
        if rc != 0:
            raise RuntimeError(f"gpo_polish_batches failed: {rc}")
        outs = []
        for i in range(n):
            if oseq[i]:
                outs.append(C.string_at(oseq[i], olen[i]))
                self.L.gpo_free(oseq[i])
            else:
                outs.append(None)
        return outs, used.value
