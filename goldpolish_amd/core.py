"""ctypes binding of libgpcore.so -- the C ABI in include/gpcore.h.

This is plumbing for tests/ and bench.py: it hands numpy buffers to the library
and gets bytes back.  There is no CPU fallback anywhere: if the shared library
is missing, or no GPU is present when a context is created, it raises.
"""
from __future__ import annotations

import ctypes as C
import os
import pathlib
import sys

import numpy as np

# (the library asks the HIP runtime for eight hardware queues instead of four when it is loaded -- api.hip,
# gp_runtime_defaults -- which only counts if the runtime has not been initialised yet; a host that brings
# the runtime up first -- torch does, on its first CUDA call -- gets the same by importing this module first)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_HERE = pathlib.Path(__file__).resolve().parent
LIB_PATH = _HERE / "libgpcore.so"

GP_MAX_K = 8
u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)

# every symbol include/gpcore.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "gp_last_error", "gp_version", "gp_device_bytes_in_use", "gp_default_params", "gp_ctx_create", "gp_ctx_destroy",
    "gp_ctx_stream", "gp_kmer_threshold", "gp_rank_reads", "gp_partition_batches", "gp_job_create", "gp_job_destroy",
    "gp_job_polish", "gp_job_build_filters", "gp_job_ntedit_pass", "gp_job_ntedit_guard",
    "gp_job_mask", "gp_job_sealer", "gp_job_upper", "gp_job_sync", "gp_job_get_filter",
    "gp_job_set_filters", "gp_job_get_lengths", "gp_job_get_sequences",
    "gp_job_get_sequences_device", "gp_job_get_counters", "gp_job_kernel_time",
    "gp_seqindex_build", "gp_seqindex_load", "gp_seqindex_save", "gp_seqindex_count",
    "gp_seqindex_entry", "gp_seqindex_find", "gp_seqindex_read", "gp_seqindex_destroy",
    "gp_mappings_load", "gp_mappings_get", "gp_mappings_destroy", "gp_job_create_from_files",
    "gp_target_extract", "gp_target_update_paf", "gp_target_reinsert",
    "gp_jobs_polish", "gp_filter_save", "gp_filter_save_btllib", "gp_filter_load", "gp_job_write_fasta",
    "gp_sealer_consensus", "gp_device_pool_bytes", "gp_device_pool_trim",
    "gp_pipeline_create", "gp_pipeline_submit", "gp_pipeline_collect", "gp_pipeline_kernel_time",
    "gp_pipeline_destroy", "gp_job_reset", "gp_debug_fail_alloc", "gp_debug_nthash", "gp_ctx_create_masked",
]


class GpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libgpcore error {code}: {msg}")
        self.code = code


class Params(C.Structure):
    _fields_ = [
        ("n_k", C.c_uint32), ("k", C.c_uint32 * GP_MAX_K), ("hash_num", C.c_uint32),
        ("cbf_bytes", C.c_uint64), ("bf_bytes", C.c_uint64),
        ("max_insertions", C.c_uint32), ("max_deletions", C.c_uint32), ("mode", C.c_int32),
        ("missing_ratio", C.c_float), ("edit_ratio", C.c_float), ("jump", C.c_uint32),
        ("min_contig_len", C.c_uint32), ("soft_mask", C.c_int32),
        ("sealer_max_gap", C.c_uint32), ("sealer_max_paths", C.c_uint32),
        ("sealer_max_branches", C.c_uint32), ("sealer_max_cost", C.c_uint32),
        ("run_sealer", C.c_int32),
    ]


class BatchSet(C.Structure):
    _fields_ = [
        ("n_batches", C.c_uint32), ("batch_contig_begin", u32p),
        ("n_contigs", C.c_uint32), ("contig_buf", u8p), ("contig_off", u64p),
        ("contig_name_len", u32p), ("contig_read_begin", u32p),
        ("contig_kmer_threshold", u32p),
        ("n_reads", C.c_uint32), ("read_buf", u8p), ("read_off", u64p),
    ]


class Counters(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in (
        "a_positions", "a_ops", "a_ordered_ops", "a_rounds", "b_scan_kmers", "b_sites",
        "b_probes", "b_edits", "c_gaps", "c_closed", "c_vertex_tests", "a_reservations",
        "c_no_start_goal", "c_no_path", "c_unique_path", "c_multiple_paths", "c_too_many_paths",
        "c_too_many_branches", "c_max_cost", "a_replayed", "a_settled_hits", "a_counted")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Polished(C.Structure):
    _fields_ = [
        ("tag", C.c_uint64), ("n_contigs", C.c_uint32), ("lengths", u64p), ("dropped", u8p),
        ("host_bytes", u8p), ("device_bytes", C.c_void_p), ("total_bytes", C.c_uint64),
        ("counters", Counters), ("upload_ms", C.c_double), ("polish_ms", C.c_double),
        ("download_ms", C.c_double), ("device_bytes_in_use", C.c_uint64),
    ]


def _share_hip_runtime_with_torch():
    """PyTorch-ROCm wheels carry their own libamdhip64 (SONAME libamdhip64.so.7, the one
    libgpcore asks for).  Two HIP/HSA runtimes in one process do not both see the GPU, so when
    a torch wheel is installed its copy is mapped first and libgpcore binds to it whichever
    of the two is imported first; the torch tensors of the multi-GPU gather and libgpcore's
    kernels then share one runtime and one address space.  GP_HIP_RUNTIME=system skips this."""
    if os.environ.get("GP_HIP_RUNTIME") == "system" or "torch" in sys.modules:
        return
    import importlib.util
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.origin:
        return
    cand = pathlib.Path(spec.origin).parent / "lib" / "libamdhip64.so"
    if cand.exists():
        try:
            C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load_library(path=None):
    path = pathlib.Path(path) if path else LIB_PATH
    if not path.exists():
        raise FileNotFoundError(
            f"{path} is missing: build it with `make -C goldpolish_amd/csrc` "
            "(or __graft_entry__.build()); libgpcore has no CPU fallback")
    _share_hip_runtime_with_torch()
    L = C.CDLL(str(path))
    L.gp_last_error.restype = C.c_char_p
    L.gp_version.restype = C.c_int
    L.gp_device_bytes_in_use.restype = C.c_uint64
    L.gp_default_params.argtypes = [C.POINTER(Params)]
    L.gp_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    L.gp_ctx_create_masked.argtypes = [C.c_int, u32p, C.c_uint32, C.POINTER(C.c_void_p)]
    L.gp_ctx_destroy.argtypes = [C.c_void_p]
    L.gp_ctx_stream.restype = C.c_void_p
    L.gp_ctx_stream.argtypes = [C.c_void_p]
    L.gp_kmer_threshold.restype = C.c_int
    L.gp_kmer_threshold.argtypes = [C.c_uint64]
    L.gp_rank_reads.restype = C.c_size_t
    L.gp_rank_reads.argtypes = [C.c_size_t, C.POINTER(C.c_char_p), C.POINTER(C.c_double), u64p,
                                C.c_uint64, C.c_double, u32p, u64p]
    L.gp_partition_batches.argtypes = [C.c_uint32, u64p, C.c_uint32, u32p, u64p]
    L.gp_job_create.argtypes = [C.c_void_p, C.POINTER(Params), C.POINTER(BatchSet),
                                C.POINTER(C.c_void_p)]
    L.gp_job_destroy.argtypes = [C.c_void_p]
    for name in ("gp_job_polish", "gp_job_build_filters", "gp_job_ntedit_guard", "gp_job_mask",
                 "gp_job_sealer", "gp_job_upper", "gp_job_reset"):
        getattr(L, name).argtypes = [C.c_void_p, C.c_int]
    L.gp_job_ntedit_pass.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
    L.gp_jobs_polish.argtypes = [C.POINTER(C.c_void_p), C.c_uint32]
    L.gp_job_sync.argtypes = [C.c_void_p]
    L.gp_job_get_filter.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, u8p]
    L.gp_job_set_filters.argtypes = [C.c_void_p, u8p]
    L.gp_job_get_lengths.argtypes = [C.c_void_p, u64p, u8p, u64p]
    L.gp_job_get_sequences.argtypes = [C.c_void_p, u8p, C.c_uint64]
    L.gp_job_get_sequences_device.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
    L.gp_job_get_counters.argtypes = [C.c_void_p, C.POINTER(Counters)]
    L.gp_job_kernel_time.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), u64p, u64p]
    vpp = C.POINTER(C.c_void_p)
    dp = C.POINTER(C.c_double)
    L.gp_seqindex_build.argtypes = [C.c_char_p, vpp]
    L.gp_seqindex_load.argtypes = [C.c_char_p, C.c_char_p, vpp]
    L.gp_seqindex_save.argtypes = [C.c_void_p, C.c_char_p]
    L.gp_seqindex_count.restype = C.c_uint64
    L.gp_seqindex_count.argtypes = [C.c_void_p]
    L.gp_seqindex_entry.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_char_p), u64p, u64p, dp]
    L.gp_seqindex_find.argtypes = [C.c_void_p, C.c_char_p, u64p, u64p, dp]
    L.gp_seqindex_read.argtypes = [C.c_void_p, C.c_char_p, u8p, C.c_uint64, u64p]
    L.gp_seqindex_destroy.argtypes = [C.c_void_p]
    L.gp_mappings_load.argtypes = [C.c_char_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_double, vpp]
    L.gp_mappings_get.argtypes = [C.c_void_p, C.c_char_p, u64p, C.POINTER(C.POINTER(C.c_char_p))]
    L.gp_mappings_destroy.argtypes = [C.c_void_p]
    L.gp_job_create_from_files.argtypes = [C.c_void_p, C.POINTER(Params), C.c_void_p, C.c_void_p,
                                           C.c_void_p, C.c_double, C.c_uint32, u32p,
                                           C.POINTER(C.c_char_p), vpp]
    L.gp_target_extract.argtypes = [C.c_char_p, C.c_char_p, C.c_uint32, C.c_char_p]
    L.gp_target_update_paf.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    L.gp_target_reinsert.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p]
    L.gp_job_write_fasta.argtypes = [C.c_void_p, C.POINTER(C.c_char_p), C.c_char_p, C.c_int]
    L.gp_filter_save.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint64, u8p]
    L.gp_filter_save_btllib.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint64, u8p]
    L.gp_filter_load.argtypes = [C.c_char_p, u32p, u32p, u64p, u8p, C.c_uint64]
    L.gp_sealer_consensus.argtypes = [C.c_void_p, C.c_uint32, u32p, u8p, u64p, u8p, C.c_uint64, u32p]
    L.gp_device_pool_bytes.restype = C.c_uint64
    L.gp_device_pool_trim.restype = None
    L.gp_pipeline_create.argtypes = [C.c_int, C.POINTER(Params), C.c_uint32, vpp]
    L.gp_pipeline_submit.argtypes = [C.c_void_p, C.POINTER(BatchSet), C.c_uint64]
    L.gp_pipeline_collect.argtypes = [C.c_void_p, C.POINTER(Polished)]
    L.gp_pipeline_kernel_time.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_double), u64p, u64p]
    L.gp_pipeline_destroy.argtypes = [C.c_void_p]
    L.gp_pipeline_destroy.restype = None
    L.gp_debug_fail_alloc.argtypes = [C.c_uint64]
    L.gp_debug_fail_alloc.restype = C.c_uint64
    L.gp_debug_nthash.restype = C.c_int64
    L.gp_debug_nthash.argtypes = [C.c_void_p, C.c_int, u8p, C.c_uint64, C.c_uint32, C.c_uint32, u64p, u64p, u64p, u8p]
    return L


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = load_library()
    return _lib


def _check(rc):
    if rc != 0:
        raise GpError(rc, lib().gp_last_error().decode(errors="replace"))


def default_params(k_values=None, **kw) -> Params:
    p = Params()
    lib().gp_default_params(C.byref(p))
    if k_values is not None:
        p.n_k = len(k_values)
        for i, k in enumerate(k_values):
            p.k[i] = k
    for a, b in kw.items():
        setattr(p, a, b)
    return p


def device_bytes_in_use() -> int:
    return int(lib().gp_device_bytes_in_use())


def device_pool_bytes() -> int:
    return int(lib().gp_device_pool_bytes())


def device_pool_trim():
    lib().gp_device_pool_trim()


def debug_fail_alloc(nth: int):
    """the n-th device allocation from now on fails (0: off) -- tests of the out-of-memory paths;
    returns the number of device allocations made so far"""
    return int(lib().gp_debug_fail_alloc(int(nth)))


def debug_nthash(ctx, variant, seq: bytes, k, hash_num=4):
    """The kernels' own ntHash over one sequence (include/gpcore.h: gp_debug_nthash):
    (fwd[n], rev[n], hashes[n, hash_num], valid[n]) as numpy arrays."""
    buf = np.frombuffer(seq, dtype=np.uint8).copy()
    n_max = max(len(seq), 8)
    f = np.zeros(n_max, dtype=np.uint64)
    r = np.zeros(n_max, dtype=np.uint64)
    h = np.zeros(n_max * hash_num, dtype=np.uint64)
    v = np.zeros(n_max, dtype=np.uint8)
    n = lib().gp_debug_nthash(ctx._h, int(variant), buf.ctypes.data_as(u8p), buf.size, int(k), int(hash_num),
                              f.ctypes.data_as(u64p), r.ctypes.data_as(u64p), h.ctypes.data_as(u64p),
                              v.ctypes.data_as(u8p))
    if n < 0:
        _check(int(n))
    return f[:n], r[:n], h[:n * hash_num].reshape(n, hash_num), v[:n]


def kmer_threshold(bases: int) -> int:
    return lib().gp_kmer_threshold(int(bases))


def rank_reads(ids, phred, lens, target_len, s):
    n = len(ids)
    cids = (C.c_char_p * max(n, 1))(*[i if isinstance(i, bytes) else i.encode() for i in ids])
    ph = (C.c_double * max(n, 1))(*phred)
    ln = (C.c_uint64 * max(n, 1))(*lens)
    order = (C.c_uint32 * max(n, 1))()
    bases = C.c_uint64(0)
    m = lib().gp_rank_reads(n, cids, ph, ln, int(target_len), float(s), order, C.byref(bases))
    return list(order[:m]), bases.value


def partition_batches(cost, n_ranks):
    """rank per batch, longest-first onto the least-loaded rank; returns (rank_of, rank_load)"""
    cost = np.ascontiguousarray(cost, dtype=np.uint64)
    rank_of = np.zeros(cost.size, dtype=np.uint32)
    load = np.zeros(n_ranks, dtype=np.uint64)
    _check(lib().gp_partition_batches(cost.size, cost.ctypes.data_as(C.POINTER(C.c_uint64)), n_ranks,
                                      rank_of.ctypes.data_as(C.POINTER(C.c_uint32)),
                                      load.ctypes.data_as(C.POINTER(C.c_uint64))))
    return rank_of, load


class Context:
    def __init__(self, device=0, cu_mask=None):
        """cu_mask: iterable of CU indices the context's streams may use (None: all)"""
        self._h = C.c_void_p()
        if cu_mask is None:
            _check(lib().gp_ctx_create(device, C.byref(self._h)))
        else:
            cus = sorted(set(int(x) for x in cu_mask))
            words = np.zeros(max(cus) // 32 + 1, dtype=np.uint32)
            for cu in cus:
                words[cu // 32] |= np.uint32(1 << (cu % 32))
            _check(lib().gp_ctx_create_masked(device, words.ctypes.data_as(u32p), words.size, C.byref(self._h)))

    @property
    def stream(self):
        return lib().gp_ctx_stream(self._h)

    def close(self):
        if self._h:
            lib().gp_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def sealer_consensus(ctx, sets):
    """align() of Sealer (alignGlobal.h:73-83) for every set of connecting sequences, on the device."""
    flat = [_bytes(q) for st in sets for q in st]
    begin = np.zeros(len(sets) + 1, dtype=np.uint32)
    begin[1:] = np.cumsum([len(st) for st in sets])
    off = np.zeros(len(flat) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(q) for q in flat])
    buf = np.frombuffer(b"".join(flat) or b"\0", dtype=np.uint8).copy()
    stride = max([sum(len(_bytes(q)) for q in st) for st in sets] + [1])
    out = np.zeros(len(sets) * stride, dtype=np.uint8)
    out_len = np.zeros(len(sets), dtype=np.uint32)
    _check(lib().gp_sealer_consensus(ctx._h, len(sets), begin.ctypes.data_as(u32p), buf.ctypes.data_as(u8p),
                                     off.ctypes.data_as(u64p), out.ctypes.data_as(u8p), stride,
                                     out_len.ctypes.data_as(u32p)))
    return [out[i * stride:i * stride + int(out_len[i])].tobytes() for i in range(len(sets))]


def _arr(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def _bytes(x):
    return x if isinstance(x, bytes) else str(x).encode()


def filter_save(path, k, hash_num, data: np.ndarray, btllib_header=False):
    d = _arr(data, np.uint8)
    f = lib().gp_filter_save_btllib if btllib_header else lib().gp_filter_save
    _check(f(_bytes(path), k, hash_num, d.size, d.ctypes.data_as(u8p)))


def filter_load(path):
    """(k, hash_num, bytes as uint8 array)"""
    k, h, n = C.c_uint32(), C.c_uint32(), C.c_uint64()
    _check(lib().gp_filter_load(_bytes(path), C.byref(k), C.byref(h), C.byref(n), None, 0))
    buf = np.zeros(max(n.value, 1), dtype=np.uint8)
    _check(lib().gp_filter_load(_bytes(path), None, None, None, buf.ctypes.data_as(u8p), buf.size))
    return k.value, h.value, buf[:n.value]


def target_extract(fasta_path, out_path, flank=64, bed_path=None):
    _check(lib().gp_target_extract(_bytes(fasta_path), _bytes(bed_path) if bed_path else None,
                                   int(flank), _bytes(out_path)))


def target_update_paf(gaps_fasta_path, paf_path, out_path):
    _check(lib().gp_target_update_paf(_bytes(gaps_fasta_path), _bytes(paf_path), _bytes(out_path)))


def target_reinsert(fasta_path, polished_gaps_path, out_path):
    _check(lib().gp_target_reinsert(_bytes(fasta_path), _bytes(polished_gaps_path), _bytes(out_path)))


class SeqIndex:
    """gp_seqindex: byte-offset index of a 2-line FASTA / 4-line FASTQ file (host side)."""

    def __init__(self, seqs_path, index_path=None):
        self._h = C.c_void_p()
        if index_path is None:
            _check(lib().gp_seqindex_build(_bytes(seqs_path), C.byref(self._h)))
        else:
            _check(lib().gp_seqindex_load(_bytes(index_path), _bytes(seqs_path), C.byref(self._h)))

    def save(self, index_path):
        _check(lib().gp_seqindex_save(self._h, _bytes(index_path)))

    def __len__(self):
        return int(lib().gp_seqindex_count(self._h))

    def entries(self):
        """[(id, start, length, avg_phred)] in file order"""
        out = []
        for i in range(len(self)):
            sid, st, ln, ph = C.c_char_p(), C.c_uint64(), C.c_uint64(), C.c_double()
            _check(lib().gp_seqindex_entry(self._h, i, C.byref(sid), C.byref(st), C.byref(ln),
                                           C.byref(ph)))
            out.append((sid.value, st.value, ln.value, ph.value))
        return out

    def find(self, sid):
        st, ln, ph = C.c_uint64(), C.c_uint64(), C.c_double()
        rc = lib().gp_seqindex_find(self._h, _bytes(sid), C.byref(st), C.byref(ln), C.byref(ph))
        if rc == -8:
            return None
        _check(rc)
        return st.value, ln.value, ph.value

    def read(self, sid) -> bytes:
        n = C.c_uint64()
        _check(lib().gp_seqindex_read(self._h, _bytes(sid), None, 0, C.byref(n)))
        buf = np.zeros(max(n.value, 1), dtype=np.uint8)
        _check(lib().gp_seqindex_read(self._h, _bytes(sid), buf.ctypes.data_as(u8p), buf.size,
                                      C.byref(n)))
        return buf[:n.value].tobytes()

    def close(self):
        if self._h:
            lib().gp_seqindex_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


class Mappings:
    """gp_mappings: target id -> mapped read ids, from a .sam / .paf / ntLink table."""

    def __init__(self, path, targets: SeqIndex, mx_min=1, mx_max=30, mx_max_mapped_per_10kbp=100.0):
        self._h = C.c_void_p()
        _check(lib().gp_mappings_load(_bytes(path), targets._h, mx_min, mx_max,
                                      float(mx_max_mapped_per_10kbp), C.byref(self._h)))

    def get(self, target_id):
        n = C.c_uint64()
        ids = C.POINTER(C.c_char_p)()
        _check(lib().gp_mappings_get(self._h, _bytes(target_id), C.byref(n), C.byref(ids)))
        return [ids[i] for i in range(n.value)]

    def close(self):
        if self._h:
            lib().gp_mappings_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def polish_jobs(jobs):
    """gp_jobs_polish: every job polished, all of them in flight at once (one context per job)"""
    arr = (C.c_void_p * max(len(jobs), 1))(*[j._h for j in jobs])
    _check(lib().gp_jobs_polish(arr, len(jobs)))


def slice_workload(w, c0, c1):
    """keyword arguments of Job() for contigs [c0, c1) of a synth.Workload (one batch per contig)"""
    r0, r1 = int(w.contig_read_begin[c0]), int(w.contig_read_begin[c1])
    return dict(contig_buf=w.contig_buf[int(w.contig_off[c0]):int(w.contig_off[c1])],
                contig_off=w.contig_off[c0:c1 + 1] - w.contig_off[c0],
                contig_read_begin=w.contig_read_begin[c0:c1 + 1] - w.contig_read_begin[c0],
                kmer_threshold=w.kmer_threshold[c0:c1],
                read_buf=w.read_buf[int(w.read_off[r0]):int(w.read_off[r1])],
                read_off=w.read_off[r0:r1 + 1] - w.read_off[r0],
                name_len=[len(n) for n in w.names[c0:c1]])


def make_batch_set(*, contig_buf, contig_off, contig_read_begin, kmer_threshold, read_buf, read_off,
                   batch_contig_begin=None, name_len=None):
    """(BatchSet, arrays it points into -- keep them alive as long as the struct is used)"""
    n_contigs = len(contig_off) - 1
    if batch_contig_begin is None:
        batch_contig_begin = np.arange(n_contigs + 1)
    k = dict(
        bcb=_arr(batch_contig_begin, np.uint32), cbuf=_arr(contig_buf, np.uint8),
        coff=_arr(contig_off, np.uint64), crb=_arr(contig_read_begin, np.uint32),
        thr=_arr(kmer_threshold, np.uint32), rbuf=_arr(read_buf, np.uint8),
        roff=_arr(read_off, np.uint64),
        nlen=_arr(name_len if name_len is not None else np.zeros(n_contigs), np.uint32))
    bs = BatchSet()
    bs.n_batches = len(k["bcb"]) - 1
    bs.batch_contig_begin = k["bcb"].ctypes.data_as(u32p)
    bs.n_contigs = n_contigs
    bs.contig_buf = k["cbuf"].ctypes.data_as(u8p)
    bs.contig_off = k["coff"].ctypes.data_as(u64p)
    bs.contig_name_len = k["nlen"].ctypes.data_as(u32p)
    bs.contig_read_begin = k["crb"].ctypes.data_as(u32p)
    bs.contig_kmer_threshold = k["thr"].ctypes.data_as(u32p)
    bs.n_reads = len(k["roff"]) - 1
    bs.read_buf = k["rbuf"].ctypes.data_as(u8p)
    bs.read_off = k["roff"].ctypes.data_as(u64p)
    return bs, k


def batch_set_of_workload(w):
    return make_batch_set(contig_buf=w.contig_buf, contig_off=w.contig_off,
                          contig_read_begin=w.contig_read_begin, kmer_threshold=w.kmer_threshold,
                          read_buf=w.read_buf, read_off=w.read_off, name_len=[len(n) for n in w.names])


class Pipeline:
    """gp_pipeline_*: batch sets from host memory to polished bytes in host memory, the upload of
    one set beside the polishing of the one before."""

    def __init__(self, device, params: Params, depth=2):
        self._h = C.c_void_p()
        self.params = params
        self._sets = {}
        _check(lib().gp_pipeline_create(int(device), C.byref(params), int(depth), C.byref(self._h)))

    def submit(self, batch_set, keep, tag):
        """batch_set, keep = make_batch_set(...); the arrays stay referenced until the set is collected"""
        self._sets[int(tag)] = (batch_set, keep)
        _check(lib().gp_pipeline_submit(self._h, C.byref(batch_set), int(tag)))

    def collect(self, copy=True):
        """the oldest submitted set: dict with tag, lengths, dropped, bytes (numpy views of the
        pipeline's pinned buffer -- valid until the next collect -- unless copy), counters, times"""
        r = Polished()
        rc = lib().gp_pipeline_collect(self._h, C.byref(r))
        self._sets.pop(int(r.tag), None)
        _check(rc)
        n = int(r.n_contigs)
        lengths = np.ctypeslib.as_array(r.lengths, shape=(n,)) if n else np.zeros(0, np.uint64)
        dropped = np.ctypeslib.as_array(r.dropped, shape=(n,)) if n else np.zeros(0, np.uint8)
        total = int(r.total_bytes)
        data = np.ctypeslib.as_array(r.host_bytes, shape=(total,)) if total else np.zeros(0, np.uint8)
        if copy:
            lengths, dropped, data = lengths.copy(), dropped.copy(), data.copy()
        return {"tag": int(r.tag), "lengths": lengths, "dropped": dropped, "bytes": data,
                "device_ptr": int(r.device_bytes or 0), "total_bytes": total,
                "counters": r.counters.as_dict(), "upload_ms": float(r.upload_ms),
                "polish_ms": float(r.polish_ms), "download_ms": float(r.download_ms),
                "device_bytes_in_use": int(r.device_bytes_in_use)}

    def kernel_time(self, name: str):
        ms, n, units = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
        _check(lib().gp_pipeline_kernel_time(self._h, name.encode(), C.byref(ms), C.byref(n), C.byref(units)))
        return ms.value, int(n.value), int(units.value)

    def close(self):
        if self._h:
            lib().gp_pipeline_destroy(self._h)
            self._h = C.c_void_p()
        self._sets.clear()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def split_polished(res):
    """list of bytes (None for dropped contigs) from a Pipeline.collect() result"""
    out, at = [], 0
    b = res["bytes"].tobytes()
    for n, d in zip(res["lengths"], res["dropped"]):
        out.append(None if d else b[at:at + int(n)])
        at += int(n)
    return out


class Job:
    """A set of batches resident on the device."""

    def __init__(self, ctx: Context, params: Params, *, contig_buf, contig_off, contig_read_begin,
                 kmer_threshold, read_buf, read_off, batch_contig_begin=None, name_len=None):
        self.ctx = ctx
        self.params = params
        n_contigs = len(contig_off) - 1
        if batch_contig_begin is None:
            batch_contig_begin = np.arange(n_contigs + 1)
        self._keep = dict(
            bcb=_arr(batch_contig_begin, np.uint32), cbuf=_arr(contig_buf, np.uint8),
            coff=_arr(contig_off, np.uint64), crb=_arr(contig_read_begin, np.uint32),
            thr=_arr(kmer_threshold, np.uint32), rbuf=_arr(read_buf, np.uint8),
            roff=_arr(read_off, np.uint64),
            nlen=_arr(name_len if name_len is not None else np.zeros(n_contigs), np.uint32))
        k = self._keep
        bs = BatchSet()
        bs.n_batches = len(k["bcb"]) - 1
        bs.batch_contig_begin = k["bcb"].ctypes.data_as(u32p)
        bs.n_contigs = n_contigs
        bs.contig_buf = k["cbuf"].ctypes.data_as(u8p)
        bs.contig_off = k["coff"].ctypes.data_as(u64p)
        bs.contig_name_len = k["nlen"].ctypes.data_as(u32p)
        bs.contig_read_begin = k["crb"].ctypes.data_as(u32p)
        bs.contig_kmer_threshold = k["thr"].ctypes.data_as(u32p)
        bs.n_reads = len(k["roff"]) - 1
        bs.read_buf = k["rbuf"].ctypes.data_as(u8p)
        bs.read_off = k["roff"].ctypes.data_as(u64p)
        self.n_batches, self.n_contigs = bs.n_batches, n_contigs
        self._h = C.c_void_p()
        _check(lib().gp_job_create(ctx._h, C.byref(params), C.byref(bs), C.byref(self._h)))

    @classmethod
    def from_files(cls, ctx, params, targets: SeqIndex, reads: SeqIndex, mappings: Mappings,
                   batches, max_reads_per_10kbp=40.0):
        """batches: list of lists of target ids (what scripts/goldpolish puts into one batch.fa)"""
        self = cls.__new__(cls)
        self.ctx, self.params = ctx, params
        ids = [_bytes(t) for b in batches for t in b]
        begin = np.zeros(len(batches) + 1, dtype=np.uint32)
        begin[1:] = np.cumsum([len(b) for b in batches])
        arr = (C.c_char_p * max(len(ids), 1))(*ids)
        self._keep = dict(begin=begin, ids=arr)
        self.n_batches, self.n_contigs = len(batches), len(ids)
        self._h = C.c_void_p()
        _check(lib().gp_job_create_from_files(ctx._h, C.byref(params), targets._h, reads._h,
                                              mappings._h, float(max_reads_per_10kbp),
                                              len(batches), begin.ctypes.data_as(u32p), arr,
                                              C.byref(self._h)))
        return self

    @classmethod
    def from_workload(cls, ctx, params, w):
        return cls(ctx, params, contig_buf=w.contig_buf, contig_off=w.contig_off,
                   contig_read_begin=w.contig_read_begin, kmer_threshold=w.kmer_threshold,
                   read_buf=w.read_buf, read_off=w.read_off,
                   name_len=[len(n) for n in w.names])

    def close(self):
        if self._h:
            lib().gp_job_destroy(self._h)
            self._h = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def polish(self, sync=True):
        _check(lib().gp_job_polish(self._h, int(sync)))

    def build_filters(self, sync=True):
        _check(lib().gp_job_build_filters(self._h, int(sync)))

    def ntedit_pass(self, ki, sync=True):
        _check(lib().gp_job_ntedit_pass(self._h, ki, int(sync)))

    def reset(self, sync=True):
        _check(lib().gp_job_reset(self._h, int(sync)))

    def ntedit_guard(self, sync=True):
        _check(lib().gp_job_ntedit_guard(self._h, int(sync)))

    def mask(self, sync=True):
        _check(lib().gp_job_mask(self._h, int(sync)))

    def sealer(self, sync=True):
        _check(lib().gp_job_sealer(self._h, int(sync)))

    def upper(self, sync=True):
        _check(lib().gp_job_upper(self._h, int(sync)))

    def sync(self):
        _check(lib().gp_job_sync(self._h))

    def get_filter(self, batch, ki) -> np.ndarray:
        out = np.zeros(self.params.bf_bytes, dtype=np.uint8)
        _check(lib().gp_job_get_filter(self._h, batch, ki, out.ctypes.data_as(u8p)))
        return out

    def set_filters(self, filters: np.ndarray):
        f = _arr(filters, np.uint8).reshape(-1)
        assert f.size == self.n_batches * self.params.n_k * self.params.bf_bytes
        _check(lib().gp_job_set_filters(self._h, f.ctypes.data_as(u8p)))

    def sequences(self):
        """list of bytes (None for contigs ntEdit dropped), in contig order"""
        lens = np.zeros(self.n_contigs, dtype=np.uint64)
        dropped = np.zeros(self.n_contigs, dtype=np.uint8)
        total = C.c_uint64(0)
        _check(lib().gp_job_get_lengths(self._h, lens.ctypes.data_as(u64p),
                                        dropped.ctypes.data_as(u8p), C.byref(total)))
        buf = np.zeros(max(int(total.value), 1), dtype=np.uint8)
        _check(lib().gp_job_get_sequences(self._h, buf.ctypes.data_as(u8p), buf.size))
        out, at = [], 0
        for i in range(self.n_contigs):
            n = int(lens[i])
            out.append(None if dropped[i] else buf[at:at + n].tobytes())
            at += n
        return out

    def lengths(self):
        """(lengths[n_contigs] uint64 with 0 for dropped records, dropped[n_contigs] uint8)"""
        lens = np.zeros(self.n_contigs, dtype=np.uint64)
        dropped = np.zeros(self.n_contigs, dtype=np.uint8)
        total = C.c_uint64(0)
        _check(lib().gp_job_get_lengths(self._h, lens.ctypes.data_as(u64p),
                                        dropped.ctypes.data_as(u8p), C.byref(total)))
        return lens, dropped

    def sequences_into_device(self, device_ptr: int, nbytes: int):
        """Packs the surviving records at a device pointer on this job's GPU (no host hop)."""
        _check(lib().gp_job_get_sequences_device(self._h, C.c_void_p(device_ptr), nbytes))

    def write_fasta(self, path, names=None, append=False):
        arr = None
        if names is not None:
            arr = (C.c_char_p * max(len(names), 1))(*[_bytes(n) for n in names])
        _check(lib().gp_job_write_fasta(self._h, arr, _bytes(path), int(append)))

    def counters(self) -> dict:
        c = Counters()
        _check(lib().gp_job_get_counters(self._h, C.byref(c)))
        return c.as_dict()

    def kernel_time(self, name: str):
        ms, n, u = C.c_double(0), C.c_uint64(0), C.c_uint64(0)
        _check(lib().gp_job_kernel_time(self._h, name.encode(), C.byref(ms), C.byref(n),
                                        C.byref(u)))
        return ms.value, n.value, u.value
