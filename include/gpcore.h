/*
 * gpcore.h -- C ABI of libgpcore, the MI355X-native GoldPolish polishing core.
 *
 * Drop-in boundary for the reference's per-batch k-mer correction path
 * (SURVEY.md section 8b): everything `goldpolish-targeted-bfs` (filter build),
 * `ntedit-gr` (substitution / indel sweep), `goldpolish-mask`, `sealer-gr`
 * (gap-fill BFS) and `goldpolish-to-upper` do for a batch, as one in-process
 * call on device-resident data.  Plain pointers and sizes only; no C++ or
 * torch types cross this boundary.  Every function returns 0 (GP_OK) or a
 * negative gp_status and never throws; the message for the calling thread's
 * last failure is gp_last_error().  A gp_ctx may be used by one thread at a
 * time; different contexts are independent.
 *
 * There is no CPU fallback: gp_ctx_create fails with GP_ERR_NO_DEVICE when no
 * gfx950 device is usable.
 */
#ifndef GPCORE_H
#define GPCORE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GP_MAX_K_VALUES 8

typedef enum gp_status {
    GP_OK = 0,
    GP_ERR_INVALID_ARG = -1,   /* bad pointer / size / parameter combination */
    GP_ERR_NO_DEVICE = -2,     /* no usable HIP device (no CPU fallback exists) */
    GP_ERR_HIP = -3,           /* a HIP runtime call failed; see gp_last_error() */
    GP_ERR_OOM = -4,           /* device or host allocation failed */
    GP_ERR_THRESHOLD = -5,     /* kmer_threshold < 4 (reference: src/utils.cpp:105-107) */
    GP_ERR_STATE = -6,         /* stage called out of order */
    GP_ERR_UNSUPPORTED = -7,   /* parameter outside what the kernels implement */
    GP_ERR_NOT_FOUND = -8,     /* a sequence id is not in the index */
    GP_ERR_CAPACITY = -9,      /* an edited contig outgrew its device slot (gp_job_polish retries by itself) */
    GP_ERR_IO = -10            /* an output file could not be written whole (it is removed / cut back) */
} gp_status;

typedef struct gp_ctx gp_ctx; /* one per device */
typedef struct gp_job gp_job; /* a set of batches resident on the device */
typedef struct gp_seqindex gp_seqindex; /* byte-offset index of a 2-line FASTA / 4-line FASTQ file */
typedef struct gp_mappings gp_mappings; /* target id -> mapped read ids */

/* Parameters of the path.  Defaults = what GoldPolish passes:
 *   k list 32,28,24,20                       scripts/goldpolish:189-190
 *   hash_num 4, cbf 10 MiB, bf 512 KiB       src/goldpolish_targeted_bfs.cpp:270-272
 *   ntEdit -d5 -i5 -m1 -X0.5 -Y0.5 -a1 -j3 -z100   scripts/goldpolish-ntedit:27,
 *                                            subprojects/ntedit/ntedit.cpp:85-115
 *   mask -s -k<first k>                      scripts/goldpolish-make:65-66
 *   Sealer -L<first k> -P10 -B1000 --lower, -G800 -C100000
 *                                            scripts/goldpolish-make:68-72,
 *                                            subprojects/sealer/sealer.cpp:131-165 */
typedef struct gp_params {
    uint32_t n_k;
    uint32_t k[GP_MAX_K_VALUES]; /* descending, each 4..63 */
    uint32_t hash_num;           /* 1..8 */
    uint64_t cbf_bytes;          /* counting filter: one u8 counter per byte */
    uint64_t bf_bytes;           /* Bloom filter bytes (bits = 8x) */
    /* ntEdit */
    uint32_t max_insertions;     /* -i 0..5 */
    uint32_t max_deletions;      /* -d 0..10 */
    int32_t mode;                /* -m; only 1 (GoldPolish's) is implemented on device */
    float missing_ratio;         /* -X */
    float edit_ratio;            /* -Y */
    uint32_t jump;               /* -j */
    uint32_t min_contig_len;     /* -z */
    int32_t soft_mask;           /* -a */
    /* Sealer */
    uint32_t sealer_max_gap;     /* -G */
    uint32_t sealer_max_paths;   /* -P */
    uint32_t sealer_max_branches;/* -B */
    uint32_t sealer_max_cost;    /* -C */
    int32_t run_sealer;          /* 0 skips stage C (mask and upper still run) */
} gp_params;

/* A set of batches in flat arrays (host memory, caller-owned, read during
 * gp_job_create only).  Batch b holds contigs [batch_contig_begin[b],
 * batch_contig_begin[b+1]); contig c holds bytes contig_buf[contig_off[c] ..
 * contig_off[c+1]) and uses reads [contig_read_begin[c], contig_read_begin[c+1])
 * -- ALREADY ranked and truncated as serve_batch does
 * (src/goldpolish_targeted_bfs.cpp:96-123; gp_rank_reads implements it) -- with
 * counting threshold contig_kmer_threshold[c] (gp_kmer_threshold).
 * contig_name_len[c] = strlen("id[ comment]") feeds the 0.75 byte-ratio guard of
 * scripts/goldpolish-ntedit:31-40. */
typedef struct gp_batch_set {
    uint32_t n_batches;
    const uint32_t* batch_contig_begin; /* [n_batches+1] */
    uint32_t n_contigs;
    const uint8_t* contig_buf;
    const uint64_t* contig_off;         /* [n_contigs+1] */
    const uint32_t* contig_name_len;    /* [n_contigs] */
    const uint32_t* contig_read_begin;  /* [n_contigs+1] */
    const uint32_t* contig_kmer_threshold; /* [n_contigs] */
    uint32_t n_reads;
    const uint8_t* read_buf;
    const uint64_t* read_off;           /* [n_reads+1] */
} gp_batch_set;

/* Work counters, filled by the kernels themselves (the
 * data-dependent terms of the algorithmic-bytes formula). */
typedef struct gp_counters {
    uint64_t a_positions;    /* read bases streamed by the filter build, summed over k */
    uint64_t a_ops;          /* valid k-mers = insert_thresh_contains calls */
    uint64_t a_ordered_ops;  /* ops that needed the ordered (reservation) path */
    uint64_t a_rounds;       /* reservation rounds launched */
    uint64_t b_scan_kmers;   /* k-mers tested by the membership scan */
    uint64_t b_sites;        /* positions where an absent k-mer was confirmed */
    uint64_t b_probes;       /* Bloom filter look-ups in the trial sweep */
    uint64_t b_edits;        /* accepted substitutions + indels */
    uint64_t c_gaps;         /* gaps found */
    uint64_t c_closed;       /* gaps closed */
    uint64_t c_vertex_tests; /* k-mer membership tests during BFS */
    uint64_t a_reservations; /* device-scope atomics spent on reservations (grid-wide schedule) */
    /* Sealer's per-k log lines (subprojects/sealer/sealer.cpp:656-683), summed over the k runs */
    uint64_t c_no_start_goal;     /* "No start/goal kmer" */
    uint64_t c_no_path;           /* "No path" */
    uint64_t c_unique_path;       /* "Unique path" */
    uint64_t c_multiple_paths;    /* "Multiple paths" */
    uint64_t c_too_many_paths;    /* "Too many paths" */
    uint64_t c_too_many_branches; /* "Too many branches" */
    uint64_t c_max_cost;          /* "Max cost exceeded" */
    uint64_t a_replayed;          /* operations cbf_local replayed: every one of a plain pair, the candidates of a census pair */
    uint64_t a_settled_hits;      /* ... of which found in the table of k-mers settled for good: one look-up each, nothing else */
    uint64_t a_counted;           /* candidate operations that needed no replay: their k-mer shares no counter with another candidate k-mer, its filter bits follow from the number of its copies */
} gp_counters;

/* ---- library / context ------------------------------------------------- */
const char* gp_last_error(void);
int gp_version(void);
/* Device memory (bytes, all devices) that live jobs and contexts of this library hold right now:
 * lets a caller size its jobs.  Blocks kept for re-use are NOT in it (gp_device_pool_bytes): what
 * the process really holds on its devices is the sum of the two.
 * A job needs about n_batches * n_k * (cbf_bytes + bf_bytes) for the filters, 30 bytes per contig
 * base and 1 byte per read base, plus 3 to 6 GiB of Sealer search arenas per job with run_sealer (39 GiB where contigs of 200 kbp and more are polished). */
uint64_t gp_device_bytes_in_use(void);
/* Device memory the library keeps for re-use: blocks of destroyed jobs go to a pool PER DEVICE, so
 * that the next job of a like shape costs no hipMalloc / hipFree (a stream of jobs:
 * gp_pipeline_*).  A device's pool never holds more than the most the library has had in use on
 * that device at one time, nor more than GP_POOL_GIB GiB (environment, default 200; 0 = no
 * pooling); it is emptied before an allocation of this library on that device is allowed to
 * fail, and when the last gp_ctx on the device is destroyed.  Other allocators in the process
 * (PyTorch's caching allocator, RCCL) cannot make this library give memory back: a caller that
 * mixes them in calls gp_device_pool_trim (all devices, everything) at its own quiet points. */
uint64_t gp_device_pool_bytes(void);
void gp_device_pool_trim(void);
/* Test hooks, not part of the reference's interface.  gp_debug_fail_alloc(n): the n-th device
 * allocation of this library from now on fails with GP_ERR_OOM (0 = off; tests of the
 * out-of-memory paths); returns the number of device allocations the library has made so far.
 * gp_debug_nthash: the kernels' OWN ntHash code over one sequence, so that it can be compared with
 * values made from the reference's header (subprojects/ntedit/lib/nthash.hpp:100-191,305-325)
 * directly and not only through the filters it fills.  variant 0: stage A as cbf_first_pass /
 * cbf_local hash (first k-mer of a lane's run from the table of rotated seeds, then rolls; ext_hash);
 * 1: stage A as census_scatter hashes (roll tables in LDS, extension hashes 1-3 from one multiply);
 * 2: stage B (ntedit_scan / the walker: seed classes of raw bytes, from scratch + roll_f / roll_r);
 * 3: stage C (kmer_of, then successor hashes as the forward search derives them; every value is also
 * derived as a predecessor, the backward search's step: a disagreement gives valid = 2);
 * 4: stage B's changelast -- the first k-mer with its last base replaced by A, C, G, T, a, N, Y: seven
 * entries (the output arrays must hold max(len, 8) entries).
 * For every k-mer start i in [0, len - k]: out_fwd[i], out_rev[i], out_hashes[i * hash_num + h]
 * (h = 0 canonical = fwd + rev, then the extensions); valid[i] = 0 where the k-mer holds a byte
 * outside ACGTacgt (stage A skips those; their values are unspecified for variants 0, 1, 3).
 * Returns the number of entries written, or a negative GP_ERR_*.  k <= 63. */
uint64_t gp_debug_fail_alloc(uint64_t nth);
int64_t gp_debug_nthash(gp_ctx* ctx, int variant, const uint8_t* seq, uint64_t len, uint32_t k, uint32_t hash_num,
                        uint64_t* out_fwd, uint64_t* out_rev, uint64_t* out_hashes, uint8_t* valid);
void gp_default_params(gp_params* p);
int gp_ctx_create(int device, gp_ctx** out);
/* The same with every stream of the context restricted to the compute units whose bits are set in
 * cu_mask (hipExtStreamCreateWithCUMask: bit i of word i / 32 = CU i): two contexts with
 * complementary masks share the chip side by side, e.g. the filter build of one job on three
 * quarters of it beside the latency-bound ntEdit / Sealer kernels of another (gp_jobs_polish). */
int gp_ctx_create_masked(int device, const uint32_t* cu_mask, uint32_t mask_words, gp_ctx** out);
void gp_ctx_destroy(gp_ctx* ctx);
/* HIP stream (hipStream_t) every kernel of this context is launched on, so a
 * caller can bracket gp_job_* calls with its own HIP events. */
void* gp_ctx_stream(gp_ctx* ctx);

/* ---- host-side helpers that belong to the path -------------------------- */
/* mappings_bases_to_kmer_threshold (src/goldpolish_targeted_bfs.cpp:45-53) */
int gp_kmer_threshold(uint64_t mapped_bases);
/* read ranking + cap of serve_batch (src/goldpolish_targeted_bfs.cpp:96-123):
 * order[] receives the indices of the reads to use, best first; returns how
 * many; *bases = sum of their lengths.  phred is truncated to an integer first,
 * as the reference does. */
size_t gp_rank_reads(size_t n, const char* const* ids, const double* phred_avg,
                     const uint64_t* lens, uint64_t target_len, double max_reads_per_10kbp,
                     uint32_t* order, uint64_t* bases);
/* Which rank polishes which batch (the reference has no counterpart: its batches
 * go to whichever worker holds a token, scripts/goldpolish:496-514): batches are dealt
 * longest-first -- cost[b] = contig bases + mapped read bases of batch b -- to the rank with the
 * least load so far; ties go to the lower batch index and the lower rank, so every rank computes
 * the same answer without talking.  rank_of[n_batches] out; rank_load[n_ranks] out, may be NULL. */
int gp_partition_batches(uint32_t n_batches, const uint64_t* cost, uint32_t n_ranks,
                         uint32_t* rank_of, uint64_t* rank_load);

/* ---- jobs --------------------------------------------------------------- */
/* Copies the batch set to the device (host -> HBM over PCIe, through the context's two pinned
 * staging chunks and its stream; the call returns when the copies are complete).  Filters and
 * work buffers are allocated by the first call that needs them (gp_job_polish, a stage, a
 * getter), not here: a job that has only been created holds its inputs and nothing else, so it
 * can be created beside another job's polishing (gp_pipeline_* does) without both holding tens
 * of GiB.  An allocation failure therefore shows up at that first call (GP_ERR_OOM; the call may
 * be repeated once memory has been freed). */
int gp_job_create(gp_ctx* ctx, const gp_params* params, const gp_batch_set* set, gp_job** out);
void gp_job_destroy(gp_job* job);
/* The whole path for every batch of the job: filters -> ntEdit cascade (+guard)
 * -> mask -> Sealer -> upper.  Asynchronous on gp_ctx_stream unless sync != 0.
 * May be called again: it restarts from the uploaded inputs. */
int gp_job_polish(gp_job* job, int sync);
/* gp_job_polish(job, 1) for several jobs, pipelined: the filter build of job i + 1 runs beside
 * the ntEdit cascade, mask, Sealer and upper of job i (two host threads), so the memory-bound
 * stage hides the latency-bound ones -- what the reference gets from running polishing batches
 * beside its filter server (scripts/goldpolish:496-514).  Every job must have been created on a
 * gp_ctx of its own, all on one device.  The contexts are BORROWED for the duration of the call:
 * each job's ctx->stream is re-pointed -- to the first job's stream while its filters are built,
 * to the second job's for everything behind them -- and put back before the call returns; the
 * filter build also uses a private second stream per job (census mark / pack, reservation
 * rounds).  So no other thread may touch any of these contexts or jobs (gp_job_sync included)
 * while the call runs.  Returns the first failing job's status, its message prefixed with the
 * job's index. */
int gp_jobs_polish(gp_job* const* jobs, uint32_t n_jobs);
/* Individual stages, in order: for tests, profiling and the per-tool programs (examples/gp_tools.cpp). */
/* current sequences := the uploaded contigs (what gp_job_ntedit_pass(job, 0, ..) does before its
 * pass): lets a later stage run on the inputs as they are, e.g. Sealer alone as `sealer-gr` does */
int gp_job_reset(gp_job* job, int sync);
int gp_job_build_filters(gp_job* job, int sync);          /* stage A */
int gp_job_ntedit_pass(gp_job* job, uint32_t k_index, int sync); /* stage B, one k */
int gp_job_ntedit_guard(gp_job* job, int sync);           /* 0.75 byte-ratio guard */
int gp_job_mask(gp_job* job, int sync);
int gp_job_sealer(gp_job* job, int sync);                 /* stage C */
int gp_job_upper(gp_job* job, int sync);
int gp_job_sync(gp_job* job);

/* Sealer's consensus step alone: align() of subprojects/sealer/Align/alignGlobal.h:73-83 (alignPair
 * for two connecting sequences, alignMulti's left fold for more; the Gotoh aligner and its
 * ambiguity-code consensus of Align/alignGlobal.cc:28-198, Common/Sequence.h:97-106) as
 * Konnector/konnector.h:393 calls it for a gap with several paths -- run by the same device code
 * stage C uses (sealer_consensus), so that it can be checked against vectors made from the
 * reference's own aligner (tests/golden/nw_kat.json).  Set s holds sequences
 * [set_seq_begin[s], set_seq_begin[s+1]) (two at least), sequence q the bytes
 * seq_buf[seq_off[q] .. seq_off[q+1]); its consensus goes to out_buf + s * out_stride,
 * out_len[s] bytes.  Sequences and running consensus are limited to 1800 bases (a connecting
 * sequence is at most -G + 2 * -L = 864): longer ones give GP_ERR_UNSUPPORTED. */
int gp_sealer_consensus(gp_ctx* ctx, uint32_t n_sets, const uint32_t* set_seq_begin,
                        const uint8_t* seq_buf, const uint64_t* seq_off, uint8_t* out_buf,
                        uint64_t out_stride, uint32_t* out_len);

/* ---- a stream of batch sets, host memory to host memory ------------------------------------
 * SURVEY.md 8d's timed region -- reads, mappings and contigs resident in host memory to polished
 * bytes in host memory -- with the three things that can overlap overlapping: while set i is
 * polished, set i + 1 is uploaded (gp_job_create on a second context: pinned staging, DMA on its
 * own stream) and the caller still reads the result of set i - 1.  Results come back in
 * submission order (the ordered collection of scripts/goldpolish-reaper:51-73).  The reference
 * overlaps the same steps with processes: its filter server works on the next batches while the
 * polishing workers run (scripts/goldpolish:496-514).  A set that holds a long contig is taken in
 * two halves -- its filter build, then everything behind it -- by threads of their own, so that the
 * build of one such set runs beside the back halves of the two before it (a depth of 4 keeps all of
 * them busy; INTEGRATION.md, GP_PIPELINE_*).
 *
 * gp_pipeline_submit never blocks: with `depth` sets submitted and not yet collected it returns
 * GP_ERR_STATE.  The arrays of a submitted gp_batch_set must stay valid and unchanged until the
 * gp_pipeline_collect that returns that set.  gp_pipeline_collect blocks until the oldest
 * submitted set is done; what it hands out (pinned host memory and an HBM copy of the same bytes,
 * e.g. for an RCCL send) belongs to the pipeline and stays valid until the next
 * gp_pipeline_collect or gp_pipeline_destroy.  A set that failed gives its status and message
 * here, prefixed with its tag; the pipeline goes on with the next one.  One thread at a time may
 * use a pipeline. */
typedef struct gp_pipeline gp_pipeline;
typedef struct gp_polished {
    uint64_t tag;                /* as submitted */
    uint32_t n_contigs;
    const uint64_t* lengths;     /* [n_contigs], 0 for a contig ntEdit dropped */
    const uint8_t* dropped;      /* [n_contigs] */
    const uint8_t* host_bytes;   /* polished sequences back to back, in contig order (pinned host memory) */
    const void* device_bytes;    /* the same bytes in HBM */
    uint64_t total_bytes;
    gp_counters counters;
    double upload_ms;            /* gp_job_create of this set (ran beside the polishing of the set before) */
    double polish_ms;            /* the whole path on the device */
    double download_ms;          /* packing + the copy to host_bytes */
    uint64_t device_bytes_in_use;/* gp_device_bytes_in_use() when this set was done */
} gp_polished;
int gp_pipeline_create(int device, const gp_params* params, uint32_t depth, gp_pipeline** out);
int gp_pipeline_submit(gp_pipeline* p, const gp_batch_set* set, uint64_t tag);
int gp_pipeline_collect(gp_pipeline* p, gp_polished* out);
/* gp_job_kernel_time for the set the last gp_pipeline_collect returned */
int gp_pipeline_kernel_time(gp_pipeline* p, const char* kernel, double* total_ms, uint64_t* launches,
                            uint64_t* units);
void gp_pipeline_destroy(gp_pipeline* p);

/* ---- results (device -> host copies) ------------------------------------ */
/* Bloom filter of batch b for k[k_index]: bf_bytes bytes into out. */
int gp_job_get_filter(gp_job* job, uint32_t batch, uint32_t k_index, uint8_t* out);
/* Replace the device filters with caller-supplied bytes (n_batches*n_k*bf_bytes,
 * batch-major): lets stage B/C run against a known filter. */
int gp_job_set_filters(gp_job* job, const uint8_t* filters);
/* Current sequence of every contig: lengths[c] == 0 and dropped[c] == 1 for
 * contigs ntEdit dropped (shorter than min_contig_len).  Sequences are returned
 * concatenated in contig order; call with buf == NULL to get total bytes. */
int gp_job_get_lengths(gp_job* job, uint64_t* lengths, uint8_t* dropped, uint64_t* total);
int gp_job_get_sequences(gp_job* job, uint8_t* buf, uint64_t buf_bytes);
/* Same bytes, written to a DEVICE pointer on the job's GPU (e.g. a buffer that goes
 * straight into an RCCL send): no host hop. */
int gp_job_get_sequences_device(gp_job* job, void* device_buf, uint64_t buf_bytes);
/* The job's final FASTA, records in input order, dropped ones left out (what
 * scripts/goldpolish-reaper:51-73 collects).  names[c] = "id[ comment]"; NULL = the header lines of
 * the target file for jobs made by gp_job_create_from_files.  append != 0 appends.  Every write
 * is checked: on failure (disk full in /dev/shm ...) the answer is GP_ERR_IO and the file is left
 * as it was before the call (a new one removed, an appended one cut back). */
int gp_job_write_fasta(gp_job* job, const char* const* names, const char* path, int append);
int gp_job_get_counters(gp_job* job, gp_counters* out);
/* Device time (ms, HIP events on the stream the kernels run on) of a kernel family since the last
 * gp_job_build_filters / gp_job_polish, the number of timed spans and the units they covered.
 * kernel: "cbf_first_pass", "cbf_round" (grid-wide schedule), "cbf_census" (census_scatter / _mark /
 * _pack and, inside the same span, "cbf_ckc": the candidates-by-k-mer kernels), "cbf_local",
 * "ntedit_scan" (with ntedit_site_eval), "ntedit_walk", "mask", "sealer_bfs", "upper". */
int gp_job_kernel_time(gp_job* job, const char* kernel, double* total_ms, uint64_t* launches,
                       uint64_t* units);


/* ---- the step in front of the path: which reads reach the filter build -----
 * Host-side (no device work), SURVEY.md 8f-2.
 *
 * gp_seqindex = SeqIndex (src/seqindex.cpp:12-137, seqindex.hpp:59-102) and the
 * `goldpolish-index` tool (src/goldpolish_index.cpp): build from a sequence file
 * or load a saved index (lines "id<TAB>start<TAB>length<TAB>avg_phred"), save,
 * look up, fetch the bytes of one sequence. */
int gp_seqindex_build(const char* seqs_path, gp_seqindex** out);
int gp_seqindex_load(const char* index_path, const char* seqs_path, gp_seqindex** out);
int gp_seqindex_save(const gp_seqindex* ix, const char* index_path);
uint64_t gp_seqindex_count(const gp_seqindex* ix);
int gp_seqindex_entry(const gp_seqindex* ix, uint64_t i, const char** id, uint64_t* start,
                      uint64_t* len, double* phred); /* i-th record in file order */
int gp_seqindex_find(const gp_seqindex* ix, const char* id, uint64_t* start, uint64_t* len,
                     double* phred);                 /* GP_ERR_NOT_FOUND if absent */
int gp_seqindex_read(const gp_seqindex* ix, const char* id, uint8_t* buf, uint64_t buf_bytes,
                     uint64_t* len);                 /* buf == NULL: length only */
void gp_seqindex_destroy(gp_seqindex* ix);
/* gp_mappings = AllMappings (src/mappings.cpp:15-345): ".sam" -> SAM, ".paf" -> PAF,
 * anything else -> ntLink "read target minimizers" table, filtered per target by
 * the common-minimizer threshold found in [mx_min, mx_max] for at most
 * mx_max_mapped_per_10kbp reads per 10 kbp of target (GoldPolish: 1, 30, -x). */
int gp_mappings_load(const char* path, const gp_seqindex* targets, uint32_t mx_min, uint32_t mx_max,
                     double mx_max_mapped_per_10kbp, gp_mappings** out);
int gp_mappings_get(gp_mappings* m, const char* target_id, uint64_t* n, const char* const** ids);
void gp_mappings_destroy(gp_mappings* m);
/* serve_batch's read selection (src/goldpolish_targeted_bfs.cpp:86-131) for every target of
 * every batch, and the batch's contigs as scripts/goldpolish:344-355 writes them: batch b holds
 * target_ids[batch_target_begin[b] .. batch_target_begin[b+1]).  The job is then the same as one
 * made by gp_job_create from the same bytes. */
int gp_job_create_from_files(gp_ctx* ctx, const gp_params* params, const gp_seqindex* targets,
                             const gp_seqindex* reads, gp_mappings* mappings,
                             double max_reads_per_10kbp, uint32_t n_batches,
                             const uint32_t* batch_target_begin, const char* const* target_ids,
                             gp_job** out);

/* One Bloom filter to / from a file, in this build's own container ("GPBF1 k hash_num bytes\n" +
 * the bytes of gp_job_get_filter): what `bfs[i]->save("<batch>-k<k>.bf")` and ntEdit's / Sealer's
 * filter loading do for the reference, for tools that are all this build's.  btllib's own .bf
 * header is never written; gp_filter_load also accepts a "[BTLKmerBloomFilter_v*]" / key = value /
 * "[HeaderEnd]" header as SURVEY.md 8a-A7 lays it out, UNPINNED (no btllib-written file here,
 * SURVEY.md 8f-1).  buf == NULL: header only. */
int gp_filter_save(const char* path, uint32_t k, uint32_t hash_num, uint64_t bytes, const uint8_t* data);
/* The same bytes under btllib's published header ("[BTLKmerBloomFilter_v2]" / bytes, hash_fn =
 * "ntHash_v2", hash_num, k / "[HeaderEnd]"): what a reference-built ntedit-gr / sealer-gr would be
 * handed as <batch>-k<k>.bf (scripts/goldpolish:36,391-396).  UNPINNED, as the reader is. */
int gp_filter_save_btllib(const char* path, uint32_t k, uint32_t hash_num, uint64_t bytes, const uint8_t* data);
int gp_filter_load(const char* path, uint32_t* k, uint32_t* hash_num, uint64_t* bytes, uint8_t* buf,
                   uint64_t buf_bytes);

/* ---- GoldPolish-Target's file-to-file steps around the path, host side -------
 * gp_target_extract: windows to polish = soft-masked runs (bed_path == NULL) or BED intervals,
 * each with `flank` bases either side, written upper-case as ">contig.N start-end"
 * (scripts/goldpolish-target-extract-seq.py:60-148).
 * gp_target_update_paf: PAF rows on the assembly -> rows on the windows, coordinates clipped
 * (scripts/goldpolish-target-update-mapping.py:70-155).
 * gp_target_reinsert: polished windows back into the assembly
 * (scripts/goldpolish-target-post-processing.py:28-94). */
int gp_target_extract(const char* fasta_path, const char* bed_path, uint32_t flank, const char* out_path);
int gp_target_update_paf(const char* gaps_fasta_path, const char* paf_path, const char* out_path);
int gp_target_reinsert(const char* fasta_path, const char* polished_gaps_path, const char* out_path);

#ifdef __cplusplus
}
#endif
#endif /* GPCORE_H */
