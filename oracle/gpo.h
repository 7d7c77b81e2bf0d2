/*
 * gpo.h -- CPU ORACLE for the GoldPolish per-batch k-mer correction path.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a plain-C restatement of the reference
 * algorithm (bcgsc/goldpolish @ /root/reference) used as the *checker* by
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.  Nothing in
 * goldpolish_amd/ (the product) may include, link or call it.
 *
 * Parity status (see DESIGN.md "Oracle pinning"):
 *   - ntHash arithmetic: PINNED against the reference's own in-tree header
 *     subprojects/ntedit/lib/nthash.hpp compiled where it lies
 *     (oracle/_ref/nthash_ref, vectors in tests/golden/nthash_kat.json).
 *   - mask / to-upper: restated from scripts/goldpolish-mask:44-72 and
 *     scripts/goldpolish-to-upper:17-21; hand-derived vectors only.
 *   - stage A counting filter / Bloom filter bit mapping (btllib, not in tree),
 *     stage B (ntedit.cpp needs btllib headers -> unbuildable here) and
 *     stage C (sealer needs boost + btllib -> unbuildable here):
 *     **parity unpinned** -- restated by reading; the reference's only tests for
 *     them are end-to-end goldens whose reads are not in the tree.
 */
#ifndef GPO_H
#define GPO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- ntHash */
/* subprojects/ntedit/lib/nthash.hpp:24-63 (seeds), :66-91 (rotations),
 * :100-119 (base hashes), :122-152 (rolls), :180-191 (canonical = fh + rh),
 * :305-314 (extension hashes). */
uint64_t gpo_seed(unsigned char c);
uint64_t gpo_ntf64(const char* kmer, unsigned k);
uint64_t gpo_ntr64(const char* kmer, unsigned k);
uint64_t gpo_ntf64_roll(uint64_t fh, unsigned k, unsigned char out, unsigned char in);
uint64_t gpo_ntr64_roll(uint64_t rh, unsigned k, unsigned char out, unsigned char in);
uint64_t gpo_ntf64_changelast(uint64_t fh, unsigned k, unsigned char out, unsigned char in);
uint64_t gpo_ntr64_changelast(uint64_t rh, unsigned k, unsigned char out, unsigned char in);
/* h[0] = base, h[i] = base * (i ^ k*multiSeed); h[i] ^= h[i] >> 27 */
void gpo_extend(uint64_t base, unsigned k, unsigned h, uint64_t* out);
/* Canonical base hash of every k-mer of seq; valid[p]=0 where the k-mer holds a
 * byte whose seed is 0 (btllib NtHash skips those).  Returns number of k-mer
 * positions (len-k+1 or 0). */
size_t gpo_nthash_all(const char* seq, size_t len, unsigned k, uint64_t* out_hash,
                      uint8_t* out_valid);

/* --------------------------------------------------------------- stage A */
/* src/goldpolish_targeted_bfs.cpp:45-53 */
int gpo_kmer_threshold(uint64_t mappings_bases);
/* src/goldpolish_targeted_bfs.cpp:96-123: order = indices of the reads to use,
 * already sorted (phred truncated to size_t, desc; id asc) and truncated to
 * min(n, size_t(target_len*s/10000)).  ids are compared as C strings.
 * Returns the number of selected reads; *bases = sum of their lengths. */
size_t gpo_rank_reads(size_t n, const char* const* ids, const double* phred_avg,
                      const uint64_t* lens, uint64_t target_len, double s_per_10kbp,
                      uint32_t* order, uint64_t* bases);
/* src/utils.cpp:96-123 with btllib KmerCountingBloomFilter8::insert_thresh_contains
 * (conservative update; returns the count AFTER insertion) and
 * KmerBloomFilter::insert (bit = h % (bytes*8); byte n/8, mask 1<<(n%8)).
 * cbfs[i] / bfs[i] are caller-owned, zero-initialised arrays.
 * Returns 0, or -1 if kmer_threshold < 4 (utils.cpp:105-107). */
int gpo_fill_bfs(const char* seq, size_t seq_len, unsigned hash_num, unsigned nk,
                 const unsigned* k_values, unsigned kmer_threshold, uint8_t* const* cbfs,
                 size_t cbf_bytes, uint8_t* const* bfs, size_t bf_bytes);
/* membership test used by ntEdit / Sealer (btllib KmerBloomFilter::contains) */
int gpo_bf_contains(const uint8_t* bf, size_t bf_bytes, const uint64_t* hashes, unsigned h);

/* --------------------------------------------------------------- stage B */
typedef struct {
    unsigned k;              /* from the BF header (ntedit.cpp:2013-2022) */
    unsigned h;
    unsigned max_insertions; /* -i, <=5 */
    unsigned max_deletions;  /* -d, <=10 */
    int mode;                /* -m 0|1|2 (GoldPolish: 1) */
    int use_ratio;           /* set when -X or -Y given */
    float missing_ratio;     /* -X */
    float edit_ratio;        /* -Y */
    float missing_threshold; /* -x */
    float edit_threshold;    /* -y */
    unsigned jump;           /* -j, default 3 */
    unsigned min_contig_len; /* -z, default 100 */
    int mask;                /* -a */
    unsigned insertion_cap;  /* unsigned(float(k)*1.5f) (ntedit.cpp:2024-2025) */
} gpo_ntedit_params;

void gpo_ntedit_default_params(gpo_ntedit_params* p, unsigned k, unsigned h);

/* kmerizeAndCorrect + writeEditsToFile (ntedit.cpp:1415-1771, 781-936) for ONE
 * contig of length >= min_contig_len.  out must hold at least
 * gpo_ntedit_out_bound(len) bytes; returns the edited length. */
size_t gpo_ntedit_out_bound(size_t len);
/* the i-th (0..340) insertion trial string starting with `first` (ntedit.cpp:198-343), for tests */
unsigned gpo_ntedit_indel_string(unsigned char first, int i, char out[6]);
/* the same with a malloc'ed result of exactly the edited length + 1 (free with gpo_free) */
size_t gpo_ntedit_contig_dyn(const char* seq, size_t len, const uint8_t* bf, size_t bf_bytes,
                             const gpo_ntedit_params* p, char** out, uint64_t* n_probes);
size_t gpo_ntedit_contig(const char* seq, size_t len, const uint8_t* bf, size_t bf_bytes,
                         const gpo_ntedit_params* p, char* out, uint64_t* n_probes);

/* ---------------------------------------------------- mask / upper (B->C) */
/* scripts/goldpolish-mask:44-72 (soft mask).  out needs max(len,1)+1 bytes. */
size_t gpo_mask(const char* seq, size_t len, unsigned k, char* out);
/* scripts/goldpolish-to-upper:17-21 */
void gpo_upper(char* seq, size_t len);

/* --------------------------------------------------------------- stage C */
typedef struct {
    unsigned flank_len;     /* -L (first k) */
    unsigned max_gap_len;   /* -G 800 */
    unsigned max_paths;     /* -P 10 */
    unsigned max_branches;  /* -B 1000 */
    unsigned max_cost;      /* -C 100000 */
    uint64_t mem_limit;     /* -s 500M */
    int lower;              /* --lower */
} gpo_sealer_params;
void gpo_sealer_default_params(gpo_sealer_params* p, unsigned first_k);

typedef struct {
    uint64_t gaps_found, gaps_closed, no_start_goal, no_path, unique_path, multiple_paths,
        too_many_paths, too_many_branches, max_cost_exceeded, vertex_tests;
} gpo_sealer_stats;

/* sealer.cpp main (:738-1115) for ONE scaffold record: findFlanks, per k kRun,
 * insertIntoScaffold.  bfs[i] is the filter for k_values[i].  Returns the new
 * length; out must hold gpo_sealer_out_bound(len) bytes. */
size_t gpo_sealer_out_bound(size_t len);
size_t gpo_sealer_record(const char* seq, size_t len, unsigned nk, const unsigned* k_values,
                         const uint8_t* const* bfs, size_t bf_bytes, unsigned hash_num,
                         const gpo_sealer_params* p, char* out, gpo_sealer_stats* st);

/* Sealer's consensus step alone (Align/alignGlobal.cc:55-198, alignGlobal.h:44-83,
 * Common/Sequence.cpp:21-57,141-204): pinned by tests/golden/nw_kat.json.  out of
 * gpo_nw_consensus needs la + lb + 1 bytes, of gpo_align_multi sum(lens) + 1. */
size_t gpo_nw_consensus(const char* a, size_t la, const char* b, size_t lb, char* out, unsigned* matches);
size_t gpo_align_multi(unsigned n, const char* const* seqs, const size_t* lens, char* out, unsigned* matches);
void gpo_revcomp(const char* s, size_t n, char* out);
char gpo_amb_or(char a, char b);

/* ------------------------------------------------------ whole batch (A->C) */
typedef struct {
    unsigned nk;
    unsigned k_values[8];
    unsigned hash_num;  /* 4  (goldpolish_targeted_bfs.cpp:272) */
    size_t cbf_bytes;   /* 10 MiB (:270) */
    size_t bf_bytes;    /* 512 KiB (:271) */
    float ntedit_x;     /* NTEDIT_X=0.5 (scripts/goldpolish-make:16) */
    float ntedit_y;     /* NTEDIT_Y=0.5 */
    unsigned sealer_P;  /* 10 */
    unsigned sealer_B;  /* 1000 */
    int run_sealer;     /* 1 */
} gpo_batch_params;
void gpo_batch_default_params(gpo_batch_params* p);

typedef struct {
    uint64_t a_ops;        /* k-mer insert_thresh_contains calls */
    uint64_t b_probes;     /* ntEdit contains() calls */
    uint64_t c_vertex_tests;
    uint64_t gaps_found, gaps_closed;
    int ntedit_skipped;    /* 0.75 guard fired (scripts/goldpolish-ntedit:31-40) */
} gpo_batch_stats;

/* One batch: targets (contigs) with, per target, its already ranked+truncated
 * reads and kmer_threshold (serve_batch, goldpolish_targeted_bfs.cpp:55-146),
 * then the ntEdit cascade + guard, mask, Sealer, upper
 * (scripts/goldpolish-make:62-72).  Contig names are "id[ comment]".
 * out_seq[i] is malloc'd by the oracle (free with gpo_free); out_len[i]==0 and
 * out_seq[i]==NULL for contigs dropped by ntEdit (< min_contig_len) when the
 * guard did not fire. */
int gpo_polish_batch(const gpo_batch_params* bp, unsigned n_contigs, const char* const* names,
                     const char* const* seqs, const uint64_t* lens,
                     const uint32_t* read_begin /* n_contigs+1 */, const char* const* read_seqs,
                     const uint64_t* read_lens, const uint32_t* kmer_threshold /* per contig */,
                     char** out_seq, uint64_t* out_len, uint8_t** out_bfs /* nk, optional */,
                     gpo_batch_stats* st);
/* Many batches the way the reference runs them on a multi-core host: one OpenMP task per batch
 * in the filter server (serve, goldpolish_targeted_bfs.cpp:189-246) and one polishing worker per
 * batch (scripts/goldpolish:496-514), i.e. batches in parallel, each batch serial.  Batch b holds
 * contigs [batch_contig_begin[b], batch_contig_begin[b+1]); read_begin indexes the one global
 * read list.  n_threads <= 0: every core OpenMP sees.  Returns 0, or the first failing batch's
 * code; *threads_used reports the team size.  This is bench.py's `cpu_baseline` leg. */
int gpo_polish_batches(const gpo_batch_params* bp, unsigned n_batches,
                       const uint32_t* batch_contig_begin, const char* const* names,
                       const char* const* seqs, const uint64_t* lens, const uint32_t* read_begin,
                       const char* const* read_seqs, const uint64_t* read_lens,
                       const uint32_t* kmer_threshold, char** out_seq, uint64_t* out_len,
                       int n_threads, int* threads_used);
void gpo_free(void* p);

#ifdef __cplusplus
}
#endif
#endif
