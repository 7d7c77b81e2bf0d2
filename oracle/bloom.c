/*
 * oracle/bloom.c -- CPU ORACLE (test infrastructure, never shipped).
 * Stage A of the path: per-batch counting-filter thresholding + Bloom filter
 * build.  Follows
 *   /root/reference/src/utils.cpp:96-123                  (fill_bfs)
 *   /root/reference/src/goldpolish_targeted_bfs.cpp:45-53 (threshold formula)
 *   /root/reference/src/goldpolish_targeted_bfs.cpp:96-123 (read ranking)
 * The filter arithmetic itself lives in btllib (>= 1.6.2, not in the tree and
 * not installed): PARITY UNPINNED.  Published semantics restated here:
 *   KmerCountingBloomFilter8::insert_thresh_contains(h, thr):
 *       c = min_i cnt[h_i % bytes];
 *       if (c < thr) { every cnt[h_i % bytes] == c becomes c+1; return c+1; }
 *       return c;
 *   KmerBloomFilter::insert / contains: bit n = h_i % (bytes*8) lives in
 *       byte n/8 under mask 1 << (n % 8).
 *   NtHash::roll skips every k-mer holding a byte outside ACGTacgt.
 */
#include "gpo.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

int gpo_kmer_threshold(uint64_t mappings_bases)
{
    const double a = 4.66943, b = 2.11391e-07;
    int t = (int)round(a + (double)mappings_bases * b);
    return t < 13 ? t : 13;
}

typedef struct {
    uint32_t idx;
    uint64_t phred; /* double -> size_t truncation happens on purpose (:102-105) */
    const char* id;
} rank_item;

static int rank_cmp(const void* pa, const void* pb)
{
    const rank_item* a = (const rank_item*)pa;
    const rank_item* b = (const rank_item*)pb;
    if (a->phred != b->phred)
        return a->phred > b->phred ? -1 : 1;
    return strcmp(a->id, b->id); /* std::string operator< on ASCII ids */
}

size_t gpo_rank_reads(size_t n, const char* const* ids, const double* phred_avg,
                      const uint64_t* lens, uint64_t target_len, double s_per_10kbp,
                      uint32_t* order, uint64_t* bases)
{
    *bases = 0;
    if (n == 0)
        return 0;
    size_t n_max = (size_t)((double)target_len * s_per_10kbp / 10000.0);
    size_t n_use = n < n_max ? n : n_max;
    rank_item* it = (rank_item*)malloc(n * sizeof *it);
    for (size_t i = 0; i < n; i++) {
        it[i].idx = (uint32_t)i;
        it[i].phred = (uint64_t)phred_avg[i];
        it[i].id = ids[i];
    }
    qsort(it, n, sizeof *it, rank_cmp); /* total order: ids are unique */
    for (size_t i = 0; i < n_use; i++) {
        order[i] = it[i].idx;
        *bases += lens[it[i].idx];
    }
    free(it);
    return n_use;
}

int gpo_bf_contains(const uint8_t* bf, size_t bf_bytes, const uint64_t* hashes, unsigned h)
{
    const uint64_t bits = (uint64_t)bf_bytes * 8u;
    for (unsigned i = 0; i < h; i++) {
        uint64_t n = hashes[i] % bits;
        if (!(bf[n >> 3] & (1u << (n & 7u))))
            return 0;
    }
    return 1;
}

static void bf_insert(uint8_t* bf, size_t bf_bytes, const uint64_t* hashes, unsigned h)
{
    const uint64_t bits = (uint64_t)bf_bytes * 8u;
    for (unsigned i = 0; i < h; i++) {
        uint64_t n = hashes[i] % bits;
        bf[n >> 3] |= (uint8_t)(1u << (n & 7u));
    }
}

static unsigned cbf_insert_thresh_contains(uint8_t* cbf, size_t cbf_bytes,
                                           const uint64_t* hashes, unsigned h, unsigned thr)
{
    size_t at[16];
    unsigned c = 255;
    for (unsigned i = 0; i < h; i++) {
        at[i] = (size_t)(hashes[i] % cbf_bytes);
        if (cbf[at[i]] < c)
            c = cbf[at[i]];
    }
    if (c < thr) {
        for (unsigned i = 0; i < h; i++)
            if (cbf[at[i]] == c)
                cbf[at[i]] = (uint8_t)(c + 1);
        return c + 1;
    }
    return c;
}

int gpo_fill_bfs(const char* seq, size_t seq_len, unsigned hash_num, unsigned nk,
                 const unsigned* k_values, unsigned kmer_threshold, uint8_t* const* cbfs,
                 size_t cbf_bytes, uint8_t* const* bfs, size_t bf_bytes)
{
    if (kmer_threshold < 4 || hash_num == 0 || hash_num > 16)
        return -1;
    unsigned thr = kmer_threshold - 2; /* then +1 per k in argument order */
    uint64_t* base = (uint64_t*)malloc((seq_len + 1) * sizeof *base);
    uint8_t* ok = (uint8_t*)malloc(seq_len + 1);
    for (unsigned ki = 0; ki < nk; ki++, thr++) {
        unsigned k = k_values[ki];
        size_t n = gpo_nthash_all(seq, seq_len, k, base, ok);
        uint64_t hv[16];
        for (size_t p = 0; p < n; p++) {
            if (!ok[p])
                continue;
            gpo_extend(base[p], k, hash_num, hv);
            if (cbf_insert_thresh_contains(cbfs[ki], cbf_bytes, hv, hash_num, thr) >= thr)
                bf_insert(bfs[ki], bf_bytes, hv, hash_num);
        }
    }
    free(base);
    free(ok);
    return 0;
}
