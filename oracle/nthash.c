/*
 * oracle/nthash.c -- CPU ORACLE (test infrastructure, never shipped).
 * ntHash as the reference's in-tree header states it:
 *   /root/reference/subprojects/ntedit/lib/nthash.hpp
 *     seeds :24-28, table :30-63, split rotate :66-91, base hashes :100-119,
 *     rolling :122-131,143-152, change-last :134-140,154-170,
 *     canonical = forward + reverse :180-191, extension hashes :305-314.
 * Pinned by tests/golden/nthash_kat.json (made by oracle/_ref/nthash_ref, a
 * driver that includes that header from where it lies).
 */
#include "gpo.h"

#include <string.h>

#define SEED_A 0x3c8bfbb395c60474ULL
#define SEED_C 0x3193c18562a02b4cULL
#define SEED_G 0x20323ed082572324ULL
#define SEED_T 0x295549f54be24456ULL
#define MULTISEED 0x90b45d39fb6da1faULL
#define MULTISHIFT 27

/* forward seed: only A,C,G,T in either case are non-zero (nthash.hpp:30-63) */
uint64_t gpo_seed(unsigned char c)
{
    switch (c) {
    case 'A': case 'a': return SEED_A;
    case 'C': case 'c': return SEED_C;
    case 'G': case 'g': return SEED_G;
    case 'T': case 't': return SEED_T;
    default: return 0;
    }
}

/* complement seed: the header indexes its table with (c & 0x07)
 * (nthash.hpp:14,115,144): slots 1,3,4,7 hold T,G,A,C; the rest are 0.  This
 * makes some IUPAC letters (e.g. 'Y' -> T, 'D' -> A) hash non-zero on the
 * reverse strand only -- kept on purpose. */
static uint64_t rc_seed(unsigned char c)
{
    static const uint64_t low3[8] = { 0, SEED_T, 0, SEED_G, SEED_A, 0, 0, SEED_C };
    return low3[c & 7u];
}

/* one-step split rotate: low 33 bits and high 31 bits rotate independently
 * (rol1 + swapbits033, nthash.hpp:66-91) */
static inline uint64_t srol1(uint64_t x)
{
    uint64_t carry = ((x & 0x8000000000000000ULL) >> 30) | ((x & 0x100000000ULL) >> 32);
    return ((x << 1) & 0xFFFFFFFDFFFFFFFFULL) | carry;
}

static inline uint64_t sror1(uint64_t x)
{
    uint64_t carry = ((x & 1ULL) << 32) | ((x & 0x200000000ULL) << 30);
    return ((x >> 1) & 0xFFFFFFFEFFFFFFFFULL) | carry;
}

/* n-step split rotate (rol31/rol33, nthash.hpp:76-85) */
static inline uint64_t sroln(uint64_t x, unsigned n)
{
    uint64_t hi = x >> 33, lo = x & 0x1FFFFFFFFULL;
    unsigned a = n % 31u, b = n % 33u;
    hi = ((hi << a) | (hi >> (31u - a))) & 0x7FFFFFFFULL;
    lo = ((lo << b) | (lo >> (33u - b))) & 0x1FFFFFFFFULL;
    return (hi << 33) | lo;
}

uint64_t gpo_ntf64(const char* kmer, unsigned k)
{
    uint64_t h = 0;
    for (unsigned i = 0; i < k; i++)
        h = srol1(h) ^ gpo_seed((unsigned char)kmer[i]);
    return h;
}

uint64_t gpo_ntr64(const char* kmer, unsigned k)
{
    uint64_t h = 0;
    for (unsigned i = 0; i < k; i++)
        h = srol1(h) ^ rc_seed((unsigned char)kmer[k - 1 - i]);
    return h;
}

uint64_t gpo_ntf64_roll(uint64_t fh, unsigned k, unsigned char out, unsigned char in)
{
    return srol1(fh) ^ gpo_seed(in) ^ sroln(gpo_seed(out), k);
}

uint64_t gpo_ntr64_roll(uint64_t rh, unsigned k, unsigned char out, unsigned char in)
{
    return sror1(rh ^ sroln(rc_seed(in), k) ^ rc_seed(out));
}

uint64_t gpo_ntf64_changelast(uint64_t fh, unsigned k, unsigned char out, unsigned char in)
{
    (void)k;
    return fh ^ gpo_seed(out) ^ gpo_seed(in);
}

uint64_t gpo_ntr64_changelast(uint64_t rh, unsigned k, unsigned char out, unsigned char in)
{
    uint64_t h = srol1(rh);
    h ^= sroln(rc_seed(out), k);
    h ^= sroln(rc_seed(in), k);
    return sror1(h);
}

void gpo_extend(uint64_t base, unsigned k, unsigned h, uint64_t* out)
{
    if (h == 0)
        return;
    out[0] = base;
    for (unsigned i = 1; i < h; i++) {
        uint64_t t = base * ((uint64_t)i ^ ((uint64_t)k * MULTISEED));
        t ^= t >> MULTISHIFT;
        out[i] = t;
    }
}

size_t gpo_nthash_all(const char* seq, size_t len, unsigned k, uint64_t* out_hash,
                      uint8_t* out_valid)
{
    if (k == 0 || len < k)
        return 0;
    size_t n = len - k + 1;
    uint64_t fh = gpo_ntf64(seq, k), rh = gpo_ntr64(seq, k);
    /* bad = number of seed-0 bytes inside the current window */
    unsigned bad = 0;
    for (unsigned i = 0; i < k; i++)
        bad += gpo_seed((unsigned char)seq[i]) == 0;
    for (size_t p = 0;; p++) {
        out_hash[p] = fh + rh;
        out_valid[p] = (bad == 0);
        if (p + 1 == n)
            break;
        unsigned char o = (unsigned char)seq[p], in = (unsigned char)seq[p + k];
        fh = gpo_ntf64_roll(fh, k, o, in);
        rh = gpo_ntr64_roll(rh, k, o, in);
        bad -= gpo_seed(o) == 0;
        bad += gpo_seed(in) == 0;
    }
    return n;
}
