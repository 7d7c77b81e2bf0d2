// gp_device.h -- device-side primitives shared by the stage kernels (gfx950).
//
// ntHash (the hash the reference pipeline agrees on end to end; arithmetic
// documented in /root/reference/subprojects/ntedit/lib/nthash.hpp:24-28,66-91,
// 100-152,180-191,305-314): 64-bit words whose low 33 and high 31 bits rotate
// independently, canonical value = forward + reverse, extra hashes by a
// multiply / xor-shift of the canonical value.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gp {

constexpr uint64_t kSeedA = 0x3c8bfbb395c60474ULL;
constexpr uint64_t kSeedC = 0x3193c18562a02b4cULL;
constexpr uint64_t kSeedG = 0x20323ed082572324ULL;
constexpr uint64_t kSeedT = 0x295549f54be24456ULL;
constexpr uint64_t kMultiSeed = 0x90b45d39fb6da1faULL;
constexpr int kMultiShift = 27;
constexpr int kMaxHash = 8;
constexpr uint32_t kNoResv = 0xFFFFFFFFu;

// 2-bit code used inside kernels: A0 C1 T2 G3 (= (c >> 1) & 3 for ACGTacgt), so
// the complement is code ^ 2.  kBadCode marks every other byte.
constexpr uint8_t kBadCode = 4;

__host__ __device__ inline uint8_t base_code(uint8_t c)
{
    uint8_t u = c & 0xDF;
    bool ok = (u == 'A') | (u == 'C') | (u == 'G') | (u == 'T');
    return ok ? (uint8_t)((c >> 1) & 3) : kBadCode;
}

__host__ __device__ inline uint64_t seed_of_code(uint32_t code)
{
    // branch-free select chain; code in 0..3
    uint64_t lo = (code & 1) ? kSeedC : kSeedA;
    uint64_t hi = (code & 1) ? kSeedG : kSeedT;
    return (code & 2) ? hi : lo;
}

// forward seed of a raw byte (0 for anything but ACGTacgt)
__host__ __device__ inline uint64_t seed_fwd(uint8_t c)
{
    uint8_t code = base_code(c);
    return code == kBadCode ? 0ULL : seed_of_code(code);
}

// reverse-strand seed of a raw byte: the reference indexes its table with the
// low three bits of the byte (nthash.hpp:14,115), which gives the complement
// for ACGTacgt and a few non-zero values for IUPAC letters; reproduced as is.
__host__ __device__ inline uint64_t seed_rc(uint8_t c)
{
    switch (c & 7) {
    case 1: return kSeedT;
    case 3: return kSeedG;
    case 4: return kSeedA;
    case 7: return kSeedC;
    default: return 0ULL;
    }
}

__host__ __device__ inline uint64_t srol1(uint64_t x)
{
    uint64_t carry = ((x & 0x8000000000000000ULL) >> 30) | ((x & 0x100000000ULL) >> 32);
    return ((x << 1) & 0xFFFFFFFDFFFFFFFFULL) | carry;
}

__host__ __device__ inline uint64_t sror1(uint64_t x)
{
    uint64_t carry = ((x & 1ULL) << 32) | ((x & 0x200000000ULL) << 30);
    return ((x >> 1) & 0xFFFFFFFEFFFFFFFFULL) | carry;
}

__host__ __device__ inline uint64_t sroln(uint64_t x, uint32_t n)
{
    uint64_t hi = x >> 33, lo = x & 0x1FFFFFFFFULL;
    uint32_t a = n % 31u, b = n % 33u;
    hi = ((hi << a) | (hi >> (31u - a))) & 0x7FFFFFFFULL;
    lo = ((lo << b) | (lo >> (33u - b))) & 0x1FFFFFFFFULL;
    return (hi << 33) | lo;
}

// i-th extension hash of canonical value `base` for k-mer size k (i >= 1)
#ifdef __HIPCC__
// Workgroups go round the eight XCDs by blockIdx, each XCD with an L2 of its own.  For a grid whose
// neighbouring units read the same data (tiles of one contig probing that contig's 512 KiB Bloom
// filter), unit = xcd_contiguous(blockIdx.x, gridDim.x) gives every XCD one contiguous eighth of
// the units, so that a filter is fetched into one L2, not into eight.
__device__ inline uint32_t xcd_contiguous(uint32_t i, uint32_t n)
{
    const uint32_t per = n / 8u;
    return i < per * 8u ? (i % 8u) * per + i / 8u : i;
}
#endif

// Seeds rotated in advance: T[j * 8 + code] = srol^j(seed[code]) for j < 64 (k <= 64), zero in the
// columns of kBadCode and its "complement" kBadCode ^ 2.  srol is a permutation of bits, so the
// hash of a k-mer from scratch -- k rotate-select-xor steps per strand, a few hundred
// instructions of every stage-A kernel per operation -- is the XOR of k table entries per strand:
//   forward = XOR_i srol^(k-1-i)(seed[c_i]),  reverse = XOR_i srol^i(seed[c_i ^ 2]).
constexpr int kSeedTableWords = 64 * 8;

#ifdef __HIPCC__
// every lane of the workgroup calls it; a workgroup barrier must follow before the first use
__device__ inline void fill_seed_table(uint64_t* T)
{
    for (int i = (int)threadIdx.x; i < kSeedTableWords; i += (int)blockDim.x) {
        const uint32_t code = (uint32_t)i & 7u;
        T[i] = code < 4 ? sroln(seed_of_code(code), (uint32_t)i >> 3) : 0ULL;
    }
}

// both strands of the k-mer at codes[0..k), and the index of its last byte that is not ACGT (-1: none)
__device__ inline void kmer_hash_from_table(const uint64_t* T, const uint8_t* codes, uint32_t k, uint64_t& fh,
                                            uint64_t& rh, int& last_bad)
{
    uint64_t f = 0, r = 0;
    int lb = -1;
    // four bases at a time: their codes in one LDS round trip, their eight table entries in another
    // (one base after the other the loop was a chain of 3 k dependent LDS look-ups)
    uint32_t i = 0;
    for (; i + 4 <= k; i += 4) {
        uint32_t c[4];
#pragma unroll
        for (int u = 0; u < 4; u++)
            c[u] = codes[i + u];
        uint64_t tf[4], tr[4];
#pragma unroll
        for (int u = 0; u < 4; u++) {
            tf[u] = T[(k - 1 - i - u) * 8 + c[u]];
            tr[u] = T[(i + u) * 8 + (c[u] ^ 2u)];
        }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            f ^= tf[u];
            r ^= tr[u];
            if (c[u] >= 4)
                lb = (int)(i + u);
        }
    }
    for (; i < k; i++) {
        const uint32_t c = codes[i];
        f ^= T[(k - 1 - i) * 8 + c];
        r ^= T[i * 8 + (c ^ 2u)];
        if (c >= 4)
            lb = (int)i;
    }
    fh = f;
    rh = r;
    last_bad = lb;
}
#endif

__host__ __device__ inline uint64_t ext_hash(uint64_t base, uint32_t k, uint32_t i)
{
    uint64_t t = base * ((uint64_t)i ^ ((uint64_t)k * kMultiSeed));
    return t ^ (t >> kMultiShift);
}

// x % d for a divisor fixed per launch.  d = c * 2^s with c odd: the low s bits
// pass through and the rest is reduced modulo c with 32-bit Lemire reductions
// (exact for c < 65536); other divisors fall back to the 64-bit remainder.
struct FastMod {
    uint64_t d;
    uint64_t magic;   // floor((2^64-1)/c) + 1
    uint32_t c;       // odd part
    uint32_t s;       // power-of-two part
    uint32_t two32;   // 2^32 mod c
    uint32_t fast;    // 1: use the split path
};

inline FastMod make_fastmod(uint64_t d)
{
    FastMod m{};
    m.d = d;
    uint64_t c = d;
    uint32_t s = 0;
    while (c && (c & 1) == 0) {
        c >>= 1;
        s++;
    }
    m.s = s;
    m.c = (uint32_t)c;
    m.fast = (c >= 1 && c < 65536) ? 1u : 0u;
    if (m.fast) {
        m.magic = c == 1 ? 0 : (0xFFFFFFFFFFFFFFFFULL / c + 1);
        m.two32 = (uint32_t)((1ULL << 32) % c);
    }
    return m;
}

__device__ inline uint32_t mod32(uint32_t a, const FastMod& m)
{
    uint64_t low = m.magic * (uint64_t)a;
    return (uint32_t)__umul64hi(low, (uint64_t)m.c);
}

// x % (5 * 2^s): GoldPolish's counting filter is 10 MiB = 5 * 2^21.  2^32, 2^16 and 2^8 are all 1 modulo 5,
// so x >> s folds to a number below 1024 with adds and shifts, and t / 5 == (t * 205) >> 10 there.
__device__ inline uint32_t mod_5_pow2(uint64_t x, uint32_t s)
{
    const uint64_t q = x >> s;
    const uint32_t lo = (uint32_t)q, hi = (uint32_t)(q >> 32);
    const uint32_t t2 = (lo & 0xFFFFu) + (lo >> 16) + (hi & 0xFFFFu) + (hi >> 16);   // < 2^18: 32-bit adds of 16-bit halves
    const uint32_t t3 = (t2 & 0xFFu) + ((t2 >> 8) & 0xFFu) + (t2 >> 16);            // < 768
    const uint32_t r5 = t3 - 5u * ((t3 * 205u) >> 10);
    return (r5 << s) | ((uint32_t)x & ((1u << s) - 1u));
}

// The extension hashes 1..3 of one k-mer from ONE 64-bit multiply: their multipliers i ^ (k * seed)
// differ from M = k * seed in the two lowest bits only, i.e. by a small signed d_i, and
// base * (M + d_i) = base * M + d_i * base.  d[i] = ((M & 3) ^ i) - (M & 3), uniform per launch.
struct ExtHash3 {
    int d[4];
};
__host__ __device__ inline ExtHash3 make_ext_hash3(uint32_t k)
{
    ExtHash3 e{};
    const int m2 = (int)(((uint64_t)k * kMultiSeed) & 3u);
    for (int i = 0; i < 4; i++)
        e.d[i] = (m2 ^ i) - m2;
    return e;
}
// d * base for d in [-3, 3] (uniform): what the multiplier of extension hash i adds to base * M
__device__ inline uint64_t ext_addend(uint64_t base, int d)
{
    const uint64_t m = d < 0 ? (uint64_t)(-d) : (uint64_t)d;
    const uint64_t v = (m & 1 ? base : 0) + (m & 2 ? base << 1 : 0);
    return d < 0 ? 0 - v : v;
}

__device__ inline uint64_t fastmod(uint64_t x, const FastMod& m)
{
    if (!m.fast)
        return x % m.d;
    uint64_t low = x & ((1ULL << m.s) - 1);
    if (m.c == 1)
        return low;
    uint64_t q = x >> m.s;
    if (m.c == 5) {
        // GoldPolish's counting filter: 10 MiB = 5 * 2^21.  2^32, 2^16 and 2^8 are all 1 modulo 5, so
        // q folds to a number below 1024 with adds and shifts, and t / 5 == (t * 205) >> 10 there:
        // a dozen full-rate instructions instead of the eight quarter-rate multiplies of the general
        // reduction below -- per counter touch, four times per operation of every stage-A kernel.
        const uint64_t t64 = (q & 0xFFFFFFFFULL) + (q >> 32);                   // < 2^33
        const uint32_t t2 = ((uint32_t)t64 & 0xFFFFu) + (uint32_t)(t64 >> 16);  // < 2^16 + 2^17
        const uint32_t t3 = (t2 & 0xFFu) + (t2 >> 8);                           // < 1023
        const uint32_t r5 = t3 - 5u * ((t3 * 205u) >> 10);
        return ((uint64_t)r5 << m.s) | low;
    }
    if (m.s < 16) {
        // q may have more than 48 bits: fold its top half first (an exact 32-bit reduction); what is
        // left is below c * 2^32 < 2^48
        const uint32_t r1 = mod32((uint32_t)(q >> 32), m);
        q = ((uint64_t)r1 << 32) | (uint32_t)q;
    }
    // One step of Lemire's reduction with the 64-bit magic is exact for q < 2^48 and c < 2^16
    // (precision needed: 48 + 16 bits).  GoldPolish's 10 MiB = 5 * 2^21 takes this path alone.
    const uint64_t r = __umul64hi(m.magic * q, (uint64_t)m.c);
    return (r << m.s) | low;
}

// first lane of the active set appends `n_active` items with one atomic
__device__ inline uint32_t wave_append(uint32_t* counter)
{
    uint64_t mask = __ballot(1);
    int lane = (int)(threadIdx.x & 63);
    uint32_t leader = (uint32_t)__ffsll((long long)mask) - 1;
    uint32_t base = 0;
    if ((uint32_t)lane == leader)
        base = atomicAdd(counter, (uint32_t)__popcll(mask));
    base = __shfl(base, (int)leader);
    uint32_t rank = (uint32_t)__popcll(mask & ((1ULL << lane) - 1));
    return base + rank;
}

} // namespace gp
