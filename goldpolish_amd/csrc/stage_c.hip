// stage_c.hip -- stage C of the path on MI355X: Sealer gap filling, i.e.
// Konnector's constrained bidirectional BFS through the Bloom-filter de Bruijn
// graph between the flanks of every soft-masked run, for each k in turn.
//
// What it replaces (see oracle/sealer.c for the line-by-line map):
//   /root/reference/subprojects/sealer/sealer.cpp findFlanks :695-733,
//       makePseudoReads :560-577, kRun :579-685, merge :425-517,
//       insertIntoScaffold :525-558
//   Konnector/konnector.h connectPairs :234-438
//   Konnector/DBGBloomAlgorithms.h getStartKmerPos :51-108
//   Graph/BidirectionalBFS.h, Graph/ConstrainedBidiBFSVisitor.h,
//   Graph/AllPathsSearch.h, Graph/HashGraph.h, Align/alignGlobal.cc
//
// Shape of the work: a search is a pointer-chasing, latency-bound walk (hash
// table of visited k-mers, 4 membership tests per vertex), searches are
// independent of each other, and there are tens of thousands of them.  So:
//   * find_gaps / prep_gaps: one workgroup per contig finds the gap runs and
//     their flanks in parallel, and an exact hash set (keys compared by the
//     flank bytes themselves) reproduces the reference's map semantics:
//     identical flank pairs collapse to the last gap, and a left flank whose
//     gap closed leaves the pool for the smaller k.
//   * connect: persistent workers (one wave each, every wave drains a shared
//     queue and exits when it is empty) run one search at a time in a private
//     HBM arena; neighbour k-mers are hashed incrementally in O(1) from the
//     parent's forward/reverse ntHash instead of re-hashing k bases.
//   * consensus: the rare multi-path gaps are folded with the affine-gap
//     aligner in a second queue.
//   * apply_fills: per contig, fills are spliced right to left exactly as
//     std::string::replace would, using a right-aligned gap buffer.
#include "gp_bc.h"

#include <algorithm>
#include <cstring>
#include <ctime>

namespace gp {

constexpr uint32_t kNoLimit = 0xFFFFFFFFu;
constexpr uint32_t kNoMatch = 0xFFFFFFFFu;
constexpr int kSlotBits = 18;
constexpr uint32_t kSlots = 1u << kSlotBits;
constexpr uint32_t kVertCap = 1u << 17;     // <= max_cost + 2 vertices per search
constexpr uint32_t kEdgeCap = 2u * kVertCap + 64;
constexpr uint32_t kMaxPathsDev = 16;
constexpr uint32_t kPathCap = 1024;          // vertices per stored path
// lane-per-gap first attempt: a small private arena per lane; what outgrows it, or finds paths
// to enumerate, is done again by the one-gap-per-wave workers with their big arenas
constexpr uint32_t kSmallSlotBits = 8, kSmallVerts = 96, kSmallEdges = 256;
constexpr uint32_t kSimtWaves = 4096;   // waves of the lane-per-gap kernel (four per SIMD: compiled for 128 registers it spills 13 and is still faster than three) ...
constexpr uint32_t kSimtLanes = 64;     // ... arenas per wave; CArgs::simt_lanes of them take a gap (GP_SEALER_LANES, default 32: divergent walks serialise)
constexpr uint32_t kNwMax = 1800;            // longest sequence the consensus aligner takes
constexpr uint32_t kMaxWorkers = 4096;  // one-wave workers of sealer_connect (CState::n_workers: by job size)
// Their arenas come in two sizes.  Nearly every search a worker takes stays below a few thousand
// vertices (limit 1000 open branches), so the 4096 workers of the first pass hold 8192 vertices each
// (1 MB with edges, slots, queues and path scratch: 3 GiB), and what outgrows that -- the cost cap of
// 100 000 allows 131 072 vertices -- is done again by at most 256 workers with full arenas (13 MB
// each: 3.3 GiB).  Round 2 gave all 3072 workers full arenas: 39 GiB for a stage of 0.15 s.
constexpr uint32_t kMedSlotBits = 14, kMedVertCap = 1u << 13, kMedEdgeCap = 2u * kMedVertCap + 64;
constexpr uint32_t kBigWorkers = 256;
constexpr int kNwWorkers = 16;

enum { FWD = 0, REV = 1 };
enum { R_FOUND = 0, R_TOO_MANY_PATHS, R_TOO_MANY_BRANCHES, R_CYCLE, R_MAX_COST, R_MEM, R_NO_PATH, R_DEFER, R_SIMPLE };
enum { V_SUCCESS = 0, V_SKIP, V_ABORT };

struct Gap {
    uint32_t s, e;      // gap run [s, e)
    uint32_t contig;
    uint32_t pair_rep;  // global index of the first gap with the same (left, right) flanks
    uint32_t left_rep;  // ... with the same left flank
};

struct Vert {
    uint64_t km[2];     // 2-bit packed k-mer, last base in the low bits (A0 C1 G2 T3)
    uint64_t fh, rh;
    uint32_t head[2], tail[2]; // successor lists of the two traversal-history graphs
    uint16_t depth[2];
    uint8_t color[2];
    uint8_t pad[2];
    uint32_t stamp_visit, stamp_cyc;
};

struct Edge {
    uint32_t to, next;
};

struct MultiHdr {
    uint32_t gap, sp, gp, n_paths;
    uint32_t len[kMaxPathsDev];
    uint64_t off[kMaxPathsDev]; // into the pool
};

struct CState {
    DevBuf<Gap> gaps;
    DevBuf<uint32_t> gap_off;      // [n_contigs+1] slice of `gaps` per contig
    DevBuf<uint32_t> n_gaps;       // [n_contigs]
    DevBuf<uint32_t> pair_last;    // [G] per pair representative: last gap + 1
    DevBuf<uint32_t> closed_at;    // [G] per left representative: first k index that closed it
    DevBuf<uint64_t> fill_off;     // [G]
    DevBuf<uint32_t> fill_len;     // [G] 0 = not closed
    DevBuf<uint32_t> table_left, table_pair; // exact hash sets, value = gap index + 1
    DevBuf<uint32_t> work;         // valid gaps
    DevBuf<uint32_t> defer;        // gaps left for the one-per-wave workers
    DevBuf<uint8_t> small;         // kSimtWaves x kSimtLanes lane arenas (SmallArena)
    DevBuf<uint32_t> queue;        // [0] work count [1] head [2] multi count [3] multi head
    DevBuf<uint8_t> pool;          // fills and connecting sequences
    DevBuf<unsigned long long> pool_head;
    DevBuf<MultiHdr> multi;
    // worker arenas
    DevBuf<uint32_t> slots;
    DevBuf<Vert> verts;
    DevBuf<Edge> edges;
    DevBuf<uint32_t> scratch;      // queues, stacks, paths (the full arenas of kBigWorkers workers)
    DevBuf<uint32_t> m_slots, m_scratch, m_epoch; // the medium arenas of n_workers workers
    DevBuf<Vert> m_verts;
    DevBuf<Edge> m_edges;
    DevBuf<uint32_t> defer2;       // gaps that outgrew a medium arena
    DevBuf<int32_t> nw;
    DevBuf<uint32_t> epoch;        // per worker
    std::vector<uint32_t> h_gap_off;
    uint32_t gap_cap = 0, table_bits = 0, multi_cap = 0;
    uint32_t lanes = 32;           // lanes of a sealer_connect_lanes wave that take a gap (the small arenas are allocated for that many)
    uint32_t n_big = 0;            // workers with full arenas (all of them where long contigs are polished, else kBigWorkers)
    uint32_t n_workers = 0;        // sealer_connect's grid: 1024 took 0.15 s per configs[2] step, 2048 0.10 s, 3072 (three waves per SIMD) 0.08 s, 4096 (four: the kernel fits 128 registers without a spill) 12 ms less
    uint64_t pool_cap = 0;
};

constexpr uint32_t kPathScratchWords = 4 * kPathCap                   // DFS path + iterator stacks
                                       + 3 * kMaxPathsDev * kPathCap; // left, right, found paths
constexpr uint32_t kScratchWords = 2 * kVertCap + kPathScratchWords;       // two BFS queues + path scratch (full arena)
constexpr uint32_t kMedScratchWords = 2 * kMedVertCap + kPathScratchWords; // (medium arena)

struct CArgs {
    const uint8_t* seq;      // masked sequences (stage input)
    uint8_t* out;            // spliced sequences
    const uint64_t* off;
    const uint32_t* len_cur;
    uint32_t* len_next;
    const uint8_t* dropped;
    const uint32_t* contig_batch;
    const uint32_t* bf;
    uint32_t n_contigs, n_k, ki, k, hash_num, bf_words;
    FastMod bf_mod;
    uint32_t flank_len, max_gap, max_paths, max_branches, max_cost;
    Gap* gaps;
    const uint32_t* gap_off;
    uint32_t* n_gaps;
    uint32_t* pair_last;
    uint32_t* closed_at;
    uint64_t* fill_off;
    uint32_t* fill_len;
    uint32_t* table_left;
    uint32_t* table_pair;
    uint32_t table_bits;
    uint32_t* work;
    uint32_t* defer;     // gaps phase 1 (lane per gap) left for phase 2, count in queue[4]
    uint32_t* queue;
    uint8_t* pool;
    unsigned long long* pool_head;
    uint64_t pool_cap;
    MultiHdr* multi;
    uint32_t multi_cap;
    uint32_t* slots;
    Vert* verts;
    Edge* edges;
    uint32_t* scratch;
    uint32_t* m_slots;   // medium arenas (first pass of sealer_connect), see kMedVertCap
    Vert* m_verts;
    Edge* m_edges;
    uint32_t* m_scratch;
    uint32_t* m_epoch;
    uint32_t* defer2;    // gaps the medium arenas could not hold, count in queue[6]
    uint32_t simt_lanes; // sealer_connect_lanes: lanes of a wave that take a gap (<= kSimtLanes)
    uint32_t tier;       // sealer_connect: 0 medium arenas (what overflows goes to defer2), 1 full arenas over defer2, 2 full arenas over defer (no medium pass)
    int32_t* nw;
    uint32_t* epoch;
    uint32_t* err;
    unsigned long long* stats; // [4] gaps [5] closed [6] vertex tests [7] no start/goal [8] no path [9] unique
                               // [10] multiple [11] too many paths [12] too many branches [13] max cost
    const uint32_t* flags;     // classify_kernel: bit 1 = the contig holds a gap character
};

__device__ inline bool is_gap_char(uint8_t c)
{
    switch (c) {
    case 'N': case 'n': case 'a': case 't': case 'c': case 'g': case 'u': case 'r': case 'y':
    case 'k': case 'm': case 's': case 'w': case 'b': case 'd': case 'h': case 'v': case 'x':
        return true;
    default:
        return false;
    }
}

__device__ inline uint8_t up(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }

__device__ inline uint8_t comp_base(uint8_t c)
{
    uint8_t r;
    switch (up(c)) {
    case 'A': r = 'T'; break;
    case 'C': r = 'G'; break;
    case 'G': r = 'C'; break;
    case 'T': r = 'A'; break;
    case 'M': r = 'K'; break;
    case 'R': r = 'Y'; break;
    case 'W': r = 'W'; break;
    case 'S': r = 'S'; break;
    case 'Y': r = 'R'; break;
    case 'K': r = 'M'; break;
    case 'V': r = 'B'; break;
    case 'H': r = 'D'; break;
    case 'D': r = 'H'; break;
    case 'B': r = 'V'; break;
    default: r = 'N'; break;
    }
    return (c >= 'a' && c <= 'z') ? (uint8_t)(r + 32) : r;
}

// ------------------------------------------------------------ gap discovery
constexpr int kGapThreads = 256;

__global__ __launch_bounds__(kGapThreads) void find_gaps(CArgs a)
{
    __shared__ uint32_t cnt[kGapThreads + 1];
    const uint32_t cid = blockIdx.x, tid = threadIdx.x;
    if (a.dropped[cid]) {
        if (tid == 0)
            a.n_gaps[cid] = 0;
        return;
    }
    const uint32_t len = a.len_cur[cid];
    const uint8_t* s = a.seq + a.off[cid];
    if (!(a.flags[cid] & 2u)) {
        // no gap character anywhere (classify_kernel): nothing to seal
        if (tid == 0)
            a.n_gaps[cid] = 0;
        return;
    }
    const uint32_t chunk = (len + kGapThreads - 1) / kGapThreads;
    const uint32_t b = min(len, tid * chunk), e = min(len, b + chunk);
    uint32_t n = 0;
    for (uint32_t i = b; i < e; i++)
        if (is_gap_char(s[i]) && (i == 0 || !is_gap_char(s[i - 1])))
            n++;
    cnt[tid] = n;
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (int t = 0; t < kGapThreads; t++) {
            uint32_t v = cnt[t];
            cnt[t] = acc;
            acc += v;
        }
        cnt[kGapThreads] = acc;
        const uint32_t cap = a.gap_off[cid + 1] - a.gap_off[cid];
        if (acc > cap) {
            atomicOr(a.err, 8u);
            acc = cap;
        }
        a.n_gaps[cid] = acc;
        atomicAdd(&a.stats[4], (unsigned long long)cnt[kGapThreads]);
    }
    __syncthreads();
    const uint32_t cap = a.gap_off[cid + 1] - a.gap_off[cid];
    Gap* g = a.gaps + a.gap_off[cid];
    uint32_t at = cnt[tid];
    for (uint32_t i = b; i < e; i++)
        if (is_gap_char(s[i]) && (i == 0 || !is_gap_char(s[i - 1]))) {
            uint32_t j = i + 1;
            while (j < len && is_gap_char(s[j]))
                j++;
            if (at < cap) {
                g[at].s = i;
                g[at].e = j;
                g[at].contig = cid;
            }
            at++;
        }
}

struct Flank {
    uint32_t ls, le, rs, re;
    bool valid;
};

__device__ inline Flank flank_of(const CArgs& a, uint32_t gi)
{
    const Gap g = a.gaps[gi];
    const uint32_t base = a.gap_off[g.contig], n = a.n_gaps[g.contig];
    const uint32_t local = gi - base;
    const uint32_t len = a.len_cur[g.contig];
    Flank f;
    f.valid = g.e < len; // a sequence that ends in a gap leaves it alone (:711-714)
    const uint32_t prev_e = local > 0 ? a.gaps[gi - 1].e : 0u;
    const uint32_t next_s = local + 1 < n ? a.gaps[gi + 1].s : len;
    const long ls = (long)g.s - (long)a.flank_len;
    f.ls = (uint32_t)max((long)prev_e, ls);
    f.le = g.s;
    f.rs = g.e;
    f.re = min(next_s, g.e + a.flank_len);
    return f;
}

__device__ inline uint64_t fnv_upper(const uint8_t* s, uint32_t n, uint64_t h)
{
    for (uint32_t i = 0; i < n; i++) {
        h ^= up(s[i]);
        h *= 1099511628211ULL;
    }
    return h;
}

__device__ inline bool same_upper(const uint8_t* a, uint32_t la, const uint8_t* b, uint32_t lb)
{
    if (la != lb)
        return false;
    for (uint32_t i = 0; i < la; i++)
        if (up(a[i]) != up(b[i]))
            return false;
    return true;
}

// exact set: returns the representative (first claimant) of my key
__device__ uint32_t find_or_insert(const CArgs& a, uint32_t* table, uint64_t h, uint32_t me,
                                   bool pair)
{
    const uint32_t mask = (1u << a.table_bits) - 1;
    uint32_t slot = (uint32_t)(h >> 20) & mask;
    const Gap gm = a.gaps[me];
    const Flank fm = flank_of(a, me);
    const uint8_t* sm = a.seq + a.off[gm.contig];
    for (uint32_t probe = 0; probe <= mask; probe++, slot = (slot + 1) & mask) {
        uint32_t v = atomicCAS(&table[slot], 0u, me + 1);
        if (v == 0)
            return me;
        const uint32_t other = v - 1;
        if (other == me)
            return me;
        const Gap go = a.gaps[other];
        if (go.contig != gm.contig)
            continue;
        const Flank fo = flank_of(a, other);
        if (!same_upper(sm + fm.ls, fm.le - fm.ls, sm + fo.ls, fo.le - fo.ls))
            continue;
        if (pair && !same_upper(sm + fm.rs, fm.re - fm.rs, sm + fo.rs, fo.re - fo.rs))
            continue;
        return other;
    }
    atomicOr(a.err, 16u);
    return me;
}

__global__ void prep_gaps(CArgs a)
{
    const uint32_t cid = blockIdx.x;
    if (a.dropped[cid])
        return;
    const uint32_t n = a.n_gaps[cid], base = a.gap_off[cid];
    const uint8_t* s = a.seq + a.off[cid];
    for (uint32_t i = threadIdx.x; i < n; i += blockDim.x) {
        const uint32_t gi = base + i;
        const Flank f = flank_of(a, gi);
        a.fill_len[gi] = 0;
        a.closed_at[gi] = 0xFFu;
        if (!f.valid) {
            a.gaps[gi].pair_rep = a.gaps[gi].left_rep = gi;
            continue;
        }
        uint64_t hl = fnv_upper(s + f.ls, f.le - f.ls, 1469598103934665603ULL ^ ((uint64_t)cid << 32));
        uint64_t hp = fnv_upper(s + f.rs, f.re - f.rs, hl ^ 0x9E3779B97F4A7C15ULL);
        const uint32_t lr = find_or_insert(a, a.table_left, hl, gi, false);
        const uint32_t pr = find_or_insert(a, a.table_pair, hp, gi, true);
        a.gaps[gi].left_rep = lr;
        a.gaps[gi].pair_rep = pr;
        atomicMax(&a.pair_last[pr], gi + 1);
        a.work[atomicAdd(&a.queue[0], 1u)] = gi;
    }
}

// ------------------------------------------------------------------ search
struct SeedTab {
    uint64_t s[4], sk[4], sk1[4]; // seed, rotated by k, rotated by k-1; index A0 C1 G2 T3
};

struct Search {
    // (what bf_has needs of the launch's arguments, by value: a pointer to the kernel's argument block
    // here forced every lane to keep a copy of the whole block in scratch)
    uint32_t hash_num;
    FastMod bf_mod;
    const uint32_t* bf;
    uint32_t k;
    SeedTab st;
    uint32_t* slots;
    uint32_t epoch;
    uint32_t slot_bits, vert_cap, edge_cap; // of the arena in use (a worker's big one or a lane's small one)
    bool small;                              // lane-per-gap attempt: no room for path enumeration
    bool retry_on_overflow;                  // a larger arena takes the gap if this one runs out: whatever else was flagged is not final
    Vert* v;
    Edge* e;
    uint32_t nv, ne;
    uint32_t* q[2];
    uint32_t qh[2], qt[2];
    uint32_t* dfs_path;
    uint32_t* dfs_it;
    uint32_t* left;  // kMaxPathsDev x kPathCap
    uint32_t* right;
    uint32_t* found;
    uint32_t left_n, right_n, found_n;
    uint32_t left_len[kMaxPathsDev], right_len[kMaxPathsDev], found_len[kMaxPathsDev];
    uint32_t start, goal;
    uint32_t max_paths, max_branches, max_cost, cost;
    uint32_t min_len, max_len, max_depth[2];
    bool too_many_branches, max_cost_exceeded, too_many_paths, overflow;
    uint32_t ce_u[kMaxPathsDev + 2], ce_v[kMaxPathsDev + 2], n_ce;
    uint32_t stamp;
    unsigned long long tests;
};

typedef unsigned __int128 u128;

__device__ inline u128 km_load(const uint64_t km[2]) { return ((u128)km[1] << 64) | km[0]; }

__device__ inline void km_store(u128 x, uint64_t km[2])
{
    km[0] = (uint64_t)x;
    km[1] = (uint64_t)(x >> 64);
}

__device__ inline uint32_t code_acgt(uint8_t c)
{
    switch (c) {
    case 'A': return 0;
    case 'C': return 1;
    case 'G': return 2;
    case 'T': return 3;
    default: return 4;
    }
}

__device__ inline bool bf_has(Search& S, uint64_t base)
{
    S.tests++;
    bool all = true;
    for (uint32_t i = 0; i < S.hash_num; i++) {
        uint64_t h = i ? ext_hash(base, S.k, i) : base;
        uint64_t n = fastmod(h, S.bf_mod);
        all &= ((S.bf[n >> 5] >> (uint32_t)(n & 31)) & 1u) != 0;
    }
    return all;
}

// hash + pack an upper-case ACGT k-mer; false if it holds another byte
__device__ bool kmer_of(const Search& S, const uint8_t* s, uint64_t km[2], uint64_t& fh,
                        uint64_t& rh)
{
    const uint32_t k = S.k;
    uint64_t f = 0, r = 0;
    u128 x = 0;
    for (uint32_t i = 0; i < k; i++) {
        uint32_t c = code_acgt(s[i]);
        uint32_t cr = code_acgt(s[k - 1 - i]);
        if (c > 3 || cr > 3)
            return false;
        f = srol1(f) ^ S.st.s[c];
        r = srol1(r) ^ S.st.s[3 - cr];
        x = (x << 2) | c;
    }
    km_store(x, km);
    fh = f;
    rh = r;
    return true;
}

// hashes of a vertex's neighbours in O(1) from its own: the successor drops the k-mer's first base and
// appends b, the predecessor drops its last base and puts b in front (codes A0 C1 G2 T3)
__device__ inline void succ_hash(const SeedTab& st, uint64_t pfh, uint64_t prh, uint32_t first, uint32_t b,
                                 uint64_t& fh, uint64_t& rh)
{
    fh = srol1(pfh) ^ st.s[b] ^ st.sk[first];
    rh = sror1(prh ^ st.sk[3 - b] ^ st.s[3 - first]);
}

__device__ inline void pred_hash(const SeedTab& st, uint64_t pfh, uint64_t prh, uint32_t last, uint32_t b,
                                 uint64_t& fh, uint64_t& rh)
{
    fh = sror1(pfh ^ st.s[last]) ^ st.sk1[b];
    rh = srol1(prh ^ st.sk1[3 - last]) ^ st.s[3 - b];
}

__device__ inline void seed_tab_init(SeedTab& st, uint32_t k)
{
    const uint64_t seeds[4] = { kSeedA, kSeedC, kSeedG, kSeedT };
    for (int c = 0; c < 4; c++) {
        st.s[c] = seeds[c];
        st.sk[c] = sroln(seeds[c], k);
        st.sk1[c] = sroln(seeds[c], k - 1);
    }
}

// gp_debug_nthash, variant 3: stage C's hashing over one upper-case sequence.  Lane i of a run of four
// positions: the first from scratch (kmer_of), the next three as successors of the one before
// (succ_hash: the forward search's step); and, as a check of the backward search's step, every k-mer
// is ALSO derived as the predecessor of the k-mer one to its right (pred_hash) -- a position whose two
// derivations differ is reported invalid.
__global__ __launch_bounds__(256) void debug_nthash_c(const uint8_t* seq, uint64_t len, uint32_t k, uint32_t hash_num,
                                                      uint64_t* out_f, uint64_t* out_r, uint64_t* out_h, uint8_t* valid)
{
    __shared__ Search S; // (only k and the seed table are used)
    if (threadIdx.x == 0) {
        S.k = k;
        seed_tab_init(S.st, k);
    }
    __syncthreads();
    const uint64_t p0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    uint64_t fh = 0, rh = 0, km[2];
    bool run_ok = false;
    for (int j = 0; j < 4; j++) {
        const uint64_t p = p0 + j;
        if (p + k > len)
            break;
        bool ok;
        if (j == 0 || !run_ok) {
            ok = kmer_of(S, seq + p, km, fh, rh);
        } else {
            const uint32_t first = code_acgt(seq[p - 1]), b = code_acgt(seq[p + k - 1]);
            ok = b < 4;
            if (ok)
                succ_hash(S.st, fh, rh, first, b, fh, rh);
        }
        run_ok = ok;
        if (ok && p + k < len && code_acgt(seq[p + k]) < 4) {
            uint64_t km2[2], f2, r2, pf, pr;
            if (kmer_of(S, seq + p + 1, km2, f2, r2)) {
                pred_hash(S.st, f2, r2, code_acgt(seq[p + k]), code_acgt(seq[p]), pf, pr);
                ok = pf == fh && pr == rh;
                if (!ok) { // a real disagreement, not a non-ACGT byte: make it visible as a wrong value
                    out_f[p] = pf;
                    out_r[p] = pr;
                    valid[p] = 2;
                    continue;
                }
            }
        }
        out_f[p] = fh;
        out_r[p] = rh;
        valid[p] = ok ? 1 : 0;
        const uint64_t base = fh + rh;
        for (uint32_t h = 0; h < hash_num; h++)
            out_h[p * hash_num + h] = h ? ext_hash(base, k, h) : base;
    }
}

int debug_nthash_stage_c(hipStream_t s, const uint8_t* d_seq, uint64_t len, uint32_t k, uint32_t hash_num,
                         uint64_t* d_f, uint64_t* d_r, uint64_t* d_h, uint8_t* d_valid)
{
    const uint32_t grid = (uint32_t)((len + 1023) / 1024);
    hipLaunchKernelGGL(debug_nthash_c, dim3(grid), dim3(256), 0, s, d_seq, len, k, hash_num, d_f, d_r, d_h, d_valid);
    GP_HIP(hipGetLastError());
    return GP_OK;
}

// getStartKmerPos, FORWARD direction (both call sites), rolling from the flank end backwards
__device__ uint32_t start_kmer_pos(Search& S, const uint8_t* seq, uint32_t len, uint32_t need)
{
    const uint32_t k = S.k;
    if (len < k)
        return kNoMatch;
    uint32_t run = 0, best_len = 0, best_pos = 0;
    for (int pos = (int)(len - k); pos >= 0; pos--) {
        uint64_t km[2], fh, rh;
        bool ok = kmer_of(S, seq + pos, km, fh, rh);
        if (ok)
            ok = bf_has(S, fh + rh);
        else
            S.tests++;
        if (!ok) {
            if (run > best_len) {
                best_pos = (uint32_t)pos + 1;
                best_len = run;
            }
            run = 0;
        } else {
            run++;
            if (run >= need)
                return (uint32_t)pos;
        }
    }
    if (run > best_len) {
        best_pos = 0;
        best_len = run;
    }
    return best_len == 0 ? kNoMatch : best_pos;
}

__device__ inline uint64_t mix_km(uint64_t a, uint64_t b)
{
    uint64_t h = a * 0x9E3779B97F4A7C15ULL ^ (b + 0x7F4A7C15ULL) * 0xC2B2AE3D27D4EB4FULL;
    h ^= h >> 29;
    return h * 0x165667B19E3779F9ULL;
}

__device__ uint32_t vert_get(Search& S, const uint64_t km[2], uint64_t fh, uint64_t rh)
{
    const uint32_t n_slots = 1u << S.slot_bits;
    uint32_t slot = (uint32_t)(mix_km(km[0], km[1]) >> 40) & (n_slots - 1);
    const uint32_t tag = S.epoch << S.slot_bits;
    for (;;) {
        uint32_t sv = S.slots[slot];
        if ((sv & ~(n_slots - 1)) != tag || sv == 0) {
            // empty for this search
            if (S.nv >= S.vert_cap - 1) {
                S.overflow = true;
                return 0;
            }
            uint32_t id = S.nv++;
            Vert& x = S.v[id];
            x.km[0] = km[0];
            x.km[1] = km[1];
            x.fh = fh;
            x.rh = rh;
            x.head[0] = x.head[1] = x.tail[0] = x.tail[1] = 0xFFFFFFFFu;
            x.depth[0] = x.depth[1] = 0;
            x.color[0] = x.color[1] = 0;
            x.stamp_visit = x.stamp_cyc = 0;
            S.slots[slot] = tag | (id + 1);
            return id;
        }
        uint32_t id = (sv & (n_slots - 1)) - 1;
        if (S.v[id].km[0] == km[0] && S.v[id].km[1] == km[1])
            return id;
        slot = (slot + 1) & (n_slots - 1);
    }
}

// HashGraph::add_edge on history graph gi: append unless already there
__device__ void hist_add_edge(Search& S, int gi, uint32_t u, uint32_t v)
{
    Vert& U = S.v[u];
    for (uint32_t e = U.head[gi]; e != 0xFFFFFFFFu; e = S.e[e].next)
        if (S.e[e].to == v)
            return;
    if (S.ne >= S.edge_cap) {
        S.overflow = true;
        return;
    }
    uint32_t id = S.ne++;
    S.e[id].to = v;
    S.e[id].next = 0xFFFFFFFFu;
    if (U.tail[gi] == 0xFFFFFFFFu)
        U.head[gi] = id;
    else
        S.e[U.tail[gi]].next = id;
    U.tail[gi] = id;
}

__device__ inline void record_edge(Search& S, uint32_t src, uint32_t dst, int dir)
{
    if (dir == FWD)
        hist_add_edge(S, FWD, dst, src);
    else
        hist_add_edge(S, REV, src, dst);
}

__device__ inline int bump_cost(Search& S)
{
    if (S.cost >= S.max_cost) {
        S.max_cost_exceeded = true;
        return V_ABORT;
    }
    S.cost++;
    return V_SUCCESS;
}

__device__ int visit_edge(Search& S, uint32_t src, uint32_t dst, int dir)
{
    const uint32_t v = dir == FWD ? dst : src;
    const uint32_t parent = dir == FWD ? src : dst;
    if (S.v[v].color[!dir] != 0) {
        if (bump_cost(S) == V_ABORT)
            return V_ABORT;
        if (S.v[parent].depth[dir] >= S.max_depth[dir])
            return V_SKIP;
        bool dup = false;
        for (uint32_t i = 0; i < S.n_ce; i++)
            dup |= (S.ce_u[i] == src && S.ce_v[i] == dst);
        if (!dup && S.n_ce < kMaxPathsDev + 2) {
            S.ce_u[S.n_ce] = src;
            S.ce_v[S.n_ce] = dst;
            S.n_ce++;
        }
        if (S.max_paths != kNoLimit && S.n_ce > S.max_paths) {
            S.too_many_paths = true;
            return V_ABORT;
        }
        record_edge(S, src, dst, FWD);
        record_edge(S, src, dst, REV);
        return V_SUCCESS;
    }
    if (S.v[v].color[dir] == 0) {
        const uint32_t active = S.qt[dir] - S.qh[dir];
        if (S.max_branches != kNoLimit && active >= S.max_branches) {
            S.too_many_branches = true;
            return V_ABORT;
        }
        if (bump_cost(S) == V_ABORT)
            return V_ABORT;
        const uint16_t pd = S.v[parent].depth[dir];
        if (pd == S.max_depth[dir])
            return V_SKIP;
        S.v[v].depth[dir] = (uint16_t)(pd + 1);
        record_edge(S, src, dst, dir);
        S.v[v].color[dir] = 1;
        S.q[dir][S.qt[dir]++] = v;
        return V_SUCCESS;
    }
    if (bump_cost(S) == V_ABORT)
        return V_ABORT;
    record_edge(S, src, dst, dir);
    return V_SUCCESS;
}

// Graph/AllPathsSearch.h:28-104 over history graph gi; paths into out (kPathCap each)
__device__ int all_paths(Search& S, int gi, uint32_t from, uint32_t to, uint32_t max_paths,
                         uint32_t max_depth, uint32_t max_cost, uint32_t* out, uint32_t* out_len,
                         uint32_t& out_n, uint32_t& cost_out)
{
    uint32_t* path = S.dfs_path;
    uint32_t* it = S.dfs_it;
    const uint32_t stamp = ++S.stamp;
    uint32_t n_cyc = 0, n = 0, cost = 0;
    int rc = -1;
    out_n = 0;
    path[0] = from;
    it[0] = S.v[from].head[gi];
    n = 1;
    S.v[from].stamp_visit = stamp;
    while (n > 0 && cost <= max_cost) {
        if (path[n - 1] == to) {
            if (max_paths != kNoLimit && out_n >= max_paths) {
                rc = R_TOO_MANY_PATHS;
                break;
            }
            if (n_cyc) {
                rc = R_CYCLE;
                break;
            }
            if (out_n >= kMaxPathsDev || n > kPathCap) {
                S.overflow = true;
                rc = R_TOO_MANY_PATHS;
                break;
            }
            for (uint32_t i = 0; i < n; i++)
                out[out_n * kPathCap + i] = path[i];
            out_len[out_n++] = n;
        }
        while (n > 0) {
            const uint32_t cur = path[n - 1];
            if ((max_depth != kNoLimit && (n - 1) >= max_depth) || it[n - 1] == 0xFFFFFFFFu) {
                S.v[cur].stamp_visit = 0;
                if (S.v[cur].stamp_cyc == stamp) {
                    S.v[cur].stamp_cyc = 0;
                    n_cyc--;
                }
                n--;
                if (n > 0)
                    it[n - 1] = S.e[it[n - 1]].next;
            } else {
                const uint32_t v = S.e[it[n - 1]].to;
                if (S.v[v].stamp_visit == stamp) {
                    if (S.v[v].stamp_cyc != stamp) {
                        S.v[v].stamp_cyc = stamp;
                        n_cyc++;
                    }
                    it[n - 1] = S.e[it[n - 1]].next;
                } else {
                    if (n >= kPathCap) {
                        S.overflow = true;
                        rc = R_TOO_MANY_PATHS;
                        n = 0;
                        break;
                    }
                    path[n] = v;
                    it[n] = S.v[v].head[gi];
                    n++;
                    S.v[v].stamp_visit = stamp;
                    cost++;
                    break;
                }
            }
        }
        if (rc >= 0)
            break;
    }
    if (rc < 0) {
        if (cost > max_cost)
            rc = R_MAX_COST;
        else if (out_n == 0)
            rc = R_NO_PATH;
        else
            rc = R_FOUND;
    }
    cost_out = cost;
    return rc;
}

__device__ int build_paths_edge(Search& S, uint32_t u, uint32_t v)
{
    if (S.cost > S.max_cost) {
        S.max_cost_exceeded = true;
        return R_MAX_COST;
    }
    uint32_t c = 0;
    const uint32_t max_to_start = S.max_paths - S.found_n;
    int rc = all_paths(S, FWD, u, S.start, max_to_start, S.max_depth[FWD], S.max_cost - S.cost,
                       S.left, S.left_len, S.left_n, c);
    S.cost += c;
    if (rc == R_FOUND) {
        const uint32_t max_to_goal = (S.max_paths - S.found_n) / S.left_n;
        rc = all_paths(S, REV, v, S.goal, max_to_goal, S.max_depth[REV], S.max_cost - S.cost,
                       S.right, S.right_len, S.right_n, c);
        S.cost += c;
        if (rc == R_FOUND) {
            bool added = false;
            for (uint32_t x = 0; x < S.left_n; x++)
                for (uint32_t y = 0; y < S.right_n; y++) {
                    const uint32_t tot = S.left_len[x] + S.right_len[y];
                    if (tot < S.min_len || tot > S.max_len)
                        continue;
                    if (S.found_n >= kMaxPathsDev || tot > kPathCap) {
                        S.overflow = true;
                        continue;
                    }
                    uint32_t* m = S.found + S.found_n * kPathCap;
                    for (uint32_t i = 0; i < S.left_len[x]; i++)
                        m[i] = S.left[x * kPathCap + S.left_len[x] - 1 - i];
                    for (uint32_t i = 0; i < S.right_len[y]; i++)
                        m[S.left_len[x] + i] = S.right[y * kPathCap + i];
                    S.found_len[S.found_n++] = tot;
                    added = true;
                }
            rc = added ? R_FOUND : R_NO_PATH;
        }
    }
    if (rc == R_MAX_COST)
        S.max_cost_exceeded = true;
    else if (rc == R_TOO_MANY_PATHS)
        S.too_many_paths = true;
    return rc;
}

__device__ int bidi_search(Search& S)
{
    const uint32_t k = S.k;
    const u128 all = (k == 64) ? ~(u128)0 : (((u128)1 << (2 * k)) - 1);
    S.v[S.start].color[FWD] = 1;
    S.v[S.goal].color[REV] = 1;
    S.q[FWD][S.qt[FWD]++] = S.start;
    S.q[REV][S.qt[REV]++] = S.goal;
    int dir = FWD;
    bool aborted = false;
    while (!aborted && (S.qh[FWD] < S.qt[FWD] || S.qh[REV] < S.qt[REV])) {
        const uint32_t u = S.q[dir][S.qh[dir]++];
        const Vert pu = S.v[u];
        const u128 pk = km_load(pu.km);
        // base leaving the k-mer: first (forward extension) or last (backward extension)
        const uint32_t first = (uint32_t)(pk >> (2 * (k - 1))) & 3;
        const uint32_t last = (uint32_t)pk & 3;
        // (measured, round 4: issuing the four neighbours' filter look-ups together ahead of this loop made the
        // long searches of SURVEY 8d config 3's shape SLOWER, 1.71 s against 1.53 per step -- a search is one
        // lane's instruction stream, not a chain of load latencies, and the look-ups of neighbours a search
        // never reaches are instructions too)
        for (uint32_t b = 0; b < 4 && !aborted; b++) {
            uint64_t km[2], fh, rh;
            if (dir == FWD) {
                km_store(((pk << 2) | b) & all, km);
                succ_hash(S.st, pu.fh, pu.rh, first, b, fh, rh);
            } else {
                km_store((pk >> 2) | ((u128)b << (2 * (k - 1))), km);
                pred_hash(S.st, pu.fh, pu.rh, last, b, fh, rh);
            }
            if (!bf_has(S, fh + rh))
                continue;
            const uint32_t w = vert_get(S, km, fh, rh);
            if (S.overflow) {
                aborted = true;
                break;
            }
            const int r = dir == FWD ? visit_edge(S, u, w, FWD) : visit_edge(S, w, u, REV);
            if (r == V_ABORT || S.overflow)
                aborted = true;
        }
        if (aborted)
            break;
        S.v[u].color[dir] = 2;
        if (dir == REV && S.qh[FWD] < S.qt[FWD])
            dir = FWD;
        else if (dir == FWD && S.qh[REV] < S.qt[REV])
            dir = REV;
    }
    if (S.retry_on_overflow && S.overflow)
        return R_MEM; // the arena ran out and a larger one follows: whatever else was flagged is not final
    if (S.too_many_paths)
        return R_TOO_MANY_PATHS;
    if (S.too_many_branches)
        return R_TOO_MANY_BRANCHES;
    if (S.max_cost_exceeded)
        return R_MAX_COST;
    if (S.overflow)
        return R_MEM;
    if (S.small && S.found_n == 0 && S.n_ce == 1)
        return R_SIMPLE; // one common edge: small_unique_path tries it without a worker's scratch
    if (S.small && (S.found_n || S.n_ce))
        return R_DEFER; // paths to enumerate: that needs a worker's scratch
    int overall = S.found_n ? R_FOUND : R_NO_PATH;
    for (uint32_t i = 0; i < S.n_ce; i++) {
        int rc = build_paths_edge(S, S.ce_u[i], S.ce_v[i]);
        if (rc == R_FOUND)
            overall = R_FOUND;
        else if (rc != R_NO_PATH) {
            overall = rc;
            break;
        }
    }
    return overall;
}

__device__ inline uint64_t pool_alloc(const CArgs& a, uint32_t bytes)
{
    unsigned long long at = atomicAdd(a.pool_head, (unsigned long long)((bytes + 15) & ~15u));
    if (at + bytes > a.pool_cap) {
        atomicOr(a.err, 32u);
        return ~0ULL;
    }
    return at;
}

// connecting sequence of found path p into dst; returns its length
__device__ uint32_t path_to_seq(const Search& S, uint32_t p, uint8_t* dst)
{
    const uint32_t k = S.k;
    const uint32_t* P = S.found + p * kPathCap;
    const u128 x0 = km_load(S.v[P[0]].km);
    for (uint32_t i = 0; i < k; i++)
        dst[i] = (uint8_t)("ACGT"[(uint32_t)(x0 >> (2 * (k - 1 - i))) & 3]);
    for (uint32_t j = 1; j < S.found_len[p]; j++)
        dst[k + j - 1] = (uint8_t)("ACGT"[S.v[P[j]].km[0] & 3]);
    return k + S.found_len[p] - 1;
}

// merged = read1[0..sp) + cons + revcomp(read2[0..gp)), then maskNew(mask = 0)
__device__ void write_merged(const CArgs& a, uint32_t gi, const uint8_t* r1, uint32_t l1,
                             const uint8_t* r2, uint32_t l2, uint32_t sp, uint32_t gp,
                             const uint8_t* cons, uint32_t lc)
{
    const uint32_t lm = sp + lc + gp;
    const uint64_t at = pool_alloc(a, lm);
    if (at == ~0ULL)
        return;
    uint8_t* m = a.pool + at;
    for (uint32_t i = 0; i < sp; i++)
        m[i] = r1[i];
    for (uint32_t i = 0; i < lc; i++)
        m[sp + i] = cons[i];
    for (uint32_t i = 0; i < gp; i++)
        m[sp + lc + i] = comp_base(r2[gp - 1 - i]);
    for (uint32_t i = 0; i < l1 && i < lm; i++)
        if (r1[i] == m[i])
            m[i] = up(r1[i]);
    if (l2 <= lm)
        for (uint32_t i = 0; i < l2; i++) {
            const uint8_t c = comp_base(r2[l2 - 1 - i]);
            const uint32_t pos = i + lm - l2;
            if (c == m[pos])
                m[pos] = up(c);
        }
    a.fill_off[gi] = at;
    a.fill_len[gi] = lm;
}

// The lane-per-gap phase's own pathsToGoal for the case nearly every closed gap is (one in seven
// searches of configs[2] ends with a path, every one of those a plain chain): ONE common edge (u, v),
// and from u back to the start and from v on to the goal the traversal-history graphs are simple
// chains -- every vertex on the way has exactly one successor, the end vertex has none.
// build_paths_edge / all_paths (Graph/AllPathsSearch.h:28-104) then find exactly one path each
// way, push every vertex once (cost = edges walked, checked against what is left of max_cost after
// every push as the DFS does; no vertex deeper than max_depth is extended), and the gap's connecting
// sequence is the start k-mer followed by the last base of every later vertex -- written straight
// into the pool, so that no path has to be stored.  Anything else (a fork, a dead end, a limit, a
// length outside [min_len, max_len]) returns false and the gap goes to a worker as before.
__device__ bool small_unique_path(const CArgs& a, Search& S, uint32_t gi, const uint8_t* r1, uint32_t l1,
                                  const uint8_t* r2, uint32_t l2, uint32_t sp, uint32_t gp)
{
    const uint32_t NONE = 0xFFFFFFFFu, k = S.k;
    const uint32_t u = S.ce_u[0], v = S.ce_v[0];
    if (S.cost > S.max_cost)
        return false;
    uint32_t cost = S.cost;
    // chain length (vertices) from `from` to `to` in history graph gi, 0 if it is not a simple chain within the limits
    auto chain = [&](int gi_, uint32_t from, uint32_t to) -> uint32_t {
        const uint32_t budget = S.max_cost - cost;
        uint32_t n = 1, cur = from, c = 0;
        while (cur != to) {
            const uint32_t h = S.v[cur].head[gi_];
            if (h == NONE || S.e[h].next != NONE)
                return 0; // dead end or fork
            if (n - 1 >= S.max_depth[gi_])
                return 0; // the DFS does not extend a vertex this deep
            cur = S.e[h].to;
            n++;
            c++;
            if (c > budget || n > S.nv)
                return 0; // cost limit; (n > nv: a cycle, which a chain cannot hold -- belt and braces)
        }
        if (S.v[to].head[gi_] != NONE)
            return 0; // the DFS would go on past the target
        cost += c;
        return n;
    };
    const uint32_t nL = chain(FWD, u, S.start);
    if (!nL)
        return false;
    const uint32_t nR = chain(REV, v, S.goal);
    if (!nR)
        return false;
    const uint32_t tot = nL + nR;
    if (tot < S.min_len || tot > S.max_len || tot > kPathCap)
        return false;
    const uint32_t lc = k + tot - 1, lm = sp + lc + gp;
    const uint64_t at = pool_alloc(a, lm);
    if (at == ~0ULL)
        return true; // (error flagged; nothing to write, as write_merged)
    uint8_t* m = a.pool + at;
    for (uint32_t i = 0; i < sp; i++)
        m[i] = r1[i];
    uint8_t* cons = m + sp;
    // left part, walked from u (path index nL - 1) down to the start (index 0)
    for (uint32_t cur = u, idx = nL - 1;; idx--) {
        const Vert& x = S.v[cur];
        if (idx == 0) {
            const u128 x0 = km_load(x.km);
            for (uint32_t i = 0; i < k; i++)
                cons[i] = (uint8_t)("ACGT"[(uint32_t)(x0 >> (2 * (k - 1 - i))) & 3]);
            break;
        }
        cons[k - 1 + idx] = (uint8_t)("ACGT"[x.km[0] & 3]);
        cur = S.e[x.head[FWD]].to;
    }
    for (uint32_t cur = v, idx = nL;; idx++) {
        const Vert& x = S.v[cur];
        cons[k - 1 + idx] = (uint8_t)("ACGT"[x.km[0] & 3]);
        if (cur == S.goal)
            break;
        cur = S.e[x.head[REV]].to;
    }
    for (uint32_t i = 0; i < gp; i++)
        m[sp + lc + i] = comp_base(r2[gp - 1 - i]);
    // maskNew(mask = 0), as write_merged
    for (uint32_t i = 0; i < l1 && i < lm; i++)
        if (r1[i] == m[i])
            m[i] = up(r1[i]);
    if (l2 <= lm)
        for (uint32_t i = 0; i < l2; i++) {
            const uint8_t c = comp_base(r2[l2 - 1 - i]);
            const uint32_t pos = i + lm - l2;
            if (c == m[pos])
                m[pos] = up(c);
        }
    a.fill_off[gi] = at;
    a.fill_len[gi] = lm;
    return true;
}

struct SmallArena {
    Vert v[kSmallVerts];
    Edge e[kSmallEdges];
    uint32_t slots[1u << kSmallSlotBits];
    uint32_t q[2][kSmallVerts];
};

// One gap at one k.  `sm` != null: lane-per-gap attempt in that lane's small arena; returns true
// when the gap has to be done again by a worker (arena too small, or paths to enumerate) --
// nothing of such an attempt is counted or written.
__device__ bool connect_one(const CArgs& a, Search& S, uint32_t gi, uint32_t wid, uint8_t* r1,
                            uint8_t* r2, uint8_t* tmp, SmallArena* sm)
{
    const Gap g = a.gaps[gi];
    // still in `flanks`?  (last gap of its flank pair, left flank not closed at a larger k)
    if (a.pair_last[g.pair_rep] != gi + 1 || a.closed_at[g.left_rep] <= a.ki)
        return false;
    const Flank f = flank_of(a, gi);
    const uint32_t k = a.k;
    const uint32_t l1 = f.le - f.ls, l2 = f.re - f.rs;
    if (l1 < k || l2 < k) {
        atomicAdd(&a.stats[7], 1ull); // a flank shorter than k: "No start/goal kmer"
        return false;
    }
    const uint8_t* s = a.seq + a.off[g.contig];
    for (uint32_t i = 0; i < l1; i++)
        r1[i] = up(s[f.ls + i]);
    for (uint32_t i = 0; i < l2; i++)
        r2[i] = comp_base(up(s[f.re - 1 - i]));

    S.hash_num = a.hash_num;
    S.bf_mod = a.bf_mod;
    S.bf = a.bf + ((uint64_t)a.contig_batch[g.contig] * a.n_k + a.ki) * a.bf_words;
    S.k = k;
    seed_tab_init(S.st, k);
    S.tests = 0;
    uint32_t sp = start_kmer_pos(S, r1, l1, 3);
    uint32_t gp = kNoMatch;
    if (sp != kNoMatch)
        gp = start_kmer_pos(S, r2, l2, 3);
    else
        (void)start_kmer_pos(S, r2, l2, 3); // the reference evaluates both (probe count)
    if (sp == kNoMatch || gp == kNoMatch) {
        atomicAdd(&a.stats[6], S.tests);
        atomicAdd(&a.stats[7], 1ull);
        return false;
    }
    S.small = sm != nullptr;
    S.retry_on_overflow = sm != nullptr || a.tier == 0;
    if (sm) {
        // ---- lane-private small arena
        for (uint32_t i = 0; i < (1u << kSmallSlotBits); i++)
            sm->slots[i] = 0;
        S.slots = sm->slots;
        S.epoch = 1;
        S.slot_bits = kSmallSlotBits;
        S.vert_cap = kSmallVerts;
        S.edge_cap = kSmallEdges;
        S.v = sm->v;
        S.e = sm->e;
        S.q[0] = sm->q[0];
        S.q[1] = sm->q[1];
        S.dfs_path = S.dfs_it = S.left = S.right = S.found = nullptr;
    } else {
        // ---- this worker's arena: a medium one in the first pass, a full one for what outgrew that
        const bool med = a.tier == 0;  // (tiers 1 and 2: a full arena)
        const uint32_t sbits = med ? kMedSlotBits : kSlotBits, vcap = med ? kMedVertCap : kVertCap;
        const uint32_t ecap = med ? kMedEdgeCap : kEdgeCap;
        uint32_t* const epoch_of = (med ? a.m_epoch : a.epoch) + wid;
        uint32_t ep = *epoch_of + 1;
        S.slots = med ? a.m_slots + ((uint64_t)wid << kMedSlotBits) : a.slots + (uint64_t)wid * kSlots;
        if (ep >= (1u << (32 - sbits))) {
            for (uint32_t i = 0; i < (1u << sbits); i++)
                S.slots[i] = 0;
            ep = 1;
        }
        *epoch_of = ep;
        S.epoch = ep;
        S.slot_bits = sbits;
        S.vert_cap = vcap;
        S.edge_cap = ecap;
        S.v = med ? a.m_verts + (uint64_t)wid * kMedVertCap : a.verts + (uint64_t)wid * kVertCap;
        S.e = med ? a.m_edges + (uint64_t)wid * kMedEdgeCap : a.edges + (uint64_t)wid * kEdgeCap;
        uint32_t* sc = med ? a.m_scratch + (uint64_t)wid * kMedScratchWords : a.scratch + (uint64_t)wid * kScratchWords;
        S.q[0] = sc;
        S.q[1] = sc + vcap;
        S.dfs_path = sc + 2 * vcap;
        S.dfs_it = S.dfs_path + 2 * kPathCap;
        S.left = S.dfs_it + 2 * kPathCap;
        S.right = S.left + kMaxPathsDev * kPathCap;
        S.found = S.right + kMaxPathsDev * kPathCap;
    }
    S.nv = S.ne = 0;
    S.qh[0] = S.qh[1] = S.qt[0] = S.qt[1] = 0;
    S.left_n = S.right_n = S.found_n = 0;
    S.n_ce = 0;
    S.stamp = 0;
    S.cost = 0;
    S.too_many_branches = S.max_cost_exceeded = S.too_many_paths = S.overflow = false;
    S.max_paths = a.max_paths;
    S.max_branches = a.max_branches;
    S.max_cost = a.max_cost;

    uint64_t km[2], fh, rh;
    kmer_of(S, r1 + sp, km, fh, rh);
    S.start = vert_get(S, km, fh, rh);
    uint8_t gk[64];
    for (uint32_t i = 0; i < k; i++)
        gk[i] = comp_base(r2[gp + k - 1 - i]);
    kmer_of(S, gk, km, fh, rh);
    S.goal = vert_get(S, km, fh, rh);

    const uint32_t max_merged = a.max_gap + 2 * a.flank_len;
    const uint32_t max_path = max_merged - k + 1 - sp - gp;
    const int min_i = (int)(0u - k + 1 - sp - gp);
    uint32_t min_path = (uint32_t)(min_i > 0 ? min_i : 0);
    const uint32_t need = max(l1 - k + 1 - sp, l2 - k + 1 - gp);
    min_path = max(min_path, need);
    S.min_len = (uint16_t)min_path;
    S.max_len = (uint16_t)max_path;
    const uint16_t md = (uint16_t)(S.max_len - 1);
    S.max_depth[FWD] = (uint16_t)(md / 2 + md % 2);
    S.max_depth[REV] = (uint16_t)(md / 2);
    if (S.overflow)
        return true; // (small arena) not even the two end vertices fitted
    if (S.start == S.goal && 1 >= S.min_len) {
        if (sm)
            return true; // a path already: a worker's business
        S.found[0] = S.start;
        S.found_len[0] = 1;
        S.found_n = 1;
    }
    int rc = bidi_search(S);
    if (sm && (rc == R_DEFER || rc == R_MEM))
        return true;
    bool written = false;
    if (rc == R_SIMPLE) {
        if (!small_unique_path(a, S, gi, r1, l1, r2, l2, sp, gp))
            return true; // not the plain case after all: a worker's business
        rc = R_FOUND;
        S.found_n = 1;
        written = true;
    }
    if (!sm && a.tier == 0 && rc == R_MEM)
        return true; // outgrew the medium arena: a worker with a full one does it again (nothing counted, nothing written)
    atomicAdd(&a.stats[6], S.tests);
    if (S.overflow)
        atomicOr(a.err, 64u);
    // the reference's per-k log lines (sealer.cpp:656-683), summed over the k runs
    if (rc != R_FOUND) {
        const int slot = rc == R_NO_PATH ? 8 : rc == R_TOO_MANY_PATHS ? 11 : rc == R_TOO_MANY_BRANCHES ? 12
                         : rc == R_MAX_COST ? 13 : -1;
        if (slot >= 0)
            atomicAdd(&a.stats[slot], 1ull);
        return false;
    }
    atomicAdd(&a.stats[S.found_n == 1 ? 9 : 10], 1ull);
    // closing removes every gap that shares this left flank from the later k runs
    atomicMin(&a.closed_at[g.left_rep], a.ki + 1);
    atomicAdd(&a.stats[5], 1ull);
    if (written) {
        // (small_unique_path wrote the merged sequence already)
    } else if (S.found_n == 1) {
        const uint32_t lc = path_to_seq(S, 0, tmp);
        write_merged(a, gi, r1, l1, r2, l2, sp, gp, tmp, lc);
    } else {
        // keep the alternative connecting sequences for the consensus pass
        const uint32_t mi = atomicAdd(&a.queue[2], 1u);
        if (mi >= a.multi_cap) {
            atomicOr(a.err, 128u);
            return false;
        }
        MultiHdr& h = a.multi[mi];
        h.gap = gi;
        h.sp = sp;
        h.gp = gp;
        h.n_paths = S.found_n;
        for (uint32_t p = 0; p < S.found_n; p++) {
            const uint32_t lc = path_to_seq(S, p, tmp);
            const uint64_t at = pool_alloc(a, lc);
            h.len[p] = lc;
            h.off[p] = at;
            if (at != ~0ULL)
                for (uint32_t i = 0; i < lc; i++)
                    a.pool[at + i] = tmp[i];
        }
    }
    return false;
}

// Phase 1: one gap per LANE.  A lane walks its gap's graph in a small arena of its own; the
// sixty-four walks of a wave diverge freely (each is a handful of dependent look-ups, so there
// is little to lose and a factor of tens in gaps in flight to gain).  What does not fit, or
// finds paths to enumerate, goes on the list for phase 2.
__global__ __launch_bounds__(64, 4) void sealer_connect_lanes(CArgs a, SmallArena* arenas)
{
    const uint32_t lane = threadIdx.x;
    const uint32_t n_work = a.queue[0];
    SmallArena* mine = arenas + (uint64_t)blockIdx.x * a.simt_lanes + (lane % a.simt_lanes);
    Search S;
    uint8_t r1[64], r2[64];
    for (;;) {
        uint32_t base = 0;
        if (lane == 0)
            base = atomicAdd(&a.queue[1], a.simt_lanes);
        base = __shfl(base, 0);
        if (base >= n_work)
            break; // every wave leaves here once the queue has drained
        const uint32_t it = base + lane;
        if (lane < a.simt_lanes && it < n_work) {
            const uint32_t gi = a.work[it];
            if (connect_one(a, S, gi, 0, r1, r2, nullptr, mine))
                a.defer[atomicAdd(&a.queue[4], 1u)] = gi;
        }
    }
}

// Phase 2: one gap per wave, lane 0 walking: whatever phase 1 left.  Launched twice: tier 0, every
// worker with a medium arena over phase 1's list (what outgrows it goes to a second list); tier 1,
// a few workers with full arenas over that list.
__global__ __launch_bounds__(64, 4) void sealer_connect(CArgs a)
{
    __shared__ Search S;
    __shared__ uint8_t r1[64], r2[64], tmp[kPathCap + 64];
    __shared__ uint32_t item;
    const uint32_t wid = blockIdx.x, lane = threadIdx.x;
    const uint32_t* const list = a.tier == 1 ? a.defer2 : a.defer; // (tier 2: full arenas straight over phase 1's list)
    uint32_t* const head = a.queue + (a.tier == 1 ? 7 : 5);
    const uint32_t n_work = a.queue[a.tier == 1 ? 6 : 4];
    for (;;) {
        if (lane == 0)
            item = atomicAdd(head, 1u);
        __syncthreads();
        const uint32_t it = item;
        __syncthreads();
        if (it >= n_work)
            break; // every wave leaves here once the list has drained
        // one lane walks the graph; the wave reconverges before the next fetch
        if (lane == 0 && connect_one(a, S, list[it], wid, r1, r2, tmp, nullptr))
            a.defer2[atomicAdd(&a.queue[6], 1u)] = list[it];
    }
}

// ------------------------------------------------------------- consensus
__device__ inline uint32_t amb_mask(uint8_t c)
{
    const uint32_t tab[26] = { 0x1, 0xe, 0x2, 0xd, 0, 0, 0x4, 0xb, 0, 0, 0xc, 0, 0x3,
                               0xf, 0, 0, 0, 0x5, 0x6, 0x8, 0, 0x7, 0x9, 0, 0xa, 0 };
    const uint32_t i = (uint32_t)(up(c) - 'A');
    return i < 26 ? tab[i] : 0;
}

__device__ inline uint8_t amb_or(uint8_t x, uint8_t y)
{
    const char* back = "NACMGRSVTWYHKDBN";
    uint8_t c = (uint8_t)back[(amb_mask(x) | amb_mask(y)) & 15];
    const bool lower = (x >= 'a' && x <= 'z') || (y >= 'a' && y <= 'z');
    return lower ? (uint8_t)(c + 32) : c;
}

__device__ inline int nw_score(uint8_t x, uint8_t y, uint8_t& c)
{
    if (x == y) {
        c = x;
        return 5;
    }
    c = amb_or(x, y);
    return (c == x || c == y) ? 5 : -4;
}

// LDS of one consensus worker (one wave): the two sequences and three anti-diagonals of the
// three score matrices
struct NwLds {
    int32_t dg[3][3][kNwMax + 2]; // [diagonal % 3][F, G, H][row]
    uint8_t A[kNwMax + 4], B[kNwMax + 4];
    uint32_t n;
};

// Gotoh global alignment + consensus (Align/alignGlobal.cc:55-198); out gets the
// consensus, returns its length.  Called by every lane of the worker's wave: the matrices are
// filled anti-diagonal by anti-diagonal, a cell per lane, with the two previous diagonals in LDS
// (a cell needs its upper and left neighbours of the diagonal before and the upper-left one of
// the diagonal before that); every cell also goes to memory, where lane 0 then walks back
// through them exactly as the serial code does.  (One lane filling 300 x 300 cells from memory
// took tens of milliseconds: the multi-path gaps were a fifth of stage C's time.)
__device__ uint32_t nw_consensus(NwLds& L, const uint8_t* A_in, uint32_t la, const uint8_t* B_in, uint32_t lb,
                                 int32_t* mat, uint8_t* out)
{
    const int OPEN = -12, EXT = -4, NEG = -(1 << 29);
    const uint32_t W = lb + 1;
    const uint32_t lane = threadIdx.x;
    int32_t* f = mat;
    int32_t* g = mat + (uint64_t)(kNwMax + 1) * (kNwMax + 1);
    int32_t* h = g + (uint64_t)(kNwMax + 1) * (kNwMax + 1);
#define F(i, j) f[(uint64_t)(i) * W + (j)]
#define G(i, j) g[(uint64_t)(i) * W + (j)]
#define H(i, j) h[(uint64_t)(i) * W + (j)]
    __syncthreads(); // the previous alignment's LDS is free
    for (uint32_t i = lane; i < la; i += 64)
        L.A[i] = A_in[i];
    for (uint32_t j = lane; j < lb; j += 64)
        L.B[j] = B_in[j];
    __syncthreads();
    uint8_t c;
    for (uint32_t d = 0; d <= la + lb; d++) {
        const uint32_t cur = d % 3, p1 = (d + 2) % 3, p2 = (d + 1) % 3;
        const uint32_t i_lo = d > lb ? d - lb : 0u, i_hi = min(d, la);
        for (uint32_t i = i_lo + lane; i <= i_hi; i += 64) {
            const uint32_t j = d - i;
            int fv, gv, hv;
            if (i == 0) {
                fv = hv = j == 0 ? 0 : OPEN + EXT * ((int)j - 1);
                gv = NEG;
            } else if (j == 0) {
                fv = gv = OPEN + EXT * ((int)i - 1);
                hv = NEG;
            } else {
                gv = max(L.dg[p1][0][i - 1] + OPEN, L.dg[p1][1][i - 1] + EXT);
                hv = max(L.dg[p1][0][i] + OPEN, L.dg[p1][2][i] + EXT);
                const int dv = L.dg[p2][0][i - 1] + nw_score(L.A[i - 1], L.B[j - 1], c);
                fv = max(dv, max(gv, hv));
            }
            L.dg[cur][0][i] = fv;
            L.dg[cur][1][i] = gv;
            L.dg[cur][2][i] = hv;
            F(i, j) = fv;
            G(i, j) = gv;
            H(i, j) = hv;
        }
        __syncthreads();
    }
    if (lane == 0) {
        const uint8_t* A = L.A;
        const uint8_t* B = L.B;
        uint32_t n = 0, i = la, j = lb;
        while (i > 0 && j > 0) {
            const int fij = F(i, j);
            const uint8_t x = A[i - 1], y = B[j - 1];
            const int sc = nw_score(x, y, c);
            if (fij == F(i - 1, j - 1) + sc) {
                out[n++] = c;
                i--;
                j--;
            } else if (fij == F(i - 1, j) + OPEN || fij == G(i - 1, j) + EXT) {
                while (i > 1 && G(i, j) == G(i - 1, j) + EXT) {
                    out[n++] = (uint8_t)(A[i - 1] | 0x20);
                    i--;
                }
                out[n++] = (uint8_t)(A[i - 1] | 0x20);
                i--;
            } else {
                while (j > 1 && H(i, j) == H(i, j - 1) + EXT) {
                    out[n++] = (uint8_t)(B[j - 1] | 0x20);
                    j--;
                }
                out[n++] = (uint8_t)(B[j - 1] | 0x20);
                j--;
            }
        }
        while (i > 0)
            out[n++] = (uint8_t)(A[--i] | 0x20);
        while (j > 0)
            out[n++] = (uint8_t)(B[--j] | 0x20);
        for (uint32_t x = 0, y = n; x + 1 < y; x++, y--) {
            const uint8_t t = out[x];
            out[x] = out[y - 1];
            out[y - 1] = t;
        }
        L.n = n;
    }
    __syncthreads(); // `out` (memory, written by lane 0) and L.n are there for every lane
#undef F
#undef G
#undef H
    return L.n;
}

// every lane of the worker's wave (the alignments are filled by all of them)
__device__ void consensus_one(const CArgs& a, NwLds& L, uint32_t it, int32_t* mat, uint8_t* bufA,
                              uint8_t* bufB, uint8_t* r1, uint8_t* r2)
{
    const MultiHdr h = a.multi[it];
    bool ok = true;
    for (uint32_t p = 0; p < h.n_paths; p++)
        ok &= (h.off[p] != ~0ULL) && h.len[p] <= kNwMax;
    if (!ok) { // (uniform: every lane read the same header)
        if (threadIdx.x == 0)
            atomicOr(a.err, 256u);
        return;
    }
    // alignPair for two, alignMulti's left fold for more
    uint32_t lc = nw_consensus(L, a.pool + h.off[0], h.len[0], a.pool + h.off[1], h.len[1], mat, bufA);
    uint8_t* cur = bufA;
    uint8_t* nxt = bufB;
    for (uint32_t p = 2; p < h.n_paths && ok; p++) {
        if (lc > kNwMax) {
            ok = false;
            break;
        }
        lc = nw_consensus(L, cur, lc, a.pool + h.off[p], h.len[p], mat, nxt);
        uint8_t* t = cur;
        cur = nxt;
        nxt = t;
    }
    if (threadIdx.x != 0)
        return;
    if (!ok) {
        atomicOr(a.err, 256u);
        return;
    }
    const Gap g = a.gaps[h.gap];
    const Flank f = flank_of(a, h.gap);
    const uint32_t l1 = f.le - f.ls, l2 = f.re - f.rs;
    const uint8_t* s = a.seq + a.off[g.contig];
    for (uint32_t i = 0; i < l1; i++)
        r1[i] = up(s[f.ls + i]);
    for (uint32_t i = 0; i < l2; i++)
        r2[i] = comp_base(up(s[f.re - 1 - i]));
    write_merged(a, h.gap, r1, l1, r2, l2, h.sp, h.gp, cur, lc);
}

__global__ __launch_bounds__(64) void sealer_consensus(CArgs a)
{
    __shared__ NwLds L;
    __shared__ uint32_t item;
    __shared__ uint8_t r1[64], r2[64];
    const uint32_t wid = blockIdx.x, lane = threadIdx.x;
    const uint32_t n_multi = min(a.queue[2], a.multi_cap);
    // per worker: three score matrices, then two consensus buffers
    const uint64_t mat_words = 3ull * (kNwMax + 1) * (kNwMax + 1);
    const uint64_t stride = mat_words + 2 * (2 * kNwMax + 64) / 4 + 64;
    int32_t* mat = a.nw + (uint64_t)wid * stride;
    uint8_t* bufA = (uint8_t*)(mat + mat_words);
    uint8_t* bufB = bufA + 2 * kNwMax + 64;
    for (;;) {
        if (lane == 0)
            item = atomicAdd(&a.queue[3], 1u);
        __syncthreads();
        const uint32_t it = item;
        __syncthreads();
        if (it >= n_multi)
            break;
        consensus_one(a, L, it, mat, bufA, bufB, r1, r2);
    }
}

// align() alone, over caller-supplied sets of connecting sequences (gp_sealer_consensus): the
// same worker layout and the same nw_consensus as sealer_consensus above
__global__ __launch_bounds__(64) void consensus_sets(const uint8_t* seq, const uint64_t* seq_off,
                                                     const uint32_t* set_begin, uint32_t n_sets, int32_t* nw,
                                                     uint8_t* out, uint32_t* out_len, uint32_t* err,
                                                     volatile uint32_t* progress)
{
    __shared__ NwLds L;
    const uint32_t wid = blockIdx.x, lane = threadIdx.x;
    const uint64_t mat_words = 3ull * (kNwMax + 1) * (kNwMax + 1);
    const uint64_t stride = mat_words + 2 * (2 * kNwMax + 64) / 4 + 64;
    const uint32_t out_stride = 2 * kNwMax + 64;
    int32_t* mat = nw + (uint64_t)wid * stride;
    uint8_t* bufA = (uint8_t*)(mat + mat_words);
    uint8_t* bufB = bufA + 2 * kNwMax + 64;
    // the sets are dealt round the workers (no queue: every worker knows its sets from its number)
    for (uint32_t it = wid; it < n_sets; it += gridDim.x) {
        const uint32_t s0 = set_begin[it], s1 = set_begin[it + 1];
        bool ok = s1 - s0 >= 2;
        for (uint32_t p = s0; p < s1; p++)
            ok &= seq_off[p + 1] - seq_off[p] <= kNwMax && seq_off[p + 1] > seq_off[p];
        uint32_t lc = 0;
        uint8_t* cur = bufA;
        uint8_t* nxt = bufB;
        if (ok) // (uniform: every lane read the same offsets)
            lc = nw_consensus(L, seq + seq_off[s0], (uint32_t)(seq_off[s0 + 1] - seq_off[s0]), seq + seq_off[s0 + 1],
                              (uint32_t)(seq_off[s0 + 2] - seq_off[s0 + 1]), mat, bufA);
        for (uint32_t p = s0 + 2; p < s1 && ok; p++) {
            if (lc > kNwMax) {
                ok = false;
                break;
            }
            lc = nw_consensus(L, cur, lc, seq + seq_off[p], (uint32_t)(seq_off[p + 1] - seq_off[p]), mat, nxt);
            uint8_t* t = cur;
            cur = nxt;
            nxt = t;
        }
        if (ok)
            for (uint32_t i = lane; i < lc; i += 64)
                out[(uint64_t)it * out_stride + i] = cur[i];
        if (lane == 0) {
            out_len[it] = ok ? lc : 0u;
            if (!ok)
                atomicOr(err, 256u);
            progress[wid] = it + 1; // (mapped host memory: what the host reports if the launch does not come back)
        }
    }
}

int sealer_consensus_sets(gp_ctx* ctx, uint32_t n_sets, const uint32_t* set_seq_begin, const uint8_t* seq_buf,
                          const uint64_t* seq_off, uint8_t* out_buf, uint64_t out_stride, uint32_t* out_len)
{
    const uint32_t n_seqs = set_seq_begin[n_sets];
    const uint64_t bytes = seq_off[n_seqs];
    const uint32_t dev_stride = 2 * kNwMax + 64;
    const uint32_t workers = std::min<uint32_t>(kNwWorkers, std::max<uint32_t>(1u, n_sets));
    DevBuf<uint8_t> d_seq, d_out;
    DevBuf<uint64_t> d_off;
    DevBuf<uint32_t> d_begin, d_len, d_err;
    DevBuf<int32_t> d_nw;
    int rc;
    if ((rc = d_seq.alloc(bytes)) != GP_OK || (rc = d_off.alloc((size_t)n_seqs + 1)) != GP_OK ||
        (rc = d_begin.alloc((size_t)n_sets + 1)) != GP_OK || (rc = d_len.alloc(n_sets)) != GP_OK ||
        (rc = d_out.alloc((uint64_t)n_sets * dev_stride)) != GP_OK || (rc = d_err.alloc(1)) != GP_OK ||
        (rc = d_nw.alloc((uint64_t)workers *
                         (3ull * (kNwMax + 1) * (kNwMax + 1) + 2 * (2 * kNwMax + 64) / 4 + 64))) != GP_OK)
        return rc;
    // progress words in mapped host memory, one per worker
    uint32_t* h_prog = nullptr;
    GP_HIP(hipHostMalloc((void**)&h_prog, kNwWorkers * sizeof(uint32_t), hipHostMallocMapped));
    struct FreeProg {
        uint32_t* p;
        ~FreeProg() { if (p) (void)hipHostFree(p); }
    } free_prog{ h_prog };
    std::memset(h_prog, 0, kNwWorkers * sizeof(uint32_t));
    uint32_t* d_prog = nullptr;
    GP_HIP(hipHostGetDevicePointer((void**)&d_prog, h_prog, 0));
    const bool dbg = getenv("GP_CONSENSUS_DEBUG") != nullptr;
    hipStream_t s = ctx->stream;
    GP_HIP(hipMemcpyAsync(d_seq.p, seq_buf, bytes, hipMemcpyHostToDevice, s));
    GP_HIP(hipMemcpyAsync(d_off.p, seq_off, ((size_t)n_seqs + 1) * 8, hipMemcpyHostToDevice, s));
    GP_HIP(hipMemcpyAsync(d_begin.p, set_seq_begin, ((size_t)n_sets + 1) * 4, hipMemcpyHostToDevice, s));
    GP_HIP(hipMemsetAsync(d_err.p, 0, 4, s));
    GP_HIP(hipStreamSynchronize(s));
    if (dbg)
        fprintf(stderr, "consensus: %u sets, %u sequences, %llu bytes uploaded; launching %u workers\n", n_sets, n_seqs,
                (unsigned long long)bytes, workers);
    hipLaunchKernelGGL(consensus_sets, dim3(workers), dim3(64), 0, s, d_seq.p, d_off.p, d_begin.p, n_sets, d_nw.p,
                       d_out.p, d_len.p, d_err.p, d_prog);
    GP_HIP(hipGetLastError());
    // a consensus takes milliseconds; a launch that has not come back after a minute is reported,
    // with how far every worker got.  The buffers of a launch that may still be running must not be
    // recycled (the pool hands blocks out again without the device-wide wait of hipFree, and the
    // progress page is host memory): the report goes to stderr at once, then the stream is waited
    // for; if that wait fails too the buffers are leaked on purpose and the context is unusable.
    auto give_up = [&](const std::string& msg) {
        fprintf(stderr, "libgpcore: %s\n", msg.c_str());
        if (hipStreamSynchronize(s) != hipSuccess) {
            (void)hipGetLastError();
            auto leak = [](auto& b) { b.p = nullptr; b.n = 0; b.cap_bytes = 0; };
            leak(d_seq); leak(d_out); leak(d_off); leak(d_begin); leak(d_len); leak(d_err); leak(d_nw);
            free_prog.p = nullptr;
        }
        set_error(msg);
        return GP_ERR_HIP;
    };
    for (int waited_ms = 0;; waited_ms += 2) {
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess)
            break;
        if (q != hipErrorNotReady) {
            (void)hip_fail(q, "consensus_sets", __FILE__, __LINE__);
            return give_up(gp_last_error());
        }
        if (waited_ms >= 60000) {
            std::string msg = "gp_sealer_consensus: the device did not finish within a minute; sets done per worker:";
            for (uint32_t w = 0; w < workers; w++)
                msg += " " + std::to_string(h_prog[w]);
            return give_up(msg);
        }
        struct timespec ts = { 0, 2000000 };
        nanosleep(&ts, nullptr);
    }
    if (dbg)
        fprintf(stderr, "consensus: kernel done\n");
    std::vector<uint32_t> h_len(n_sets);
    std::vector<uint8_t> h_out((size_t)n_sets * dev_stride);
    uint32_t h_err = 0;
    GP_HIP(hipMemcpyAsync(h_len.data(), d_len.p, (size_t)n_sets * 4, hipMemcpyDeviceToHost, s));
    GP_HIP(hipMemcpyAsync(h_out.data(), d_out.p, h_out.size(), hipMemcpyDeviceToHost, s));
    GP_HIP(hipMemcpyAsync(&h_err, d_err.p, 4, hipMemcpyDeviceToHost, s));
    GP_HIP(hipStreamSynchronize(s));
    if (h_err) {
        set_error("gp_sealer_consensus: a set with fewer than two sequences, an empty sequence, or a sequence / "
                  "running consensus longer than " + std::to_string(kNwMax) + " bases");
        return GP_ERR_UNSUPPORTED;
    }
    for (uint32_t i = 0; i < n_sets; i++) {
        if (h_len[i] > out_stride) {
            set_error("gp_sealer_consensus: out_stride is smaller than a consensus");
            return GP_ERR_INVALID_ARG;
        }
        std::memcpy(out_buf + (uint64_t)i * out_stride, h_out.data() + (size_t)i * dev_stride, h_len[i]);
        out_len[i] = h_len[i];
    }
    return GP_OK;
}

// --------------------------------------------------------------- splicing
constexpr int kApplyThreads = 256;

__global__ __launch_bounds__(kApplyThreads) void apply_fills(CArgs a)
{
    const uint32_t cid = blockIdx.x, tid = threadIdx.x;
    if (a.dropped[cid] || !(a.flags[cid] & 2u))
        return; // gap-free contigs were copied by copy_flagged
    const uint32_t len = a.len_cur[cid];
    const uint8_t* src = a.seq + a.off[cid];
    uint8_t* dst = a.out + a.off[cid];
    const uint32_t cap = (uint32_t)(a.off[cid + 1] - a.off[cid]);
    const uint32_t n = a.n_gaps[cid], base = a.gap_off[cid];
    // current string = src[0..P) + dst[sb..cap): the suffix grows leftwards
    uint32_t P = len, sb = cap;
    bool overflow = false;
    for (uint32_t ii = n; ii-- > 0;) {
        const uint32_t gi = base + ii;
        const uint32_t fl = a.fill_len[gi];
        if (!fl)
            continue;
        // A fill whose range reaches into the previous (right-hand) fill overwrites the start of
        // it: every lane must be done writing that one first (found by tools/fuzz_parity.py:
        // neighbouring gaps closer than two flanks, both closed).  Everything that decides
        // whether this point is reached is uniform across the workgroup.
        __syncthreads();
        const Flank f = flank_of(a, gi);
        const uint32_t pos = f.ls;
        const uint32_t size = P + (cap - sb);
        if (pos > size)
            continue; // std::out_of_range in the reference
        uint32_t cnt = f.re - f.ls;
        if (cnt > size - pos)
            cnt = size - pos;
        // pos < P always (flanks lie left of everything spliced so far)
        if (pos + cnt >= P) {
            sb += pos + cnt - P; // the replaced range eats into the suffix
        } else {
            const uint32_t keep = P - (pos + cnt);
            if (sb < keep + fl + pos) {
                overflow = true;
                break;
            }
            sb -= keep;
            for (uint32_t i = tid; i < keep; i += kApplyThreads)
                dst[sb + i] = src[pos + cnt + i];
        }
        if (sb < fl + pos) {
            overflow = true;
            break;
        }
        sb -= fl;
        const uint8_t* fill = a.pool + a.fill_off[gi];
        for (uint32_t i = tid; i < fl; i += kApplyThreads)
            dst[sb + i] = fill[i];
        P = pos;
    }
    if (overflow) {
        if (tid == 0)
            atomicOr(a.err, 512u);
        return;
    }
    __syncthreads();
    // result = src[0..P) + suffix; move the suffix down to P tile by tile (it only moves left)
    const uint32_t ns = cap - sb;
    if (sb != P) {
        for (uint32_t t0 = 0; t0 < ns; t0 += kApplyThreads * 16) {
            uint8_t v[16];
            uint32_t cnt = 0;
            for (uint32_t q = 0; q < 16; q++) {
                const uint32_t i = t0 + q * kApplyThreads + tid;
                if (i < ns) {
                    v[q] = dst[sb + i];
                    cnt = q + 1;
                }
            }
            __syncthreads();
            for (uint32_t q = 0; q < cnt; q++)
                dst[P + t0 + q * kApplyThreads + tid] = v[q];
            __syncthreads();
        }
    }
    {
        const uint32_t n16 = P / 16;
        const uint4* s4 = reinterpret_cast<const uint4*>(src);
        uint4* d4 = reinterpret_cast<uint4*>(dst);
        for (uint32_t i = tid; i < n16; i += kApplyThreads)
            d4[i] = s4[i];
        for (uint32_t i = n16 * 16 + tid; i < P; i += kApplyThreads)
            dst[i] = src[i];
    }
    if (tid == 0)
        a.len_next[cid] = P + ns;
}

// ------------------------------------------------------------------ host side
static CState* cstate(gp_job* job) { return (CState*)job->bc->c_state; }

// lanes of a sealer_connect_lanes wave that take a gap, each with a small arena of its own (read once, when the job's arenas are allocated)
static uint32_t sealer_lanes() { return (uint32_t)std::min<uint64_t>(kSimtLanes, std::max<uint64_t>(1, env_u64("GP_SEALER_LANES", 32))); }

int stage_c_prepare(gp_job* job)
{
    if (!job->prm.run_sealer)
        return GP_OK;
    const gp_params& p = job->prm;
    if (p.sealer_max_paths > kMaxPathsDev - 2 || p.sealer_max_cost > kVertCap - 8 ||
        p.sealer_max_gap + 2 * p.k[0] - p.k[p.n_k - 1] + 2 > kPathCap) {
        set_error("Sealer limits beyond device arenas: max_paths <= 14, max_cost <= 131064");
        return GP_ERR_UNSUPPORTED;
    }
    CState* c = new (std::nothrow) CState();
    if (!c)
        return GP_ERR_OOM;
    job->bc->c_state = c;
    const uint32_t n = job->n_contigs;
    c->h_gap_off.resize((size_t)n + 1);
    uint64_t g = 0, total_len = 0;
    for (uint32_t i = 0; i < n; i++) {
        c->h_gap_off[i] = (uint32_t)g;
        uint64_t cap = job->bc->h_off[i + 1] - job->bc->h_off[i];
        g += cap / 4 + 64;
        total_len += cap;
    }
    if (g >= (1ull << 31)) {
        set_error("too many potential gaps in one job");
        return GP_ERR_UNSUPPORTED;
    }
    c->h_gap_off[n] = (uint32_t)g;
    c->gap_cap = (uint32_t)g;
    uint32_t tb = 10;
    while ((1ull << tb) < 2 * g)
        tb++;
    c->table_bits = tb;
    c->multi_cap = (uint32_t)std::min<uint64_t>(g, 1u << 20);
    c->pool_cap = 2 * total_len + (16ull << 20);
    // a worker per 4 kbp of contigs, 256 at least, kMaxWorkers from 12 Mbp on (3 GiB of medium arenas + 3.3 GiB of full ones)
    c->n_workers = (uint32_t)std::min<uint64_t>(
        kMaxWorkers, std::max<uint64_t>(256, env_u64("GP_SEALER_WORKERS", total_len / 4096)));
    // (where long contigs are polished every worker gets a full arena, below: 13 MB each.  2048 of them are as
    // fast as 4096 there -- a launch lasts as long as its longest chain of searches -- and leave room for the
    // second back half gp_pipeline runs beside the first: SURVEY 8d config 3's shape 210 -> 137 GiB in use,
    // 5.81 and 5.78 s per step; 1024: Sealer 2.3 -> 3.1 s)
    {
        uint64_t longest0 = 0;
        for (uint32_t i = 0; i < n; i++)
            longest0 = std::max<uint64_t>(longest0, job->h_contig_off[i + 1] - job->h_contig_off[i]);
        if (longest0 >= env_u64("GP_SEALER_LONG", 200000) && !getenv("GP_SEALER_WORKERS"))
            c->n_workers = std::min<uint32_t>(c->n_workers, 2048);
    }
    // Full arenas for every worker where long contigs are polished: their near-saturated filters make
    // searches that neither find a path nor hit the branch limit explore tens of thousands of
    // vertices (160 000 such searches per step on SURVEY 8d's 100 Mbp of megabase contigs; with 256
    // full arenas behind the medium ones Sealer took 6.4 s there instead of 2.0).  Read-length contigs
    // get the two tiers: 6.3 GiB instead of 39.
    uint64_t longest = 0;
    for (uint32_t i = 0; i < n; i++)
        longest = std::max<uint64_t>(longest, job->h_contig_off[i + 1] - job->h_contig_off[i]);
    const uint64_t n_workers = c->n_workers;
    const uint64_t n_big = longest >= env_u64("GP_SEALER_LONG", 200000) ? n_workers : std::min<uint64_t>(n_workers, kBigWorkers);
    c->n_big = (uint32_t)n_big;
    const uint64_t n_med = n_big < n_workers ? n_workers : 1; // (no medium pass when every worker has a full arena)
    int rc;
    if ((rc = c->gaps.alloc(g)) != GP_OK || (rc = c->gap_off.alloc((size_t)n + 1)) != GP_OK ||
        (rc = c->n_gaps.alloc(n)) != GP_OK || (rc = c->pair_last.alloc(g)) != GP_OK ||
        (rc = c->closed_at.alloc(g)) != GP_OK || (rc = c->fill_off.alloc(g)) != GP_OK ||
        (rc = c->fill_len.alloc(g)) != GP_OK || (rc = c->table_left.alloc(1ull << tb)) != GP_OK ||
        (rc = c->table_pair.alloc(1ull << tb)) != GP_OK || (rc = c->work.alloc(g)) != GP_OK ||
        (rc = c->defer.alloc(g)) != GP_OK ||
        (rc = c->small.alloc((uint64_t)kSimtWaves * (c->lanes = sealer_lanes()) * sizeof(SmallArena))) != GP_OK ||
        (rc = c->queue.alloc(8)) != GP_OK || (rc = c->pool.alloc(c->pool_cap)) != GP_OK ||
        (rc = c->pool_head.alloc(1)) != GP_OK || (rc = c->multi.alloc(c->multi_cap)) != GP_OK ||
        (rc = c->slots.alloc(n_big * kSlots)) != GP_OK || (rc = c->verts.alloc(n_big * kVertCap)) != GP_OK ||
        (rc = c->edges.alloc(n_big * kEdgeCap)) != GP_OK || (rc = c->scratch.alloc(n_big * kScratchWords)) != GP_OK ||
        (rc = c->m_slots.alloc(n_med << kMedSlotBits)) != GP_OK || (rc = c->m_verts.alloc(n_med * kMedVertCap)) != GP_OK ||
        (rc = c->m_edges.alloc(n_med * kMedEdgeCap)) != GP_OK ||
        (rc = c->m_scratch.alloc(n_med * kMedScratchWords)) != GP_OK || (rc = c->m_epoch.alloc(n_med)) != GP_OK ||
        (rc = c->defer2.alloc(g)) != GP_OK ||
        (rc = c->nw.alloc((uint64_t)kNwWorkers *
                          (3ull * (kNwMax + 1) * (kNwMax + 1) + 2 * (2 * kNwMax + 64) / 4 + 64))) != GP_OK ||
        (rc = c->epoch.alloc(n_big)) != GP_OK)
        return rc;
    // (on the context's stream, not the null stream: see stage_bc_prepare)
    hipStream_t sq = job->ctx->stream;
    GP_HIP(hipMemcpyAsync(c->gap_off.p, c->h_gap_off.data(), (n + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, sq));
    GP_HIP(hipMemsetAsync(c->slots.p, 0, n_big * kSlots * sizeof(uint32_t), sq));
    GP_HIP(hipMemsetAsync(c->epoch.p, 0, n_big * sizeof(uint32_t), sq));
    GP_HIP(hipMemsetAsync(c->m_slots.p, 0, (n_med << kMedSlotBits) * sizeof(uint32_t), sq));
    GP_HIP(hipMemsetAsync(c->m_epoch.p, 0, n_med * sizeof(uint32_t), sq));
    GP_HIP(hipStreamSynchronize(sq));
    return GP_OK;
}

void stage_c_release(gp_job* job)
{
    if (job->bc && job->bc->c_state) {
        delete (CState*)job->bc->c_state;
        job->bc->c_state = nullptr;
    }
}

int stage_c_run(gp_job* job)
{
    StageBC* b = job->bc;
    CState* c = cstate(job);
    const gp_params& p = job->prm;
    if (!c || job->n_contigs == 0)
        return GP_OK;
    hipStream_t s = job->ctx->stream;
    CArgs a{};
    a.seq = b->seq[b->cur].p;
    a.out = b->seq[b->cur ^ 1].p;
    a.off = b->off.p;
    a.len_cur = b->len[b->cur].p;
    a.len_next = b->len[b->cur ^ 1].p;
    a.dropped = b->dropped.p;
    a.contig_batch = b->contig_batch.p;
    a.bf = job->d_bf.p;
    a.n_contigs = job->n_contigs;
    a.n_k = p.n_k;
    a.hash_num = p.hash_num;
    a.bf_words = (uint32_t)(p.bf_bytes / 4);
    a.bf_mod = make_fastmod(p.bf_bytes * 8);
    a.flank_len = p.k[0]; // -L$(firstword K)
    a.max_gap = p.sealer_max_gap;
    a.max_paths = p.sealer_max_paths;
    a.max_branches = p.sealer_max_branches;
    a.max_cost = p.sealer_max_cost;
    a.gaps = c->gaps.p;
    a.gap_off = c->gap_off.p;
    a.n_gaps = c->n_gaps.p;
    a.pair_last = c->pair_last.p;
    a.closed_at = c->closed_at.p;
    a.fill_off = c->fill_off.p;
    a.fill_len = c->fill_len.p;
    a.table_left = c->table_left.p;
    a.table_pair = c->table_pair.p;
    a.table_bits = c->table_bits;
    a.work = c->work.p;
    a.defer = c->defer.p;
    a.defer2 = c->defer2.p;
    a.m_slots = c->m_slots.p;
    a.m_verts = c->m_verts.p;
    a.m_edges = c->m_edges.p;
    a.m_scratch = c->m_scratch.p;
    a.m_epoch = c->m_epoch.p;
    a.queue = c->queue.p;
    a.pool = c->pool.p;
    a.pool_head = c->pool_head.p;
    a.pool_cap = c->pool_cap;
    a.multi = c->multi.p;
    a.multi_cap = c->multi_cap;
    a.slots = c->slots.p;
    a.verts = c->verts.p;
    a.edges = c->edges.p;
    a.scratch = c->scratch.p;
    a.nw = c->nw.p;
    a.epoch = c->epoch.p;
    a.err = b->err.p;
    a.stats = b->stats.p;
    a.flags = b->flags.p;
    a.simt_lanes = c->lanes;

    GP_HIP(hipMemsetAsync(c->queue.p, 0, 8 * sizeof(uint32_t), s));
    GP_HIP(hipMemsetAsync(c->pool_head.p, 0, sizeof(unsigned long long), s));
    GP_HIP(hipMemsetAsync(c->table_left.p, 0, (1ull << c->table_bits) * sizeof(uint32_t), s));
    GP_HIP(hipMemsetAsync(c->table_pair.p, 0, (1ull << c->table_bits) * sizeof(uint32_t), s));
    GP_HIP(hipMemsetAsync(c->pair_last.p, 0, (uint64_t)c->gap_cap * sizeof(uint32_t), s));
    int rc = job->t_bfs.begin(s, job->n_contigs);
    if (rc != GP_OK)
        return rc;
    if ((rc = launch_classify(job)) != GP_OK || (rc = launch_copy_flagged(job, 2u)) != GP_OK)
        return rc;
    hipLaunchKernelGGL(find_gaps, dim3(job->n_contigs), dim3(kGapThreads), 0, s, a);
    hipLaunchKernelGGL(prep_gaps, dim3(job->n_contigs), dim3(256), 0, s, a);
    for (uint32_t ki = 0; ki < p.n_k; ki++) {
        a.ki = ki;
        a.k = p.k[ki];
        // queue heads back to 0, counts stay
        GP_HIP(hipMemsetAsync(c->queue.p + 1, 0, sizeof(uint32_t), s));
        GP_HIP(hipMemsetAsync(c->queue.p + 2, 0, 2 * sizeof(uint32_t), s));
        GP_HIP(hipMemsetAsync(c->queue.p + 4, 0, 4 * sizeof(uint32_t), s));
        hipLaunchKernelGGL(sealer_connect_lanes, dim3(kSimtWaves), dim3(64), 0, s, a,
                           reinterpret_cast<SmallArena*>(c->small.p));
        if (c->n_big < c->n_workers) {
            a.tier = 0;
            hipLaunchKernelGGL(sealer_connect, dim3(c->n_workers), dim3(64), 0, s, a);
            a.tier = 1;
            hipLaunchKernelGGL(sealer_connect, dim3(c->n_big), dim3(64), 0, s, a);
        } else {
            a.tier = 2;
            hipLaunchKernelGGL(sealer_connect, dim3(c->n_big), dim3(64), 0, s, a);
        }
        a.tier = 0;
        hipLaunchKernelGGL(sealer_consensus, dim3(kNwWorkers), dim3(64), 0, s, a);
    }
    hipLaunchKernelGGL(apply_fills, dim3(job->n_contigs), dim3(kApplyThreads), 0, s, a);
    GP_HIP(hipGetLastError());
    if ((rc = job->t_bfs.end(s)) != GP_OK)
        return rc;
    if (getenv("GP_SEALER_DEBUG")) {
        // development aid: every closed gap of every contig, as the reference's ClosedGap
        GP_HIP(hipStreamSynchronize(s));
        std::vector<uint32_t> n_g(job->n_contigs), g_off(job->n_contigs + 1);
        GP_HIP(hipMemcpy(n_g.data(), c->n_gaps.p, n_g.size() * 4, hipMemcpyDeviceToHost));
        GP_HIP(hipMemcpy(g_off.data(), c->gap_off.p, g_off.size() * 4, hipMemcpyDeviceToHost));
        for (uint32_t cid = 0; cid < job->n_contigs; cid++) {
            std::vector<Gap> gs(n_g[cid] ? n_g[cid] : 1);
            std::vector<uint32_t> fl(gs.size()), ca(gs.size());
            std::vector<uint64_t> fo(gs.size());
            if (!n_g[cid])
                continue;
            GP_HIP(hipMemcpy(gs.data(), c->gaps.p + g_off[cid], n_g[cid] * sizeof(Gap), hipMemcpyDeviceToHost));
            GP_HIP(hipMemcpy(fl.data(), c->fill_len.p + g_off[cid], n_g[cid] * 4, hipMemcpyDeviceToHost));
            GP_HIP(hipMemcpy(ca.data(), c->closed_at.p + g_off[cid], n_g[cid] * 4, hipMemcpyDeviceToHost));
            GP_HIP(hipMemcpy(fo.data(), c->fill_off.p + g_off[cid], n_g[cid] * 8, hipMemcpyDeviceToHost));
            for (uint32_t i = 0; i < n_g[cid]; i++) {
                std::string f(fl[i], ' ');
                if (fl[i])
                    GP_HIP(hipMemcpy(&f[0], c->pool.p + fo[i], fl[i], hipMemcpyDeviceToHost));
                fprintf(stderr, "DEVGAP contig %u gap %u s %u e %u fl %u closed_at %u fill %.40s...%s\n", cid, i, gs[i].s,
                        gs[i].e, fl[i], ca[i], f.c_str(), fl[i] > 12 ? f.c_str() + fl[i] - 12 : "");
            }
        }
    }
    b->cur ^= 1;
    return GP_OK;
}

} // namespace gp
