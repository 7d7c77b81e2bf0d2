// stage_a.hip -- stage A of the path on MI355X: ntHash every k-mer of every
// mapped read and build the per-batch Bloom filters through the counting
// filter threshold.
//
// What it replaces: fill_bfs (/root/reference/src/utils.cpp:96-123) as driven
// by serve_batch (/root/reference/src/goldpolish_targeted_bfs.cpp:55-146).
// The reference feeds one counting filter per (batch, k) *serially*: reads in
// rank order, k-mers left to right, each doing
//     c = min of its 4 counters; if (c < thr) { counters == c become c+1; c++ }
//     if (c >= thr) insert the k-mer into the batch's Bloom filter
// which makes the result depend on that order.  The same result is produced
// here in parallel:
//
//   * "position in the concatenated read buffer" IS the serial order, so an
//     operation's identity is (k index, position).
//   * Counters never decrease, and no operation lifts a counter past its own
//     threshold.  Contigs of one batch share the filters but not the threshold
//     (serve_batch works it out per target, goldpolish_targeted_bfs.cpp:126-127), so for
//     an operation let thr be its own threshold and S the largest threshold of itself and
//     every LATER operation on the same filter (seg_smax: suffix maximum over the batch's
//     contigs that have reads; S = thr when bsize = 1).  An operation is "settled" -- its
//     k-mer goes in, nothing is written, no turn is needed -- when it sees, at ANY time,
//     even ahead of its turn, even from a stale read (a read never shows the future, and
//     values only grow), either
//       (a) all its counters >= thr: that is what the serial update does with them, or
//       (b) all its counters >= S-1: it and every later operation insert whatever happens,
//           and the only writes left are lifts from S-1 to S, which no decision of a
//           later operation (all of them have thresholds <= S) can tell apart.
//     An operation whose minimum is exactly its own thr-1 while a later contig has a larger
//     threshold is NOT settled: the serial update lifts those counters to thr and the later
//     contig's operations see the difference; it takes its turn like everything below thr-1.
//   * Every other operation takes a deterministic reservation: atomicMin of
//     (step tag, position) into a hashed slot for each of its counters; in the
//     next step the operations that hold all their slots are exactly those
//     with no earlier unfinished operation on any of their counters, so they
//     apply the serial update; the rest retry (round_op).  Tags shrink per
//     step (newer beats stale), two tables alternate.
//   * The work is a stream of steps.  Step s = cbf_round over the waiting list
//     (everything that has not had its turn yet) beside cbf_first_pass over the
//     next slice of read positions, which appends its waiting operations to the
//     list step s+1 consumes.  The list is never drained between slices, so
//     there is no per-slice tail of small latency-bound launches; only at the end
//     of a group of batches a few more rounds and cbf_tail (one workgroup looping
//     with L1-bypassing loads) finish what is left.  No launch waits for the
//     host; feedback (list length, waiting fraction) returns through mapped host
//     memory and steers the slice size.
//   * cbf_check_sat marks (batch, k) pairs whose counters and filter can no
//     longer change; their k-mers then cost nothing.
//
// Memory behaviour: the sequence is streamed once per k through LDS (coalesced
// 1 B/base); the counter / bit accesses are 1-byte random accesses into 10 MiB
// (counters) and 512 KiB (bits) per (batch, k) -- random-transaction and
// atomic bound, no reuse to tile for.  Integer / bitwise work only.
#include "gp_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <thread>

namespace gp {

constexpr int kTile = 1024;        // positions per workgroup
constexpr int kPerThread = 4;      // consecutive positions per lane
constexpr int kBlock = kTile / kPerThread;
constexpr int kHalo = 64;          // k <= 64
constexpr int kRoundBlocks = 2048;
constexpr uint32_t kRing = 16;         // feedback slots in mapped host memory
constexpr uint32_t kAhead = 8;         // steps the host may be ahead of the latest report
constexpr uint32_t kSteps = 16;        // ring of per-step list counters
constexpr uint32_t kDrainRounds = 12;  // grid-wide rounds launched before the tail (< kSteps - 2)
constexpr uint32_t kCountStride = 32;  // one 128-B line per step: word 0 = list length, word 1 = waiting operations the first pass added

struct AArgs {
    const uint8_t* reads;
    uint64_t n_pos;
    const uint64_t* seg_start;
    const uint32_t* seg_batch;
    const uint32_t* seg_thr;
    uint32_t n_seg;
    uint8_t* cbf;
    uint32_t* bf;
    uint32_t n_k, hash_num;
    uint64_t cbf_bytes;
    uint64_t cbf_stride; // bytes per counter array (one arena slot) in memory
    const uint32_t* batch_slot; // per batch: first arena slot of its n_k counter arrays while its group is worked on
    uint32_t* queue;     // cbf_local: next (batch, k) pair to replay
    // census (see census_scatter): per queue entry of cbf_local
    const uint32_t* pair_mode;              // 1: only the operations marked in `cand` are replayed, thresholds one lower (2: census overflowed; 3: those marked in `cand2`)
    const uint64_t* pair_bit_off;           // first bit of the pair in `cand` (multiple of 64)
    const unsigned long long* cand;         // bit (pair_bit_off + position in the batch's reads)
    const unsigned long long* cand2;        // pair_mode 3: the candidates left for the replay once the private k-mers have been counted (ckc_apply)
    // census pairs: the k-mers cbf_local has seen settled for good (see there), a table per workgroup
    unsigned long long* kset;
    uint32_t kset_mask;                     // entries per table - 1 (0: none)
    uint32_t nib;        // 1: two 4-bit counters per byte (every threshold <= 15)
    uint32_t ctr_vbits;  // cbf_local: bits of a counter byte that hold the value; the rest is the pair's epoch tag (0: no tags)
    uint32_t bf_words;
    FastMod cbf_mod, bf_mod;
    KConsts kc[GP_MAX_K_VALUES];
    // reservation slots, (step tag << 32) | position key; two tables alternate so that
    // the reservations a launch makes never disturb the ones it is still checking
    unsigned long long* resv;      // table the incoming list reserved in
    unsigned long long* resv_out;  // table this launch reserves in
    uint64_t tag_cur;    // 32-bit tag the incoming list reserved with
    uint64_t tag_next;   // tag for reservations made by this launch (always smaller: newer wins)
    uint32_t* sat;       // per (batch, k): bit 0 = every counter >= max thr-1, bit 1 = filter all ones
    const uint32_t* batch_thr_max; // per batch: largest kmer_threshold of its contigs that have reads
    const uint32_t* batch_seg;     // [n_batches+1] first contig (segment) of each batch
    const uint32_t* seg_smax;      // per contig: largest kmer_threshold of it and the later contigs of its batch (with reads)
    uint32_t tail_max;   // while draining: lists up to this long are left to the tail kernel
    uint32_t bits_in;    // log2 slots of the table being checked (what the previous step reserved with)
    uint32_t bits_out;   // log2 slots this step reserves with (<= the allocation), see stage_a_run
    uint32_t all_wait;   // 1: nearly every operation waits anyway; the first pass skips the counter look-up
    uint32_t drain;      // 1: end of a group, cbf_tail follows (no feedback, counters left for it)
    uint32_t pos_base;   // slice number << 24: orders operations of different slices
    uint4* list_in;
    uint4* list_out;
    uint32_t* count_in;   // counter line of the list this step consumes
    uint32_t* count_out;  // ... of the list it produces (cbf_round and cbf_first_pass both append)
    uint32_t* count_zero; // the line the previous step consumed, cleared for reuse (or null)
    volatile uint32_t* host_ring; // mapped host memory: 4 words per slot {seq, list length, fresh waiting, -}
    uint32_t ring_slot, ring_seq;
    // a window = the same slice [off, off + slice) of every batch of a group (batch-relative
    // positions), one or more 1024-position tiles per batch
    const uint64_t* batch_pos; // [n_batches+1] first read position of each batch
    const uint32_t* batch_order; // batches in the order the groups take them (long ones first, then the cbf_local ones)
    uint32_t grp_first, grp_n, tiles_per_batch;
    uint64_t off, slice;
    unsigned long long* stats; // [0] positions [1] ops [2] ordered ops [3] tail / local rounds [4] reservation atomics [5] replayed [6] settled-table hits [7] candidate operations settled by their k-mer's count (ckc_apply)
};

__device__ inline uint32_t resv_slot(const AArgs& a, uint32_t bits, uint32_t stream, uint32_t idx)
{
    uint64_t key = (uint64_t)stream * a.cbf_bytes + idx;
    return (uint32_t)((key * 0x9E3779B97F4A7C15ULL) >> (64 - bits));
}

// The counting filters live in an ARENA of slots, not one array per (batch, k) pair for the whole
// job: only the Bloom filters are results, a counting filter is scratch for as long as its pair
// is being replayed (the reference frees its own at the end of serve_batch).  A group of the
// grid-wide schedule holds n_k slots per batch (batch_slot); a cbf_local workgroup owns one slot
// for the whole launch.  100 Mbp in 25 kbp contigs = 16 000 pairs = 168 GB of counters at once
// before; 10 GiB of slots now, whatever the job's size.
__device__ inline uint8_t* cbf_of(const AArgs& a, uint32_t batch, uint32_t ki)
{
    return a.cbf + (uint64_t)(a.batch_slot[batch] + ki) * a.cbf_stride;
}

// Counters never pass their stream's threshold, and GoldPolish's thresholds are <= 14, so
// two counters can share a byte (GP_STAGE_A_NIBBLE=1): the 10 MiB per filter become 5 MiB,
// most of which lives in an XCD's 4 MiB L2.  Default and thresholds above 15: one byte each.
template <bool FRESH>
__device__ inline uint32_t ctr_load(const AArgs& a, const uint8_t* cb, uint32_t idx)
{
    if (a.nib) {
        uint8_t b = FRESH ? __hip_atomic_load(cb + (idx >> 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                          : cb[idx >> 1];
        return (b >> ((idx & 1u) * 4u)) & 15u;
    }
    return FRESH ? __hip_atomic_load(cb + idx, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : cb[idx];
}

// counter idx: v -> v + 1 (only ever called by the one operation that holds the counter)
__device__ inline void ctr_increment(const AArgs& a, uint8_t* cb, uint32_t idx, uint32_t v)
{
    if (a.nib) {
        // the other nibbles of the word may be incremented by other holders in this round
        atomicAdd(reinterpret_cast<uint32_t*>(cb) + (idx >> 3), 1u << ((idx & 7u) * 4u));
    } else {
        cb[idx] = (uint8_t)(v + 1);
    }
}

__device__ inline void bf_insert(const AArgs& a, uint32_t stream, const uint64_t* hv)
{
    uint32_t* w = a.bf + (uint64_t)stream * a.bf_words;
    // the words first, all in flight together, then the bits that are missing
    uint32_t* p[kMaxHash];
    uint32_t bit[kMaxHash], have[kMaxHash];
#pragma unroll
    for (int i = 0; i < kMaxHash; i++)
        if ((uint32_t)i < a.hash_num) {
            const uint64_t n = fastmod(hv[i], a.bf_mod);
            bit[i] = 1u << (uint32_t)(n & 31);
            p[i] = w + (n >> 5);
            have[i] = *p[i];
        }
#pragma unroll
    for (int i = 0; i < kMaxHash; i++)
        if ((uint32_t)i < a.hash_num && !(have[i] & bit[i]))
            atomicOr(p[i], bit[i]); // a stale read only costs a redundant atomic: bits are never cleared
}

// A reservation is (tag, position) packed so that atomicMin prefers the newer round first
// and the earlier position second; stale entries of older rounds never need clearing.
__device__ inline unsigned long long resv_word(uint64_t tag, uint32_t pos)
{
    return ((unsigned long long)tag << 32) | pos;
}

__device__ inline void reserve(const AArgs& a, uint32_t stream, const uint32_t* idx,
                               uint32_t pos)
{
    const unsigned long long w = resv_word(a.tag_next, pos);
    for (uint32_t i = 0; i < a.hash_num; i++)
        atomicMin(&a.resv_out[resv_slot(a, a.bits_out, stream, idx[i])], w);
}


// A workgroup appends `n_mine` items per lane with ONE global atomic.  Every lane of the
// workgroup must call it (n_mine may be 0).  Returns the index of the lane's first item.
// (Per-wave appends to one word serialise at about 90 per microsecond chip-wide; with four
// lists of a window in flight that was a third of a window's time.)
template <int THREADS>
__device__ inline uint32_t block_append(uint32_t* counter, uint32_t n_mine, uint32_t* s_wave /*[THREADS/64 + 1]*/,
                                        uint32_t* also = nullptr, unsigned long long* resv_stat = nullptr,
                                        uint32_t resv_each = 0)
{
    constexpr int kWaves = THREADS / 64;
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    uint32_t incl = n_mine;
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t o = __shfl_up(incl, d);
        if (lane >= d)
            incl += o;
    }
    if (lane == 63)
        s_wave[wave] = incl;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < kWaves; w++) {
            uint32_t c = s_wave[w];
            s_wave[w] = tot;
            tot += c;
        }
        s_wave[kWaves] = tot ? atomicAdd(counter, tot) : 0u;
        if (tot && also)
            atomicAdd(also, tot);
        if (tot && resv_stat)
            atomicAdd(resv_stat, (unsigned long long)tot * resv_each);
    }
    __syncthreads();
    const uint32_t at = s_wave[kWaves] + s_wave[wave] + incl - n_mine;
    __syncthreads(); // s_wave may be reused by the caller's next call
    return at;
}

// ---- first pass over a window: hash, settle what can be settled, reserve the rest
__global__ __launch_bounds__(kBlock) void cbf_first_pass(AArgs a)
{
    __shared__ uint8_t codes[kTile + kHalo];
    __shared__ uint64_t seed_t[kSeedTableWords];
    __shared__ unsigned long long s_stats[3];
    __shared__ uint32_t s_wave[kBlock / 64 + 1];
    const uint32_t ki = blockIdx.y;
    const KConsts& kc = a.kc[ki];
    const uint32_t k = kc.k;
    const uint32_t gb = a.batch_order[a.grp_first + blockIdx.x / a.tiles_per_batch];
    const uint64_t b_start = a.batch_pos[gb], b_end = a.batch_pos[gb + 1];
    const uint64_t tile0 = b_start + a.off + (uint64_t)(blockIdx.x % a.tiles_per_batch) * kTile;
    const uint64_t tile_end = min(min(b_end, b_start + a.off + a.slice), tile0 + (uint64_t)kTile);
    if (tile0 >= tile_end)
        return; // this batch has no reads left at this offset

    if (threadIdx.x < 3)
        s_stats[threadIdx.x] = 0;
    fill_seed_table(seed_t);
    for (int i = threadIdx.x; i < kTile + kHalo; i += kBlock) {
        uint64_t p = tile0 + i;
        uint8_t c = p < a.n_pos ? a.reads[p] : (uint8_t)'\n';
        codes[i] = base_code(c);
    }
    __syncthreads();

    const int t0 = threadIdx.x * kPerThread;
    const uint64_t pos0 = tile0 + t0;
    uint32_t n_ops = 0, n_ordered = 0, n_positions = 0;
    // what the lane's positions turned into: 0 nothing left to do, 1 counters looked up, 2 filter only
    uint32_t op_state[kPerThread] = {}, op_seg[kPerThread] = {}, op_min[kPerThread] = {};
    uint32_t op_idx[kPerThread][4] = {};
    uint64_t op_base[kPerThread] = {};
    if (pos0 < tile_end) {
        // segment (= contig's read block) that holds pos0
        uint32_t lo = 0, hi = a.n_seg;
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (a.seg_start[mid] <= pos0)
                lo = mid;
            else
                hi = mid;
        }
        uint32_t seg = lo;
        uint64_t seg_end = a.seg_start[seg + 1];

        // all of this lane's positions in one saturated filter pair: nothing to hash or
        // look up, only the operation count is kept (k-mers without a non-ACGT byte)
        const bool saturated = pos0 + kPerThread <= seg_end &&
                               a.sat[a.seg_batch[seg] * a.n_k + ki] == 3u;
        if (saturated) {
            int lb = -1;
            for (uint32_t i = 0; i + 1 < k; i++)
                if (codes[t0 + i] >= 4)
                    lb = (int)i;
            for (int j = 0; j < kPerThread; j++) {
                const uint64_t pos = pos0 + j;
                if (pos >= tile_end)
                    break;
                if (codes[t0 + j + k - 1] >= 4)
                    lb = j + (int)k - 1;
                n_positions++;
                if (lb >= j || pos + k > a.n_pos)
                    continue;
                n_ops++;
            }
        }

        // hash of the first window; bad bytes contribute 0 so rolling stays exact
        uint64_t fh = 0, rh = 0;
        int last_bad = -1;
        if (!saturated)
            kmer_hash_from_table(seed_t, codes + t0, k, fh, rh, last_bad);
        // pass 1: hashes and counter loads of all of the lane's positions (the loads of the
        // four positions are independent, so they are in flight together)
#pragma unroll
        for (int j = 0; j < kPerThread; j++) {
            const uint64_t pos = pos0 + j;
            if (saturated || pos >= tile_end)
                continue;
            if (j > 0) {
                uint32_t out = codes[t0 + j - 1], in = codes[t0 + j + k - 1];
                uint64_t s_in = in < 4 ? seed_of_code(in) : 0ULL;
                uint64_t s_outk = out < 4 ? kc.out_fwd[out] : 0ULL;
                uint64_t r_ink = in < 4 ? kc.in_rev[in] : 0ULL;
                uint64_t r_out = out < 4 ? seed_of_code(out ^ 2) : 0ULL;
                fh = srol1(fh) ^ s_in ^ s_outk;
                rh = sror1(rh ^ r_ink ^ r_out);
                if (in >= 4)
                    last_bad = j + (int)k - 1;
            }
            n_positions++;
            if (last_bad >= j || pos + k > a.n_pos)
                continue;
            while (pos >= seg_end) {
                seg++;
                seg_end = a.seg_start[seg + 1];
            }
            const uint32_t stream = a.seg_batch[seg] * a.n_k + ki;
            n_ops++;
            // a saturated filter pair answers for itself: every counter is already high enough
            // for any k-mer of the batch to go in, and every bit it could set is set (cbf_check_sat)
            const uint32_t sat = a.sat[stream];
            if (sat == 3u)
                continue;
            op_base[j] = fh + rh;
            op_seg[j] = seg;
            op_state[j] = (sat & 1u) ? 2u : 1u; // 2: counters saturated, only the filter is left
            if (op_state[j] == 1u) {
                const uint8_t* cb = cbf_of(a, a.seg_batch[seg], ki);
                uint32_t m = a.all_wait ? 0u : 255u;
                for (uint32_t h = 0; h < a.hash_num; h++) {
                    const uint64_t hv = h ? ext_hash(op_base[j], k, h) : op_base[j];
                    const uint32_t ix = (uint32_t)fastmod(hv, a.cbf_mod);
                    if (h < 4)
                        op_idx[j][h] = ix;
                    if (!a.all_wait)
                        m = min(m, ctr_load<false>(a, cb, ix));
                }
                op_min[j] = m;
            }
        }
        // pass 2: settle what can be settled (top of the file); count what has to wait
#pragma unroll
        for (int j = 0; j < kPerThread; j++) {
            if (!op_state[j])
                continue;
            const uint32_t seg_j = op_seg[j];
            const uint32_t batch_j = a.seg_batch[seg_j];
            const uint32_t stream = batch_j * a.n_k + ki;
            const uint32_t thr = a.seg_thr[seg_j] - 2 + ki, smax = a.seg_smax[seg_j] - 2 + ki;
            if (op_state[j] == 2u || op_min[j] >= thr || op_min[j] + 1 >= smax) { // settled, see the top of the file
                uint64_t hv[kMaxHash];
                hv[0] = op_base[j];
                for (uint32_t i = 1; i < a.hash_num; i++)
                    hv[i] = ext_hash(op_base[j], k, i);
                bf_insert(a, stream, hv);
                op_state[j] = 0;
            } else {
                n_ordered++;
            }
        }
    }
    // pass 3: one list append per workgroup, then reservations + records
    {
        uint32_t at = block_append<kBlock>(a.count_out, n_ordered, s_wave, a.count_out + 1, a.stats + 4, a.hash_num);
#pragma unroll
        for (int j = 0; j < kPerThread; j++) {
            if (!op_state[j])
                continue;
            const uint32_t seg_j = op_seg[j];
            const uint32_t stream = a.seg_batch[seg_j] * a.n_k + ki;
            // order key: tiles of a batch are consecutive blocks in increasing position order
            const uint32_t rel = a.pos_base | (blockIdx.x * (uint32_t)kTile + (uint32_t)(t0 + j));
            uint32_t idx[kMaxHash];
            for (uint32_t h = 0; h < a.hash_num; h++)
                idx[h] = h < 4 ? op_idx[j][h] : (uint32_t)fastmod(ext_hash(op_base[j], k, h), a.cbf_mod);
            reserve(a, stream, idx, rel);
            a.list_out[2 * at] = make_uint4((uint32_t)op_base[j], (uint32_t)(op_base[j] >> 32), rel,
                                            (seg_j << 3) | ki);
            a.list_out[2 * at + 1] = make_uint4(op_idx[j][0], op_idx[j][1], op_idx[j][2], op_idx[j][3]);
            at++;
        }
    }
    atomicAdd(&s_stats[0], (unsigned long long)n_positions);
    atomicAdd(&s_stats[1], (unsigned long long)n_ops);
    atomicAdd(&s_stats[2], (unsigned long long)n_ordered);
    __syncthreads();
    if (threadIdx.x < 3 && s_stats[threadIdx.x])
        atomicAdd(&a.stats[threadIdx.x], s_stats[threadIdx.x]);
}

// gp_debug_nthash, variants 0 and 1: the hashing of the stage-A kernels over one sequence, by the same
// primitives in the same arrangement -- a lane starts its run of kPerThread positions from the table of
// rotated seeds (kmer_hash_from_table) and rolls from there; variant 0 rolls and extends the way
// cbf_first_pass / cbf_local do (select chains, ext_hash), variant 1 the way census_scatter does (the
// roll tables in LDS, hashes 1-3 from one multiply: ExtHash3).  tests/test_gpu_nthash.py compares
// them with the known answers made from the reference's own header (tests/golden/nthash_kat.json).
__global__ __launch_bounds__(kBlock) void debug_nthash_a(const uint8_t* seq, uint64_t len, KConsts kc, uint32_t hash_num,
                                                         int variant, uint64_t* out_f, uint64_t* out_r,
                                                         uint64_t* out_h, uint8_t* valid)
{
    __shared__ uint8_t codes[kTile + kHalo];
    __shared__ uint64_t seed_t[kSeedTableWords];
    __shared__ ulonglong2 roll_in[8], roll_out[8];
    const uint32_t k = kc.k;
    const uint64_t tile0 = (uint64_t)blockIdx.x * kTile;
    fill_seed_table(seed_t);
    if (threadIdx.x < 8) {
        const uint32_t cd = threadIdx.x;
        roll_in[cd] = cd < 4 ? make_ulonglong2(seed_of_code(cd), kc.in_rev[cd]) : make_ulonglong2(0, 0);
        roll_out[cd] = cd < 4 ? make_ulonglong2(kc.out_fwd[cd], seed_of_code(cd ^ 2)) : make_ulonglong2(0, 0);
    }
    for (int i = threadIdx.x; i < kTile + kHalo; i += kBlock) {
        const uint64_t p = tile0 + i;
        codes[i] = base_code(p < len ? seq[p] : (uint8_t)'\n');
    }
    __syncthreads();
    const int t0 = threadIdx.x * kPerThread;
    const ExtHash3 eh = make_ext_hash3(k);
    const uint64_t ext_m = (uint64_t)k * kMultiSeed;
    uint64_t fh = 0, rh = 0;
    int last_bad = -1;
    if (tile0 + t0 + k <= len)
        kmer_hash_from_table(seed_t, codes + t0, k, fh, rh, last_bad);
    for (int j = 0; j < kPerThread; j++) {
        const uint64_t pos = tile0 + t0 + j;
        if (pos + k > len)
            break;
        if (j > 0) {
            const uint32_t out = codes[t0 + j - 1], in = codes[t0 + j + k - 1];
            if (variant == 0) {
                const uint64_t s_in = in < 4 ? seed_of_code(in) : 0ULL;
                const uint64_t s_outk = out < 4 ? kc.out_fwd[out] : 0ULL;
                const uint64_t r_ink = in < 4 ? kc.in_rev[in] : 0ULL;
                const uint64_t r_out = out < 4 ? seed_of_code(out ^ 2) : 0ULL;
                fh = srol1(fh) ^ s_in ^ s_outk;
                rh = sror1(rh ^ r_ink ^ r_out);
            } else {
                const ulonglong2 ri = roll_in[in], ro = roll_out[out];
                fh = srol1(fh) ^ ri.x ^ ro.x;
                rh = sror1(rh ^ ri.y ^ ro.y);
            }
            if (in >= 4)
                last_bad = j + (int)k - 1;
        }
        out_f[pos] = fh;
        out_r[pos] = rh;
        valid[pos] = last_bad < j ? 1 : 0;
        const uint64_t base = fh + rh;
        for (uint32_t h = 0; h < hash_num; h++) {
            uint64_t v;
            if (h == 0)
                v = base;
            else if (variant == 1 && h < 4) {
                const uint64_t t = base * ext_m + ext_addend(base, eh.d[h]);
                v = t ^ (t >> kMultiShift);
            } else
                v = ext_hash(base, k, h);
            out_h[pos * hash_num + h] = v;
        }
    }
}

int debug_nthash_stage_a(hipStream_t s, const uint8_t* d_seq, uint64_t len, uint32_t k, uint32_t hash_num, int variant,
                         uint64_t* d_f, uint64_t* d_r, uint64_t* d_h, uint8_t* d_valid)
{
    KConsts kc{};
    kc.k = k;
    for (uint32_t c = 0; c < 4; c++) {
        kc.out_fwd[c] = sroln(seed_of_code(c), k);
        kc.in_rev[c] = sroln(seed_of_code(c ^ 2), k);
    }
    const uint32_t grid = (uint32_t)((len + kTile - 1) / kTile);
    hipLaunchKernelGGL(debug_nthash_a, dim3(grid), dim3(kBlock), 0, s, d_seq, len, kc, hash_num, variant, d_f, d_r, d_h,
                       d_valid);
    GP_HIP(hipGetLastError());
    return GP_OK;
}

// Has a (batch, k) filter pair stopped changing?  Counters only matter through "is the
// minimum >= thr-1", so once every counter is >= thr-1 (for the largest threshold any contig
// of the batch uses) every later operation settles; once every bit of the Bloom filter is
// set, inserting is a no-op.  One workgroup per stream of the window; flags only ever go
// from 0 to 1, so a first pass that reads them late loses nothing but time.
__global__ __launch_bounds__(1024) void cbf_check_sat(AArgs a)
{
    __shared__ uint32_t bad_c, bad_b;
    const uint32_t stream = a.batch_order[a.grp_first + blockIdx.x / a.n_k] * a.n_k + blockIdx.x % a.n_k;
    uint32_t cur = a.sat[stream];
    if (cur == 3u)
        return;
    if (threadIdx.x == 0)
        bad_c = bad_b = 0;
    __syncthreads();
    const uint32_t batch = stream / a.n_k, ki = stream % a.n_k;
    const uint32_t need = a.batch_thr_max[batch] - 2 + ki - 1; // thr - 1
    if (!(cur & 1u)) {
        const uint8_t* cb = cbf_of(a, batch, ki);
        uint32_t bad = 0;
        if (a.nib) {
            const uint64_t nbytes = a.cbf_bytes / 2; // whole bytes; an odd last counter below
            const uint64_t n16 = nbytes / 16;
            const uint4* p4 = reinterpret_cast<const uint4*>(cb);
            for (uint64_t i = threadIdx.x; i < n16 && !bad; i += blockDim.x) {
                const uint4 v = p4[i];
                const uint32_t w[4] = { v.x, v.y, v.z, v.w };
                for (int q = 0; q < 4; q++)
                    for (int r = 0; r < 8; r++)
                        bad |= ((w[q] >> (4 * r)) & 15u) < need;
            }
            for (uint64_t i = n16 * 16 + threadIdx.x; i < nbytes; i += blockDim.x)
                bad |= ((cb[i] & 15u) < need) | ((uint32_t)(cb[i] >> 4) < need);
            if (threadIdx.x == 0 && (a.cbf_bytes & 1))
                bad |= (cb[nbytes] & 15u) < need;
        } else {
            const uint64_t n16 = a.cbf_bytes / 16;
            const uint4* p4 = reinterpret_cast<const uint4*>(cb);
            for (uint64_t i = threadIdx.x; i < n16 && !bad; i += blockDim.x) {
                const uint4 v = p4[i];
                const uint32_t w[4] = { v.x, v.y, v.z, v.w };
                for (int q = 0; q < 4; q++)
                    for (int r = 0; r < 4; r++)
                        bad |= ((w[q] >> (8 * r)) & 255u) < need;
            }
            for (uint64_t i = n16 * 16 + threadIdx.x; i < a.cbf_bytes; i += blockDim.x)
                bad |= cb[i] < need;
        }
        if (bad)
            atomicOr(&bad_c, 1u);
    }
    if (!(cur & 2u)) {
        const uint32_t* w = a.bf + (uint64_t)stream * a.bf_words;
        uint32_t bad = 0;
        for (uint32_t i = threadIdx.x; i < a.bf_words && !bad; i += blockDim.x)
            bad |= w[i] != 0xFFFFFFFFu;
        if (bad)
            atomicOr(&bad_b, 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t now = cur;
        if (!(cur & 1u) && !bad_c)
            now |= 1u;
        if (!(cur & 2u) && !bad_b)
            now |= 2u;
        if (now != cur)
            a.sat[stream] = now;
    }
}

// loads that must see what other waves of the SAME launch wrote (tail kernel): agent-scope
// atomic loads bypass the CU's vector L1
template <bool FRESH, typename T>
__device__ inline T ld(const T* p)
{
    if (FRESH)
        return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return *p;
}

// One waiting operation in one round.  Returns true when it is still waiting (it has then
// reserved for the next round with tag_next in resv_out).
template <bool FRESH>
__device__ inline bool round_op(const AArgs& a, const uint4 rec, const uint4 rix,
                                const unsigned long long* resv_in, unsigned long long* resv_out,
                                uint64_t tag_cur, uint64_t tag_next, uint32_t bits_in)
{
    const uint64_t base = ((uint64_t)rec.y << 32) | rec.x;
    const uint32_t pos = rec.z;
    const uint32_t ki = rec.w & 7, seg = rec.w >> 3;
    const uint32_t k = a.kc[ki].k;
    const uint32_t batch = a.seg_batch[seg];
    const uint32_t stream = batch * a.n_k + ki;
    const uint32_t thr = a.seg_thr[seg] - 2 + ki;
    const uint32_t smax = a.seg_smax[seg] - 2 + ki;
    uint32_t idx[kMaxHash] = { rix.x, rix.y, rix.z, rix.w };
    // only four indices travel with the record; the rest are recomputed
    for (uint32_t h = 4; h < a.hash_num; h++)
        idx[h] = (uint32_t)fastmod(ext_hash(base, k, h), a.cbf_mod);

    const unsigned long long want = resv_word(tag_cur, pos);
    bool mine = true;
    for (uint32_t h = 0; h < a.hash_num; h++)
        mine &= (ld<FRESH>(&resv_in[resv_slot(a, bits_in, stream, idx[h])]) == want);

    uint8_t* cb = cbf_of(a, batch, ki);
    uint32_t c[kMaxHash], m = 255;
    for (uint32_t h = 0; h < a.hash_num; h++) {
        c[h] = ctr_load<FRESH>(a, cb, idx[h]);
        m = min(m, c[h]);
    }
    bool insert;
    if (mine) {
        // no earlier unfinished operation touches these counters: the serial update
        if (m < thr) {
            for (uint32_t h = 0; h < a.hash_num; h++) {
                bool first = c[h] == m;
                for (uint32_t g = 0; g < h; g++)
                    first &= idx[g] != idx[h]; // a counter hit twice still moves by one
                if (first)
                    ctr_increment(a, cb, idx[h], m);
            }
            m++;
        }
        insert = m >= thr;
    } else if (m >= thr || m + 1 >= smax) {
        insert = true; // settled (top of the file): no counter needs writing
    } else {
        const unsigned long long w = resv_word(tag_next, pos);
        for (uint32_t h = 0; h < a.hash_num; h++)
            atomicMin(&resv_out[resv_slot(a, a.bits_out, stream, idx[h])], w);
        return true;
    }
    if (insert) {
        uint64_t hv[kMaxHash];
        hv[0] = base;
        for (uint32_t h = 1; h < a.hash_num; h++)
            hv[h] = ext_hash(base, k, h);
        bf_insert(a, stream, hv);
    }
    return false;
}

// ---- one reservation round over a long list: every CU
__global__ __launch_bounds__(256) void cbf_round(AArgs a)
{
    __shared__ uint32_t s_wave[256 / 64 + 1];
    const uint32_t n_in = a.count_in[0];
    if (a.drain) {
        if (n_in <= a.tail_max)
            return; // short list: cbf_tail finishes it without further launches
    } else if (blockIdx.x == 0 && threadIdx.x == 0) {
        // feedback for the host, straight into mapped host memory; the sequence number last
        volatile uint32_t* slot = a.host_ring + 4 * a.ring_slot;
        slot[1] = n_in;
        slot[2] = a.count_in[1];
        __threadfence_system();
        slot[0] = a.ring_seq;
        if (a.count_zero) {
            a.count_zero[0] = 0;
            a.count_zero[1] = 0;
        }
    }
    for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < n_in; i0 += gridDim.x * blockDim.x) {
        const uint32_t i = i0 + threadIdx.x;
        uint4 rec = {}, rix = {};
        bool waits = false;
        if (i < n_in) {
            rec = a.list_in[2 * i];
            rix = a.list_in[2 * i + 1];
            waits = round_op<false>(a, rec, rix, a.resv, a.resv_out, a.tag_cur, a.tag_next, a.bits_in);
        }
        const uint32_t at = block_append<256>(a.count_out, waits ? 1u : 0u, s_wave, nullptr, a.stats + 4, a.hash_num);
        if (waits) {
            a.list_out[2 * at] = rec;
            a.list_out[2 * at + 1] = rix;
        }
    }
}

// ---- the tail of a group: one workgroup keeps running rounds on the list the grid-wide
// drain rounds left (the first list no longer than tail_max, or whatever is left after the
// last of them) until nothing waits.  Rounds are separated by workgroup barriers; every load
// of data another wave may have written in an earlier round bypasses L1.
constexpr int kTailThreads = 1024;

struct TailArgs {
    uint4* list[2];               // indexed by step parity
    unsigned long long* resv[2];
    uint32_t* counts;     // the ring of counter lines
    uint64_t tag_max;     // step s checks reservations made with tag_max - s
    uint64_t step0;       // first drain step
    uint32_t n_rounds;    // grid-wide drain rounds that were launched
    uint32_t max_rounds;  // bug trap
    uint32_t* err;
    unsigned long long* stats; // [3] rounds run by the tail
};

__global__ __launch_bounds__(kTailThreads) void cbf_tail(AArgs a, TailArgs t)
{
    __shared__ uint32_t s_n, s_out;
    __shared__ uint64_t s_step;
    if (threadIdx.x == 0) {
        uint32_t r = 0;
        while (r < t.n_rounds && t.counts[((t.step0 + r) % kSteps) * kCountStride] > a.tail_max)
            r++;
        s_n = t.counts[((t.step0 + r) % kSteps) * kCountStride];
        s_step = t.step0 + r;
    }
    __syncthreads();
    uint32_t n = s_n;
    uint64_t step = s_step;
    __syncthreads();
    uint32_t done_rounds = 0;
    while (n > 0) {
        if (threadIdx.x == 0)
            s_out = 0;
        __syncthreads();
        const uint4* in = t.list[step & 1];
        uint4* out = t.list[(step + 1) & 1];
        const uint64_t tag_cur = t.tag_max - step, tag_next = tag_cur - 1;
        for (uint32_t i = threadIdx.x; i < n; i += kTailThreads) {
            uint4 rec, rix;
            {
                // records may have been written by another wave in the previous round
                const unsigned long long* q = reinterpret_cast<const unsigned long long*>(in + 2 * i);
                unsigned long long w0 = ld<true>(q), w1 = ld<true>(q + 1), w2 = ld<true>(q + 2),
                                   w3 = ld<true>(q + 3);
                rec = make_uint4((uint32_t)w0, (uint32_t)(w0 >> 32), (uint32_t)w1, (uint32_t)(w1 >> 32));
                rix = make_uint4((uint32_t)w2, (uint32_t)(w2 >> 32), (uint32_t)w3, (uint32_t)(w3 >> 32));
            }
            // the drain keeps the slot count fixed: bits_in == bits_out for every tail round
            if (round_op<true>(a, rec, rix, t.resv[step & 1], t.resv[(step + 1) & 1], tag_cur,
                               tag_next, a.bits_out)) {
                uint32_t at = atomicAdd(&s_out, 1u);
                out[2 * at] = rec;
                out[2 * at + 1] = rix;
            }
        }
        __threadfence();
        __syncthreads();
        n = s_out;
        step++;
        done_rounds++;
        __syncthreads();
        if (done_rounds > t.max_rounds) {
            if (threadIdx.x == 0)
                atomicOr(t.err, 1u);
            break;
        }
    }
    if (threadIdx.x == 0 && done_rounds)
        atomicAdd(&t.stats[3], (unsigned long long)done_rounds);
    // nothing waits any more: every counter line is free again
    for (uint32_t i = threadIdx.x; i < kSteps * kCountStride; i += blockDim.x)
        t.counts[i] = 0;
}


// ------------------------------------------------------------------ census
// Most operations of a sparse filter are the FIRST to touch one of their counters: that counter
// is 0, so the operation's minimum is 0, it lifts its zero counters to 1 and inserts nothing
// (every threshold is >= 2).  Which counters those are needs no order beyond "first": a counter is
// 0 exactly until its first touch.  Everything else -- the operations NONE of whose counters they
// touch first ("candidates": repeats of true k-mers, 15 % of the operations at k = 32, 33 % at
// k = 20 on 25 kbp contigs with 40x reads) -- sees only counters >= 1 and only moves them among
// levels >= 1, so it is an ordinary conservative-update replay of the candidates alone with every
// counter starting at level 1: thresholds one lower on a zero array.  The census finds the
// candidates without random memory traffic: every counter touch goes as a 4-byte record into
// the bucket of its counter range (1280 ranges of 8192 counters at 10 MiB, a run of 64 slots per
// 8192-position stretch and range: census_scatter), a workgroup per range takes the earliest
// position per counter in LDS and packs the touches that are not it into the head of their run
// (census_mark), and an operation with all its touches marked is a candidate (census_pack).
// cbf_local then replays a fifth to a third of the operations.  A run that overflows (many
// copies of one k-mer within a stretch) sends its pair back to the plain replay.
constexpr int kCSuper = 8;                // tiles of kTile positions per census_scatter workgroup: an 8192-position stretch
constexpr uint32_t kCStretch = kCSuper * kTile;
constexpr uint32_t kCBuckets = 2048;      // most counter ranges per pair (a range is a power of two wide: 1280 of 8192 for 10 MiB)
constexpr uint32_t kCRun = 64;            // records a stretch may leave in one bucket (26 expected at four hashes and 10 MiB)
constexpr uint32_t kCMaxShift = 14;       // counters per range: 14 bits of a record, 64 KB of LDS in census_mark
// Two geometries of the same record layout (the kernels are templates over them): RUN record slots per
// (range, stretch), a ring of RING slots per range in LDS, groups of RING / 2 records per store.
//   narrow: <= 2048 ranges, RUN 64, RING 8: 16-byte groups (half a 32-byte sector: PMC showed 0.59 TB
//           written per configs[2] step for 0.26 TB of records);
//   wide:   <= 1024 ranges (16384 counters each at 10 MiB), RUN 128, RING 16: 32-byte groups, whole
//           sectors, by two neighbouring lanes; census_mark's table is 64 KB, two 1024-thread
//           workgroups per CU.
// Both need RUN x ranges = the same record slots per stretch.
constexpr uint32_t kCBucketsWide = 1024, kCRunWide = 128, kCRingWide = 16;
static_assert(kCBuckets * kCRun == kCBucketsWide * kCRunWide, "the record buffers are sized alike");

// Where the run of (range `bucket`, stretch `stretch`) of a pair of `nst` stretches starts in the pair's
// record space, in runs.  Stretches come in SEGMENTS of 128: [segment][range][stretch in the segment], so
// that the ranges' runs of one stretch -- which census_scatter writes and census_pack reads side by side --
// lie at most 128 runs (64 KB) apart however long the pair is.  With one flat [range][stretch] table a pair
// of 100 M positions (12 000 stretches: the census level by level on a 2.5 Mbp contig) put them 6 MB
// apart -- 640 pages per stretch -- and the level-by-level census took 17 s where the model said 2.  A pair
// of up to 128 stretches has one segment: the layout it always had.
constexpr uint32_t kCSeg = 128;
__host__ __device__ inline uint64_t census_run_index(uint32_t nst, uint32_t n_buckets, uint32_t bucket, uint32_t stretch)
{
    const uint32_t stride = nst < kCSeg ? nst : kCSeg;
    return ((uint64_t)(stretch / kCSeg) * n_buckets + bucket) * stride + stretch % kCSeg;
}
// runs in a pair's record space
__host__ __device__ inline uint64_t census_runs(uint32_t nst, uint32_t n_buckets)
{
    const uint32_t stride = nst < kCSeg ? nst : kCSeg;
    return (uint64_t)((nst + kCSeg - 1) / kCSeg) * n_buckets * stride;
}

struct CensusArgs {
    const uint32_t* wg_pair;       // [workgroups] pair (cohort-local) of a stretch
    const uint64_t* wg_pos0;       // [workgroups] its first read position
    uint32_t n_wg;
    const uint32_t* pair_batch;    // [pairs]
    const uint32_t* pair_ki;       // [pairs]
    const uint32_t* pair_nst;      // [pairs] stretches of the pair
    const uint64_t* pair_rec_off;  // [pairs] first record of the pair: rec[bucket][stretch][kCRun]
    const uint64_t* pair_cnt_off;  // [pairs] first count of the pair: cnt[bucket][stretch]
    const uint64_t* pair_bit_off;  // [pairs] first bit of the pair in `cand` (multiple of 64)
    const uint32_t* pair_queue;    // [pairs] index of the pair in cbf_local's queue (for pair_mode)
    uint32_t* rec;                 // (position in the stretch << 16) | (counter in its range << 2) | hash number; bit 31: marked (census_mark)
    uint8_t* cnt;                  // records a stretch left in a bucket
    uint32_t* overflow;            // [pairs]
    uint32_t* pair_mode;           // [queue entries] set to 2 for pairs that overflowed
    unsigned long long* cand;
    uint32_t bucket_shift;         // a range holds 1 << bucket_shift counters
    uint32_t n_buckets;            // ranges per pair
    uint32_t seed_rows;            // census_scatter: rows of its table of rotated seeds (the largest k)
    // the census repeated level by level (long batches whose filters stay unsaturated: stage_a_levels)
    const unsigned long long* members; // only the operations whose bit is set take part (null: all of them)
    const uint32_t* pair_need;     // [pairs] census passes the pair takes (null: one); a pair is skipped from pass pair_need on
    uint32_t level;                // the pass (0: positions and operations are counted)
    uint32_t mark_prefetch;        // census_mark: slots of every run fetched before its length is known
    unsigned long long* level_count; // census_pack: operations with every touch marked, over the launch (or null)
};

// ---- census 1: every counter touch of an 8192-position stretch into the bucket of its counter
// range.  A stretch has a run of kCRun record slots of its own in every bucket (the stretch and
// the bucket are in the address: 4-byte records, no shared cursor, one hashing pass).  Ranges are
// a power of two wide, so a touch finds its bucket with a shift.  (Measured on 1.2 G positions:
// 43 ms like this; 63-68 ms with the records sorted by bucket in LDS first and every run written
// in one go, which costs a second hashing pass and most of the occupancy; 67 ms with 2048-position
// stretches, whose runs of six records reach memory as partial sectors.)
template <uint32_t RUN, uint32_t RING, int ROUNDS, bool MOD5>
__global__ __launch_bounds__(kBlock) void census_scatter(AArgs a, CensusArgs c)
{
    // MOD5: the filter holds 5 * 2^s counters (GoldPolish's 10 MiB, s < 32) and an operation has four
    // touches: the index arithmetic without the general reduction's code and branches
    // A lane hashes ROUNDS x kPerThread consecutive positions of a staged tile: the k-step start of
    // its run of positions (2k table look-ups) once, a roll per position after that; the ring is
    // emptied after every round (1024 positions, whichever they are: a record carries its position).
    // Device clocks per phase (1000 x 25 kbp, cycles per workgroup = per stretch): staging 12 k
    // (45 k before the words came a tile ahead), starts 32 k / 13 k / 7 k at 1 / 2 / 4 rounds, hashing +
    // slotting 110-120 k -- about 65 instructions per touch, which is what four SIMDs issue in the time:
    // the kernel is bound by instruction issue, not by LDS or memory -- and 27 k for the flushes.
    // Four rounds make the workgroup 15 % shorter and the launch no faster (57.9 against 60.0 ms: fewer
    // workgroups are resident with the larger staged tile), so one round is the default.
    constexpr int PER = kPerThread, SPAN = PER * ROUNDS, TILE = kBlock * SPAN;
    static_assert(kCStretch % TILE == 0, "whole tiles per stretch");
    constexpr uint32_t G = RING / 2, Q = G / 4; // records per group; 16-byte pieces of a group, a lane each
    // Records do not go to memory one by one (a 4-byte store is an L2 request of its own and, with
    // hundreds of MB of half-written runs in flight, leaves L2 as a 32-byte sector of its own: PMC
    // showed 1.35 TB written per configs[2] step for 0.26 TB of records, and the kernel bound by
    // that).  Every range has a ring of RING record slots in LDS; after each round the complete
    // groups of G records are written as 16 bytes per lane.  The window of a range is
    // [base, base + RING) record slots of its run, base a multiple of G: a record whose slot lies in
    // it goes to the ring, any other (a round that brings a range more than the ring holds)
    // straight to memory, so a group is always written by one of the two ways only.
    // (ring, counters and group bases are sized by the pair's ranges, not by the most a pair may
    // have: three workgroups per CU where the static 2048-range arrays left room for two)
    extern __shared__ uint32_t ring[]; // [ranges][RING], then hist[ranges], a spare word, then the table of rotated seeds (c.seed_rows x 8)
    __shared__ uint8_t codes[TILE + kHalo];
    // what a base brings to the two hashes when it enters a k-mer and when it leaves one, by code
    // (codes >= 4: nothing): two 16-byte LDS reads per position instead of four select chains
    __shared__ ulonglong2 roll_in[8], roll_out[8]; // {forward, reverse}
    // per range, one word: records so far (low half; at most 4 x 8192) and, in the high half, the groups
    // of the run that are not the ring's any more (base / G) -- the atomic that hands out a record's
    // rank brings the window along (a look-up less per touch)
    uint32_t* const hist = ring + (size_t)c.n_buckets * RING;
    const uint32_t spare = c.n_buckets * (RING + 1u); // a word nobody reads
    // the seeds rotated by 0 .. (largest k) - 1 only: 2 KB at k = 32 instead of the 4 KB of 64 rows, which
    // is what lets two of these workgroups and one of census_mark's (64 KB) share a CU's 160 KB
    uint64_t* const seed_t = reinterpret_cast<uint64_t*>(ring + ((spare + 2u) & ~1u));
    for (uint32_t i = threadIdx.x; i < c.seed_rows * 8u; i += kBlock) {
        const uint32_t code = i & 7u;
        seed_t[i] = code < 4 ? sroln(seed_of_code(code), i >> 3) : 0ULL;
    }
    uint32_t n_positions = 0, n_ops = 0;
    // A workgroup takes several stretches, one after the other (c.n_wg stretches over the grid): the
    // stores a stretch leaves behind -- its last groups, the run lengths, what was left in the rings --
    // drain while the next one is hashed.  With a workgroup per stretch the CU's LDS stayed taken until
    // they had: the launch went no faster when the workgroup itself was made a fifth shorter.
#pragma nounroll
    for (uint32_t wg = blockIdx.x; wg < c.n_wg; wg += gridDim.x) {
    __syncthreads(); // the stretch before is done with the rings and counters
    const uint32_t pair = c.wg_pair[wg];
    const uint32_t b = c.pair_batch[pair], ki = c.pair_ki[pair];
    const KConsts& kc = a.kc[ki];
    const uint32_t k = kc.k;
    const uint64_t b_start = a.batch_pos[b], b_end = a.batch_pos[b + 1];
    const uint64_t pos0 = c.wg_pos0[wg];
    const uint64_t stretch_end = min(b_end, pos0 + (uint64_t)kCStretch);
    const uint32_t nst = c.pair_nst[pair], stretch = (uint32_t)((pos0 - b_start) / kCStretch);
    uint32_t* const rec = c.rec + c.pair_rec_off[pair];
    const int t0 = threadIdx.x * SPAN;
    const uint32_t shift = c.bucket_shift, lmask = (1u << shift) - 1u, mod5_s = a.cbf_mod.s;
    bool over = false;
    // (the census level by level scatters once, at level 0: its later passes take the records census_mark_seg
    // left marked; a pair without a pass to run is left alone)
    if (c.pair_need && (c.level >= c.pair_need[pair] || c.overflow[pair]))
        continue; // (uniform: one pair per workgroup turn)
    for (uint32_t i = threadIdx.x; i < c.n_buckets; i += kBlock)
        hist[i] = 0;
    if (threadIdx.x < 8) {
        const uint32_t cd = threadIdx.x;
        roll_in[cd] = cd < 4 ? make_ulonglong2(seed_of_code(cd), kc.in_rev[cd]) : make_ulonglong2(0, 0);
        roll_out[cd] = cd < 4 ? make_ulonglong2(kc.out_fwd[cd], seed_of_code(cd ^ 2)) : make_ulonglong2(0, 0);
    }
    const ExtHash3 eh = make_ext_hash3(k);
    const uint64_t ext_m = (uint64_t)k * kMultiSeed;
    const bool ext_plain = eh.d[1] == 1 && eh.d[2] == 2 && eh.d[3] == 3;
    // A tile's bases come as aligned 32-bit words, all of a lane's in flight together, a tile ahead.
    constexpr int NW = (TILE + kHalo) / 4 + 1, WPT = (NW + kBlock - 1) / kBlock;
    uint32_t pre[WPT];
    auto fetch = [&](uint64_t tile0) {
        const uint64_t a0 = tile0 & ~3ull;
#pragma unroll
        for (int q = 0; q < WPT; q++) {
            const int w = (int)threadIdx.x + q * kBlock;
            const uint64_t p = a0 + 4ull * (uint64_t)w;
            uint32_t v = 0x0A0A0A0Au; // '\n' past the end of the reads
            if (w < NW) {
                if (p + 4 <= a.n_pos) {
                    v = *reinterpret_cast<const uint32_t*>(a.reads + p);
                } else {
                    for (int bb = 0; bb < 4; bb++)
                        if (p + bb < a.n_pos)
                            v = (v & ~(0xFFu << (8 * bb))) | ((uint32_t)a.reads[p + bb] << (8 * bb));
                }
            }
            pre[q] = v;
        }
    };
    if (pos0 < stretch_end)
        fetch(pos0);
    for (int st = 0; st < (int)(kCStretch / TILE); st++) {
        const uint64_t tile0 = pos0 + (uint64_t)st * TILE;
        if (tile0 >= stretch_end)
            break; // uniform
        const uint64_t tile_end = min(stretch_end, tile0 + (uint64_t)TILE);
        __syncthreads(); // codes and ring free (and the counters cleared before the first use)
        {
            const int sh = (int)(tile0 & 3ull);
#pragma unroll
            for (int q = 0; q < WPT; q++) {
                const int i0 = 4 * ((int)threadIdx.x + q * kBlock) - sh;
#pragma unroll
                for (int bb = 0; bb < 4; bb++)
                    if (i0 + bb >= 0 && i0 + bb < TILE + kHalo)
                        codes[i0 + bb] = base_code((uint8_t)(pre[q] >> (8 * bb)));
            }
        }
        if (tile0 + (uint64_t)TILE < stretch_end)
            fetch(tile0 + (uint64_t)TILE); // the next tile's bases are on their way while this one is hashed
        __syncthreads();
        const uint64_t lane_pos0 = tile0 + t0;
        const bool lane_on = lane_pos0 < tile_end;
        uint64_t fh = 0, rh = 0;
        int last_bad = -1;
        if (lane_on)
            kmer_hash_from_table(seed_t, codes + t0, k, fh, rh, last_bad);
        // cbf_local only visits the candidates of a census pair: the counters of positions
        // streamed and of operations (valid k-mers) are kept here
        n_positions += lane_on ? (uint32_t)min((uint64_t)SPAN, tile_end - lane_pos0) : 0u;
        for (int rd = 0; rd < ROUNDS; rd++) {
            if (rd > 0)
                __syncthreads(); // the round before has been flushed
#pragma unroll
            for (int j = 0; j < PER; j++) {
                const int jj = rd * PER + j;
                const uint64_t pos = lane_pos0 + jj;
                if (!lane_on || pos >= tile_end)
                    break;
                if (jj > 0) {
                    const uint32_t out = codes[t0 + jj - 1], in = codes[t0 + jj + k - 1];
                    const ulonglong2 ri = roll_in[in], ro = roll_out[out];
                    fh = srol1(fh) ^ ri.x ^ ro.x;
                    rh = sror1(rh ^ ri.y ^ ro.y);
                    if (in >= 4)
                        last_bad = jj + (int)k - 1;
                }
                if (!(last_bad < jj && pos + k <= a.n_pos))
                    continue;
                n_ops++;
                const uint64_t base = fh + rh;
                const uint32_t off = (uint32_t)(st * TILE + t0 + jj); // position in the stretch
                // the operation's touches side by side: their indices, then their LDS atomics in flight
                // together, then their records (one touch after the other the loop was a chain of
                // multiply, reduction, atomic, wait and store per touch, at three waves per SIMD)
                const uint32_t hn = MOD5 ? 4u : a.hash_num; // (MOD5 is launched for four hashes only)
                uint32_t ix[4], word[4];
                const uint64_t base_m = base * ext_m; // the one 64-bit multiply of the operation (see ExtHash3)
                uint64_t hv[4];
                hv[0] = base;
                if (ext_plain) { // even k: the multipliers are M + 1, M + 2, M + 3
                    hv[1] = base_m + base;
                    hv[2] = hv[1] + base;
                    hv[3] = hv[2] + base;
                } else {
#pragma unroll
                    for (int h = 1; h < 4; h++)
                        hv[h] = base_m + ext_addend(base, eh.d[h]);
                }
#pragma unroll
                for (uint32_t h = 0; h < 4; h++) {
                    if (h < hn) {
                        const uint64_t v = h ? hv[h] ^ (hv[h] >> kMultiShift) : base;
                        ix[h] = MOD5 ? mod_5_pow2(v, mod5_s) : (uint32_t)fastmod(v, a.cbf_mod);
                    }
                }
#pragma unroll
                for (uint32_t h = 0; h < 4; h++)
                    if (h < hn)
                        word[h] = atomicAdd(&hist[ix[h] >> shift], 1u);
#pragma unroll
                for (uint32_t h = 0; h < 4; h++) {
                    if (h < hn) {
                        const uint32_t bucket = ix[h] >> shift, rank = word[h] & 0xFFFFu;
                        const uint32_t r = (off << 16) | ((ix[h] & lmask) << 2) | h;
                        // (a rank past the run that still lies in the window lands in a ring slot of a
                        // group the flush never writes; the pair is given up anyway)
                        over |= rank >= RUN;
                        const bool in_win = rank - G * (word[h] >> 16) < RING;
                        // (a lane outside the window writes the spare word behind the counters)
                        ring[in_win ? bucket * RING + (rank & (RING - 1u)) : spare] = r;
                        if (__builtin_expect(__ballot(!in_win && rank < RUN) != 0ull, 0))
                            if (!in_win && rank < RUN)
                                rec[census_run_index(nst, c.n_buckets, bucket, stretch) * RUN + rank] = r;
                    }
                }
            }
            __syncthreads();
            // the groups this round completed: Q neighbouring lanes per range, 16 bytes of a group each
            // (they are lanes of one wave: all of them have read the range's word before the first of them writes it)
            for (uint32_t idx = threadIdx.x; idx < c.n_buckets * Q; idx += kBlock) {
                const uint32_t i = idx / Q, part = idx % Q;
                const uint32_t word = hist[i], cnt = min(word & 0xFFFFu, RUN), g0 = word >> 16;
                uint32_t ng, g_next;
                if (cnt > G * g0 + RING) { // more than the ring holds: its two groups are full, the rest is in memory
                    ng = 2;
                    g_next = (cnt + G - 1u) / G;
                } else {
                    ng = cnt > G * g0 ? (cnt - G * g0) / G : 0u; // (after such a round the window starts past the records)
                    g_next = g0 + ng;
                }
                uint4* const dst = reinterpret_cast<uint4*>(rec + census_run_index(nst, c.n_buckets, i, stretch) * RUN);
                const uint4* const src = reinterpret_cast<const uint4*>(ring + i * RING);
                for (uint32_t g = g0; g < g0 + ng; g++)
                    dst[g * Q + part] = src[(g & 1u) * Q + part];
                if (part == 0)
                    hist[i] = (word & 0xFFFFu) | (g_next << 16);
            }
        }
    }
    __syncthreads();
    uint8_t* const cnt = c.cnt + c.pair_cnt_off[pair];
    for (uint32_t i = threadIdx.x; i < c.n_buckets; i += kBlock) {
        const uint32_t n = min(hist[i] & 0xFFFFu, RUN);
        cnt[(uint64_t)i * nst + stretch] = (uint8_t)n;
        // what is left in the ring: less than a group
        for (uint32_t r = G * (hist[i] >> 16); r < n; r++)
            rec[census_run_index(nst, c.n_buckets, i, stretch) * RUN + r] = ring[i * RING + (r & (RING - 1u))];
    }
    if (over)
        c.overflow[pair] = 1u; // many copies of one k-mer within the stretch: this pair is replayed whole
    }
    // per-wave sums, one atomic each
    for (int o = 32; o > 0; o >>= 1) {
        n_positions += (uint32_t)__shfl_down((int)n_positions, o);
        n_ops += (uint32_t)__shfl_down((int)n_ops, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&a.stats[0], (unsigned long long)n_positions);
        atomicAdd(&a.stats[1], (unsigned long long)n_ops);
    }
}

// ---- census 2: per bucket, the earliest position per counter; every later touch is marked.
// A wave takes a stretch's run at a time (256 bytes of record slots, as many lanes as it
// holds); the runs' lengths are fetched once, the records once (kept in registers between the
// two passes when the bucket has no more than kKeep runs per wave).
constexpr int kMarkBlock = 512;       // eight waves share a range's table: 32 waves per CU at 10 MiB (narrow ranges)
constexpr int kMarkBlockWide = 1024;  // wide ranges: a 64 KB table, two workgroups per CU, sixteen waves each
template <uint32_t RUN, int BLOCK>
__global__ __launch_bounds__(BLOCK, 8) void census_mark(CensusArgs c)
{
    constexpr int kMarkWaves = BLOCK / 64;
    constexpr int PER = RUN / 64;         // records of a run per lane
    extern __shared__ uint32_t first_t[]; // a word per counter of the range (32 KB at GoldPolish's 10 MiB; 64 KB for wide ranges), then the run lengths
    const uint32_t pair = blockIdx.x / c.n_buckets, bucket = blockIdx.x % c.n_buckets;
    const uint32_t bucket_w = 1u << c.bucket_shift;
    if (c.overflow[pair])
        return;
    const uint32_t nst = c.pair_nst[pair];
    uint32_t* const rec0 = c.rec + c.pair_rec_off[pair];
    auto run_of = [&](uint32_t st) -> uint32_t* { return rec0 + census_run_index(nst, c.n_buckets, bucket, st) * RUN; };
    const uint8_t* cnt = c.cnt + c.pair_cnt_off[pair] + (uint64_t)bucket * nst;
    uint8_t* const s_cnt = reinterpret_cast<uint8_t*>(first_t + bucket_w);
    constexpr int kKeep = 128 / kMarkWaves; // 128 runs: a pair of a million positions
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    uint32_t keep[kKeep][PER];
    // The first slots of a run are fetched beside the run lengths, not behind them; what a length rules out
    // is dropped below, what it adds is fetched there.  How many: c.mark_prefetch, all 64 by default.  A wide
    // range's run holds 38 records on average (three touches per position), more than 48 in one run in
    // twenty, so the fourth 64-byte line is fetched for nothing nineteen times in twenty -- 80 GB per configs[2]
    // step -- and NOT fetching it is slower: stage A 654.6 ms with 64, 661.4 with 48, 667.1 with 40, 668.8 with 32
    // (GP_STAGE_A_MARK_PREFETCH).  The census is not short of bandwidth; the late fetch of the runs that do
    // need the line is a second round trip in the one workgroup the others of its CU wait for.
    const uint32_t pre = c.mark_prefetch;
#pragma unroll
    for (int q = 0; q < kKeep; q++) {
        const uint32_t st = wave + (uint32_t)kMarkWaves * (uint32_t)q;
        keep[q][0] = (st < nst && lane < pre) ? run_of(st)[lane] : 0xFFFFFFFFu;
    }
    for (uint32_t i = threadIdx.x; i < nst; i += BLOCK)
        s_cnt[i] = cnt[i];
    for (uint32_t i = threadIdx.x; i < bucket_w; i += BLOCK)
        first_t[i] = 0xFFFFFFFFu;
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kKeep; q++) {
        const uint32_t st = wave + (uint32_t)kMarkWaves * (uint32_t)q;
        const uint32_t n = st < nst ? s_cnt[st] : 0u;
        if (lane >= pre && lane < n)
            keep[q][0] = run_of(st)[lane];
        if (lane >= n)
            keep[q][0] = 0xFFFFFFFFu;
#pragma unroll
        for (int e = 1; e < PER; e++)
            keep[q][e] = lane + 64u * e < n ? run_of(st)[lane + 64u * e] : 0xFFFFFFFFu;
    }
#pragma unroll
    for (int q = 0; q < kKeep; q++)
#pragma unroll
        for (int e = 0; e < PER; e++)
            if (keep[q][e] != 0xFFFFFFFFu)
                atomicMin(&first_t[(keep[q][e] >> 2) & 0x3FFFu],
                          (wave + (uint32_t)kMarkWaves * (uint32_t)q) * kCStretch + (keep[q][e] >> 16));
#pragma unroll 4
    for (uint32_t st = wave + (uint32_t)kMarkWaves * kKeep; st < nst; st += kMarkWaves)
#pragma unroll
        for (int e = 0; e < PER; e++)
            if (lane + 64u * e < s_cnt[st]) {
                const uint32_t r = run_of(st)[lane + 64u * e];
                atomicMin(&first_t[(r >> 2) & 0x3FFFu], st * kCStretch + (r >> 16));
            }
    __syncthreads();
    // The touches that are not their counter's first go back into the head of their own run, bit 31
    // set, packed (the records are in registers, the run is this wave's alone): a sector or two per
    // run, next to each other, instead of a flag byte per touch scattered over the pair's positions
    // (19 G bytes per configs[2] step that left L2 as 0.64 TB of partial sectors).  census_pack reads
    // a run up to the first record without the bit; a run whose touches are all marked gets a zero
    // behind them (what lies past a run's records is left over from an earlier cohort).
    auto mark = [&](const uint32_t (&r)[PER], uint32_t st) { // every lane of the wave; r[e] == 0xFFFFFFFF: no record
        uint32_t* const run = run_of(st);
        uint32_t n = 0;
#pragma unroll
        for (int e = 0; e < PER; e++) {
            const uint32_t t = st * kCStretch + ((r[e] >> 16) & (kCStretch - 1u));
            const bool later = r[e] != 0xFFFFFFFFu && first_t[(r[e] >> 2) & 0x3FFFu] != t; // (a counter hit twice by one operation: both touches are its first)
            const unsigned long long m = __ballot(later);
            if (later)
                run[n + __popcll(m & ((1ull << lane) - 1ull))] = r[e] | 0x80000000u;
            n += (uint32_t)__popcll(m);
        }
        if (lane == 0 && n == s_cnt[st] && n < RUN)
            run[n] = 0;
    };
#pragma unroll
    for (int q = 0; q < kKeep; q++) {
        const uint32_t st = wave + (uint32_t)kMarkWaves * (uint32_t)q;
        if (st < nst)
            mark(keep[q], st);
    }
    for (uint32_t st = wave + (uint32_t)kMarkWaves * kKeep; st < nst; st += kMarkWaves) {
        uint32_t r[PER];
#pragma unroll
        for (int e = 0; e < PER; e++)
            r[e] = lane + 64u * e < s_cnt[st] ? run_of(st)[lane + 64u * e] : 0xFFFFFFFFu;
        mark(r, st);
    }
}

// ---- census 2 for the census LEVEL BY LEVEL (stage_a_levels): the same marks, a pair of any length taken a
// segment of 128 runs at a time, and from the second pass on fed by the pass before.
template <uint32_t RUN, int BLOCK>
__global__ __launch_bounds__(BLOCK, 8) void census_mark_seg(CensusArgs c)
{
    constexpr int kMarkWaves = BLOCK / 64;
    constexpr int PER = RUN / 64;         // records of a run per lane
    extern __shared__ uint32_t first_t[]; // a word per counter of the range (32 KB at GoldPolish's 10 MiB; 64 KB for wide ranges), then the run lengths
    const uint32_t pair = blockIdx.x / c.n_buckets, bucket = blockIdx.x % c.n_buckets;
    const uint32_t bucket_w = 1u << c.bucket_shift;
    if (c.overflow[pair] || (c.pair_need && c.level >= c.pair_need[pair]))
        return;
    const uint32_t nst = c.pair_nst[pair];
    uint32_t* const rec0 = c.rec + c.pair_rec_off[pair];
    auto run_of = [&](uint32_t st) -> uint32_t* { return rec0 + census_run_index(nst, c.n_buckets, bucket, st) * RUN; };
    uint8_t* const cnt = c.cnt + c.pair_cnt_off[pair] + (uint64_t)bucket * nst;
    // Level by level (stage_a_levels), from the second pass on: NOTHING is scattered again.  The records of
    // the operations of L(j + 1) are already here -- an operation is in L(j + 1) iff all its touches were
    // marked in pass j, and the marked records lie packed at the head of their runs -- so a pass's input is
    // the previous pass's marked records whose operation made it into the set (one bit per record, from the
    // bitmap census_pack has just written), and the run lengths are rewritten to what this pass marks.
    const unsigned long long* const members =
        (c.level > 0 && c.members) ? c.members + (c.pair_bit_off[pair] >> 6) : nullptr;
    auto take = [&](uint32_t r, uint32_t st) -> uint32_t { // a record of the pass before -> this pass's (or none)
        if (!members || r == 0xFFFFFFFFu)
            return r;
        const uint64_t pos = (uint64_t)st * kCStretch + ((r >> 16) & (kCStretch - 1u));
        return ((members[pos >> 6] >> (pos & 63)) & 1ull) ? (r & 0x7FFFFFFFu) : 0xFFFFFFFFu;
    };
    uint8_t* const s_cnt = reinterpret_cast<uint8_t*>(first_t + bucket_w);
    constexpr int kKeep = 128 / kMarkWaves; // a segment of 128 runs (a million positions): a run per wave and turn, kept in registers
    const uint32_t wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (uint32_t i = threadIdx.x; i < nst; i += BLOCK)
        s_cnt[i] = cnt[i];
    for (uint32_t i = threadIdx.x; i < bucket_w; i += BLOCK)
        first_t[i] = 0xFFFFFFFFu;
    // A pair longer than a segment (the census level by level on a long contig: thousands of stretches) is
    // taken a segment at a time, the table carried along: positions only grow from segment to segment, so
    // after a segment's minima are in, the table is final for every counter that segment touches -- a touch
    // is its counter's first iff the table holds its own position -- and the next segment can only fill
    // entries nobody has touched yet.  (First version: one pass over ALL the pair's runs for the minima,
    // a second for the marks, both re-reading the records: 17 s for the level-by-level census of SURVEY
    // 8d's long contigs where the grid-wide schedule took 6.6.)
    for (uint32_t seg0 = 0; seg0 < nst; seg0 += 128) {
        uint32_t keep[kKeep][PER];
        // The first 64 slots of a run are fetched beside the run lengths, not behind them (a run of
        // 51 records on average spans those four lines anyway); what a length rules out is dropped below.
        // (from the second pass of the level-by-level census on most runs are empty: there nothing is fetched
        // before the length is known -- fetching the heads of all runs again in every pass was 320 GB a pass
        // on SURVEY 8d's long contigs)
#pragma unroll
        for (int q = 0; q < kKeep; q++) {
            const uint32_t st = seg0 + wave + (uint32_t)kMarkWaves * (uint32_t)q;
            keep[q][0] = (st < nst && !members) ? run_of(st)[lane] : 0xFFFFFFFFu;
        }
        if (seg0 == 0)
            __syncthreads(); // the run lengths and the empty table are there
#pragma unroll
        for (int q = 0; q < kKeep; q++) {
            const uint32_t st = seg0 + wave + (uint32_t)kMarkWaves * (uint32_t)q;
            const uint32_t n = st < nst ? s_cnt[st] : 0u;
            if (members && lane < n)
                keep[q][0] = run_of(st)[lane];
            if (lane >= n)
                keep[q][0] = 0xFFFFFFFFu;
#pragma unroll
            for (int e = 1; e < PER; e++)
                keep[q][e] = lane + 64u * e < n ? run_of(st)[lane + 64u * e] : 0xFFFFFFFFu;
            if (members) {
#pragma unroll
                for (int e = 0; e < PER; e++)
                    keep[q][e] = take(keep[q][e], st);
            }
        }
#pragma unroll
        for (int q = 0; q < kKeep; q++)
#pragma unroll
            for (int e = 0; e < PER; e++)
                if (keep[q][e] != 0xFFFFFFFFu)
                    atomicMin(&first_t[(keep[q][e] >> 2) & 0x3FFFu],
                              (seg0 + wave + (uint32_t)kMarkWaves * (uint32_t)q) * kCStretch + (keep[q][e] >> 16));
        __syncthreads();
        // The touches that are not their counter's first go back into the head of their own run, bit 31
        // set, packed (the records are in registers, the run is this wave's alone): a sector or two per
        // run, next to each other, instead of a flag byte per touch scattered over the pair's positions
        // (19 G bytes per configs[2] step that left L2 as 0.64 TB of partial sectors).  census_pack reads
        // a run up to the first record without the bit; a run whose touches are all marked gets a zero
        // behind them (what lies past a run's records is left over from an earlier cohort).
#pragma unroll
        for (int q = 0; q < kKeep; q++) {
            const uint32_t st = seg0 + wave + (uint32_t)kMarkWaves * (uint32_t)q;
            if (st >= nst)
                continue;
            uint32_t* const run = run_of(st);
            uint32_t n = 0;
#pragma unroll
            for (int e = 0; e < PER; e++) {
                const uint32_t r = keep[q][e];
                const uint32_t t = st * kCStretch + ((r >> 16) & (kCStretch - 1u));
                const bool later = r != 0xFFFFFFFFu && first_t[(r >> 2) & 0x3FFFu] != t; // (a counter hit twice by one operation: both touches are its first)
                const unsigned long long m = __ballot(later);
                if (later)
                    run[n + __popcll(m & ((1ull << lane) - 1ull))] = r | 0x80000000u;
                n += (uint32_t)__popcll(m);
            }
            // (behind a pass's marked records: an unmarked original at the first pass unless every touch was
            // marked; from the second pass on whatever the pass before left there)
            if (lane == 0 && (members || n == s_cnt[st]) && n < RUN)
                run[n] = 0;
            if (lane == 0 && c.pair_need)
                cnt[st] = (uint8_t)n; // the next pass's input: this pass's marked records
        }
    }
}

// ---- census 3: operations with every touch marked -> one bit each.  A workgroup per stretch: a lane
// per range reads the marked records at the head of the stretch's run there and counts them per
// position in LDS (a byte each).
template <uint32_t RUN>
__global__ __launch_bounds__(256) void census_pack(AArgs a, CensusArgs c)
{
    __shared__ uint32_t s_touch[kCStretch / 4];
    __shared__ uint32_t s_ncand;
    const uint32_t pair = c.wg_pair[blockIdx.x];
    if (threadIdx.x == 0)
        s_ncand = 0;
    if (c.overflow[pair]) {
        if (threadIdx.x == 0)
            c.pair_mode[c.pair_queue[pair]] = 2u; // the plain replay for this pair (its operations are counted already)
        return;
    }
    if (c.pair_need && c.level >= c.pair_need[pair])
        return;
    const uint32_t b = c.pair_batch[pair];
    const uint64_t b_start = a.batch_pos[b], b_end = a.batch_pos[b + 1];
    const uint64_t pos0 = c.wg_pos0[blockIdx.x];
    const uint64_t stretch_end = min(b_end, pos0 + (uint64_t)kCStretch);
    const uint32_t nst = c.pair_nst[pair], stretch = (uint32_t)((pos0 - b_start) / kCStretch);
    if (c.members) {
        // a stretch without a member of this level has nobody in the next one either: zeros, no run is read
        const unsigned long long* const mem = c.members + ((c.pair_bit_off[pair] + (pos0 - b_start)) >> 6);
        int any = 0;
        if (threadIdx.x < kCStretch / 64 && pos0 + (uint64_t)threadIdx.x * 64 < stretch_end)
            any = mem[threadIdx.x] != 0ull;
        if (!__syncthreads_or(any)) {
            if (threadIdx.x < kCStretch / 64 && pos0 + (uint64_t)threadIdx.x * 64 < stretch_end)
                c.cand[((c.pair_bit_off[pair] + (pos0 - b_start)) >> 6) + threadIdx.x] = 0ull;
            return;
        }
    }
    const uint8_t* const cnt = c.cnt + c.pair_cnt_off[pair];
    for (uint32_t i = threadIdx.x; i < kCStretch / 4; i += 256)
        s_touch[i] = 0;
    __syncthreads();
    const uint32_t* const rec = c.rec + c.pair_rec_off[pair];
    for (uint32_t bu = threadIdx.x; bu < c.n_buckets; bu += 256) {
        if (c.members && cnt[(uint64_t)bu * nst + stretch] == 0)
            continue; // (deeper levels: most runs are empty; a byte says so before a line is fetched)
        // a 64-byte line of the run at a time (four loads in flight instead of one behind the other)
        const uint4* run = reinterpret_cast<const uint4*>(rec + census_run_index(nst, c.n_buckets, bu, stretch) * RUN);
        for (uint32_t q = 0; q < RUN / 16; q++) {
            const uint4 v0 = run[4 * q], v1 = run[4 * q + 1], v2 = run[4 * q + 2], v3 = run[4 * q + 3];
            const uint32_t w[16] = { v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w,
                                     v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w };
            bool more = true;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (more && (w[i] >> 31)) {
                    const uint32_t off = (w[i] >> 16) & (kCStretch - 1u);
                    atomicAdd(&s_touch[off >> 2], 1u << (8u * (off & 3u)));
                } else {
                    more = false;
                }
            }
            if (!more)
                break;
        }
    }
    __syncthreads();
    uint32_t n_cand = 0;
    for (uint64_t p0 = pos0; p0 < stretch_end; p0 += 256) {
        const uint64_t p = p0 + threadIdx.x;
        const uint32_t rel = (uint32_t)(p - pos0);
        const bool is_cand = p < stretch_end && ((s_touch[rel >> 2] >> (8u * (rel & 3u))) & 255u) == a.hash_num;
        const unsigned long long m = __ballot(is_cand);
        if ((threadIdx.x & 63) == 0 && p < stretch_end) {
            c.cand[(c.pair_bit_off[pair] + (p - b_start)) >> 6] = m; // stretches and batches start on multiples of 64 bits
            n_cand += (uint32_t)__popcll(m);
        }
    }
    // (one atomic per workgroup that found any: an atomic per 64 positions on the one word was 73 ms of
    // a 75 ms launch -- 1.5 M of them in a row -- in the first version of the level-by-level census)
    if (c.level_count) {
        if ((threadIdx.x & 63) == 0 && n_cand)
            atomicAdd(&s_ncand, n_cand);
        __syncthreads();
        if (threadIdx.x == 0 && s_ncand)
            atomicAdd(c.level_count, (unsigned long long)s_ncand);
    }
}

// ------------------------------------------------- candidates by k-mer: most of the replay is a count
// The candidates of a census pair are a conservative-update replay among themselves (census_scatter).
// Call a candidate k-mer PRIVATE when none of its counters is touched by a candidate of a different
// k-mer: its counters then only ever see its own copies, its i-th candidate copy finds all of them at
// min(i, thr') and goes into the filter iff i >= thr' - 1, so the k-mer is in the Bloom filter iff it has
// >= thr - 1 candidate copies (thr' = thr - 1 the candidates' threshold, uniform over the pair) -- a
// count, no replay.  The k-mers that are not private are closed under "shares a counter": they are
// replayed by themselves, on the arena slot as before.  On 25 kbp contigs at 40x, 91-93 % of the
// candidate operations belong to private k-mers (tests/test_private_kmer_model.py checks the closed form
// against the serial replay and measures the fraction); what cbf_local replays goes from 23 % of the
// operations to under 2 %, and with it most of its random counter transactions.
// Streaming passes, all in the record space the census has just finished with:
//   ckc_scatter  candidates (64-bit hash, position) into hash buckets of their pair: LDS staging, whole
//                chunks per bucket and quarter-stretch, one cursor atomic per chunk;
//   ckc_group    a workgroup per bucket: an LDS hash set of its distinct k-mers (a slot each, written back
//                into the records); every new k-mer sends its counters, tagged (bucket, slot), to the
//                group of their counter range;
//   ckc_share    a workgroup per counter group: an exact LDS table counter -> k-mer; a second k-mer on a
//                counter flags both (a bit per (bucket, slot));
//   ckc_apply    a workgroup per bucket again: copies per slot; unflagged with enough copies -> the
//                filter's bits; flagged -> their copies' bits in the SECOND candidate bitmap, which is
//                what cbf_local replays for the pair (pair_mode 3).
// Anything that does not fit (a bucket, a table, a group) sets the pair's overflow word and the pair
// keeps the whole candidate replay.
__device__ inline uint32_t select64(unsigned long long w, uint32_t r); // (below, with cbf_local)

constexpr uint32_t kCkcSlots = 2048;        // distinct k-mers a bucket's LDS set holds (load <= 7/8)
constexpr uint32_t kCkcShareSlots = 8192;   // (counter, k-mer) entries a counter group's LDS table holds (load <= 3/4)
constexpr uint32_t kCkcStage = 3072;        // twice the records ckc_scatter stages in LDS per quarter-stretch (12 bytes each)
constexpr uint32_t kCkcMaxBuckets = 128, kCkcMaxGroups = 256;
constexpr uint32_t kCkcHdrWords = 1024;     // [0,256) bucket cursors, [256,512) group cursors, [512] overflow

struct CkcArgs {
    const uint32_t* pair_nb;    // [pairs] hash buckets of the pair (a power of two; 0: not eligible, the pair keeps the whole candidate replay)
    const uint32_t* pair_capb;  // [pairs] records a bucket holds
    const uint32_t* pair_capg;  // [pairs] entries a counter group holds
    const uint32_t* pair_need;  // [pairs] candidate copies from which a private k-mer is in the filter (thr - 1)
    uint32_t gshift, n_groups;  // a counter group holds 1 << gshift counters; groups per pair
    unsigned long long* cand2;  // the candidates that still need the replay (same bit positions as `cand`)
};

struct CkcRegion {
    uint32_t* cur_b;       // bucket cursors
    uint32_t* cur_g;       // group cursors
    uint32_t* over;        // != 0: the pair keeps the whole candidate replay
    uint32_t* flags;       // [nb][kCkcSlots / 32]: k-mers that share a counter with another
    uint4* rec;            // [nb][capb]: {hash lo, hash hi, position in the pair, slot}
    unsigned long long* ent; // [n_groups][capg]: (counter within its group << 32) | (bucket << 11 | slot)
};

__device__ inline CkcRegion ckc_region(const CensusArgs& c, const CkcArgs& g, uint32_t pair)
{
    uint32_t* base = c.rec + c.pair_rec_off[pair];
    const uint32_t nb = g.pair_nb[pair];
    CkcRegion r;
    r.cur_b = base;
    r.cur_g = base + 256;
    r.over = base + 512;
    r.flags = base + kCkcHdrWords;
    r.rec = reinterpret_cast<uint4*>(base + kCkcHdrWords + nb * (kCkcSlots / 32));
    r.ent = reinterpret_cast<unsigned long long*>(r.rec + (uint64_t)nb * g.pair_capb[pair]);
    return r;
}

__device__ inline uint64_t ckc_mix(uint64_t base) { return base * 0x9E3779B97F4A7C15ULL; }

__global__ __launch_bounds__(256) void ckc_init(CensusArgs c, CkcArgs g, uint32_t n_pairs)
{
    const uint32_t pair = blockIdx.x;
    if (pair >= n_pairs || !g.pair_nb[pair])
        return;
    uint32_t* base = c.rec + c.pair_rec_off[pair];
    const uint32_t n = kCkcHdrWords + g.pair_nb[pair] * (kCkcSlots / 32);
    for (uint32_t i = threadIdx.x; i < n; i += 256)
        base[i] = 0;
    if (threadIdx.x == 0 && c.overflow[pair])
        base[512] = 1u; // the census gave the pair up: nothing to group
}

// A workgroup per stretch (the census's table of stretches), a quarter of it (2048 positions) at a time:
// a lane rolls over eight consecutive positions -- the k-step start from the table of rotated seeds
// once, a roll per position after that, nothing at all where its eight candidate bits are clear -- and
// stages its candidates per bucket; a quarter's chunks leave with one cursor atomic per bucket.
// (First version: a candidate per lane and turn, each hashed from scratch, half-stretches, 63 KB of
// LDS: two workgroups per CU, 32.8 ms per 1000 contigs of 25 kbp at 40x; this one: 24 KB.)
constexpr uint32_t kCkcQuarter = kCStretch / 4, kCkcPerLane = kCkcQuarter / 256;
static_assert(kCkcPerLane == 8, "a byte of candidate bits per lane");

__global__ __launch_bounds__(256) void ckc_scatter(AArgs a, CensusArgs c, CkcArgs g)
{
    __shared__ uint8_t codes[kCkcQuarter + kHalo];
    __shared__ uint64_t seed_t[kSeedTableWords];
    __shared__ ulonglong2 roll_in[8], roll_out[8]; // what a base brings to {forward, reverse} when it enters / leaves a k-mer
    __shared__ uint32_t st_h0[kCkcStage / 2], st_h1[kCkcStage / 2], st_pos[kCkcStage / 2];
    __shared__ uint32_t s_cnt[kCkcMaxBuckets], s_base[kCkcMaxBuckets];
    const uint32_t pair = c.wg_pair[blockIdx.x];
    const uint32_t nb = g.pair_nb[pair];
    if (!nb)
        return;
    const CkcRegion R = ckc_region(c, g, pair);
    if (*R.over)
        return;
    const uint32_t b = c.pair_batch[pair], ki = c.pair_ki[pair];
    const KConsts& kc = a.kc[ki];
    const uint32_t k = kc.k;
    const uint64_t b_start = a.batch_pos[b], b_end = a.batch_pos[b + 1];
    const uint64_t pos0 = c.wg_pos0[blockIdx.x];
    const uint64_t stretch_end = min(b_end, pos0 + (uint64_t)kCStretch);
    const uint32_t capb = g.pair_capb[pair], cap_s = (kCkcStage / 2) / nb, tid = threadIdx.x;
    fill_seed_table(seed_t);
    if (tid < 8) {
        roll_in[tid] = tid < 4 ? make_ulonglong2(seed_of_code(tid), kc.in_rev[tid]) : make_ulonglong2(0, 0);
        roll_out[tid] = tid < 4 ? make_ulonglong2(kc.out_fwd[tid], seed_of_code(tid ^ 2)) : make_ulonglong2(0, 0);
    }
    if (tid < nb)
        s_cnt[tid] = 0;
    const uint8_t* const cand = reinterpret_cast<const uint8_t*>(c.cand + ((c.pair_bit_off[pair] + (pos0 - b_start)) >> 6));
    for (uint32_t q = 0; q < 4; q++) {
        const uint64_t q0 = pos0 + (uint64_t)q * kCkcQuarter;
        if (q0 >= stretch_end)
            break; // uniform
        __syncthreads(); // the quarter before is flushed (and the tables are set)
        for (uint32_t i = tid; i < kCkcQuarter + kHalo; i += 256) {
            const uint64_t p = q0 + i;
            codes[i] = base_code(p < a.n_pos ? a.reads[p] : (uint8_t)'\n');
        }
        // the lane's eight candidate bits (little endian: byte j of the bitmap = positions 8 j .. 8 j + 7);
        // nothing is read past the stretch's own bits
        const uint64_t lane_pos0 = q0 + (uint64_t)tid * kCkcPerLane;
        uint32_t bits = lane_pos0 < stretch_end ? cand[q * (kCkcQuarter / 8) + tid] : 0u;
        if (lane_pos0 < stretch_end && stretch_end - lane_pos0 < 8)
            bits &= (1u << (stretch_end - lane_pos0)) - 1u;
        __syncthreads();
        if (bits) {
            const uint32_t t0 = tid * kCkcPerLane;
            const uint32_t j0 = (uint32_t)__ffs((int)bits) - 1u; // the rolling starts at the lane's first candidate
            uint64_t fh = 0, rh = 0;
            int no_bad = -1; // a candidate is a k-mer: no byte of it is bad (bytes between candidates may be: they hash as 0, and so do they in the rolls)
            kmer_hash_from_table(seed_t, codes + t0 + j0, k, fh, rh, no_bad);
            for (uint32_t j = j0; j < kCkcPerLane; j++) {
                if (j > j0) {
                    const ulonglong2 ri = roll_in[codes[t0 + j + k - 1]], ro = roll_out[codes[t0 + j - 1]];
                    fh = srol1(fh) ^ ri.x ^ ro.x;
                    rh = sror1(rh ^ ri.y ^ ro.y);
                }
                if (!((bits >> j) & 1u))
                    continue;
                const uint64_t base = fh + rh;
                const uint32_t bkt = (uint32_t)(ckc_mix(base) >> 56) & (nb - 1u);
                const uint32_t prel = (uint32_t)(q0 + t0 + j - b_start);
                const uint32_t r = atomicAdd(&s_cnt[bkt], 1u);
                if (r < cap_s) {
                    st_h0[bkt * cap_s + r] = (uint32_t)base;
                    st_h1[bkt * cap_s + r] = (uint32_t)(base >> 32);
                    st_pos[bkt * cap_s + r] = prel;
                } else { // more of one bucket than the staging holds: straight to memory
                    const uint32_t at = atomicAdd(&R.cur_b[bkt], 1u);
                    if (at < capb)
                        R.rec[(uint64_t)bkt * capb + at] = make_uint4((uint32_t)base, (uint32_t)(base >> 32), prel, 0u);
                    else
                        *R.over = 1u;
                }
            }
        }
        __syncthreads();
        if (tid < nb) {
            const uint32_t n = min(s_cnt[tid], cap_s);
            uint32_t at = n ? atomicAdd(&R.cur_b[tid], n) : 0u;
            if (at + n > capb) {
                *R.over = 1u;
                at = 0xFFFFFFFFu;
            }
            s_base[tid] = at;
        }
        __syncthreads();
        for (uint32_t idx = tid; idx < nb * cap_s; idx += 256) {
            const uint32_t bkt = idx / cap_s, i = idx - bkt * cap_s;
            if (i < min(s_cnt[bkt], cap_s) && s_base[bkt] != 0xFFFFFFFFu)
                R.rec[(uint64_t)bkt * capb + s_base[bkt] + i] = make_uint4(st_h0[idx], st_h1[idx], st_pos[idx], 0u);
        }
        __syncthreads();
        if (tid < nb)
            s_cnt[tid] = 0;
    }
}

// A workgroup per (pair, bucket): the bucket's distinct k-mers.  Three passes: the records into the
// LDS set (their slot goes back into the record); the set's k-mers counted per counter group, a range
// of each group reserved with one atomic; the set's k-mers once more, their (counter, k-mer) entries
// written straight into those ranges.  (First version: entries staged per group in 40 KB of LDS -- two
// workgroups per CU, 20.6 ms per 1000 contigs; the counters are cheap to work out twice.)
__global__ __launch_bounds__(256) void ckc_group(AArgs a, CensusArgs c, CkcArgs g, uint32_t span)
{
    __shared__ unsigned long long keys[kCkcSlots];
    __shared__ uint32_t s_cnt[kCkcMaxGroups], s_base[kCkcMaxGroups], s_ndist;
    const uint32_t pair = blockIdx.x / span, bkt = blockIdx.x % span, tid = threadIdx.x; // (a 1-D grid: pairs x the most buckets a pair has)
    const uint32_t nb = g.pair_nb[pair];
    if (bkt >= nb)
        return;
    const CkcRegion R = ckc_region(c, g, pair);
    if (*R.over)
        return;
    const uint32_t capb = g.pair_capb[pair], capg = g.pair_capg[pair];
    const uint32_t n = R.cur_b[bkt]; // (<= capb: a chunk that did not fit set the overflow word)
    const uint32_t ng = g.n_groups, gmask = (1u << g.gshift) - 1u;
    const uint32_t k = a.kc[c.pair_ki[pair]].k;
    const unsigned long long EMPTY = ~0ull;
    for (uint32_t i = tid; i < kCkcSlots; i += 256)
        keys[i] = EMPTY;
    for (uint32_t i = tid; i < ng; i += 256)
        s_cnt[i] = 0;
    if (tid == 0)
        s_ndist = 0;
    __syncthreads();
    uint4* const rec = R.rec + (uint64_t)bkt * capb;
    bool bad = false;
    for (uint32_t i = tid; i < n; i += 256) {
        const uint4 r = rec[i];
        const unsigned long long key = ((unsigned long long)r.y << 32) | r.x;
        if (key == EMPTY) {
            bad = true; // (the one hash the set cannot hold)
            continue;
        }
        uint32_t sl = (uint32_t)(ckc_mix(key) >> 40) & (kCkcSlots - 1u);
        bool placed = false;
        for (uint32_t probe = 0; probe < kCkcSlots; probe++, sl = (sl + 1u) & (kCkcSlots - 1u)) {
            const unsigned long long old = atomicCAS(&keys[sl], EMPTY, key);
            if (old == EMPTY) {
                if (atomicAdd(&s_ndist, 1u) >= kCkcSlots / 8 * 7)
                    bad = true;
                placed = true;
                break;
            }
            if (old == key) {
                placed = true;
                break;
            }
        }
        if (!placed) {
            bad = true;
            continue;
        }
        rec[i].w = sl;
    }
    __syncthreads();
    // the set's k-mers per counter group
    for (uint32_t sl = tid; sl < kCkcSlots; sl += 256) {
        const unsigned long long key = keys[sl];
        if (key == EMPTY)
            continue;
        for (uint32_t h = 0; h < a.hash_num; h++)
            atomicAdd(&s_cnt[(uint32_t)fastmod(h ? ext_hash(key, k, h) : key, a.cbf_mod) >> g.gshift], 1u);
    }
    __syncthreads();
    for (uint32_t grp = tid; grp < ng; grp += 256) {
        const uint32_t m = s_cnt[grp];
        uint32_t at = m ? atomicAdd(&R.cur_g[grp], m) : 0u;
        if (at + m > capg) {
            bad = true;
            at = 0xFFFFFFFFu;
        }
        s_base[grp] = at;
        s_cnt[grp] = 0; // (the next pass hands out the places of the range)
    }
    if (bad)
        *R.over = 1u;
    __syncthreads();
    for (uint32_t sl = tid; sl < kCkcSlots; sl += 256) {
        const unsigned long long key = keys[sl];
        if (key == EMPTY)
            continue;
        const uint32_t ref = (bkt << 11) | sl;
        for (uint32_t h = 0; h < a.hash_num; h++) {
            const uint32_t ix = (uint32_t)fastmod(h ? ext_hash(key, k, h) : key, a.cbf_mod);
            const uint32_t grp = ix >> g.gshift;
            const uint32_t q = atomicAdd(&s_cnt[grp], 1u);
            if (s_base[grp] != 0xFFFFFFFFu)
                R.ent[(uint64_t)grp * capg + s_base[grp] + q] = ((unsigned long long)(ix & gmask) << 32) | ref;
        }
    }
}

// Which counters do two different candidate k-mers touch?  A unit = (pair, counter group): an exact LDS
// table counter -> k-mer.  A persistent grid: a workgroup takes units one after the other and never clears
// its table -- an entry carries the unit's epoch in its top bits, one of another epoch is an empty slot --
// because a unit is a few hundred to a few thousand entries and clearing 64 KB for each of the 320 000
// units of 1000 contigs was most of the kernel (11.4 ms per 1000 contigs with a workgroup per unit).
// entry in memory: (counter within its group << 32) | (bucket << 11 | slot); in the table:
// (epoch << 44) | (counter within its group << 20) | (bucket << 11 | slot)
__global__ __launch_bounds__(256) void ckc_share(CensusArgs c, CkcArgs g, uint32_t n_pairs)
{
    __shared__ unsigned long long tab[kCkcShareSlots];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < kCkcShareSlots; i += 256)
        tab[i] = 0; // epoch 0: never a unit's
    unsigned long long ep = 0;
    const uint32_t n_units = n_pairs * g.n_groups;
    for (uint32_t unit = blockIdx.x; unit < n_units; unit += gridDim.x) {
        __syncthreads(); // the unit before is done with the table
        if (++ep == (1ull << 20)) { // (a million units per workgroup: the tags start over on a clean table; before any `continue`: the epoch is the workgroup's, not a thread's)
            for (uint32_t i = tid; i < kCkcShareSlots; i += 256)
                tab[i] = 0;
            ep = 1;
            __syncthreads();
        }
        const uint32_t pair = unit / g.n_groups, grp = unit % g.n_groups;
        if (!g.pair_nb[pair])
            continue;
        const CkcRegion R = ckc_region(c, g, pair);
        if (*R.over)
            continue;
        const uint32_t capg = g.pair_capg[pair];
        const uint32_t n = R.cur_g[grp];
        if (n > kCkcShareSlots / 4 * 3) {
            if (tid == 0)
                *R.over = 1u;
            continue;
        }
        const unsigned long long* const ent = R.ent + (uint64_t)grp * capg;
        for (uint32_t i = tid; i < n; i += 256) {
            const unsigned long long e = ent[i];
            const uint32_t key = (uint32_t)(e >> 32), ref = (uint32_t)e;
            const unsigned long long mine = (ep << 44) | ((unsigned long long)key << 20) | ref;
            uint32_t sl = (key * 0x9E3779B1u) >> (32 - 13);
            unsigned long long old = tab[sl];
            for (uint32_t probe = 0; probe < kCkcShareSlots;) {
                if ((old >> 44) != ep) { // free for this unit
                    const unsigned long long prev = atomicCAS(&tab[sl], old, mine);
                    if (prev == old)
                        break; // in
                    old = prev; // somebody else took the slot meanwhile: look at what is there now
                    continue;
                }
                if (((uint32_t)(old >> 20) & 0xFFFFFFu) == key) { // (same epoch, same counter)
                    const uint32_t other = (uint32_t)old & 0xFFFFFu;
                    if (other != ref) { // two k-mers on one counter: neither is private
                        atomicOr(&R.flags[(ref >> 11) * (kCkcSlots / 32) + ((ref & 2047u) >> 5)], 1u << (ref & 31u));
                        atomicOr(&R.flags[(other >> 11) * (kCkcSlots / 32) + ((other & 2047u) >> 5)], 1u << (other & 31u));
                    }
                    break;
                }
                sl = (sl + 1u) & (kCkcShareSlots - 1u);
                old = tab[sl];
                probe++;
            }
        }
    }
}

// A workgroup per (pair, bucket): the count for the private k-mers, the second bitmap for the others.
__global__ __launch_bounds__(256) void ckc_apply(AArgs a, CensusArgs c, CkcArgs g, uint32_t span)
{
    __shared__ uint32_t cnt[kCkcSlots], hlo[kCkcSlots], hhi[kCkcSlots], flg[kCkcSlots / 32];
    __shared__ unsigned long long s_private;
    const uint32_t pair = blockIdx.x / span, bkt = blockIdx.x % span, tid = threadIdx.x;
    const uint32_t nb = g.pair_nb[pair];
    if (bkt >= nb)
        return;
    const CkcRegion R = ckc_region(c, g, pair);
    if (*R.over)
        return; // the pair keeps the whole candidate replay (pair_mode stays 1)
    const uint32_t capb = g.pair_capb[pair], n = R.cur_b[bkt], need = g.pair_need[pair];
    const uint32_t b = c.pair_batch[pair], ki = c.pair_ki[pair], k = a.kc[ki].k, stream = b * a.n_k + ki;
    for (uint32_t i = tid; i < kCkcSlots; i += 256)
        cnt[i] = 0;
    if (tid < kCkcSlots / 32)
        flg[tid] = R.flags[bkt * (kCkcSlots / 32) + tid];
    if (tid == 0)
        s_private = 0;
    __syncthreads();
    const uint4* const rec = R.rec + (uint64_t)bkt * capb;
    const uint64_t bit0 = c.pair_bit_off[pair];
    for (uint32_t i = tid; i < n; i += 256) {
        const uint4 r = rec[i];
        if ((flg[r.w >> 5] >> (r.w & 31u)) & 1u) { // a copy of a k-mer that shares a counter: replayed
            const uint64_t bit = bit0 + r.z;
            atomicOr(&g.cand2[bit >> 6], 1ULL << (bit & 63));
        } else if (atomicAdd(&cnt[r.w], 1u) == 0u) { // (every record of a slot carries the same hash)
            hlo[r.w] = r.x;
            hhi[r.w] = r.y;
        }
    }
    __syncthreads();
    unsigned long long mine = 0;
    uint32_t* const bf = a.bf + (uint64_t)stream * a.bf_words;
    for (uint32_t sl = tid; sl < kCkcSlots; sl += 256) {
        const uint32_t m = cnt[sl];
        if (!m)
            continue;
        mine += m;
        if (m >= need) {
            // (no look at the word first, as bf_insert does: this is the k-mer's one insertion, and an
            // atomic whose result nobody waits for costs the lane nothing)
            const uint64_t base = ((uint64_t)hhi[sl] << 32) | hlo[sl];
            for (uint32_t i = 0; i < a.hash_num; i++) {
                const uint64_t nbit = fastmod(i ? ext_hash(base, k, i) : base, a.bf_mod);
                atomicOr(&bf[nbit >> 5], 1u << (uint32_t)(nbit & 31));
            }
        }
    }
    if (mine)
        atomicAdd(&s_private, mine);
    __syncthreads();
    if (tid == 0) {
        if (s_private)
            atomicAdd(&a.stats[7], s_private); // candidate operations settled by their k-mer's count
        if (bkt == 0)
            c.pair_mode[c.pair_queue[pair]] = 3u; // cbf_local replays the second bitmap's candidates
    }
}

// ---- the census level by level (stage_a_levels): the operations that are in the set after a pair's last
// pass go into its filter.  A workgroup per stretch, a quarter of it at a time, hashing as ckc_scatter does.
__global__ __launch_bounds__(256) void level_insert(AArgs a, CensusArgs c, const unsigned long long* bits)
{
    __shared__ uint8_t codes[kCkcQuarter + kHalo];
    __shared__ uint64_t seed_t[kSeedTableWords];
    __shared__ ulonglong2 roll_in[8], roll_out[8];
    const uint32_t pair = c.wg_pair[blockIdx.x];
    if (c.overflow[pair] || c.pair_need[pair] != c.level + 1)
        return; // (only the pairs whose last pass this was)
    const uint32_t b = c.pair_batch[pair], ki = c.pair_ki[pair], tid = threadIdx.x;
    const KConsts& kc = a.kc[ki];
    const uint32_t k = kc.k;
    const uint64_t b_start = a.batch_pos[b], b_end = a.batch_pos[b + 1];
    const uint64_t pos0 = c.wg_pos0[blockIdx.x];
    const uint64_t stretch_end = min(b_end, pos0 + (uint64_t)kCStretch);
    const uint8_t* const mem = reinterpret_cast<const uint8_t*>(bits + ((c.pair_bit_off[pair] + (pos0 - b_start)) >> 6));
    uint32_t* const bf = a.bf + ((uint64_t)b * a.n_k + ki) * a.bf_words;
    fill_seed_table(seed_t);
    if (tid < 8) {
        roll_in[tid] = tid < 4 ? make_ulonglong2(seed_of_code(tid), kc.in_rev[tid]) : make_ulonglong2(0, 0);
        roll_out[tid] = tid < 4 ? make_ulonglong2(kc.out_fwd[tid], seed_of_code(tid ^ 2)) : make_ulonglong2(0, 0);
    }
    for (uint32_t q = 0; q < 4; q++) {
        const uint64_t q0 = pos0 + (uint64_t)q * kCkcQuarter;
        if (q0 >= stretch_end)
            break; // uniform
        const uint64_t lane_pos0 = q0 + (uint64_t)tid * kCkcPerLane;
        uint32_t mb = lane_pos0 < stretch_end ? mem[q * (kCkcQuarter / 8) + tid] : 0u;
        if (lane_pos0 < stretch_end && stretch_end - lane_pos0 < 8)
            mb &= (1u << (stretch_end - lane_pos0)) - 1u;
        if (!__syncthreads_or((int)mb))
            continue; // nobody of this quarter is in the set (and the barrier the staging below needs)
        for (uint32_t i = tid; i < kCkcQuarter + kHalo; i += 256) {
            const uint64_t p = q0 + i;
            codes[i] = base_code(p < a.n_pos ? a.reads[p] : (uint8_t)'\n');
        }
        __syncthreads();
        if (mb) {
            const uint32_t t0 = tid * kCkcPerLane;
            const uint32_t j0 = (uint32_t)__ffs((int)mb) - 1u;
            uint64_t fh = 0, rh = 0;
            int no_bad = -1;
            kmer_hash_from_table(seed_t, codes + t0 + j0, k, fh, rh, no_bad);
            for (uint32_t j = j0; j < kCkcPerLane; j++) {
                if (j > j0) {
                    const ulonglong2 ri = roll_in[codes[t0 + j + k - 1]], ro = roll_out[codes[t0 + j - 1]];
                    fh = srol1(fh) ^ ri.x ^ ro.x;
                    rh = sror1(rh ^ ri.y ^ ro.y);
                }
                if (!((mb >> j) & 1u))
                    continue;
                const uint64_t base = fh + rh;
                for (uint32_t i = 0; i < a.hash_num; i++) {
                    const uint64_t nbit = fastmod(i ? ext_hash(base, k, i) : base, a.bf_mod);
                    atomicOr(&bf[nbit >> 5], 1u << (uint32_t)(nbit & 31));
                }
            }
        }
    }
}

// ---- many small filters: one workgroup per (batch, k) pair replays its operations alone.
// Nobody else touches the pair's counters, so the whole ordering problem stays inside the
// workgroup: a tile of 512 consecutive positions is hashed, what is settled goes straight
// into the filter, and the rest is ordered through an LDS hash table keyed by the counter
// index itself (exact: only operations that really share a counter wait for each other),
// with workgroup barriers instead of launches between the rounds and without a single
// global atomic on the way (device-scope atomics run at 27 G/s chip-wide whatever their
// target).  Tiles follow each other in position order, which IS the serial order.  What
// another wave of the workgroup wrote before a workgroup barrier is visible after it (the
// waves share the CU's L1); no device-scope fence is needed, and none is wanted: an
// agent-scope release writes the XCD's L2 back.  A round-0 winner has no earlier waiting
// operation of its tile on any of its counters, so the values loaded for the settle test
// are still the current ones: in a sparse filter an operation costs its four counter
// loads and the stores, nothing else.  (Replaying EVERY operation the kernel is bound by those
// random line fetches; replaying the census's candidates it was bound by hashing them from
// scratch and by one memory round trip behind the other, until the seed table, the settled
// table and the merged loads below: DESIGN.md section 4.)
// A fresh counting filter for the pair a cbf_local workgroup takes next.  The slot is that
// workgroup's alone; its stores are visible to its own waves after the next workgroup barrier
// (one CU, one L1).  Kept out of line, and the statistics flushed per pair: inlined, or with
// counters alive across it, the kernel needs 133 VGPRs instead of 127 and loses its fourth
// workgroup per CU (-15 % on 400 x 25 kbp).
__device__ __attribute__((noinline)) void zero_slot(uint8_t* cb, uint64_t bytes)
{
    uint4* z = reinterpret_cast<uint4*>(cb);
    const uint64_t n16 = bytes / 16;
    for (uint64_t i = threadIdx.x; i < n16; i += blockDim.x)
        z[i] = make_uint4(0, 0, 0, 0);
}

// The cheap way back to a clean slot after a SHORT pair (a GoldPolish-Target window is a few
// hundred bases with a handful of reads): walk its positions once more and store a zero at every
// counter a k-mer of it hashes to.  A pair of 4 k positions touches 16 k counters; clearing the
// whole 10 MiB slot would write 164 k lines for it.  One position per lane, hashed from scratch.
// (scalars by value: a reference to the kernel's argument block would force a copy of it into scratch)
__device__ __attribute__((noinline)) void unwind_slot(const uint8_t* reads, uint64_t n_pos, uint8_t* cb, uint32_t k,
                                                      uint32_t hash_num, FastMod cbf_mod, uint32_t nib,
                                                      uint64_t b_start, uint64_t b_end)
{
    for (uint64_t p0 = b_start; p0 < b_end; p0 += blockDim.x) {
        const uint64_t pos = p0 + threadIdx.x;
        if (pos >= b_end || pos + k > n_pos)
            continue;
        uint64_t fh = 0, rh = 0;
        bool bad = false;
        for (uint32_t i = 0; i < k; i++) {
            const uint32_t cf = base_code(reads[pos + i]);
            bad |= cf >= 4;
            fh = srol1(fh) ^ (cf < 4 ? seed_of_code(cf) : 0ULL);
        }
        if (bad)
            continue;
        for (uint32_t i = 0; i < k; i++) {
            const uint32_t cr = base_code(reads[pos + k - 1 - i]);
            rh = srol1(rh) ^ seed_of_code(cr ^ 2);
        }
        const uint64_t base = fh + rh;
        for (uint32_t h = 0; h < hash_num; h++) {
            const uint32_t ix = (uint32_t)fastmod(h ? ext_hash(base, k, h) : base, cbf_mod);
            cb[nib ? ix >> 1 : ix] = 0; // (packed: the neighbour nibble is on its way to zero too)
        }
    }
}

// position of the r-th (0-based) set bit of w; r < popcount(w)
__device__ inline uint32_t select64(unsigned long long w, uint32_t r)
{
    uint32_t pos = 0;
    uint32_t c = (uint32_t)__popc((uint32_t)w);
    if (r >= c) {
        r -= c;
        w >>= 32;
        pos = 32;
    }
    uint32_t lo = (uint32_t)w;
    c = (uint32_t)__popc(lo & 0xFFFFu);
    if (r >= c) {
        r -= c;
        lo >>= 16;
        pos += 16;
    }
    c = (uint32_t)__popc(lo & 0xFFu);
    if (r >= c) {
        r -= c;
        lo >>= 8;
        pos += 8;
    }
    c = (uint32_t)__popc(lo & 0xFu);
    if (r >= c) {
        r -= c;
        lo >>= 4;
        pos += 4;
    }
    c = (uint32_t)__popc(lo & 0x3u);
    if (r >= c) {
        r -= c;
        lo >>= 2;
        pos += 2;
    }
    return pos + (r >= (lo & 1u) ? 1u : 0u);
}

constexpr int kLTile = 512;            // operations per turn (positions of a tile, or candidates of a window)
constexpr int kLPer = 1;               // ... one per lane: 512-thread workgroups, two per CU (32 KB table + 16 KB filter each),
constexpr int kLBlock = kLTile / kLPer; // at 128 registers 16 waves per CU.  (Two operations per lane in 256-thread
                                        // workgroups -- 8 waves per CU -- took 0.54 s for configs[2]'s replay, this 0.48 s.)
constexpr int kLWin = 2048;            // census pairs: positions whose candidates are gathered at a time
constexpr int kLWinSparse = 8192;      // ... where only the candidates of shared k-mers are left (pair_mode 3: under 2 % of the positions)
constexpr int kLocalSlotBits = 12;
constexpr int kLocalSlots = 1 << kLocalSlotBits; // >= 2 x the counter touches of a tile (hash_num <= 4)
constexpr uint32_t kLocalEmpty = 0xFFFFFFFFu;
constexpr uint32_t kLSetBits = 1u << 17;   // LDS filter in front of the settled table: a candidate whose bit is clear is not in it

// `entry` = the k-mer's mixed hash with the bits its slot already fixes replaced by the pair's
// epoch (see cbf_local)
__device__ inline void kset_put(unsigned long long* kset, uint32_t* s_kbits, uint32_t slot, uint64_t base,
                                unsigned long long entry)
{
    kset[slot] = entry;
    const uint32_t kb = (uint32_t)((base * 0x9E3779B97F4A7C15ULL) >> 10) & (kLSetBits - 1u);
    atomicOr(&s_kbits[kb >> 5], 1u << (kb & 31u));
}

__global__ __launch_bounds__(kLBlock, 4) void cbf_local(AArgs a, uint32_t first, uint32_t n_pairs) // two 8-wave workgroups per CU: <= 128 VGPRs
{
    __shared__ uint8_t codes[kLWinSparse + kHalo];
    __shared__ uint64_t seed_t[kSeedTableWords];
    __shared__ uint32_t keys[kLocalSlots]; // counter index held by the slot
    __shared__ uint32_t vals[kLocalSlots]; // earliest waiting operation of the tile on that counter
    __shared__ unsigned long long s_stats[7]; // [5]: operations replayed (all of a plain pair, the candidates of a census pair), [6]: settled-table hits
    __shared__ uint32_t s_kbits[kLSetBits / 32];    // census: a bit per k-mer hash that has gone into the settled table
    __shared__ unsigned long long s_cw[kLWinSparse / 64]; // census: candidate bits of the window ...
    __shared__ uint32_t s_cpre[kLWinSparse / 64], s_ncand, s_wtot[2];   // ... candidates before each word, and in all
    // This workgroup's arena slot for the whole launch: the counting filter of whichever pair it
    // is replaying.  Pairs come from a queue (longest batches first), so the grid is as large as
    // the chip holds and no larger, and a short pair's workgroup goes on to the next one instead
    // of leaving its CU idle.  Every workgroup reaches the end of the queue and exits.
    uint8_t* const cb = a.cbf + (uint64_t)blockIdx.x * a.cbf_stride;
    __shared__ uint32_t s_pair;
    if (threadIdx.x < 7)
        s_stats[threadIdx.x] = 0;
    fill_seed_table(seed_t); // (the barriers at the top of the pair loop come before its first use)
    const int t0 = threadIdx.x * kLPer;
    bool slot_clean = false; // uniform: the previous pair was short and has been unwound
    const uint32_t vb = a.nib ? 0u : a.ctr_vbits, vmask = (1u << vb) - 1u, n_ep = vb ? (256u >> vb) - 1u : 0u;
    uint32_t ep = 0, ep_k = 0; // epoch tags of the counters and of the settled table (0: not cleared yet)
    // a counter of this workgroup's slot, as the current pair sees it / one step up
    auto ctr_get = [&](uint32_t idx) -> uint32_t {
        if (!vb)
            return ctr_load<false>(a, cb, idx);
        const uint32_t raw = cb[idx];
        return (raw >> vb) == ep ? raw & vmask : 0u;
    };
    auto ctr_up = [&](uint32_t idx, uint32_t v) {
        if (!vb)
            ctr_increment(a, cb, idx, v);
        else
            cb[idx] = (uint8_t)((ep << vb) | (v + 1u));
    };

    for (;;) {
    uint32_t n_ops = 0, n_ordered = 0, n_positions = 0, n_rounds = 0, n_replayed = 0, n_hits = 0;
    __syncthreads(); // the previous pair is finished by every wave
    if (threadIdx.x == 0)
        s_pair = atomicAdd(a.queue, 1u);
    __syncthreads();
    const uint32_t pair = s_pair;
    if (pair >= n_pairs)
        break;
    const uint32_t b = a.batch_order[first + pair / a.n_k];
    const uint32_t ki = pair % a.n_k;
    const KConsts& kc = a.kc[ki];
    const uint32_t k = kc.k;
    const uint32_t stream = b * a.n_k + ki;
    const uint64_t b_start = a.batch_pos[b], b_end = a.batch_pos[b + 1];
    // census: only the candidates are replayed, every counter they see is already at level 1, so
    // the array holds level - 1 and every threshold is one lower (see census_scatter)
    const uint32_t pmode = a.pair_mode ? a.pair_mode[pair] : 0u; // 0 plain, 1 census, 2 plain after a census that overflowed
    const uint32_t census = (pmode == 1u || pmode == 3u) ? 1u : 0u;
    const unsigned long long* const cand_bits = census ? (pmode == 3u ? a.cand2 : a.cand) + (a.pair_bit_off[pair] >> 6) : nullptr;
    const uint32_t win = pmode == 3u ? (uint32_t)kLWinSparse : (uint32_t)kLWin, n_words = win / 64u;
    // Counter bytes carry the pair's epoch in their upper bits (ctr_vbits value bits below it: no
    // counter passes its threshold): a byte whose tag is not the current pair's reads as zero, so
    // the slot is cleared once per 2^(8 - vbits) - 1 pairs, not before every pair -- clearing 10 MiB
    // for each of 16 000 pairs was 168 GB of writes per configs[2] step, more than the algorithmic
    // bytes of the whole step.
    if (vb) {
        if (ep == 0 || ep == n_ep) {
            zero_slot(cb, a.cbf_stride);
            ep = 1;
        } else {
            ep++;
        }
    } else if (!slot_clean) {
        zero_slot(cb, a.cbf_stride);
    }
    // Census pairs: the candidates are mostly later copies of k-mers that reached their threshold
    // long ago -- four counter look-ups and four filter look-ups each to learn that nothing is
    // left to do.  Once an operation finds ALL its counters >= S-1 (rule (b) at the top of the
    // file: it and every later operation on these counters insert and write nothing) and has put
    // its k-mer into the filter, every later copy of that k-mer is done before it starts; its
    // 64-bit hash (which fixes the counters and the filter bits) goes into a direct-mapped table
    // of the workgroup's own, and a candidate that finds its hash there costs that one look-up.
    // A table entry lost to another k-mer only costs the full look-ups again.
    // (thresholds of a census pair are kept one lower: see census_scatter; a batch of one contig
    // has one pair of them, fetched here and not behind every candidate's hash)
    const uint32_t seg_lo = a.batch_seg[b], seg_hi = a.batch_seg[b + 1];
    const uint32_t thr_one =
        seg_lo < seg_hi ? (a.seg_thr[seg_lo] - 2 + ki - 1) | ((a.seg_smax[seg_lo] - 2 + ki - 1) << 16) : 0u;
    unsigned long long* const kset =
        census && a.kset_mask ? a.kset + (uint64_t)blockIdx.x * ((uint64_t)a.kset_mask + 1) : nullptr;
    // The table's entries are epoch-tagged too, exactly: an entry is the k-mer's mixed hash (a
    // bijection of the 64-bit hash) with bits 32.. -- which the slot index already fixes -- replaced
    // by the pair's ordinal, so an entry of an earlier pair never compares equal and the 2 MiB table
    // is cleared once per launch, not per pair.
    if (kset) {
        if (ep_k == 0 || ep_k == a.kset_mask) {
            zero_slot(reinterpret_cast<uint8_t*>(kset), ((uint64_t)a.kset_mask + 1) * 8);
            ep_k = 1;
        } else {
            ep_k++;
        }
        for (uint32_t i = threadIdx.x; i < kLSetBits / 32; i += kLBlock)
            s_kbits[i] = 0; // (the barrier at the top of the first turn comes before its first use)
    }
    const unsigned long long kfield = (unsigned long long)a.kset_mask << 32, ktag = (unsigned long long)ep_k << 32;

    // A turn of this loop replays up to 512 operations, two per lane, in serial order.  Plain
    // pairs: the next 512 positions.  Census pairs: the next 512 CANDIDATES of a 2048-position
    // window (a fifth to a third of its positions), so that a turn is as full as a plain one.
    uint64_t next0 = b_start;      // plain: next tile; census: next window
    uint32_t c0 = 0, n_c = 0;      // census: next candidate of the window, candidates in it
    uint64_t win0 = b_start;
    for (;;) {
        uint32_t op_state[kLPer] = {}, op_thr[kLPer] = {}, op_c[kLPer] = {};
        uint32_t op_idx[kLPer][4] = {};
        uint64_t op_base[kLPer] = {};
        uint32_t op_set[kLPer] = {};
        unsigned long long op_tag[kLPer] = {};
        __syncthreads(); // everyone is done with the previous turn (codes, table)
        if (!census) {
        if (next0 >= b_end)
            break;
        const uint64_t tile0 = next0;
        next0 += kLTile;
        const uint64_t tile_end = min(b_end, tile0 + (uint64_t)kLTile);
        for (int i = threadIdx.x; i < kLTile + kHalo; i += kLBlock) {
            uint64_t p = tile0 + i;
            uint8_t c = p < a.n_pos ? a.reads[p] : (uint8_t)'\n';
            codes[i] = base_code(c);
        }
        __syncthreads();

        const uint64_t pos0 = tile0 + t0;
        if (pos0 < tile_end) {
            uint32_t lo = seg_lo, hi = seg_hi; // the batch's contigs
            while (hi - lo > 1) {
                uint32_t mid = (lo + hi) >> 1;
                if (a.seg_start[mid] <= pos0)
                    lo = mid;
                else
                    hi = mid;
            }
            uint32_t seg = lo;
            uint64_t seg_end = a.seg_start[seg + 1];
            uint64_t fh = 0, rh = 0;
            int last_bad = -1;
            kmer_hash_from_table(seed_t, codes + t0, k, fh, rh, last_bad);
#pragma unroll
            for (int j = 0; j < kLPer; j++) {
                const uint64_t pos = pos0 + j;
                if (pos >= tile_end)
                    continue;
                if (j > 0) {
                    uint32_t out = codes[t0 + j - 1], in = codes[t0 + j + k - 1];
                    uint64_t s_in = in < 4 ? seed_of_code(in) : 0ULL;
                    uint64_t s_outk = out < 4 ? kc.out_fwd[out] : 0ULL;
                    uint64_t r_ink = in < 4 ? kc.in_rev[in] : 0ULL;
                    uint64_t r_out = out < 4 ? seed_of_code(out ^ 2) : 0ULL;
                    fh = srol1(fh) ^ s_in ^ s_outk;
                    rh = sror1(rh ^ r_ink ^ r_out);
                    if (in >= 4)
                        last_bad = j + (int)k - 1;
                }
                n_positions++;
                if (last_bad >= j || pos + k > a.n_pos)
                    continue;
                while (pos >= seg_end) {
                    seg++;
                    seg_end = a.seg_start[seg + 1];
                }
                n_ops++;
                op_base[j] = fh + rh;
                op_thr[j] = (a.seg_thr[seg] - 2 + ki) | ((a.seg_smax[seg] - 2 + ki) << 16); // own threshold | S
                op_state[j] = 1;
                n_replayed++;
                uint32_t pack = 0;
                for (uint32_t h = 0; h < 4; h++) { // hash_num <= 4 on this path (stage_a_run)
                    if (h >= a.hash_num)
                        break;
                    const uint64_t hv = h ? ext_hash(op_base[j], k, h) : op_base[j];
                    op_idx[j][h] = (uint32_t)fastmod(hv, a.cbf_mod);
                    pack |= ctr_get(op_idx[j][h]) << (8 * h);
                }
                op_c[j] = pack;
            }
        }
        } else {
        if (c0 >= n_c) {
            // the next window that holds a candidate
            for (;;) {
                if (next0 >= b_end)
                    break;
                win0 = next0;
                next0 += win;
                // the window's bases and its candidate bits in one round trip (the loads first, all
                // in flight together, then the stores)
                {
                    constexpr int kPer = (kLWinSparse + kHalo + kLBlock - 1) / kLBlock;
                    uint8_t raw[kPer];
#pragma unroll
                    for (int q = 0; q < kPer; q++) {
                        const uint64_t p = win0 + (uint64_t)(q * kLBlock) + threadIdx.x;
                        raw[q] = (q * kLBlock + (int)threadIdx.x < (int)win + kHalo && p < a.n_pos) ? a.reads[p] : (uint8_t)'\n';
                    }
#pragma unroll
                    for (int q = 0; q < kPer; q++)
                        if (q * kLBlock + (int)threadIdx.x < (int)win + kHalo)
                            codes[q * kLBlock + threadIdx.x] = base_code(raw[q]);
                }
                if (threadIdx.x < 128) { // waves 0 and 1: a word per lane (32 words of a plain census window, 128 of a sparse one)
                    unsigned long long w = 0;
                    if (threadIdx.x < n_words) {
                        const uint64_t first = win0 + (uint64_t)threadIdx.x * 64; // bits past the batch are not the census's
                        if (first < b_end) {
                            w = cand_bits[((win0 - b_start) >> 6) + threadIdx.x];
                            if (b_end - first < 64)
                                w &= (1ULL << (b_end - first)) - 1;
                        }
                        s_cw[threadIdx.x] = w;
                    }
                    const uint32_t cnt = (uint32_t)__popcll(w);
                    uint32_t inc = cnt;
                    for (int o = 1; o < 64; o <<= 1) {
                        const uint32_t u = (uint32_t)__shfl_up((int)inc, o);
                        if ((int)(threadIdx.x & 63) >= o)
                            inc += u;
                    }
                    if ((threadIdx.x & 63) == 63)
                        s_wtot[threadIdx.x >> 6] = inc;
                    if (threadIdx.x < 64)
                        s_cpre[threadIdx.x] = inc - cnt; // (the second wave's words get the first wave's total below)
                    else if (threadIdx.x < n_words)
                        s_cpre[threadIdx.x] = inc - cnt;
                }
                __syncthreads();
                if (threadIdx.x >= 64 && threadIdx.x < n_words)
                    s_cpre[threadIdx.x] += s_wtot[0];
                if (threadIdx.x == 0)
                    s_ncand = s_wtot[0] + (n_words > 64 ? s_wtot[1] : 0u);
                __syncthreads();
                n_c = s_ncand;
                c0 = 0;
                __syncthreads(); // s_ncand may be rewritten by the next window
                if (n_c)
                    break;
            }
            if (c0 >= n_c)
                break; // no candidate left in the batch (uniform)
        }
        // the hashes first, and the look-ups in the table of settled k-mers in flight together
        unsigned long long seen[kLPer] = {};
        uint64_t op_pos[kLPer] = {};
#pragma unroll
        for (int j = 0; j < kLPer; j++) {
            const uint32_t ci = c0 + (uint32_t)(t0 + j);
            if (ci >= n_c)
                continue;
            // the ci-th candidate of the window: its word, then its bit
            uint32_t wlo = 0;
            for (uint32_t step = n_words / 2; step; step >>= 1)
                if (s_cpre[wlo + step] <= ci)
                    wlo += step;
            const uint32_t prel = wlo * 64 + select64(s_cw[wlo], ci - s_cpre[wlo]);
            const uint64_t pos = win0 + prel;
            uint64_t fh = 0, rh = 0;
            int no_bad = -1; // a candidate is a k-mer: no byte of it is bad
            kmer_hash_from_table(seed_t, codes + prel, k, fh, rh, no_bad);
            op_base[j] = fh + rh;
            op_pos[j] = pos;
            op_state[j] = 1;
            if (kset) {
                const uint64_t mix = op_base[j] * 0x9E3779B97F4A7C15ULL;
                op_set[j] = (uint32_t)(mix >> 32) & a.kset_mask;
                const uint32_t kb = (uint32_t)(mix >> 10) & (kLSetBits - 1u);
                // (more than half of the candidates are not in the table: most of them learn it here)
                seen[j] = (s_kbits[kb >> 5] >> (kb & 31u)) & 1u ? kset[op_set[j]] : 0ull;
                op_tag[j] = (mix & ~kfield) | ktag;
            }
        }
#pragma unroll
        for (int j = 0; j < kLPer; j++) {
            if (!op_state[j])
                continue;
            n_replayed++;
            if (kset && seen[j] == op_tag[j]) {
                op_state[j] = 0; // settled for good and in the filter
                n_hits++;
                continue;
            }
            const uint64_t pos = op_pos[j];
            op_thr[j] = thr_one;
            if (seg_hi - seg_lo > 1) { // several contigs in the batch: the one whose reads hold the position
                uint32_t seg = seg_lo;
                for (uint32_t hi = seg_hi; hi - seg > 1;) {
                    const uint32_t mid = (seg + hi) >> 1;
                    if (a.seg_start[mid] <= pos)
                        seg = mid;
                    else
                        hi = mid;
                }
                op_thr[j] = (a.seg_thr[seg] - 2 + ki - 1) | ((a.seg_smax[seg] - 2 + ki - 1) << 16);
            }
            uint32_t pack = 0;
            for (uint32_t h = 0; h < 4; h++) {
                if (h >= a.hash_num)
                    break;
                const uint64_t hv = h ? ext_hash(op_base[j], k, h) : op_base[j];
                op_idx[j][h] = (uint32_t)fastmod(hv, a.cbf_mod);
                pack |= ctr_get(op_idx[j][h]) << (8 * h);
            }
            op_c[j] = pack;
        }
        c0 += kLTile;
        }

        // rounds: every waiting operation enters its counters into the table; who is the
        // earliest on all of them has no earlier waiting operation on any of its counters and
        // applies the serial update.  Round 0 works on the values loaded above.
        for (uint32_t round = 0;; round++) {
            uint32_t mine_left = 0;
#pragma unroll
            for (int j = 0; j < kLPer; j++) {
                if (!op_state[j])
                    continue;
                if (round == 0 || (round & 7u) == 7u) {
                    // settled (top of the file): goes in, nothing to write.  Waiting operations
                    // look again every eighth round: a run of one k-mer (a homopolymer) would
                    // otherwise take a round per copy long after its counters are high enough.
                    uint32_t m = 255;
                    for (uint32_t h = 0; h < a.hash_num; h++) {
                        const uint32_t c = round == 0 ? (op_c[j] >> (8 * h)) & 255u
                                                      : ctr_get(op_idx[j][h]);
                        m = min(m, c);
                    }
                    if (m >= (op_thr[j] & 0xFFFFu) || m + 1 >= (op_thr[j] >> 16)) {
                        uint64_t hv[kMaxHash];
                        hv[0] = op_base[j];
                        for (uint32_t i = 1; i < a.hash_num; i++)
                            hv[i] = ext_hash(op_base[j], k, i);
                        bf_insert(a, stream, hv);
                        if (kset && m + 1 >= (op_thr[j] >> 16))
                            kset_put(kset, s_kbits, op_set[j], op_base[j], op_tag[j]);
                        op_state[j] = 0;
                        continue;
                    }
                    n_ordered += round == 0;
                }
                mine_left++;
            }
            if (!__syncthreads_or((int)mine_left))
                break;
            n_rounds += threadIdx.x == 0;
            for (int i = threadIdx.x; i < kLocalSlots; i += kLBlock) {
                keys[i] = kLocalEmpty;
                vals[i] = kLocalEmpty;
            }
            __syncthreads();
            uint32_t op_slot[kLPer][4];
#pragma unroll
            for (int j = 0; j < kLPer; j++) {
                if (!op_state[j])
                    continue;
                for (uint32_t h = 0; h < a.hash_num; h++) {
                    const uint32_t ix = op_idx[j][h];
                    uint32_t sl = (ix * 0x9E3779B1u) >> (32 - kLocalSlotBits);
                    for (;;) { // linear probing; the table is at most half full
                        const uint32_t old = atomicCAS(&keys[sl], kLocalEmpty, ix);
                        if (old == kLocalEmpty || old == ix)
                            break;
                        sl = (sl + 1) & (kLocalSlots - 1);
                    }
                    atomicMin(&vals[sl], (uint32_t)(t0 + j));
                    op_slot[j][h] = sl;
                }
            }
            __syncthreads();
#pragma unroll
            for (int j = 0; j < kLPer; j++) {
                if (!op_state[j])
                    continue;
                bool mine = true;
                for (uint32_t h = 0; h < a.hash_num; h++)
                    mine &= vals[op_slot[j][h]] == (uint32_t)(t0 + j);
                if (!mine)
                    continue;
                uint32_t c[4], m = 255;
                for (uint32_t h = 0; h < a.hash_num; h++) {
                    c[h] = round == 0 ? (op_c[j] >> (8 * h)) & 255u : ctr_get(op_idx[j][h]);
                    m = min(m, c[h]);
                }
                const uint32_t thr = op_thr[j] & 0xFFFFu;
                if (m < thr) {
                    for (uint32_t h = 0; h < a.hash_num; h++) {
                        bool first = c[h] == m;
                        for (uint32_t g = 0; g < h; g++)
                            first &= op_idx[j][g] != op_idx[j][h]; // a counter hit twice still moves by one
                        if (first)
                            ctr_up(op_idx[j][h], m);
                    }
                    m++;
                }
                if (m >= thr) {
                    uint64_t hv[kMaxHash];
                    hv[0] = op_base[j];
                    for (uint32_t h = 1; h < a.hash_num; h++)
                        hv[h] = ext_hash(op_base[j], k, h);
                    bf_insert(a, stream, hv);
                    if (kset && m + 1 >= (op_thr[j] >> 16))
                        kset_put(kset, s_kbits, op_set[j], op_base[j], op_tag[j]); // every counter is >= m now
                }
                op_state[j] = 0;
            }
            // the barrier at the top of the loop orders these stores before the next loads
        }
    }
    if (pmode == 0u) { // (census_scatter has counted the others)
        atomicAdd(&s_stats[0], (unsigned long long)n_positions);
        atomicAdd(&s_stats[1], (unsigned long long)n_ops);
    }
    atomicAdd(&s_stats[2], (unsigned long long)n_ordered);
    atomicAdd(&s_stats[3], (unsigned long long)n_rounds);
    atomicAdd(&s_stats[5], (unsigned long long)n_replayed);
    if (n_hits)
        atomicAdd(&s_stats[6], (unsigned long long)n_hits);
    // fewer counter lines touched than a quarter of the slot holds: unwind instead of clearing
    slot_clean = !vb && (b_end - b_start) * a.hash_num * 64 * 4 < a.cbf_stride;
    if (slot_clean) {
        __syncthreads(); // the pair's last stores are done
        unwind_slot(a.reads, a.n_pos, cb, k, a.hash_num, a.cbf_mod, a.nib, b_start, b_end);
    }
    } // next pair
    __syncthreads();
    if (threadIdx.x < 7 && threadIdx.x != 4 && s_stats[threadIdx.x])
        atomicAdd(&a.stats[threadIdx.x], s_stats[threadIdx.x]);
}

// ------------------------------------------------------------------ host side
int KernelTimer::begin(hipStream_t s, uint64_t n_units)
{
    if (used == start.size()) {
        hipEvent_t a, b;
        GP_HIP(hipEventCreate(&a));
        GP_HIP(hipEventCreate(&b));
        start.push_back(a);
        stop.push_back(b);
        units.push_back(0);
    }
    units[used] = n_units;
    GP_HIP(hipEventRecord(start[used], s));
    return GP_OK;
}

int KernelTimer::end(hipStream_t s)
{
    GP_HIP(hipEventRecord(stop[used], s));
    used++;
    return GP_OK;
}

int KernelTimer::collect()
{
    for (size_t i = 0; i < used; i++) {
        float ms = 0;
        GP_HIP(hipEventElapsedTime(&ms, start[i], stop[i]));
        total_ms += ms;
        launches++;
        total_units += units[i];
    }
    used = 0;
    return GP_OK;
}

void KernelTimer::destroy()
{
    for (auto e : start)
        (void)hipEventDestroy(e);
    for (auto e : stop)
        (void)hipEventDestroy(e);
    start.clear();
    stop.clear();
    units.clear();
    used = 0;
}

int stage_a_prepare(gp_job* job)
{
    const gp_params& p = job->prm;
    // window: the most read positions one first-pass launch may cover.  The waiting lists
    // are sized for the worst case (every k-mer of every k of a window waits).
    uint64_t w = env_u64("GP_STAGE_A_WINDOW", 8ull << 20);
    w = std::max<uint64_t>(w, kTile);
    w = std::min<uint64_t>(w, std::max<uint64_t>(job->read_positions, kTile));
    w = (w + kTile - 1) / kTile * kTile;
    job->window = w;
    const uint64_t max_pending = w * p.n_k;
    if (max_pending >= (1ull << 31)) {
        set_error("stage A window too large");
        return GP_ERR_INVALID_ARG;
    }
    int rc;
    // two lists (steps alternate), two uint4 per record
    for (int i = 0; i < 2; i++)
        if ((rc = job->d_pending[i].alloc(2 * max_pending)) != GP_OK)
            return rc;
    // reservation slots: two per possible waiting counter touch, power of two, <= 512 MiB a table
    uint32_t bits = 16;
    while ((1ull << bits) < max_pending * p.hash_num * 2 && bits < 26)
        bits++;
    job->resv_bits = bits;
    if ((rc = job->d_resv.alloc(2ull << bits)) != GP_OK) // two tables
        return rc;
    if ((rc = job->d_counts.alloc((uint64_t)(kSteps + 1) * kCountStride)) != GP_OK) // + a line for the error flag
        return rc;
    if ((rc = job->d_stats.alloc(16)) != GP_OK)
        return rc;
    if ((rc = job->d_batch_pos.alloc((uint64_t)job->n_batches + 1)) != GP_OK ||
        (rc = job->d_batch_order.alloc(job->n_batches)) != GP_OK)
        return rc;
    if ((rc = job->d_sat.alloc((uint64_t)job->n_batches * p.n_k)) != GP_OK ||
        (rc = job->d_batch_thr_max.alloc(job->n_batches)) != GP_OK ||
        (rc = job->d_batch_slot.alloc(job->n_batches)) != GP_OK || (rc = job->d_queue.alloc(32)) != GP_OK)
        return rc;
    {
        // thresholds that matter are those of contigs whose reads reach the filter
        std::vector<uint32_t> tm(job->n_batches, 4), smax(std::max<uint32_t>(job->n_contigs, 1), 4);
        for (uint32_t bi = 0; bi < job->n_batches; bi++) {
            uint32_t run = 4;
            for (uint32_t c = job->h_batch_contig_begin[bi + 1]; c-- > job->h_batch_contig_begin[bi];) {
                if (job->h_seg_start[c + 1] > job->h_seg_start[c])
                    run = std::max(run, job->h_contig_thr[c]);
                smax[c] = run;
            }
            tm[bi] = run;
        }
        // (on the context's stream, never the null stream: a null-stream copy waits for every BLOCKING stream
        // of the device -- CU-masked streams are created blocking -- i.e. for another job's kernels)
        hipStream_t s = job->ctx->stream;
        if (job->n_batches)
            GP_HIP(hipMemcpyAsync(job->d_batch_thr_max.p, tm.data(), tm.size() * sizeof(uint32_t),
                                  hipMemcpyHostToDevice, s));
        if ((rc = job->d_seg_smax.alloc(smax.size())) != GP_OK || (rc = job->d_batch_seg.alloc(job->n_batches + 1)) != GP_OK)
            return rc;
        GP_HIP(hipMemcpyAsync(job->d_batch_seg.p, job->h_batch_contig_begin.data(), (job->n_batches + 1) * sizeof(uint32_t),
                              hipMemcpyHostToDevice, s));
        GP_HIP(hipMemcpyAsync(job->d_seg_smax.p, smax.data(), smax.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        GP_HIP(hipStreamSynchronize(s)); // tm and smax are locals
    }
    // streams, events and the pinned ring are the CONTEXT's (gp_ctx::StageA): made once, borrowed by every job
    gp_ctx::StageA& r = job->ctx->a;
    if (!r.ready) {
        static_assert(kRing <= 16, "gp_ctx::StageA holds 16 events per ring");
        if (!r.h_count_pinned)
            GP_HIP(hipHostMalloc((void**)&r.h_count_pinned, 4 * kRing * sizeof(uint32_t)));
        for (uint32_t i = 0; i < kRing; i++) {
            if (!r.ev[i])
                GP_HIP(hipEventCreateWithFlags(&r.ev[i], hipEventDisableTiming));
            if (!r.ev_b[i])
                GP_HIP(hipEventCreateWithFlags(&r.ev_b[i], hipEventDisableTiming));
        }
        if (!r.ev_init)
            GP_HIP(hipEventCreateWithFlags(&r.ev_init, hipEventDisableTiming));
        int rcs;
        if ((!r.stream2 && (rcs = ctx_stream_create(job->ctx, &r.stream2)) != GP_OK) ||
            (!r.stream3 && (rcs = ctx_stream_create(job->ctx, &r.stream3)) != GP_OK))
            return rcs;
        for (hipEvent_t& e : r.ev_census)
            if (!e)
                GP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        r.ready = true;
    }
    job->h_count_pinned = r.h_count_pinned;
    for (uint32_t i = 0; i < kRing; i++) {
        job->a_ev[i] = r.ev[i];
        job->a_ev_b[i] = r.ev_b[i];
    }
    job->a_ev_init = r.ev_init;
    job->a_stream2 = r.stream2;
    job->a_stream3 = r.stream3;
    for (int i = 0; i < 9; i++)
        job->a_ev_census[i] = r.ev_census[i];
    return GP_OK;
}

void stage_a_ctx_release(gp_ctx* ctx)
{
    gp_ctx::StageA& r = ctx->a;
    if (r.h_count_pinned)
        (void)hipHostFree(r.h_count_pinned);
    auto drop = [](hipEvent_t& e) {
        if (e)
            (void)hipEventDestroy(e);
        e = nullptr;
    };
    for (hipEvent_t& e : r.ev)
        drop(e);
    for (hipEvent_t& e : r.ev_b)
        drop(e);
    drop(r.ev_init);
    for (hipEvent_t& e : r.ev_census)
        drop(e);
    if (r.stream2)
        (void)hipStreamDestroy(r.stream2);
    if (r.stream3)
        (void)hipStreamDestroy(r.stream3);
    r = gp_ctx::StageA();
}

void stage_a_release(gp_job* job)
{
    // Idempotent: every handle is null once it is gone, so ensure_work's failure path followed by
    // gp_job_destroy (or by a second try) gives nothing back twice; the work buffers go back to the pool.
    // Streams, events and the pinned ring were borrowed from the context (stage_a_prepare): forgotten, not destroyed.
    job->h_count_pinned = nullptr;
    for (hipEvent_t& e : job->a_ev)
        e = nullptr;
    for (hipEvent_t& e : job->a_ev_b)
        e = nullptr;
    job->a_ev_init = nullptr;
    for (hipEvent_t& e : job->a_ev_census)
        e = nullptr;
    job->a_stream2 = job->a_stream3 = nullptr;
    job->d_resv.release();
    job->d_pending[0].release();
    job->d_pending[1].release();
    job->d_sat.release();
    job->d_batch_thr_max.release();
    job->d_batch_pos.release();
    job->d_batch_order.release();
    job->d_batch_slot.release();
    job->d_queue.release();
    job->d_cand.release();
    job->d_cand2.release();
    job->d_kset.release();
    job->d_census_cnt.release();
    job->d_census_rec.release();
    job->d_census_over.release();
    job->d_census_tab32.release();
    job->d_pair_mode.release();
    job->d_census_tab64.release();
    job->d_pair_bit_off.release();
    job->d_counts.release();
    job->d_stats.release();
    job->d_seg_smax.release();
    job->d_batch_seg.release();
}

int stage_a_run(gp_job* job)
{
    const gp_params& p = job->prm;
    hipStream_t s = job->ctx->stream;
    // Whichever way this function is left -- the early return of a job without reads, an error
    // half way, the end -- both of its streams are drained first: the caller (gp_jobs_polish runs
    // the back half on another stream) may take "returned" as "the filters are there", and the
    // copies enqueued below read from tables that live on this stack frame.
    struct Drain {
        hipStream_t a, b, c;
        ~Drain()
        {
            (void)hipStreamSynchronize(a);
            if (b)
                (void)hipStreamSynchronize(b);
            if (c)
                (void)hipStreamSynchronize(c);
        }
    } drain{ s, job->a_stream2, job->a_stream3 };
    AArgs a{};
    a.reads = job->d_reads.p;
    a.n_pos = job->read_positions;
    a.seg_start = job->d_seg_start.p;
    a.seg_batch = job->d_seg_batch.p;
    a.seg_thr = job->d_seg_thr.p;
    a.n_seg = job->n_contigs;
    a.cbf = job->d_cbf.p;
    a.bf = job->d_bf.p;
    a.n_k = p.n_k;
    a.hash_num = p.hash_num;
    a.cbf_bytes = p.cbf_bytes;
    {
        uint32_t max_thr = 0;
        for (uint32_t c = 0; c < job->n_contigs; c++)
            max_thr = std::max(max_thr, job->h_contig_thr[c] - 2 + p.n_k - 1);
        // Measured on MI355X: packing halves the fetch traffic of the counter loads but
        // turns every increment into an atomic; end to end it is 1.5 % (config 2) to 4 %
        // (25 kbp contigs) slower, so it is opt-in: GP_STAGE_A_NIBBLE=1.
        a.nib = (max_thr <= 15 && env_u64("GP_STAGE_A_NIBBLE", 0)) ? 1u : 0u;
        // cbf_local's epoch tags: the bits a counter value needs, the rest of the byte is the tag
        // (GP_STAGE_A_EPOCH=0: none, every pair clears its slot)
        uint32_t vbits = 1;
        while ((1u << vbits) <= max_thr)
            vbits++;
        a.ctr_vbits = (vbits <= 6 && !a.nib && env_u64("GP_STAGE_A_EPOCH", 1)) ? vbits : 0u;
        a.cbf_stride = ((a.nib ? (p.cbf_bytes + 1) / 2 : p.cbf_bytes) + 15) / 16 * 16;
    }
    a.bf_words = (uint32_t)(p.bf_bytes / 4);
    a.cbf_mod = make_fastmod(p.cbf_bytes);
    a.bf_mod = make_fastmod(p.bf_bytes * 8);
    for (uint32_t i = 0; i < p.n_k; i++) {
        a.kc[i].k = p.k[i];
        for (uint32_t c = 0; c < 4; c++) {
            a.kc[i].out_fwd[c] = sroln(seed_of_code(c), p.k[i]);
            a.kc[i].in_rev[c] = sroln(seed_of_code(c ^ 2), p.k[i]);
        }
    }
    a.bits_in = a.bits_out = std::min<uint32_t>(18, job->resv_bits);
    a.stats = job->d_stats.p;
    a.sat = job->d_sat.p;
    a.batch_thr_max = job->d_batch_thr_max.p;
    a.seg_smax = job->d_seg_smax.p;
    a.batch_seg = job->d_batch_seg.p;
    a.batch_slot = job->d_batch_slot.p;
    a.queue = job->d_queue.p;
    a.host_ring = job->h_count_pinned;
    a.tail_max = (uint32_t)env_u64("GP_STAGE_A_TAIL", 4096);

    job->t_first_pass.reset();
    job->t_round.reset();
    job->t_local.reset();
    job->t_census.reset();
    job->t_ckc.reset();
    job->counters.a_rounds = 0;

    // fresh filters + free reservation tables (the counting-filter arena is cleared slot by slot,
    // by whoever takes a slot: cbf_local's workgroups, and every group of the grid-wide schedule)
    GP_HIP(hipMemsetAsync(job->d_bf.p, 0, job->d_bf.n * sizeof(uint32_t), s));
    GP_HIP(hipMemsetAsync(job->d_queue.p, 0, job->d_queue.n * sizeof(uint32_t), s));
    GP_HIP(hipMemsetAsync(job->d_stats.p, 0, 16 * sizeof(unsigned long long), s));
    GP_HIP(hipMemsetAsync(job->d_sat.p, 0, job->d_sat.n * sizeof(uint32_t), s));
    GP_HIP(hipMemsetAsync(job->d_counts.p, 0, job->d_counts.n * sizeof(uint32_t), s));
    GP_HIP(hipMemsetAsync(job->d_resv.p, 0xFF, job->d_resv.n * sizeof(unsigned long long), s));
    // stream `s` runs the first passes, stream `s2` the rounds and tails
    // (`s2` is touched only when a batch really takes the grid-wide schedule: a job of short
    // batches then keeps to one stream, and gp_jobs_polish to two for any number of jobs)
    hipStream_t s2 = job->a_stream2;
    GP_HIP(hipEventRecord(job->a_ev_init, s));

    if (job->read_positions == 0 || job->n_contigs == 0) {
        job->filters_ready = true;
        return GP_OK;
    }

    std::vector<uint64_t> batch_pos(job->n_batches + 1);
    for (uint32_t bi = 0; bi < job->n_batches; bi++)
        batch_pos[bi] = job->h_seg_start[job->h_batch_contig_begin[bi]];
    batch_pos[job->n_batches] = job->read_positions;
    GP_HIP(hipMemcpyAsync(job->d_batch_pos.p, batch_pos.data(), batch_pos.size() * sizeof(uint64_t),
                          hipMemcpyHostToDevice, s));
    a.batch_pos = job->d_batch_pos.p;
    // Which batches replay alone in a workgroup (cbf_local) and which go through the grid-wide
    // reservation rounds?  The local replay pays when there are enough batches to fill the chip
    // and none of them is long; a long contig's batch needs the whole chip on its one filter.
    // GP_STAGE_A_LOCAL: 0 never, 1 every batch, default by size.
    std::vector<uint32_t> order;
    uint32_t n_big = job->n_batches, n_lvl = 0;
    std::vector<uint32_t> lvl_batches, lvl_thr; // batches of the level-by-level census (in `order` behind the grid-wide ones) and their thresholds
    // the census's geometry (counter ranges a power of two wide), as the local schedule's census takes it
    const bool lv_wide = env_u64("GP_STAGE_A_CENSUS_WIDE", 1) != 0;
    uint32_t lv_shift = 2;
    while (((p.cbf_bytes + (1ull << lv_shift) - 1) >> lv_shift) > (lv_wide ? kCBucketsWide : kCBuckets))
        lv_shift++;
    const uint32_t lv_buckets = (uint32_t)((p.cbf_bytes + (1ull << lv_shift) - 1) >> lv_shift);
    const uint64_t lv_budget = (env_u64("GP_STAGE_A_CENSUS_MB", 16384) << 20) / 4; // record slots of a cohort (one buffer set)
    {
        // A workgroup replays its pair at about 15 us per 512-position tile when it has the memory
        // system to itself; all of them together are bound by random line traffic, about 4.5 G
        // operations/s.  A batch may go local as long as its own replay stays well inside the time
        // the whole local set needs anyway (so that it is not the critical path), or is short.
        const uint64_t mode = p.hash_num <= 4 ? env_u64("GP_STAGE_A_LOCAL", 2) : 0; // the LDS table holds 4 touches per operation
        uint64_t local_max = env_u64("GP_STAGE_A_LOCAL_MAX", 3ull << 19); // positions per batch
        if (mode == 2) {
            double t_all = (double)job->read_positions * p.n_k / 4.5e9;   // s, if every batch went local
            local_max = std::max<uint64_t>(local_max, (uint64_t)(0.5 * t_all / 15e-6 * 512.0));
        }
        std::vector<uint32_t> small;
        for (uint32_t bi = 0; bi < job->n_batches; bi++) {
            const uint64_t len = batch_pos[bi + 1] - batch_pos[bi];
            if (mode == 1 || (mode == 2 && len <= local_max))
                small.push_back(bi);
            else
                order.push_back(bi);
        }
        if (mode == 2 && small.size() * p.n_k < 128) { // too few to fill the chip
            order.clear();
            small.clear();
            for (uint32_t bi = 0; bi < job->n_batches; bi++)
                order.push_back(bi);
        }
        // Long batches whose filters stay UNSATURATED take the census level by level (stage A's third
        // schedule, below): one threshold per filter (a batch of one contig always has), every pair's
        // records inside a cohort's budget, and no more than GP_STAGE_A_LEVELS_LOAD counter touches per
        // counter (default 45: a contig of about 2.9 Mbp at 40x on GoldPolish's 10 MiB; beyond that the
        // filter saturates, the grid-wide schedule settles nearly everything in its first pass and the
        // levels would take their twelve passes for nothing).  GP_STAGE_A_LEVELS=0: never.
        {
            const bool levels_on = env_u64("GP_STAGE_A_LEVELS", 1) != 0 && p.hash_num <= 4 && lv_shift <= kCMaxShift;
            const double max_load = (double)env_u64("GP_STAGE_A_LEVELS_LOAD", 45);
            std::vector<uint32_t> grid;
            for (uint32_t bi : order) {
                const uint64_t len = batch_pos[bi + 1] - batch_pos[bi];
                bool uniform = true, any = false;
                uint32_t thr0 = 0;
                for (uint32_t cc = job->h_batch_contig_begin[bi]; cc < job->h_batch_contig_begin[bi + 1]; cc++) {
                    if (job->h_seg_start[cc + 1] == job->h_seg_start[cc])
                        continue;
                    if (!any) {
                        any = true;
                        thr0 = job->h_contig_thr[cc];
                    } else if (job->h_contig_thr[cc] != thr0) {
                        uniform = false;
                    }
                }
                const uint64_t nst = (len + kCStretch - 1) / kCStretch;
                const uint64_t recs = census_runs((uint32_t)nst, lv_buckets) * (lv_wide ? kCRunWide : kCRun);
                const bool ok = levels_on && any && uniform && len > 0 && nst <= std::min<uint64_t>(31000, env_u64("GP_STAGE_A_LEVELS_NST", 20000)) && // (census_mark_seg keeps a length byte per stretch in LDS beside its 64 KB table: 96 KB asked for below)
                                (double)len * p.hash_num <= max_load * (double)p.cbf_bytes && recs <= lv_budget;
                if (ok) {
                    lvl_batches.push_back(bi);
                    lvl_thr.push_back(thr0);
                } else {
                    grid.push_back(bi);
                }
            }
            order.swap(grid);
        }
        n_big = (uint32_t)order.size();
        n_lvl = (uint32_t)lvl_batches.size();
        order.insert(order.end(), lvl_batches.begin(), lvl_batches.end());
        // the queue of cbf_local hands out the longest batches first: the last workgroups to
        // finish then work on short ones
        std::stable_sort(small.begin(), small.end(), [&](uint32_t x, uint32_t y) {
            return batch_pos[x + 1] - batch_pos[x] > batch_pos[y + 1] - batch_pos[y];
        });
        order.insert(order.end(), small.begin(), small.end());
    }
    GP_HIP(hipMemcpyAsync(job->d_batch_order.p, order.data(), order.size() * sizeof(uint32_t),
                          hipMemcpyHostToDevice, s));
    // Arena slots of the grid-wide groups: a group is a run of consecutive entries of `order`,
    // batch i of the group holds slots [i * n_k, (i + 1) * n_k).  The grouping below depends on
    // nothing but these sizes, so the whole table is known now.
    // (default: as many batches as keep the group's counting filters inside the 256 MB Infinity Cache --
    // 4 batches x 4 k x 10 MiB; SURVEY 8d's 48 long contigs in one group of 192 filters = 2 GiB took 6.92 s
    // per step for their filters, in groups of 4: 5.97 s, 8: 6.22 s, 16: 6.66 s, same bits)
    const uint64_t group_fit = std::max<uint64_t>(1, (200ull << 20) / ((uint64_t)p.n_k * a.cbf_stride));
    const uint32_t group_cap = (uint32_t)std::max<uint64_t>(
        1, std::min<uint64_t>({ std::max<uint64_t>(1, env_u64("GP_STAGE_A_GROUP", group_fit)),
                                std::max<uint64_t>(1, job->window / kTile),
                                std::max<uint64_t>(1, job->cbf_slots / p.n_k) }));
    {
        std::vector<uint32_t> slot(job->n_batches, 0);
        for (uint32_t i = 0; i < n_big; i++)
            slot[order[i]] = (i % group_cap) * p.n_k;
        GP_HIP(hipMemcpyAsync(job->d_batch_slot.p, slot.data(), slot.size() * sizeof(uint32_t),
                              hipMemcpyHostToDevice, s));
        GP_HIP(hipStreamSynchronize(s)); // batch_pos, order and slot are locals
    }
    a.batch_order = job->d_batch_order.p;
    const uint32_t n_local = job->n_batches - n_big - n_lvl, first_local = n_big + n_lvl;
    // census tables: host copies stay alive until the stream has been synchronised at the end
    struct Cohort {
        std::vector<uint32_t> wg_pair, pair_batch, pair_ki, pair_nst, pair_queue, tab32;
        std::vector<uint32_t> ckc_nb, ckc_capb, ckc_capg, ckc_need; // candidates by k-mer (ckc_*): per pair
        uint32_t ckc_max_nb = 0;
        std::vector<uint64_t> wg_pos0, pair_rec_off, pair_cnt_off, pair_bit_off, tab64;
    };
    std::vector<Cohort> cohorts;
    std::vector<uint32_t> h_pair_mode;
    std::vector<uint64_t> h_pair_bit_off;
    if (n_local) {
        uint64_t units = 0;
        for (uint32_t i = first_local; i < job->n_batches; i++)
            units += (batch_pos[order[i] + 1] - batch_pos[order[i]]) * p.n_k;
        int rc0 = GP_OK;
        // as many workgroups as the chip holds at once (each owns one arena slot), never more than
        // there are slots or pairs
        int per_cu = 0;
        GP_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, cbf_local, kLBlock, 0));
        // Two per CU by default: the kernel is bound by random memory transactions and runs at the
        // same speed with two, three or four (measured), and two leave LDS and wave slots for the
        // ntEdit / Sealer kernels of another job (gp_jobs_polish) to run beside it.
        per_cu = std::max(1, std::min<int>(per_cu, (int)env_u64("GP_STAGE_A_LOCAL_PER_CU", 2)));
        const uint64_t n_pairs = (uint64_t)n_local * p.n_k;
        uint32_t grid = (uint32_t)std::min<uint64_t>(
            { n_pairs, (uint64_t)per_cu * (uint64_t)job->ctx->prop.multiProcessorCount, (uint64_t)job->cbf_slots });
        {
            // Pairs of similar length finish in waves of `grid`: a last wave of a few workgroups
            // would leave the chip idle for a whole pair's time.  Cut the grid so that the waves
            // are equally full (a workgroup alone runs faster, so fewer at a time cost nothing).
            const uint64_t waves = (n_pairs + grid - 1) / grid;
            grid = (uint32_t)((n_pairs + waves - 1) / waves);
        }
        // ---- census of first touches for the pairs it pays for (see census_scatter)
        a.pair_mode = nullptr;
        a.pair_bit_off = nullptr;
        a.cand = nullptr;
        a.cand2 = nullptr;
        // counter ranges: a power of two wide, at most kCBuckets of them
        const bool wide = env_u64("GP_STAGE_A_CENSUS_WIDE", 1) != 0; // 32-byte groups of records (see kCBucketsWide)
        const uint64_t scatter_span = std::max<uint64_t>(1, env_u64("GP_STAGE_A_SCATTER_SPAN", 4)); // stretches per census_scatter workgroup
        const uint64_t scatter_rounds = env_u64("GP_STAGE_A_SCATTER_ROUNDS", 1); // rounds of four positions per lane and staged tile in census_scatter
        uint32_t bucket_shift = 2;
        while (((p.cbf_bytes + (1ull << bucket_shift) - 1) >> bucket_shift) > (wide ? kCBucketsWide : kCBuckets))
            bucket_shift++;
        const uint32_t n_buckets = (uint32_t)((p.cbf_bytes + (1ull << bucket_shift) - 1) >> bucket_shift);
        if (env_u64("GP_STAGE_A_CENSUS", 1) && p.hash_num <= 4 && bucket_shift <= kCMaxShift) {
            const uint64_t min_pos = env_u64("GP_STAGE_A_CENSUS_MIN", 65536);           // below: too few touches per bucket
            // Two buffer sets: census_scatter of one cohort (VALU and write bandwidth) runs beside
            // census_mark / census_pack of the one before it (latency), on two streams.  (Worth 4 % of
            // the census: both kernels are held by LDS -- 2 x 56 KB and 4 x 33 KB per CU -- so beside a
            // full scatter the mark gets one workgroup per CU.)
            // (OVERLAP=3: three sets and census_pack on a stream of its own, so that no cohort waits for the
            // pack of the one before -- in the trace the chip idles for it, a tenth of a cohort's time;
            // measured: 91.7 against 91.8 ms per 1000 contigs, the pack beside the next mark costs what it
            // saves.  =0: one set, one stream)
            const uint64_t overlap = env_u64("GP_STAGE_A_CENSUS_OVERLAP", 1);
            // candidates by k-mer (GP_STAGE_A_CKC=0: every candidate is replayed, as before round 4)
            const bool ckc_on = env_u64("GP_STAGE_A_CKC", 1) != 0 && p.hash_num <= 4 && p.cbf_bytes <= (1ull << 30); // (ckc_share's table holds 24 bits of a counter's place in its group)
            // With the candidates-by-k-mer kernels behind every cohort's pack a set of buffers is busy for four
            // kernels in a row (scatter, mark, pack, ckc_*): three sets and the ckc_* kernels on the third
            // stream, so that a cohort's grouping runs beside the next one's mark and the scatter of the one
            // after (GP_STAGE_A_CKC_STREAM=0: two sets, the ckc_* kernels behind the pack on its stream)
            const bool ckc_stream = ckc_on && overlap == 1 && env_u64("GP_STAGE_A_CKC_STREAM", 1) != 0;
            const uint32_t n_sets = overlap == 0 ? 1u : (overlap == 3 || ckc_stream) ? 3u : 2u;
            const uint64_t scatter_per_cu = env_u64("GP_STAGE_A_SCATTER_PER_CU", n_sets > 1 ? 2 : 0); // census_scatter workgroups per CU (0: no cap on its grid)
            const uint64_t rec_budget = (env_u64("GP_STAGE_A_CENSUS_MB", 16384) << 20) / 4 / n_sets; // (4-byte) record slots of a cohort
            h_pair_mode.assign(n_pairs, 0);
            h_pair_bit_off.assign(n_pairs, 0);
            std::vector<uint8_t> batch_uniform_thr(job->n_batches, 0);
            std::vector<uint32_t> batch_thr_seg(job->n_batches, 0); // a contig of the batch that has reads
            for (uint32_t bi = 0; bi < job->n_batches; bi++) {
                bool first_seen = false, uniform = true;
                for (uint32_t cc = job->h_batch_contig_begin[bi]; cc < job->h_batch_contig_begin[bi + 1]; cc++) {
                    if (job->h_seg_start[cc + 1] == job->h_seg_start[cc])
                        continue;
                    if (!first_seen) {
                        first_seen = true;
                        batch_thr_seg[bi] = cc;
                    } else if (job->h_contig_thr[cc] != job->h_contig_thr[batch_thr_seg[bi]]) {
                        uniform = false;
                    }
                }
                batch_uniform_thr[bi] = first_seen && uniform;
            }
            uint32_t ckc_gshift = 2;
            while (((p.cbf_bytes + (1ull << ckc_gshift) - 1) >> ckc_gshift) > 128)
                ckc_gshift++;
            const uint32_t ckc_groups = (uint32_t)((p.cbf_bytes + (1ull << ckc_gshift) - 1) >> ckc_gshift);
            uint64_t bits = 0, max_rec = 0, max_cnt = 0, max_pairs = 0, max_wg = 0;
            Cohort cur;
            uint64_t cur_rec = 0, cur_cnt = 0;
            auto close = [&]() {
                if (cur.pair_batch.empty())
                    return;
                max_rec = std::max(max_rec, cur_rec);
                max_cnt = std::max(max_cnt, cur_cnt);
                max_pairs = std::max<uint64_t>(max_pairs, cur.pair_batch.size());
                max_wg = std::max<uint64_t>(max_wg, cur.wg_pair.size());
                cohorts.push_back(std::move(cur));
                cur = Cohort();
                cur_rec = cur_cnt = 0;
            };
            for (uint32_t i = 0; i < n_local; i++) {
                const uint32_t b = order[first_local + i];
                const uint64_t pos = batch_pos[b + 1] - batch_pos[b];
                // every operation in a first touch when the filter is nearly empty, hardly any once
                // it has seen a few times its size: the census pays up to about two touches per counter
                if (pos < min_pos || pos >= (1ull << 32) || pos * p.hash_num > 2 * p.cbf_bytes)
                    continue;
                const uint32_t nst = (uint32_t)((pos + kCStretch - 1) / kCStretch);
                const uint64_t recs = census_runs(nst, n_buckets) * (wide ? kCRunWide : kCRun); // a run of slots per bucket and stretch
                if (recs * p.n_k > rec_budget)
                    continue; // a batch that would not fit a cohort by itself
                if (cur_rec + recs * p.n_k > rec_budget)
                    close();
                for (uint32_t ki = 0; ki < p.n_k; ki++) {
                    const uint32_t q = i * p.n_k + ki, lp = (uint32_t)cur.pair_batch.size();
                    cur.pair_batch.push_back(b);
                    cur.pair_ki.push_back(ki);
                    cur.pair_nst.push_back(nst);
                    cur.pair_queue.push_back(q);
                    cur.pair_rec_off.push_back(cur_rec);
                    cur.pair_cnt_off.push_back(cur_cnt);
                    cur.pair_bit_off.push_back(bits);
                    {
                        // candidates by k-mer (ckc_scatter ...): for pairs with ONE threshold (the closed form
                        // needs it; a batch of one contig always has), of up to 2 M positions, whose record
                        // space -- free again once census_pack has read it -- holds the buckets
                        uint32_t nb = 0, capb = 0, capg = 0;
                        if (ckc_on && batch_uniform_thr[b] && pos <= (1ull << 21)) {
                            nb = 16;
                            while (nb < kCkcMaxBuckets && (uint64_t)nb * 16384 < pos)
                                nb *= 2;
                            capb = (uint32_t)((double)pos * 0.75 * 1.3 / nb) + 64;
                            capg = (uint32_t)((double)pos * 0.08 * p.hash_num * 1.6 / ckc_groups) + 64;
                            const uint64_t need_words = kCkcHdrWords + (uint64_t)nb * (kCkcSlots / 32) + (uint64_t)nb * capb * 4 +
                                                        (uint64_t)ckc_groups * capg * 2;
                            if (need_words > recs)
                                nb = 0;
                        }
                        cur.ckc_nb.push_back(nb);
                        cur.ckc_capb.push_back(capb);
                        cur.ckc_capg.push_back(capg);
                        cur.ckc_need.push_back(job->h_contig_thr[batch_thr_seg[b]] - 2 + ki - 1);
                        cur.ckc_max_nb = std::max(cur.ckc_max_nb, nb);
                    }
                    h_pair_mode[q] = 1;
                    h_pair_bit_off[q] = bits;
                    cur_rec += recs;
                    cur_cnt += (uint64_t)n_buckets * nst;
                    bits += (pos + kLWin - 1) / kLWin * kLWin; // whole windows of cbf_local
                    for (uint64_t p0 = 0; p0 < pos; p0 += kCStretch) {
                        cur.wg_pair.push_back(lp);
                        cur.wg_pos0.push_back(batch_pos[b] + p0);
                    }
                }
            }
            close();
            if (!cohorts.empty()) {
                int rcc;
                max_rec = (max_rec + 63) / 64 * 64; // (a set's records start on a line)
                const uint64_t tab32_n = (max_wg + 8 * max_pairs + 3) / 4 * 4, tab64_n = max_wg + 3 * max_pairs;
                if ((rcc = job->d_census_rec.ensure(n_sets * max_rec)) != GP_OK ||
                    (rcc = job->d_census_cnt.ensure(n_sets * max_cnt)) != GP_OK ||
                    (rcc = job->d_census_over.ensure(n_sets * max_pairs)) != GP_OK ||
                    (rcc = job->d_census_tab32.ensure(n_sets * tab32_n)) != GP_OK ||
                    (rcc = job->d_census_tab64.ensure(n_sets * tab64_n)) != GP_OK ||
                    (rcc = job->d_cand.ensure(bits / 64)) != GP_OK || (rcc = job->d_cand2.ensure(ckc_on ? bits / 64 : 1)) != GP_OK ||
                    (rcc = job->d_pair_mode.ensure(n_pairs)) != GP_OK ||
                    (rcc = job->d_pair_bit_off.ensure(n_pairs)) != GP_OK)
                    return rcc;
                GP_HIP(hipMemcpyAsync(job->d_pair_mode.p, h_pair_mode.data(), n_pairs * sizeof(uint32_t), hipMemcpyHostToDevice, s));
                GP_HIP(hipMemcpyAsync(job->d_pair_bit_off.p, h_pair_bit_off.data(), n_pairs * sizeof(uint64_t),
                                      hipMemcpyHostToDevice, s));
                if (ckc_on) // (on `s`, ahead of the first scatter: every later stream of the census is behind it)
                    GP_HIP(hipMemsetAsync(job->d_cand2.p, 0, bits / 8, s));
                if ((rcc = job->t_census.begin(s, units)) != GP_OK)
                    return rcc;
                // census_scatter: a ring of eight record slots per range in LDS (40 KB at 1280 ranges:
                // two workgroups per CU, which is also what it likes best -- with eight per CU and
                // records stored one by one it had 670 MB of half-written runs in flight)
                uint32_t k_max = 1;
                for (uint32_t ki = 0; ki < p.n_k; ki++)
                    k_max = std::max(k_max, p.k[ki]);
                const size_t scatter_lds =
                    (((size_t)n_buckets * ((wide ? kCRingWide : 8u) + 1) + 2) & ~(size_t)1) * sizeof(uint32_t) + (size_t)k_max * 8 * sizeof(uint64_t);
                if (wide) // a 64 KB table and the run lengths: more than a workgroup gets without asking (per device)
                    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&census_mark<kCRunWide, kMarkBlockWide>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
                hipStream_t sm = n_sets > 1 ? job->a_stream2 : s;  // mark
                hipStream_t sp = (n_sets > 2 && !ckc_stream) ? job->a_stream3 : sm; // pack
                hipStream_t sc = ckc_stream ? job->a_stream3 : sp;                   // candidates by k-mer
                for (size_t ci = 0; ci < cohorts.size(); ci++) {
                    Cohort& co = cohorts[ci];
                    const uint32_t set = (uint32_t)(ci % n_sets);
                    hipEvent_t ev_scattered = job->a_ev_census[set], ev_marked = job->a_ev_census[3 + set],
                               ev_packed = job->a_ev_census[6 + set];
                    if (n_sets > 1 && ci >= n_sets)
                        GP_HIP(hipStreamWaitEvent(s, ev_packed, 0)); // the set's previous cohort has been packed
                    uint32_t* const d_tab32 = job->d_census_tab32.p + (uint64_t)set * tab32_n;
                    uint64_t* const d_tab64 = job->d_census_tab64.p + (uint64_t)set * tab64_n;
                    uint32_t* const d_over = job->d_census_over.p + (uint64_t)set * max_pairs;
                    const size_t np = co.pair_batch.size(), nw = co.wg_pair.size();
                    // one table of 32-bit and one of 64-bit columns per cohort
                    co.tab32.reserve(nw + 8 * np);
                    co.tab32.insert(co.tab32.end(), co.wg_pair.begin(), co.wg_pair.end());
                    co.tab32.insert(co.tab32.end(), co.pair_batch.begin(), co.pair_batch.end());
                    co.tab32.insert(co.tab32.end(), co.pair_ki.begin(), co.pair_ki.end());
                    co.tab32.insert(co.tab32.end(), co.pair_nst.begin(), co.pair_nst.end());
                    co.tab32.insert(co.tab32.end(), co.pair_queue.begin(), co.pair_queue.end());
                    co.tab32.insert(co.tab32.end(), co.ckc_nb.begin(), co.ckc_nb.end());
                    co.tab32.insert(co.tab32.end(), co.ckc_capb.begin(), co.ckc_capb.end());
                    co.tab32.insert(co.tab32.end(), co.ckc_capg.begin(), co.ckc_capg.end());
                    co.tab32.insert(co.tab32.end(), co.ckc_need.begin(), co.ckc_need.end());
                    co.tab64.reserve(nw + 3 * np);
                    co.tab64.insert(co.tab64.end(), co.wg_pos0.begin(), co.wg_pos0.end());
                    co.tab64.insert(co.tab64.end(), co.pair_rec_off.begin(), co.pair_rec_off.end());
                    co.tab64.insert(co.tab64.end(), co.pair_cnt_off.begin(), co.pair_cnt_off.end());
                    co.tab64.insert(co.tab64.end(), co.pair_bit_off.begin(), co.pair_bit_off.end());
                    GP_HIP(hipMemcpyAsync(d_tab32, co.tab32.data(), co.tab32.size() * sizeof(uint32_t),
                                          hipMemcpyHostToDevice, s));
                    GP_HIP(hipMemcpyAsync(d_tab64, co.tab64.data(), co.tab64.size() * sizeof(uint64_t),
                                          hipMemcpyHostToDevice, s));
                    GP_HIP(hipMemsetAsync(d_over, 0, np * sizeof(uint32_t), s));
                    CensusArgs c{};
                    c.wg_pair = d_tab32;
                    c.pair_batch = c.wg_pair + nw;
                    c.pair_ki = c.pair_batch + np;
                    c.pair_nst = c.pair_ki + np;
                    c.pair_queue = c.pair_nst + np;
                    c.wg_pos0 = d_tab64;
                    c.pair_rec_off = c.wg_pos0 + nw;
                    c.pair_cnt_off = c.pair_rec_off + np;
                    c.pair_bit_off = c.pair_cnt_off + np;
                    c.n_wg = (uint32_t)nw;
                    c.rec = job->d_census_rec.p + (uint64_t)set * max_rec;
                    c.cnt = job->d_census_cnt.p + (uint64_t)set * max_cnt;
                    c.overflow = d_over;
                    c.pair_mode = job->d_pair_mode.p;
                    c.cand = job->d_cand.p;
                    c.bucket_shift = bucket_shift;
                    c.n_buckets = n_buckets;
                    c.seed_rows = k_max;
                    c.mark_prefetch = (uint32_t)std::min<uint64_t>(64, env_u64("GP_STAGE_A_MARK_PREFETCH", 64));
                    {
                        const bool mod5 = a.cbf_mod.fast && a.cbf_mod.c == 5 && a.cbf_mod.s < 32 && a.hash_num == 4;
                        // (beside census_mark -- two buffer sets -- the grid is two workgroups per CU, each
                        // taking its share of the stretches: with the mark's workgroup that is what a CU holds)
                        uint64_t n_g = (nw + scatter_span - 1) / scatter_span;
                        if (scatter_per_cu)
                            n_g = std::min<uint64_t>(n_g, scatter_per_cu * (uint64_t)job->ctx->prop.multiProcessorCount);
                        const dim3 g((unsigned)n_g), bl(kBlock);
                        if (wide && scatter_rounds == 4 && mod5)
                            hipLaunchKernelGGL((census_scatter<kCRunWide, kCRingWide, 4, true>), g, bl, scatter_lds, s, a, c);
                        else if (wide && scatter_rounds == 4)
                            hipLaunchKernelGGL((census_scatter<kCRunWide, kCRingWide, 4, false>), g, bl, scatter_lds, s, a, c);
                        else if (wide && mod5)
                            hipLaunchKernelGGL((census_scatter<kCRunWide, kCRingWide, 1, true>), g, bl, scatter_lds, s, a, c);
                        else if (wide)
                            hipLaunchKernelGGL((census_scatter<kCRunWide, kCRingWide, 1, false>), g, bl, scatter_lds, s, a, c);
                        else
                            hipLaunchKernelGGL((census_scatter<kCRun, 8u, 1, false>), g, bl, scatter_lds, s, a, c);
                    }
                    if (n_sets > 1) {
                        GP_HIP(hipEventRecord(ev_scattered, s));
                        GP_HIP(hipStreamWaitEvent(sm, ev_scattered, 0));
                    }
                    {
                        uint32_t max_nst = 0;
                        for (uint32_t x : co.pair_nst)
                            max_nst = std::max(max_nst, x);
                        const size_t lds = ((size_t)sizeof(uint32_t) << bucket_shift) + (max_nst + 3) / 4 * 4;
                        if (wide)
                            hipLaunchKernelGGL((census_mark<kCRunWide, kMarkBlockWide>), dim3((unsigned)(np * n_buckets)),
                                               dim3(kMarkBlockWide), lds, sm, c);
                        else
                            hipLaunchKernelGGL((census_mark<kCRun, kMarkBlock>), dim3((unsigned)(np * n_buckets)), dim3(kMarkBlock), lds, sm, c);
                    }
                    if (sp != sm) {
                        GP_HIP(hipEventRecord(ev_marked, sm));
                        GP_HIP(hipStreamWaitEvent(sp, ev_marked, 0));
                    }
                    if (wide)
                        hipLaunchKernelGGL((census_pack<kCRunWide>), dim3((unsigned)nw), dim3(256), 0, sp, a, c);
                    else
                        hipLaunchKernelGGL((census_pack<kCRun>), dim3((unsigned)nw), dim3(256), 0, sp, a, c);
                    GP_HIP(hipGetLastError());
                    if (ckc_on && co.ckc_max_nb) {
                        // candidates by k-mer, behind the pack in the record space it has just read
                        CkcArgs g{};
                        g.pair_nb = c.pair_queue + np;
                        g.pair_capb = g.pair_nb + np;
                        g.pair_capg = g.pair_capb + np;
                        g.pair_need = g.pair_capg + np;
                        g.gshift = ckc_gshift;
                        g.n_groups = ckc_groups;
                        g.cand2 = job->d_cand2.p;
                        const uint32_t span = co.ckc_max_nb;
                        if (sc != sp) { // (ev_marked is free in this arrangement: mark and pack share a stream)
                            GP_HIP(hipEventRecord(ev_marked, sp));
                            GP_HIP(hipStreamWaitEvent(sc, ev_marked, 0));
                        }
                        if ((rcc = job->t_ckc.begin(sc, co.pair_batch.size())) != GP_OK)
                            return rcc;
                        hipLaunchKernelGGL(ckc_init, dim3((unsigned)np), dim3(256), 0, sc, c, g, (uint32_t)np);
                        hipLaunchKernelGGL(ckc_scatter, dim3((unsigned)nw), dim3(256), 0, sc, a, c, g);
                        hipLaunchKernelGGL(ckc_group, dim3((unsigned)(np * span)), dim3(256), 0, sc, a, c, g, span);
                        hipLaunchKernelGGL(ckc_share, dim3((unsigned)std::min<uint64_t>(np * ckc_groups, 2ull * job->ctx->prop.multiProcessorCount)), dim3(256), 0, sc, c, g, (uint32_t)np);
                        hipLaunchKernelGGL(ckc_apply, dim3((unsigned)(np * span)), dim3(256), 0, sc, a, c, g, span);
                        GP_HIP(hipGetLastError());
                        if ((rcc = job->t_ckc.end(sc)) != GP_OK)
                            return rcc;
                        if (n_sets > 1)
                            GP_HIP(hipEventRecord(ev_packed, sc)); // the set is free when its grouping is done
                    } else if (n_sets > 1) {
                        GP_HIP(hipEventRecord(ev_packed, sp));
                    }
                }
                if (n_sets > 1) // everything has been packed before the replay starts
                    for (uint32_t set = 0; set < n_sets && set < cohorts.size(); set++)
                        GP_HIP(hipStreamWaitEvent(s, job->a_ev_census[6 + set], 0));
                if ((rcc = job->t_census.end(s)) != GP_OK)
                    return rcc;
                a.pair_mode = job->d_pair_mode.p;
                a.pair_bit_off = job->d_pair_bit_off.p;
                a.cand = job->d_cand.p;
                a.cand2 = job->d_cand2.p;
            }
        }
        a.kset = nullptr;
        a.kset_mask = 0;
        if (a.pair_mode) {
            // the table of settled k-mers of a census pair: 2^18 entries (2 MiB) per workgroup (44 % of the
            // candidates of configs[2] are found there with 2^16 entries, 50 % with 2^20)
            const uint64_t lg = std::min<uint64_t>(24, env_u64("GP_STAGE_A_KSET_LOG2", 18));
            if (lg >= 4) {
                if ((rc0 = job->d_kset.ensure((uint64_t)grid << lg)) != GP_OK)
                    return rc0;
                a.kset = job->d_kset.p;
                a.kset_mask = (uint32_t)((1ull << lg) - 1);
            }
        }
        if ((rc0 = job->t_local.begin(s, units)) != GP_OK)
            return rc0;
        hipLaunchKernelGGL(cbf_local, dim3(grid), dim3(kLBlock), 0, s, a, first_local, (uint32_t)n_pairs);
        GP_HIP(hipGetLastError());
        if ((rc0 = job->t_local.end(s)) != GP_OK)
            return rc0;
    }

    // ---- The census LEVEL BY LEVEL: long batches whose filters stay unsaturated.
    // Let m(o) be the minimum an operation finds among its counters.  A counter is >= j + 1 exactly from
    // its first touch by an operation with m >= j on (that operation lifts it there, or found it there), so
    //     m(o) >= j + 1  <=>  every counter of o has an EARLIER touch by an operation with m >= j:
    // the set L(j + 1) = { m >= j + 1 } is "the candidates" of a census over the operations of L(j) alone --
    // L(0) is everybody, L(1) the census's candidates as the local schedule knows them -- and an operation
    // puts its k-mer into the filter iff m >= thr - 1, i.e. iff it is in L(thr - 1).  thr - 1 passes of
    // census_scatter (once) / census_mark_seg / census_pack, the last one's set into the filter
    // (level_insert): streaming kernels, no counting filter at all, no order to keep.  The sets shrink: in
    // the CPU model of this schedule (tests/test_levels_model.py, which also checks the claim against the
    // serial filter) the passes' sizes sum to 5.4 times the operations at 15 touches per counter and a
    // threshold of 13, to 7.9 times at 30 touches, against 5.4 reservation atomics plus two rounds of random
    // counter traffic per ordered operation in the grid-wide schedule.  A pass in which
    // nobody is left ends a cohort early.  Exact for one threshold per filter (for j + 1 <= thr an operation
    // that does not update found its counters at thr already).
    std::vector<uint32_t> lvl_fallback;
    if (n_lvl) {
        GP_HIP(hipStreamSynchronize(s)); // (the buffers below may be the local schedule's, about to be resized)
        const uint32_t RUNW = lv_wide ? kCRunWide : kCRun;
        struct LCohort {
            std::vector<uint32_t> wg_pair, pair_batch, pair_ki, pair_nst, pair_queue, pair_need, tab32;
            std::vector<uint64_t> wg_pos0, pair_rec_off, pair_cnt_off, pair_bit_off, tab64;
            uint64_t recs = 0, cnts = 0, bits = 0;
            uint32_t max_need = 0, max_nst = 0;
        };
        std::vector<LCohort> lco;
        {
            LCohort cur;
            for (uint32_t i = 0; i < n_lvl; i++) {
                const uint32_t b = lvl_batches[i];
                const uint64_t pos = batch_pos[b + 1] - batch_pos[b];
                const uint32_t nst = (uint32_t)((pos + kCStretch - 1) / kCStretch);
                const uint64_t recs = census_runs(nst, lv_buckets) * RUNW;
                for (uint32_t ki = 0; ki < p.n_k; ki++) {
                    if (!cur.pair_batch.empty() && cur.recs + recs > lv_budget) {
                        lco.push_back(std::move(cur));
                        cur = LCohort();
                    }
                    const uint32_t lp = (uint32_t)cur.pair_batch.size();
                    cur.pair_batch.push_back(b);
                    cur.pair_ki.push_back(ki);
                    cur.pair_nst.push_back(nst);
                    cur.pair_queue.push_back(lp);
                    cur.pair_need.push_back(lvl_thr[i] - 2 + ki - 1); // thr - 1 passes (thr >= 2: at least one)
                    cur.pair_rec_off.push_back(cur.recs);
                    cur.pair_cnt_off.push_back(cur.cnts);
                    cur.pair_bit_off.push_back(cur.bits);
                    cur.max_need = std::max(cur.max_need, cur.pair_need.back());
                    cur.max_nst = std::max(cur.max_nst, nst);
                    cur.recs += recs;
                    cur.cnts += (uint64_t)lv_buckets * nst;
                    cur.bits += (uint64_t)nst * kCStretch; // whole stretches: a pass reads a stretch's 128 words
                    for (uint64_t p0 = 0; p0 < pos; p0 += kCStretch) {
                        cur.wg_pair.push_back(lp);
                        cur.wg_pos0.push_back(batch_pos[b] + p0);
                    }
                }
            }
            if (!cur.pair_batch.empty())
                lco.push_back(std::move(cur));
        }
        uint64_t m_rec = 0, m_cnt = 0, m_bits = 0, m_np = 0, m_wg = 0;
        for (const LCohort& co : lco) {
            m_rec = std::max(m_rec, co.recs);
            m_cnt = std::max(m_cnt, co.cnts);
            m_bits = std::max(m_bits, co.bits);
            m_np = std::max<uint64_t>(m_np, co.pair_batch.size());
            m_wg = std::max<uint64_t>(m_wg, co.wg_pair.size());
        }
        int rcl;
        DevBuf<unsigned long long> d_bits[2], d_count;
        DevBuf<uint32_t> d_mode;
        const uint64_t t32 = (m_wg + 5 * m_np + 3) / 4 * 4, t64 = m_wg + 3 * m_np;
        if ((rcl = job->d_census_rec.ensure(m_rec + 64)) != GP_OK || (rcl = job->d_census_cnt.ensure(m_cnt + 64)) != GP_OK ||
            (rcl = job->d_census_over.ensure(m_np)) != GP_OK || (rcl = job->d_census_tab32.ensure(t32)) != GP_OK ||
            (rcl = job->d_census_tab64.ensure(t64)) != GP_OK || (rcl = d_bits[0].alloc(m_bits / 64 + 64)) != GP_OK ||
            (rcl = d_bits[1].alloc(m_bits / 64 + 64)) != GP_OK || (rcl = d_count.alloc(16)) != GP_OK ||
            (rcl = d_mode.alloc(m_np)) != GP_OK)
            return rcl;
        uint64_t lv_units = 0;
        for (uint32_t b : lvl_batches)
            lv_units += (batch_pos[b + 1] - batch_pos[b]) * p.n_k;
        if ((rcl = job->t_census.begin(s, lv_units)) != GP_OK)
            return rcl;
        uint32_t k_max = 1;
        for (uint32_t ki = 0; ki < p.n_k; ki++)
            k_max = std::max(k_max, p.k[ki]);
        const size_t scatter_lds =
            (((size_t)lv_buckets * ((lv_wide ? kCRingWide : 8u) + 1) + 2) & ~(size_t)1) * sizeof(uint32_t) + (size_t)k_max * 8 * sizeof(uint64_t);
        if (lv_wide)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&census_mark_seg<kCRunWide, kMarkBlockWide>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
        const uint64_t span = std::max<uint64_t>(1, env_u64("GP_STAGE_A_SCATTER_SPAN", 4));
        const bool mod5 = a.cbf_mod.fast && a.cbf_mod.c == 5 && a.cbf_mod.s < 32 && a.hash_num == 4;
        unsigned long long* const h_count = job->h_readback + 16; // (pinned; words 0..8 are stage A's statistics)
        uint64_t passes = 0;
        for (LCohort& co : lco) {
            const size_t np = co.pair_batch.size(), nw = co.wg_pair.size();
            co.tab32.reserve(nw + 5 * np);
            co.tab32.insert(co.tab32.end(), co.wg_pair.begin(), co.wg_pair.end());
            co.tab32.insert(co.tab32.end(), co.pair_batch.begin(), co.pair_batch.end());
            co.tab32.insert(co.tab32.end(), co.pair_ki.begin(), co.pair_ki.end());
            co.tab32.insert(co.tab32.end(), co.pair_nst.begin(), co.pair_nst.end());
            co.tab32.insert(co.tab32.end(), co.pair_queue.begin(), co.pair_queue.end());
            co.tab32.insert(co.tab32.end(), co.pair_need.begin(), co.pair_need.end());
            co.tab64.reserve(nw + 3 * np);
            co.tab64.insert(co.tab64.end(), co.wg_pos0.begin(), co.wg_pos0.end());
            co.tab64.insert(co.tab64.end(), co.pair_rec_off.begin(), co.pair_rec_off.end());
            co.tab64.insert(co.tab64.end(), co.pair_cnt_off.begin(), co.pair_cnt_off.end());
            co.tab64.insert(co.tab64.end(), co.pair_bit_off.begin(), co.pair_bit_off.end());
            GP_HIP(hipMemcpyAsync(job->d_census_tab32.p, co.tab32.data(), co.tab32.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
            GP_HIP(hipMemcpyAsync(job->d_census_tab64.p, co.tab64.data(), co.tab64.size() * sizeof(uint64_t), hipMemcpyHostToDevice, s));
            GP_HIP(hipMemsetAsync(job->d_census_over.p, 0, np * sizeof(uint32_t), s));
            CensusArgs c{};
            c.wg_pair = job->d_census_tab32.p;
            c.pair_batch = c.wg_pair + nw;
            c.pair_ki = c.pair_batch + np;
            c.pair_nst = c.pair_ki + np;
            c.pair_queue = c.pair_nst + np;
            c.pair_need = c.pair_queue + np;
            c.wg_pos0 = job->d_census_tab64.p;
            c.pair_rec_off = c.wg_pos0 + nw;
            c.pair_cnt_off = c.pair_rec_off + np;
            c.pair_bit_off = c.pair_cnt_off + np;
            c.n_wg = (uint32_t)nw;
            c.rec = job->d_census_rec.p;
            c.cnt = job->d_census_cnt.p;
            c.overflow = job->d_census_over.p;
            c.pair_mode = d_mode.p;
            c.bucket_shift = lv_shift;
            c.n_buckets = lv_buckets;
            c.seed_rows = k_max;
            c.level_count = d_count.p;
            for (uint32_t level = 0; level < co.max_need; level++) {
                c.level = level;
                c.members = level ? d_bits[(level - 1) & 1].p : nullptr;
                c.cand = d_bits[level & 1].p;
                GP_HIP(hipMemsetAsync(d_count.p, 0, sizeof(unsigned long long), s));
                if (level == 0) { // (later passes take the marked records of the pass before: census_mark)
                    uint64_t n_g = (nw + span - 1) / span;
                    n_g = std::min<uint64_t>(n_g, 3ull * (uint64_t)job->ctx->prop.multiProcessorCount);
                    const dim3 g((unsigned)n_g), bl(kBlock);
                    if (lv_wide && mod5)
                        hipLaunchKernelGGL((census_scatter<kCRunWide, kCRingWide, 1, true>), g, bl, scatter_lds, s, a, c);
                    else if (lv_wide)
                        hipLaunchKernelGGL((census_scatter<kCRunWide, kCRingWide, 1, false>), g, bl, scatter_lds, s, a, c);
                    else
                        hipLaunchKernelGGL((census_scatter<kCRun, 8u, 1, false>), g, bl, scatter_lds, s, a, c);
                }
                {
                    const size_t lds = ((size_t)sizeof(uint32_t) << lv_shift) + (co.max_nst + 3) / 4 * 4;
                    if (lv_wide)
                        hipLaunchKernelGGL((census_mark_seg<kCRunWide, kMarkBlockWide>), dim3((unsigned)(np * lv_buckets)),
                                           dim3(kMarkBlockWide), lds, s, c);
                    else
                        hipLaunchKernelGGL((census_mark_seg<kCRun, kMarkBlock>), dim3((unsigned)(np * lv_buckets)), dim3(kMarkBlock), lds, s, c);
                }
                if (lv_wide)
                    hipLaunchKernelGGL((census_pack<kCRunWide>), dim3((unsigned)nw), dim3(256), 0, s, a, c);
                else
                    hipLaunchKernelGGL((census_pack<kCRun>), dim3((unsigned)nw), dim3(256), 0, s, a, c);
                bool any_final = false;
                for (uint32_t need : co.pair_need)
                    any_final |= need == level + 1;
                if (any_final)
                    hipLaunchKernelGGL(level_insert, dim3((unsigned)nw), dim3(256), 0, s, a, c, (const unsigned long long*)d_bits[level & 1].p);
                GP_HIP(hipGetLastError());
                passes++;
                GP_HIP(hipMemcpyAsync(h_count, d_count.p, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
                GP_HIP(hipStreamSynchronize(s));
                if (*h_count == 0)
                    break; // nobody is left in any pair's set: the later sets are empty too
            }
            std::vector<uint32_t> h_over(np);
            GP_HIP(hipMemcpyAsync(h_over.data(), job->d_census_over.p, np * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            GP_HIP(hipStreamSynchronize(s));
            for (size_t q = 0; q < np; q++)
                if (h_over[q] && (lvl_fallback.empty() || lvl_fallback.back() != co.pair_batch[q]))
                    lvl_fallback.push_back(co.pair_batch[q]);
        }
        job->counters.a_rounds += passes;
        if ((rcl = job->t_census.end(s)) != GP_OK)
            return rcl;
        std::sort(lvl_fallback.begin(), lvl_fallback.end());
        lvl_fallback.erase(std::unique(lvl_fallback.begin(), lvl_fallback.end()), lvl_fallback.end());
    }

    // Slice = the read positions of one batch a step's first pass hashes.  What makes the
    // rounds expensive is many waiting operations per counter, so a slice takes at most
    // `lim` positions of any one batch (hash_num * lim touches over cbf_bytes counters) but
    // as many whole batches as fit.  Once a filter is saturated nearly everything settles
    // in the first pass; the slice then grows (feedback on the waiting fraction).
    const uint64_t load_div = std::max<uint64_t>(1, env_u64("GP_STAGE_A_LOAD_DIV", 6));
    const uint64_t lim0 = std::max<uint64_t>(kTile, p.cbf_bytes / (load_div * p.hash_num));
    const double wait_cap = (double)env_u64("GP_STAGE_A_WAITCAP", 2000000); // operations expected to wait, per step
    const double slot_mul = (double)env_u64("GP_STAGE_A_SLOTS_PER_RESV", 8);
    const bool allow_skip = env_u64("GP_STAGE_A_SKIP_LOOKUP", 1) != 0;
    const uint64_t max_pending = job->window * p.n_k; // list capacity, operations
    const uint64_t tag_max = 0xFFFFFFFFull, tail_span = 1ull << 22;

    uint32_t* const counts = job->d_counts.p;
    uint32_t* const err_flag = counts + (uint64_t)kSteps * kCountStride;
    uint4* const lists[2] = { job->d_pending[0].p, job->d_pending[1].p };
    unsigned long long* const tables[2] = { job->d_resv.p, job->d_resv.p + (1ull << job->resv_bits) };
    TailArgs t{};
    t.list[0] = lists[0];
    t.list[1] = lists[1];
    t.resv[0] = tables[0];
    t.resv[1] = tables[1];
    t.counts = counts;
    t.tag_max = tag_max;
    t.n_rounds = kDrainRounds;
    t.max_rounds = (uint32_t)tail_span - 16;
    t.err = err_flag;
    t.stats = job->d_stats.p;

    // feedback ring: what a step's cbf_round saw, read back without stalling the streams
    volatile uint32_t* hring = job->h_count_pinned;
    std::memset((void*)job->h_count_pinned, 0, 4 * kRing * sizeof(uint32_t));
    struct StepNote {
        uint64_t prev_units; // operations the previous step's first pass looked at
        bool prev_looked;    // ... and whether it looked the counters up (its waiting count is real)
        bool hashed;         // this step ran a first pass beside its round
    } notes[kRing];
    // Rounds per first pass: two.  About half of a waiting list holds all its slots in a
    // round (true conflicts on a long contig's one filter; slot collisions when hundreds of
    // small filters share the table), so hashing a new slice every step lets the list grow
    // until chains per counter are long; one round-only step between slices keeps it short
    // (measured: 47 ms instead of 63 per 10 Mbp contig, 790 instead of 1670 on 400 x 25 kbp).
    // Fixed rather than steered: feedback arrives several steps late, too late to stop growth.
    const uint32_t ahead = (uint32_t)std::min<uint64_t>(kRing, std::max<uint64_t>(1, env_u64("GP_STAGE_A_AHEAD", kAhead)));
    const uint32_t cadence = (uint32_t)std::max<uint64_t>(1, env_u64("GP_STAGE_A_CADENCE", 2));
    uint32_t rounds_since_fp = 0;
    uint32_t ring_head = 0, ring_tail = 0; // [tail, head) outstanding
    uint64_t known_in = 0;   // list length in the latest report
    uint64_t unacked = 0;    // operations of first passes whose outcome no report covers yet
    double last_frac = 1.0;  // waiting fraction of the most recent first pass we know about
    uint64_t scale = 1;
    uint32_t since_look = 0;
    auto drain_feedback = [&](bool block) -> int {
        while (ring_tail != ring_head) {
            const uint32_t i = ring_tail % kRing;
            const uint32_t want_seq = ring_tail + 1;
            if (block) {
                // the report of a step arrives within its few hundred microseconds: spin briefly, then
                // yield the core between looks, and give up after a wall-clock minute
                // (GP_STAGE_A_FEEDBACK_TIMEOUT_S), not after a number of iterations
                uint32_t spins = 0;
                const auto t_wait = std::chrono::steady_clock::now();
                const double limit_s = (double)env_u64("GP_STAGE_A_FEEDBACK_TIMEOUT_S", 60);
                while (hring[4 * i] != want_seq) {
                    if (++spins < 4096)
                        continue;
                    std::this_thread::yield();
                    if ((spins & 1023u) == 0 &&
                        std::chrono::duration<double>(std::chrono::steady_clock::now() - t_wait).count() > limit_s) {
                        set_error("stage A: device stopped reporting step feedback");
                        return GP_ERR_STATE;
                    }
                }
                block = false; // one report is enough to make progress
            } else if (hring[4 * i] != want_seq) {
                break;
            }
            known_in = hring[4 * i + 1];
            unacked -= std::min(unacked, notes[i].prev_units);
            if (notes[i].prev_looked && notes[i].prev_units) {
                const double frac = (double)hring[4 * i + 2] / (double)notes[i].prev_units;
                last_frac = frac;
                if (frac < 0.05)
                    scale = std::min<uint64_t>(scale * 4, 1u << 16);
                else if (frac < 0.15)
                    scale = std::min<uint64_t>(scale * 2, 1u << 16);
                else if (frac > 0.6 && scale > 1)
                    scale /= 2;
            }
            ring_tail++;
        }
        return GP_OK;
    };

    uint64_t step = 0;            // global step number: list/table parity and tag
    uint64_t n_first_pass = 0;
    bool span_open = false;
    uint64_t prev_units = 0;
    bool prev_looked = false;
    const auto t_host0 = std::chrono::steady_clock::now();
    const bool debug = env_u64("GP_STAGE_A_DEBUG", 0) != 0;

    // one step's cbf_round on s2 (a.drain selects the group-end flavour)
    auto launch_round = [&](uint32_t blocks) {
        a.tag_cur = tag_max - step;
        a.tag_next = a.tag_cur - 1;
        a.resv = tables[step & 1];
        a.resv_out = tables[(step + 1) & 1];
        a.list_in = lists[step & 1];
        a.list_out = lists[(step + 1) & 1];
        a.count_in = counts + (step % kSteps) * kCountStride;
        a.count_out = counts + ((step + 1) % kSteps) * kCountStride;
        a.count_zero = step ? counts + ((step - 1) % kSteps) * kCountStride : nullptr;
        hipLaunchKernelGGL(cbf_round, dim3(blocks), dim3(256), 0, s2, a);
    };

    // Batches are taken in groups; every slice covers the same range [off, off + slice) of
    // every batch of the group.  Many small batches therefore advance in lockstep: each filter
    // sees only a sliver of its reads per step, yet a step still holds enough operations to
    // fill the chip.  A group of one batch degenerates to consecutive slices of that batch.
    bool grid_used = false;
    auto grid_range = [&](uint32_t g_lo, uint32_t g_hi) -> int { // the grid-wide schedule over order[g_lo .. g_hi)
    if (g_lo < g_hi && !grid_used) {
        GP_HIP(hipStreamWaitEvent(s2, job->a_ev_init, 0)); // behind the memsets above
        grid_used = true;
    }
    for (uint32_t g_first = g_lo; g_first < g_hi;) {
        const uint32_t g_n = std::min<uint32_t>(group_cap, g_hi - g_first);
        // the group's counting filters start from zero; stream `s` is behind the previous
        // group's tail (and behind cbf_local) here, and every later step waits for `s`
        GP_HIP(hipMemsetAsync(job->d_cbf.p, 0, (uint64_t)g_n * p.n_k * a.cbf_stride, s));
        uint64_t longest = 0;
        for (uint32_t gi = g_first; gi < g_first + g_n; gi++)
            longest = std::max(longest, batch_pos[order[gi] + 1] - batch_pos[order[gi]]);
        scale = 1;
        last_frac = 1.0;
        uint32_t slice_seq = 0;
        uint64_t off = 0;
        while (off < longest) {
            int rc = drain_feedback(ring_head - ring_tail >= ahead);
            if (rc != GP_OK)
                return rc;
            // ---- this step's slice
            const uint64_t wmax = std::min<uint64_t>(
                job->window,
                std::max<uint64_t>(kTile, (uint64_t)(wait_cap / ((double)p.n_k * std::max(last_frac, 0.02)))));
            uint64_t slice = std::min<uint64_t>(lim0 * scale, std::max<uint64_t>(kTile, wmax / g_n));
            slice = std::max<uint64_t>(kTile, slice / kTile * kTile);
            slice = std::min<uint64_t>(slice, (longest - off + kTile - 1) / kTile * kTile);
            uint32_t tiles_per_batch = (uint32_t)(slice / kTile);
            if ((uint64_t)tiles_per_batch * g_n * kTile > job->window || (uint64_t)tiles_per_batch * g_n >= (1u << 14)) {
                set_error("stage A: slice does not fit the work buffers");
                return GP_ERR_STATE;
            }
            uint64_t w_positions = 0; // positions this slice really covers
            for (uint32_t gi = g_first; gi < g_first + g_n; gi++) {
                const uint64_t len = batch_pos[order[gi] + 1] - batch_pos[order[gi]];
                if (len > off)
                    w_positions += std::min<uint64_t>(slice, len - off);
            }
            uint64_t units = w_positions * p.n_k;
            // The list must never overflow: what may be waiting at most (latest report + every
            // first pass since) plus this slice has to fit.  Too much waiting also means long
            // chains per counter: then this step only runs a round.
            bool hash_now = units > 0;
            if (hash_now && known_in + unacked + units > max_pending) {
                if ((rc = drain_feedback(ring_tail != ring_head)) != GP_OK)
                    return rc;
                if (known_in + unacked + units > max_pending)
                    hash_now = false;
            }
            if (hash_now && (double)known_in > 4.0 * wait_cap)
                hash_now = false;
            if (hash_now && rounds_since_fp + 1 < cadence)
                hash_now = false;
            if (!hash_now && units == 0) {
                off += slice; // no batch of the group reaches this far
                continue;
            }

            {
                // Slots in use: about eight per expected reservation, within the allocation.  A
                // smaller table keeps the atomics at their full rate (27 G/s up to 2^24 slots,
                // 22 G/s at 2^26, measured); stale entries of other steps carry larger tags and
                // lose to any new reservation, so the mapping may change from step to step.
                const double expect = ((double)known_in + (double)unacked * std::min(1.0, last_frac * 1.5) +
                                       (hash_now ? (double)units * std::min(1.0, last_frac * 1.5) : 0.0)) * p.hash_num;
                // The count only moves when it has to (too full) or is far too large: a change
                // costs the waiting operations one round, because they are checked against the
                // table they reserved in with the old count.
                uint32_t bits = std::min<uint32_t>(18, job->resv_bits); // never beyond the allocation
                while (bits < job->resv_bits && (double)(1ull << bits) < slot_mul * expect)
                    bits++;
                a.bits_in = a.bits_out;
                if (bits > a.bits_out || bits + 2 <= a.bits_out)
                    a.bits_out = bits;
                // while nearly everything waits the look-up is wasted work; every eighth first
                // pass still does it so that the estimate follows the filter filling up
                a.all_wait = (allow_skip && last_frac > 0.9 && since_look < 7) ? 1u : 0u;
            }
            a.grp_first = g_first;
            a.grp_n = g_n;
            a.tiles_per_batch = tiles_per_batch;
            a.off = off;
            a.slice = slice;
            a.drain = 0;
            a.pos_base = slice_seq << 24;
            a.ring_slot = ring_head % kRing;
            a.ring_seq = ring_head + 1;
            notes[ring_head % kRing] = { prev_units, prev_looked, hash_now };

            if (debug && (job->counters.a_rounds < 40 || job->counters.a_rounds % 25 == 0))
                fprintf(stderr, "[stage A] step %llu: slice %llu units %llu hash %d known_in %llu unacked %llu frac %.3f "
                                "scale %llu bits %u skip_lookup %u ahead %u\n",
                        (unsigned long long)job->counters.a_rounds, (unsigned long long)slice,
                        (unsigned long long)units, (int)hash_now, (unsigned long long)known_in,
                        (unsigned long long)unacked, last_frac, (unsigned long long)scale, a.bits_out, a.all_wait,
                        ring_head - ring_tail);
            if (!span_open) {
                if ((rc = job->t_round.begin(s2, 1)) != GP_OK) // one span over all rounds
                    return rc;
                span_open = true;
            }
            launch_round(last_frac < 0.01 && known_in + unacked < 65536 ? 256u : (unsigned)kRoundBlocks);
            GP_HIP(hipGetLastError());
            if (hash_now) {
                if ((rc = job->t_first_pass.begin(s, units)) != GP_OK)
                    return rc;
                hipLaunchKernelGGL(cbf_first_pass, dim3(tiles_per_batch * g_n, p.n_k), dim3(kBlock), 0, s, a);
                GP_HIP(hipGetLastError());
                if ((rc = job->t_first_pass.end(s)) != GP_OK)
                    return rc;
                n_first_pass++;
                rounds_since_fp = 0;
                since_look = a.all_wait ? since_look + 1 : 0;
                unacked += units;
                prev_units = units;
                prev_looked = !a.all_wait;
                if (last_frac < 0.05 && g_n * p.n_k <= 64) {
                    // have the (batch, k) pairs this slice touched stopped changing?
                    hipLaunchKernelGGL(cbf_check_sat, dim3(g_n * p.n_k), dim3(1024), 0, s, a);
                    GP_HIP(hipGetLastError());
                }
                off += slice;
                slice_seq++;
            } else {
                prev_units = 0;
                prev_looked = false;
                rounds_since_fp++;
            }
            // the next step starts when both halves of this one are done
            const uint32_t e = (uint32_t)(step % kRing);
            GP_HIP(hipEventRecord(job->a_ev[e], s2));
            GP_HIP(hipEventRecord(job->a_ev_b[e], s));
            GP_HIP(hipStreamWaitEvent(s, job->a_ev[e], 0));
            GP_HIP(hipStreamWaitEvent(s2, job->a_ev_b[e], 0));
            ring_head++;
            step++;
            job->counters.a_rounds++;

            // ---- end of the group, or the slice numbers are used up: finish what waits
            if (off >= longest || slice_seq == 255) {
                a.drain = 1;
                a.bits_in = a.bits_out; // fixed from here to the end of the tail
                t.step0 = step;
                for (uint32_t r = 0; r < kDrainRounds; r++) {
                    launch_round(kRoundBlocks);
                    step++;
                }
                hipLaunchKernelGGL(cbf_tail, dim3(1), dim3(kTailThreads), 0, s2, a, t);
                GP_HIP(hipGetLastError());
                job->counters.a_rounds += kDrainRounds;
                step += tail_span + 1; // the tail's rounds use the tags in between
                const uint32_t e2 = (uint32_t)(step % kRing);
                GP_HIP(hipEventRecord(job->a_ev[e2], s2));
                GP_HIP(hipStreamWaitEvent(s, job->a_ev[e2], 0));
                slice_seq = 0;
                // collect the reports of the steps before the drain (a late one would be counted
                // against the next group's first passes); nothing waits on the device any more
                while (ring_tail != ring_head)
                    if ((rc = drain_feedback(true)) != GP_OK)
                        return rc;
                known_in = 0;
                unacked = 0;
                prev_units = 0;
                prev_looked = false;
                if (step + 2 * tail_span >= tag_max) {
                    // tags used up (thousands of groups): start over with clean tables
                    GP_HIP(hipStreamSynchronize(s2));
                    GP_HIP(hipMemsetAsync(job->d_resv.p, 0xFF, job->d_resv.n * sizeof(unsigned long long), s2));
                    GP_HIP(hipStreamSynchronize(s2));
                    step = 0;
                }
            }
        }
        g_first += g_n;
    }
    return GP_OK;
    };
    {
        int rcg = grid_range(0, n_big);
        if (rcg != GP_OK)
            return rcg;
    }
    if (!lvl_fallback.empty()) {
        // pairs of the level-by-level census that overflowed a run (many copies of one k-mer within a
        // stretch): their batches take the grid-wide schedule after all -- whole batches, every k: setting
        // the bits of a filter the levels have already finished a second time changes nothing
        GP_HIP(hipStreamSynchronize(s));
        GP_HIP(hipStreamSynchronize(s2));
        for (size_t i = 0; i < lvl_fallback.size(); i++)
            order[n_big + i] = lvl_fallback[i];
        std::vector<uint32_t> slot(job->n_batches, 0);
        for (uint32_t i = 0; i < (uint32_t)lvl_fallback.size(); i++)
            slot[lvl_fallback[i]] = (i % group_cap) * p.n_k;
        GP_HIP(hipMemcpyAsync(job->d_batch_order.p, order.data(), order.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        GP_HIP(hipMemcpyAsync(job->d_batch_slot.p, slot.data(), slot.size() * sizeof(uint32_t), hipMemcpyHostToDevice, s));
        GP_HIP(hipStreamSynchronize(s));
        int rcg = grid_range(n_big, n_big + (uint32_t)lvl_fallback.size());
        if (rcg != GP_OK)
            return rcg;
    }
    if (span_open) {
        int rc = job->t_round.end(s2);
        if (rc != GP_OK)
            return rc;
    }
    if (debug) {
        const double host_ms = std::chrono::duration<double, std::milli>(
                                   std::chrono::steady_clock::now() - t_host0).count();
        GP_HIP(hipStreamSynchronize(s2));
        const double all_ms = std::chrono::duration<double, std::milli>(
                                  std::chrono::steady_clock::now() - t_host0).count();
        fprintf(stderr, "[stage A] %llu first passes: host enqueue %.2f ms, device done after %.2f ms\n",
                (unsigned long long)n_first_pass, host_ms, all_ms);
    }
    // the rest of the pipeline (stream `s`) continues behind the last tail
    if (grid_used) {
        GP_HIP(hipEventRecord(job->a_ev_init, s2));
        GP_HIP(hipStreamWaitEvent(s, job->a_ev_init, 0));
    }
    unsigned long long* const st = job->h_readback;   // pinned: [0..4] statistics, [8] the tail's error flag
    st[8] = 0;
    GP_HIP(hipMemcpyAsync(st, job->d_stats.p, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    GP_HIP(hipMemcpyAsync(st + 8, err_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    GP_HIP(hipStreamSynchronize(s));
    const uint32_t tail_err = (uint32_t)st[8];
    job->counters.a_positions = st[0];
    job->counters.a_ops = st[1];
    job->counters.a_ordered_ops = st[2];
    job->counters.a_rounds += st[3];
    job->counters.a_reservations = st[4];
    job->counters.a_replayed = st[5];
    job->counters.a_settled_hits = st[6];
    job->counters.a_counted = st[7];
    if (tail_err) {
        set_error("stage A: reservation rounds did not converge");
        return GP_ERR_STATE;
    }
    int rc = job->t_first_pass.collect();
    if (rc != GP_OK)
        return rc;
    if ((rc = job->t_round.collect()) != GP_OK || (rc = job->t_local.collect()) != GP_OK ||
        (rc = job->t_census.collect()) != GP_OK || (rc = job->t_ckc.collect()) != GP_OK)
        return rc;
    job->filters_ready = true;
    return GP_OK;
}

} // namespace gp
