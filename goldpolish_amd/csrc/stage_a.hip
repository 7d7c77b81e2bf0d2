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
//   * Counters never exceed thr and never decrease, and a minimum of thr-1 or thr
//     both mean "insert".  An operation that sees all of its counters >= thr-1
//     -- at ANY time, even ahead of its turn, even from a stale read -- inserts
//     its k-mer and writes nothing ("settled", try_settle).
//   * Every other operation takes a deterministic reservation: atomicMin of
//     (round tag, position) into a hashed slot for each of its counters; in the
//     next launch the operations that hold all their slots are exactly those
//     with no earlier unfinished operation on any of their counters, so they
//     apply the serial update; the rest retry (round_op).  Tags shrink per
//     launch (newer beats stale), two tables alternate.
//   * A window (bounded waiting operations per counter) = cbf_first_pass, a few
//     grid-wide cbf_round launches, then cbf_tail: one workgroup that loops over
//     the short remaining list with L1-bypassing loads.  No launch waits for the
//     host; the first pass of the next window runs beside it on a second stream;
//     feedback (how much waited) returns through mapped host memory.
//   * cbf_check_sat marks (batch, k) pairs whose counters and filter can no
//     longer change; their k-mers then cost nothing.
//
// Memory behaviour: the sequence is streamed once per k through LDS (coalesced
// 1 B/base); the counter / bit accesses are 1-byte random accesses into 10 MiB
// (counters) and 512 KiB (bits) per (batch, k) -- random-transaction and
// atomic bound, no reuse to tile for.  Integer / bitwise work only.
#include "gp_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>

namespace gp {

constexpr int kTile = 1024;        // positions per workgroup
constexpr int kPerThread = 4;      // consecutive positions per lane
constexpr int kBlock = kTile / kPerThread;
constexpr int kHalo = 64;          // k <= 64
constexpr uint32_t kMaxRounds = 4094;
constexpr int kRoundBlocks = 2048;
constexpr uint32_t kRing = 16;
constexpr uint32_t kCountStride = 32; // one 128-B line per round counter: appends of different rounds never share a line

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
    uint64_t cbf_stride; // bytes per (batch, k) counter array in memory
    uint32_t nib;        // 1: two 4-bit counters per byte (every threshold <= 15)
    uint32_t bf_words;
    FastMod cbf_mod, bf_mod;
    KConsts kc[GP_MAX_K_VALUES];
    // reservation slots, (round tag << 32) | window position; two tables alternate so that
    // the reservations a launch makes never disturb the ones it is still checking
    unsigned long long* resv;      // table the incoming list reserved in
    unsigned long long* resv_out;  // table this launch reserves in
    uint64_t tag_cur;    // 40-bit tag the incoming list reserved with
    uint64_t tag_next;   // tag for reservations made by this launch (always smaller: newer wins)
    uint32_t* sat;       // per (batch, k): bit 0 = every counter >= max thr-1, bit 1 = filter all ones
    const uint32_t* batch_thr_max; // per batch: largest kmer_threshold of its contigs
    uint32_t tail_from;  // first round the tail kernel may take over (see cbf_tail)
    uint32_t tail_max;   // lists up to this long are left to the tail kernel
    uint32_t resv_bits;  // slots in use this window (<= the allocation), see stage_a_run
    uint32_t all_wait;   // 1: nearly every operation waits anyway; the first pass skips the counter look-up
    uint4* list_in;
    uint4* list_out;
    uint32_t* counts;    // counts[r] = length of the list round r consumes
    uint32_t round;
    // a window = the same slice [off, off + slice) of every batch of a group (batch-relative
    // positions), one or more 1024-position tiles per batch
    const uint64_t* batch_pos; // [n_batches+1] first read position of each batch
    uint32_t grp_first, grp_n, tiles_per_batch;
    uint64_t off, slice;
    unsigned long long* stats; // [0] positions [1] ops [2] ordered ops
};

__device__ inline uint32_t resv_slot(const AArgs& a, uint32_t stream, uint32_t idx)
{
    uint64_t key = (uint64_t)stream * a.cbf_bytes + idx;
    return (uint32_t)((key * 0x9E3779B97F4A7C15ULL) >> (64 - a.resv_bits));
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
    for (uint32_t i = 0; i < a.hash_num; i++) {
        uint64_t n = fastmod(hv[i], a.bf_mod);
        uint32_t bit = 1u << (uint32_t)(n & 31);
        uint32_t* p = w + (n >> 5);
        // a stale read only costs a redundant atomic: bits are never cleared
        if (!(*p & bit))
            atomicOr(p, bit);
    }
}

// One operation whose hash is known.  Returns true when it is finished, false
// when it has to wait for its turn (then idx[] holds its counter indices).
__device__ inline bool try_settle(const AArgs& a, uint32_t stream, uint32_t thr, uint32_t k,
                                  uint64_t base, uint32_t* idx)
{
    uint64_t hv[kMaxHash];
    hv[0] = base;
    for (uint32_t i = 1; i < a.hash_num; i++)
        hv[i] = ext_hash(base, k, i);
    const uint8_t* cb = a.cbf + (uint64_t)stream * a.cbf_stride;
    uint32_t c[kMaxHash];
    uint32_t m = 255;
    for (uint32_t i = 0; i < a.hash_num; i++) {
        idx[i] = (uint32_t)fastmod(hv[i], a.cbf_mod);
        c[i] = ctr_load<false>(a, cb, idx[i]);
        m = min(m, c[i]);
    }
    if (m + 1 < thr)
        return false;
    // Every counter is already >= thr-1: the k-mer goes in whatever happens before this
    // operation.  Its own update would only lift counters from thr-1 to thr, and no later
    // decision can tell those two values apart (a minimum of thr-1 or thr both mean
    // "insert", and neither is ever incremented from below again), so nothing is written.
    bf_insert(a, stream, hv);
    return true;
}

// A reservation is (tag, position) packed so that atomicMin prefers the newer round first
// and the earlier position second; stale entries of older rounds never need clearing.
__device__ inline unsigned long long resv_word(uint64_t tag, uint32_t pos)
{
    return ((unsigned long long)tag << 24) | pos; // window positions are < 2^24
}

__device__ inline void reserve(const AArgs& a, uint32_t stream, const uint32_t* idx,
                               uint32_t pos)
{
    const unsigned long long w = resv_word(a.tag_next, pos);
    for (uint32_t i = 0; i < a.hash_num; i++)
        atomicMin(&a.resv_out[resv_slot(a, stream, idx[i])], w);
}


// A workgroup appends `n_mine` items per lane with ONE global atomic.  Every lane of the
// workgroup must call it (n_mine may be 0).  Returns the index of the lane's first item.
// (Per-wave appends to one word serialise at about 90 per microsecond chip-wide; with four
// lists of a window in flight that was a third of a window's time.)
template <int THREADS>
__device__ inline uint32_t block_append(uint32_t* counter, uint32_t n_mine, uint32_t* s_wave /*[THREADS/64 + 1]*/)
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
    __shared__ unsigned long long s_stats[3];
    __shared__ uint32_t s_wave[kBlock / 64 + 1];
    const uint32_t ki = blockIdx.y;
    const KConsts& kc = a.kc[ki];
    const uint32_t k = kc.k;
    const uint32_t gb = a.grp_first + blockIdx.x / a.tiles_per_batch;
    const uint64_t b_start = a.batch_pos[gb], b_end = a.batch_pos[gb + 1];
    const uint64_t tile0 = b_start + a.off + (uint64_t)(blockIdx.x % a.tiles_per_batch) * kTile;
    const uint64_t tile_end = min(min(b_end, b_start + a.off + a.slice), tile0 + (uint64_t)kTile);
    if (tile0 >= tile_end)
        return; // this batch has no reads left at this offset

    if (threadIdx.x < 3)
        s_stats[threadIdx.x] = 0;
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
        for (uint32_t i = 0; i < k && !saturated; i++) {
            uint32_t cf = codes[t0 + i];
            uint32_t cr = codes[t0 + k - 1 - i];
            fh = srol1(fh) ^ (cf < 4 ? seed_of_code(cf) : 0ULL);
            rh = srol1(rh) ^ (cr < 4 ? seed_of_code(cr ^ 2) : 0ULL);
            if (cf >= 4)
                last_bad = (int)i;
        }
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
            // for any k-mer to go in, and every bit it could set is set (cbf_check_sat)
            const uint32_t sat = a.sat[stream];
            if (sat == 3u)
                continue;
            op_base[j] = fh + rh;
            op_seg[j] = seg;
            op_state[j] = (sat & 1u) ? 2u : 1u; // 2: counters saturated, only the filter is left
            if (op_state[j] == 1u) {
                const uint8_t* cb = a.cbf + (uint64_t)stream * a.cbf_stride;
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
        // pass 2: settle what can be settled (see try_settle); count what has to wait
#pragma unroll
        for (int j = 0; j < kPerThread; j++) {
            if (!op_state[j])
                continue;
            const uint32_t seg_j = op_seg[j];
            const uint32_t stream = a.seg_batch[seg_j] * a.n_k + ki;
            const uint32_t thr = a.seg_thr[seg_j] - 2 + ki;
            if (op_state[j] == 2u || op_min[j] + 1 >= thr) {
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
        uint32_t at = block_append<kBlock>(&a.counts[0], n_ordered, s_wave);
#pragma unroll
        for (int j = 0; j < kPerThread; j++) {
            if (!op_state[j])
                continue;
            const uint32_t seg_j = op_seg[j];
            const uint32_t stream = a.seg_batch[seg_j] * a.n_k + ki;
            // order key: tiles of a batch are consecutive blocks in increasing position order
            const uint32_t rel = blockIdx.x * (uint32_t)kTile + (uint32_t)(t0 + j);
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

// Has a (batch, k) filter pair stopped changing?  Counters only matter through "is the
// minimum >= thr-1", so once every counter is >= thr-1 (for the largest threshold any contig
// of the batch uses) every later operation settles; once every bit of the Bloom filter is
// set, inserting is a no-op.  One workgroup per stream of the window; flags only ever go
// from 0 to 1, so a first pass that reads them late loses nothing but time.
__global__ __launch_bounds__(1024) void cbf_check_sat(AArgs a, uint32_t stream0)
{
    __shared__ uint32_t bad_c, bad_b;
    const uint32_t stream = stream0 + blockIdx.x;
    uint32_t cur = a.sat[stream];
    if (cur == 3u)
        return;
    if (threadIdx.x == 0)
        bad_c = bad_b = 0;
    __syncthreads();
    const uint32_t batch = stream / a.n_k, ki = stream % a.n_k;
    const uint32_t need = a.batch_thr_max[batch] - 2 + ki - 1; // thr - 1
    if (!(cur & 1u)) {
        const uint8_t* cb = a.cbf + (uint64_t)stream * a.cbf_stride;
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
                                uint64_t tag_cur, uint64_t tag_next)
{
    const uint64_t base = ((uint64_t)rec.y << 32) | rec.x;
    const uint32_t pos = rec.z;
    const uint32_t ki = rec.w & 7, seg = rec.w >> 3;
    const uint32_t k = a.kc[ki].k;
    const uint32_t stream = a.seg_batch[seg] * a.n_k + ki;
    const uint32_t thr = a.seg_thr[seg] - 2 + ki;
    uint32_t idx[kMaxHash] = { rix.x, rix.y, rix.z, rix.w };
    // only four indices travel with the record; the rest are recomputed
    for (uint32_t h = 4; h < a.hash_num; h++)
        idx[h] = (uint32_t)fastmod(ext_hash(base, k, h), a.cbf_mod);

    const unsigned long long want = resv_word(tag_cur, pos);
    bool mine = true;
    for (uint32_t h = 0; h < a.hash_num; h++)
        mine &= (ld<FRESH>(&resv_in[resv_slot(a, stream, idx[h])]) == want);

    uint8_t* cb = a.cbf + (uint64_t)stream * a.cbf_stride;
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
    } else if (m + 1 >= thr) {
        insert = true; // settled (see try_settle): no counter needs writing
    } else {
        const unsigned long long w = resv_word(tag_next, pos);
        for (uint32_t h = 0; h < a.hash_num; h++)
            atomicMin(&resv_out[resv_slot(a, stream, idx[h])], w);
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
    const uint32_t n_in = a.counts[a.round * kCountStride];
    if (a.round >= a.tail_from && n_in <= a.tail_max)
        return; // short list: cbf_tail finishes it without further launches
    for (uint32_t i0 = blockIdx.x * blockDim.x; i0 < n_in; i0 += gridDim.x * blockDim.x) {
        const uint32_t i = i0 + threadIdx.x;
        uint4 rec = {}, rix = {};
        bool waits = false;
        if (i < n_in) {
            rec = a.list_in[2 * i];
            rix = a.list_in[2 * i + 1];
            waits = round_op<false>(a, rec, rix, a.resv, a.resv_out, a.tag_cur, a.tag_next);
        }
        const uint32_t at = block_append<256>(&a.counts[(a.round + 1) * kCountStride], waits ? 1u : 0u, s_wave);
        if (waits) {
            a.list_out[2 * at] = rec;
            a.list_out[2 * at + 1] = rix;
        }
    }
}

// ---- the tail of a window: one workgroup keeps running rounds on the list the grid-wide
// rounds left (the first list no longer than tail_max, or whatever is left after the last
// of them) until nothing waits.  Rounds are separated by workgroup barriers; every load of
// data another wave may have written in an earlier round bypasses L1.
constexpr int kTailThreads = 1024;

struct TailArgs {
    uint4* list[2];
    unsigned long long* resv[2];
    uint64_t tag_first;   // tag of round 0 of this window; round r uses tag_first - r
    uint32_t n_rounds;    // grid-wide rounds that were launched
    uint32_t max_rounds;  // bug trap
    uint32_t* err;
    uint32_t* info;       // [0] round at which the tail took over, [1] list length then
    volatile uint32_t* host_ring; // mapped host memory: 4 words per slot {seq, waiting, took_at, took_len}
    uint32_t ring_slot, ring_seq;
    uint32_t n_counts;    // counters to zero for the next user of this buffer set
    unsigned long long* stats; // [3] rounds run by the tail
};

__global__ __launch_bounds__(kTailThreads) void cbf_tail(AArgs a, TailArgs t)
{
    __shared__ uint32_t s_n, s_out;
    if (threadIdx.x == 0) {
        uint32_t r = 0;
        while (r < t.n_rounds && !(r >= a.tail_from && a.counts[r * kCountStride] <= a.tail_max))
            r++;
        s_n = a.counts[r * kCountStride];
        s_out = r; // reused below as "current round" until the loop starts
        t.info[0] = r;
        t.info[1] = s_n;
    }
    __syncthreads();
    uint32_t n = s_n, round = s_out;
    __syncthreads();
    uint32_t done_rounds = 0;
    while (n > 0) {
        if (threadIdx.x == 0)
            s_out = 0;
        __syncthreads();
        const uint4* in = t.list[round & 1];
        uint4* out = t.list[(round + 1) & 1];
        const uint64_t tag_cur = t.tag_first - round, tag_next = tag_cur - 1;
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
            if (round_op<true>(a, rec, rix, t.resv[round & 1], t.resv[(round + 1) & 1], tag_cur,
                               tag_next)) {
                uint32_t at = atomicAdd(&s_out, 1u);
                out[2 * at] = rec;
                out[2 * at + 1] = rix;
            }
        }
        __threadfence();
        __syncthreads();
        n = s_out;
        round++;
        done_rounds++;
        __syncthreads();
        if (done_rounds > t.max_rounds) {
            if (threadIdx.x == 0)
                atomicOr(t.err, 1u);
            break;
        }
    }
    if (threadIdx.x == 0) {
        if (done_rounds)
            atomicAdd(&t.stats[3], (unsigned long long)done_rounds);
        // feedback for the host (how much waited, where the tail took over), written straight
        // into mapped host memory; the sequence number goes last
        volatile uint32_t* slot = t.host_ring + 4 * t.ring_slot;
        slot[1] = a.counts[0];
        slot[2] = t.info[0];
        slot[3] = t.info[1];
        __threadfence_system();
        slot[0] = t.ring_seq;
        __threadfence_system();
    }
    __syncthreads();
    // leave the counter array clean for the window that uses this buffer set next
    for (uint32_t i = threadIdx.x; i < t.n_counts; i += blockDim.x)
        a.counts[i * kCountStride] = 0;
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

static uint64_t env_u64(const char* name, uint64_t dflt)
{
    const char* v = getenv(name);
    if (!v || !*v)
        return dflt;
    return strtoull(v, nullptr, 10);
}

int stage_a_prepare(gp_job* job)
{
    const gp_params& p = job->prm;
    // window: how many read positions are hashed per launch.  The pending lists
    // are sized for the worst case (every k-mer of every k must wait).
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
    // two sets (windows alternate between them so that the first pass of window w+1 can
    // run beside the rounds of window w), two lists per set, two uint4 per record
    for (int i = 0; i < 4; i++)
        if ((rc = job->d_pending[i].alloc(2 * max_pending)) != GP_OK)
            return rc;
    // reservation slots: two per possible waiting counter touch, power of two, <= 1 GiB
    uint32_t bits = 16;
    while ((1ull << bits) < max_pending * p.hash_num * 2 && bits < 26)
        bits++;
    job->resv_bits = bits;
    if ((rc = job->d_resv.alloc(4ull << bits)) != GP_OK) // two tables per set
        return rc;
    if ((rc = job->d_counts.alloc(2ull * (kMaxRounds + 2) * kCountStride)) != GP_OK)
        return rc;
    if ((rc = job->d_stats.alloc(16)) != GP_OK)
        return rc;
    if ((rc = job->d_batch_pos.alloc((uint64_t)job->n_batches + 1)) != GP_OK)
        return rc;
    if ((rc = job->d_sat.alloc((uint64_t)job->n_batches * p.n_k)) != GP_OK ||
        (rc = job->d_batch_thr_max.alloc(job->n_batches)) != GP_OK)
        return rc;
    {
        std::vector<uint32_t> tm(job->n_batches, 4);
        for (uint32_t bi = 0; bi < job->n_batches; bi++)
            for (uint32_t c = job->h_batch_contig_begin[bi]; c < job->h_batch_contig_begin[bi + 1]; c++)
                tm[bi] = std::max(tm[bi], job->h_contig_thr[c]);
        if (job->n_batches)
            GP_HIP(hipMemcpy(job->d_batch_thr_max.p, tm.data(), tm.size() * sizeof(uint32_t),
                             hipMemcpyHostToDevice));
    }
    GP_HIP(hipHostMalloc((void**)&job->h_count_pinned, 256));
    for (uint32_t i = 0; i < kRing; i++)
        GP_HIP(hipEventCreateWithFlags(&job->a_ev[i], hipEventDisableTiming));
    for (int i = 0; i < 2; i++) {
        GP_HIP(hipEventCreateWithFlags(&job->a_ev_fp[i], hipEventDisableTiming));
        GP_HIP(hipEventCreateWithFlags(&job->a_ev_tail[i], hipEventDisableTiming));
    }
    GP_HIP(hipEventCreateWithFlags(&job->a_ev_init, hipEventDisableTiming));
    GP_HIP(hipStreamCreateWithFlags(&job->a_stream2, hipStreamNonBlocking));
    return GP_OK;
}

void stage_a_release(gp_job* job)
{
    if (job->h_count_pinned)
        (void)hipHostFree(job->h_count_pinned);
    job->h_count_pinned = nullptr;
    for (uint32_t i = 0; i < kRing; i++)
        if (job->a_ev[i])
            (void)hipEventDestroy(job->a_ev[i]);
    for (int i = 0; i < 2; i++) {
        if (job->a_ev_fp[i])
            (void)hipEventDestroy(job->a_ev_fp[i]);
        if (job->a_ev_tail[i])
            (void)hipEventDestroy(job->a_ev_tail[i]);
    }
    if (job->a_ev_init)
        (void)hipEventDestroy(job->a_ev_init);
    if (job->a_stream2)
        (void)hipStreamDestroy(job->a_stream2);
}

int stage_a_run(gp_job* job)
{
    const gp_params& p = job->prm;
    hipStream_t s = job->ctx->stream;
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
    a.resv_bits = job->resv_bits;
    a.stats = job->d_stats.p;
    a.sat = job->d_sat.p;
    a.batch_thr_max = job->d_batch_thr_max.p;

    job->t_first_pass.reset();
    job->t_round.reset();
    job->counters.a_rounds = 0;

    // fresh filters + free reservation tables
    GP_HIP(hipMemsetAsync(job->d_cbf.p, 0,
                          std::min<uint64_t>(job->d_cbf.n, (uint64_t)job->n_batches * p.n_k * a.cbf_stride + 8), s));
    GP_HIP(hipMemsetAsync(job->d_bf.p, 0, job->d_bf.n * sizeof(uint32_t), s));
    GP_HIP(hipMemsetAsync(job->d_stats.p, 0, 16 * sizeof(unsigned long long), s));
    GP_HIP(hipMemsetAsync(job->d_sat.p, 0, job->d_sat.n * sizeof(uint32_t), s));
    GP_HIP(hipMemsetAsync(job->d_counts.p, 0, job->d_counts.n * sizeof(uint32_t), s));
    GP_HIP(hipMemsetAsync(job->d_resv.p, 0xFF, job->d_resv.n * sizeof(unsigned long long), s));
    // stream `s` runs the first passes, stream `s2` the rounds and tails
    hipStream_t s2 = env_u64("GP_STAGE_A_SERIAL", 0) ? s : job->a_stream2; // debugging aid: one stream
    GP_HIP(hipEventRecord(job->a_ev_init, s));
    GP_HIP(hipStreamWaitEvent(s2, job->a_ev_init, 0));

    if (job->read_positions == 0 || job->n_contigs == 0) {
        job->filters_ready = true;
        return GP_OK;
    }

    // Window = the slice of read positions hashed per first-pass launch.  What makes the
    // reservation rounds expensive is many waiting operations per counter, so a window
    // takes at most `lim` positions of any one batch (hash_num * lim touches over
    // cbf_bytes counters: a load of 1/4 at scale 1) but as many whole batches as fit.
    // Once a filter is saturated nearly everything settles in the first pass; the
    // window then grows again (feedback on the waiting fraction of the last window).
    std::vector<uint64_t> batch_pos(job->n_batches + 1);
    for (uint32_t bi = 0; bi < job->n_batches; bi++)
        batch_pos[bi] = job->h_seg_start[job->h_batch_contig_begin[bi]];
    batch_pos[job->n_batches] = job->read_positions;
    GP_HIP(hipMemcpyAsync(job->d_batch_pos.p, batch_pos.data(), batch_pos.size() * sizeof(uint64_t),
                          hipMemcpyHostToDevice, s));
    GP_HIP(hipStreamSynchronize(s)); // batch_pos is a local
    a.batch_pos = job->d_batch_pos.p;
    const uint64_t load_div = std::max<uint64_t>(1, env_u64("GP_STAGE_A_LOAD_DIV", 6));
    const uint64_t lim0 = std::max<uint64_t>(kTile, p.cbf_bytes / (load_div * p.hash_num));
    const uint32_t n_rounds = (uint32_t)std::min<uint64_t>(env_u64("GP_STAGE_A_ROUNDS", 6), kMaxRounds);
    const uint64_t tag_max = (1ull << 40) - 1, tail_span = 1ull << 22;
    a.tail_from = 2;
    a.tail_max = (uint32_t)env_u64("GP_STAGE_A_TAIL", 4096);
    TailArgs t{};
    t.n_rounds = n_rounds;
    t.max_rounds = (uint32_t)tail_span - 16;
    t.err = job->d_counts.p + (uint64_t)(kMaxRounds + 1) * kCountStride; // last line of set 0's counter array is a flag
    t.stats = job->d_stats.p;
    t.host_ring = job->h_count_pinned;
    std::memset((void*)job->h_count_pinned, 0, 256);
    uint64_t wi = 0; // window index; its parity picks the buffer set
    const auto t_host0 = std::chrono::steady_clock::now();
    uint64_t scale = 1, gen = 1;
    double last_frac = 1.0; // waiting fraction of the most recent window we know about
    uint32_t want_rounds = n_rounds; // grid-wide rounds per window, steered by the same feedback
    // feedback ring: waiting count of a window, read back without stalling the stream
    uint32_t ring_head = 0, ring_tail = 0, ring_slot = 0; // [tail, head) outstanding
    uint64_t ring_units[kRing];
    bool ring_looked[kRing]; // did that window's first pass look the counters up (frac is real)?
    uint32_t since_look = 0;
    volatile uint32_t* hring = job->h_count_pinned;
    auto drain = [&](bool block) -> int {
        while (ring_tail != ring_head) {
            const uint32_t i = ring_tail % kRing;
            const uint32_t want_seq = ring_tail + 1;
            if (block) {
                // only when the host is a full ring ahead of the device
                uint64_t spins = 0;
                while (hring[4 * i] != want_seq) {
                    if (++spins > (1ull << 34)) {
                        set_error("stage A: device stopped reporting window feedback");
                        return GP_ERR_STATE;
                    }
                }
            } else if (hring[4 * i] != want_seq) {
                break;
            }
            // a window that skipped the look-up reports everything as waiting: keep the estimate
            const double frac = ring_looked[i] ? (double)hring[4 * i + 1] / (double)ring_units[i] : last_frac;
            last_frac = frac;
            // did the grid-wide rounds hand the tail a short list, and how early?
            const uint32_t took_at = hring[4 * i + 2];
            const uint32_t took_len = hring[4 * i + 3];
            if (took_len > a.tail_max)
                want_rounds = std::min<uint32_t>(want_rounds * 2, 512);
            else
                want_rounds = std::max<uint32_t>(2, std::min<uint32_t>(want_rounds, took_at + 2));
            if (took_len > 16 * a.tail_max && scale > 1)
                scale /= 2; // the single-workgroup tail got far too much: smaller windows
            if (frac < 0.05)
                scale = std::min<uint64_t>(scale * 4, 1u << 16);
            else if (frac < 0.15)
                scale = std::min<uint64_t>(scale * 2, 1u << 16);
            else if (frac > 0.6 && scale > 1)
                scale /= 2;
            ring_tail++;
        }
        return GP_OK;
    };
    // Batches are taken in groups; every window covers the same slice [off, off + slice) of
    // every batch of the group.  Many small batches therefore advance in lockstep: each filter
    // sees only a sliver of its reads per window (no chains of waiting operations to speak
    // of), yet a window still holds enough operations to fill the chip.  A group of one batch
    // degenerates to consecutive slices of that batch (the long-contig case).
    const uint32_t group_max = (uint32_t)std::max<uint64_t>(1, env_u64("GP_STAGE_A_GROUP", 2048));
    const double slot_mul = (double)env_u64("GP_STAGE_A_SLOTS_PER_RESV", 8);
    const bool allow_skip = env_u64("GP_STAGE_A_SKIP_LOOKUP", 1) != 0;
    for (uint32_t g_first = 0; g_first < job->n_batches;) {
        uint32_t g_n = std::min<uint32_t>(group_max, job->n_batches - g_first);
        g_n = (uint32_t)std::min<uint64_t>(g_n, std::max<uint64_t>(1, job->window / kTile));
        uint64_t longest = 0;
        for (uint32_t bi = g_first; bi < g_first + g_n; bi++)
            longest = std::max(longest, batch_pos[bi + 1] - batch_pos[bi]);
        scale = 1;
        last_frac = 1.0;
      for (uint64_t off = 0; off < longest;) {
        int rc = drain(ring_head - ring_tail >= kRing);
        if (rc != GP_OK)
            return rc;
        // per-batch slice: bounded by the per-filter load (lim), by the total number of
        // operations expected to wait (about 3 M per window), and by the buffers
        const uint64_t wmax = std::min<uint64_t>(
            job->window,
            std::max<uint64_t>(kTile, (uint64_t)(3.0e6 / ((double)p.n_k * std::max(last_frac, 0.02)))));
        uint64_t slice = std::min<uint64_t>(lim0 * scale, std::max<uint64_t>(kTile, wmax / g_n));
        slice = std::max<uint64_t>(kTile, slice / kTile * kTile);
        slice = std::min<uint64_t>(slice, (longest - off + kTile - 1) / kTile * kTile);
        const uint32_t tiles_per_batch = (uint32_t)(slice / kTile);
        if ((uint64_t)tiles_per_batch * g_n * kTile > job->window || (uint64_t)tiles_per_batch * g_n >= (1u << 14)) {
            set_error("stage A: window does not fit the work buffers");
            return GP_ERR_STATE;
        }
        uint64_t w_positions = 0; // positions this window really covers
        for (uint32_t bi = g_first; bi < g_first + g_n; bi++) {
            const uint64_t len = batch_pos[bi + 1] - batch_pos[bi];
            if (len > off)
                w_positions += std::min<uint64_t>(slice, len - off);
        }
        if (w_positions == 0) {
            off += slice;
            continue;
        }
        const uint32_t bcur = g_first;
        const uint64_t w0 = 0, w1 = w_positions; // only their difference is used below
        const uint32_t set = (uint32_t)(wi & 1);
        uint32_t* counts = job->d_counts.p + (uint64_t)set * (kMaxRounds + 2) * kCountStride;
        t.list[0] = job->d_pending[2 * set].p;
        t.list[1] = job->d_pending[2 * set + 1].p;
        t.resv[0] = job->d_resv.p + ((uint64_t)(2 * set) << job->resv_bits);
        t.resv[1] = job->d_resv.p + ((uint64_t)(2 * set + 1) << job->resv_bits);
        t.info = counts + (uint64_t)kMaxRounds * kCountStride; // a line no round ever counts in
        // this set's buffers were last used by window wi-2: wait for its tail
        if (wi >= 2)
            GP_HIP(hipStreamWaitEvent(s, job->a_ev_tail[set], 0));
        const uint64_t g0 = gen;
        {
            // Slots in use: about eight per expected reservation, within the allocation.  A
            // smaller table keeps the atomics at their full rate (27 G/s up to 2^24 slots,
            // 22 G/s at 2^26, measured); stale entries of other windows carry larger tags and
            // lose to any new reservation, so the mapping may change from window to window.
            const double expect = (double)w_positions * p.n_k * std::min(1.0, last_frac * 1.5) * p.hash_num;
            uint32_t bits = std::min<uint32_t>(18, job->resv_bits); // never beyond the allocation
            while (bits < job->resv_bits && (double)(1ull << bits) < slot_mul * expect)
                bits++;
            a.resv_bits = bits;
            // while nearly everything waits the look-up is wasted work; every eighth window
            // still does it so that the estimate follows the filter filling up
            a.all_wait = (allow_skip && last_frac > 0.9 && since_look < 7) ? 1u : 0u;
            since_look = a.all_wait ? since_look + 1 : 0;
        }
        a.grp_first = g_first;
        a.grp_n = g_n;
        a.tiles_per_batch = tiles_per_batch;
        a.off = off;
        a.slice = slice;
        a.counts = counts;
        a.resv = nullptr;
        a.resv_out = t.resv[0];
        a.tag_cur = 0;
        a.tag_next = tag_max - g0;
        a.list_in = nullptr;
        a.list_out = t.list[0];
        a.round = 0;
        dim3 grid(tiles_per_batch * g_n, p.n_k);
        if ((rc = job->t_first_pass.begin(s, (w1 - w0) * p.n_k)) != GP_OK)
            return rc;
        hipLaunchKernelGGL(cbf_first_pass, grid, dim3(kBlock), 0, s, a);
        GP_HIP(hipGetLastError());
        if ((rc = job->t_first_pass.end(s)) != GP_OK)
            return rc;
        GP_HIP(hipEventRecord(job->a_ev_fp[set], s));
        {
            const uint32_t i = ring_head % kRing;
            ring_units[i] = (w1 - w0) * p.n_k;
            ring_looked[i] = !a.all_wait;
            ring_slot = i;
        }
        // Rounds and tail on the second stream: they need this window's first pass and, by
        // stream order, the previous window's tail.  Nothing here waits for the device.
        GP_HIP(hipStreamWaitEvent(s2, job->a_ev_fp[set], 0));
        const uint32_t rounds_now = std::min<uint32_t>(want_rounds, kMaxRounds - 8);
        t.n_rounds = rounds_now;
        t.n_counts = rounds_now + 2;
        t.ring_slot = ring_slot;
        t.ring_seq = ring_head + 1;
        if (wi == 0 && (rc = job->t_round.begin(s2, 1)) != GP_OK) // one span over all rounds
            return rc;
        for (uint32_t round = 0; round < rounds_now; round++) {
            a.round = round;
            a.tag_cur = tag_max - (g0 + round);
            a.tag_next = a.tag_cur - 1;
            a.resv = t.resv[round & 1];
            a.resv_out = t.resv[(round + 1) & 1];
            a.list_in = t.list[round & 1];
            a.list_out = t.list[(round + 1) & 1];
            const unsigned blocks = last_frac < 0.01 ? 256u : (unsigned)kRoundBlocks;
            hipLaunchKernelGGL(cbf_round, dim3(blocks), dim3(256), 0, s2, a);
        }
        t.tag_first = tag_max - g0;
        hipLaunchKernelGGL(cbf_tail, dim3(1), dim3(kTailThreads), 0, s2, a, t);
        GP_HIP(hipGetLastError());
        if (last_frac < 0.05) {
            // the (batch, k) pairs this window touched
            const uint32_t n_streams = g_n * p.n_k;
            if (n_streams <= 64) {
                hipLaunchKernelGGL(cbf_check_sat, dim3(n_streams), dim3(1024), 0, s2, a,
                                   bcur * p.n_k);
                GP_HIP(hipGetLastError());
            }
        }
        GP_HIP(hipEventRecord(job->a_ev_tail[set], s2));
        ring_head++;
        wi++;
        job->counters.a_rounds += rounds_now + 1;
        gen = g0 + rounds_now + tail_span + 1;
        off += slice;
      }
        g_first += g_n;
    }
    if (wi > 0) {
        int rc = job->t_round.end(s2);
        if (rc != GP_OK)
            return rc;
    }
    if (env_u64("GP_STAGE_A_DEBUG", 0)) {
        const double host_ms = std::chrono::duration<double, std::milli>(
                                   std::chrono::steady_clock::now() - t_host0).count();
        GP_HIP(hipStreamSynchronize(s2));
        const double all_ms = std::chrono::duration<double, std::milli>(
                                  std::chrono::steady_clock::now() - t_host0).count();
        fprintf(stderr, "[stage A] %llu windows: host enqueue %.2f ms, device done after %.2f ms\n",
                (unsigned long long)wi, host_ms, all_ms);
    }
    // the rest of the pipeline (stream `s`) continues behind the last tail
    GP_HIP(hipEventRecord(job->a_ev_init, s2));
    GP_HIP(hipStreamWaitEvent(s, job->a_ev_init, 0));
    unsigned long long st[4];
    uint32_t tail_err = 0;
    GP_HIP(hipMemcpyAsync(st, job->d_stats.p, sizeof st, hipMemcpyDeviceToHost, s));
    GP_HIP(hipMemcpyAsync(&tail_err, job->d_counts.p + (uint64_t)(kMaxRounds + 1) * kCountStride, sizeof tail_err,
                          hipMemcpyDeviceToHost, s));
    GP_HIP(hipStreamSynchronize(s));
    job->counters.a_positions = st[0];
    job->counters.a_ops = st[1];
    job->counters.a_ordered_ops = st[2];
    job->counters.a_rounds += st[3];
    if (tail_err) {
        set_error("stage A: reservation rounds did not converge");
        return GP_ERR_STATE;
    }
    int rc = job->t_first_pass.collect();
    if (rc != GP_OK)
        return rc;
    if ((rc = job->t_round.collect()) != GP_OK)
        return rc;
    job->filters_ready = true;
    return GP_OK;
}

} // namespace gp
