// stage_b.hip -- stage B of the path on MI355X: the ntEdit substitution / indel
// trial sweep for every contig of a job against the batch's Bloom filter for
// one k, plus the cascade guard and the two string transforms around stage C
// (soft mask, upper-case).
//
// What it replaces: kmerizeAndCorrect and its helpers
// (/root/reference/subprojects/ntedit/ntedit.cpp:393-1771) as invoked by
// scripts/goldpolish-ntedit:20-40 (-d5 -i5 -m1 -X -Y -a1), then
// scripts/goldpolish-mask:44-72 and scripts/goldpolish-to-upper:17-21.
//
// ntEdit is a strictly left-to-right state machine: an accepted edit changes
// what every later position sees.  The device version keeps that order exact
// and finds its parallelism inside each step instead:
//   * one 256-thread workgroup per contig; thread 0 owns the edit state (the
//     reference's node list of position runs and inserted characters, kept in
//     HBM; cursors and hashes in LDS),
//   * "scan": when everything right of the tail is untouched contig, the next
//     256 k-mers are hashed and probed at once, one per lane, and the first
//     absent one (ballot + ffs) is the next site -- nothing between is
//     observable, so the walker jumps there,
//   * "site": the confirm count, the 3 substitution trials and, when needed,
//     the 341 insertion strings + deletions run one candidate per lane, each
//     lane rolling its own hash over a shared LDS window and counting filter
//     hits; the winner is picked with the reference's tie rule (later wins),
//   * thresholds are evaluated on the host with the reference's float
//     expressions and handed over as integers.
// Filter probes are 4-byte random reads into 512 KiB (L2-resident); sequence
// traffic is one byte per base per pass.  Integer / bitwise work only.
#include "gp_internal.h"
#include "gp_bc.h"

#include <chrono>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace gp {

constexpr int kWalkThreads = 256;
constexpr int kCacheWords = 64;                 // 4096 contig positions of both bitmaps ...
constexpr int kCacheBases = kCacheWords * 64 + 256; // ... and their bases plus one site window
constexpr int kMatTile = 4096; // bytes per workgroup in the tile-parallel copy kernels
constexpr int kMaxK = 64;
constexpr int kZMax = 2 * kMaxK + 16 + kWalkThreads; // window chars in LDS
constexpr int kGSegMax = 40;                          // position runs of a gathered window copied by all lanes

struct SiteRec;

struct WalkArgs {
    uint8_t* seq;            // current sequences (mutated in place)
    uint8_t* out;            // next sequences
    const uint64_t* off;
    const uint32_t* len_cur;
    uint32_t* len_next;
    uint8_t* dropped;
    Node* nodes;
    const uint64_t* node_off;
    const uint32_t* bf;      // all filters
    const uint32_t* contig_batch;
    uint32_t n_k, ki, k, hash_num, bf_words;
    FastMod bf_mod;
    uint32_t max_ins_tries;  // num_tries[max_insertions]
    uint32_t max_del, jump, min_contig_len, soft_mask;
    uint32_t need_missing, need_edit, need_del; // integer forms of the float tests
    uint32_t insertion_cap;
    const unsigned long long* bm_site; // bit (off[c] + p): k-mer at p is absent from the filter
    const unsigned long long* bm_ok;   // bit (off[c] + p): bases p .. p+k-1 are all accepted
    uint32_t* n_nodes;                 // live nodes per contig after the walk
    SiteRec* site_rec;                 // [slot position] what the trials at a confirmed site decide (or null)
    uint32_t* err;
    unsigned long long* stats;
    unsigned long long* prof;          // development aid (GP_WALK_PROFILE): per contig, 100 MHz ticks and trips per request type
    // long contigs walked as segments (see SegDesc); null: every contig by one workgroup
    const struct SegDesc* segs;
    struct SegOut* seg_out;
    struct SegState* seg_state;        // [descriptor] where a failed segment leaves its walker
    const uint32_t* contig_mode;       // per contig: 0 = walk it whole, 1 = walked as segments (the whole-contig workgroup leaves),
                                       // 2 = a segment failed, walk it whole now (second launch)
    uint32_t walk_phase;               // 0: first launch (whole contigs with mode 0, segments); 1: the rerun of contigs with mode 2
    uint32_t n_prof_contigs;           // GP_WALK_PROFILE: rows of `prof` before the segments' rows
    uint32_t seg_base;                 // first descriptor of this launch of the segment walk
    const uint32_t* seg_order;         // merge_segments: a contig's descriptors, left to right
};

// A LONG CONTIG AS SEGMENTS.  The walk is serial per contig -- an edit changes the k-mers after it --
// and one workgroup per contig leaves a chip of 256 CUs idle when contigs are megabases long
// (SURVEY 8d's configs 3 and 4: 47 s per 100 Mbp, 37 of them in this kernel).  But while everything
// right of the walker's head is untouched contig and the head is past what the last edit touched
// ("clean"), the walker has no memory: what happens from a k-mer start p on depends on p, the
// bases from p on and the site records alone.  So a contig is cut at positions chosen inside long
// stretches where no record says "edit" (pick_boundaries), every segment [p_begin, p_end) is
// walked by a workgroup of its own starting clean at p_begin, and a segment's walk is VALID only
// if it reaches p_end clean without having worked on anything within seg_guard(k) of it -- then it
// arrives exactly as its right neighbour assumed.  A segment that does not (an edit chain running
// into the boundary) fails its contig: the contig's bases are restored from the copy the pass
// keeps in the output slot and the contig is walked whole (second launch).  k-mer starts
// [p_begin, p_end) map to bases [p_begin + k - 1, p_end + k - 1): a k-mer's edits touch its LAST
// base, so neighbouring segments write disjoint bases and the node lists concatenate.
struct SegDesc {
    uint32_t cid;
    uint32_t p_begin, p_end;   // k-mer starts of the segment; p_end == 0xFFFFFFFF: to the end of the contig
    uint32_t node_first;       // first node slot of the segment inside the contig's node region
    uint32_t node_cap;
    uint32_t resume_from;      // 0xFFFFFFFF: start clean at p_begin; else: go on from the state descriptor `resume_from` left
    uint32_t pad;
};
struct SegOut {
    uint32_t failed, n_nodes, out_len, pad;
    unsigned long long st[4];  // scan, sites, probes, edits
};
// the walker of a segment that failed, as it stood: a failed segment is not walked again, it goes ON
// through its right neighbours' stretch (whose bases come back from the pass's copy first)
struct SegState {
    uint32_t hi, ti, hn, tn, nn, char_in, min_tested_ti, clean_from, need_hash, fast_head, confirmed;
    int32_t cur_tested;
    uint64_t fh, rh;
    unsigned long long st[4];
};
// How close to its end a segment's walker may still WORK (anything but sweep over records): the
// base a k-mer start q owns is q + k - 1, the right neighbour writes bases from seg_end + k - 1 on,
// and a site worked out at head h reads bases up to h + 2k + 12 (a deletion up to 10 past the
// tail).  With h + k + 32 < seg_end nothing the walker reads or writes can be the neighbour's;
// scans near the end are shortened to match (see REQ_SCAN's n).
__host__ __device__ inline uint32_t seg_guard(uint32_t k) { return k + 32; }
constexpr uint32_t kSegGuard = kMaxK + 32; // its largest value (host-side spacing of cuts)

__device__ inline uint32_t __reduce_or_sync_compat(uint32_t v)
{
    for (int o = 32; o > 0; o >>= 1)
        v |= (uint32_t)__shfl_xor((int)v, o);
    return v;
}

// ---------------------------------------------------------------- hashing
struct HashTab {
    uint64_t f[5], fk[5], r[5], rk[5]; // seeds (index 4 = 0) plain and rotated by k
};

__device__ inline uint32_t fcode(uint8_t c)
{
    uint8_t b = base_code(c);
    return b == kBadCode ? 4u : b;
}

__device__ inline uint32_t rcode(uint8_t c)
{
    // reverse-strand seed class of a raw byte (reference: table[c & 7]); returns the
    // code of the base whose FORWARD seed it is, or 4 for zero
    switch (c & 7) {
    case 1: return 2; // T
    case 3: return 3; // G
    case 4: return 0; // A
    case 7: return 1; // C
    default: return 4;
    }
}

// What the out-of-line parts of the walk (make_edit, indel_sweep, walk_site) need of the launch's
// arguments, kept in LDS.  Nothing takes the address of the kernel's argument block: while a Ctx held
// `&a` and was handed to those functions by reference, the whole WalkArgs lived in scratch -- every
// lane stored its 330 bytes at kernel entry and every later `a.field` was a per-lane scratch load
// (117 of them in the trip loop: most of the 57 GB the walk fetched per launch, r03 PMC) instead of a
// scalar load from the kernel-argument segment.
struct WalkCfg {
    uint32_t k, hash_num, jump;
    uint32_t max_ins_tries, max_del;
    uint32_t need_missing, need_edit, need_del;
    uint32_t insertion_cap, soft_mask;
    FastMod bf_mod;
    uint32_t* err;
};

struct Best {
    uint32_t type; // 0 none, 1 sub, 2 ins, 3 del
    uint8_t sub_base;
    uint8_t indel[8];
    uint32_t indel_len;
    uint32_t sup;
};

struct Ctx {
    // per-workgroup constants
    uint8_t* seq;       // this contig (mutable)
    Node* nd;
    uint32_t node_cap;
    const uint32_t* bf;
    uint32_t len, k;
};

struct Shared {
    HashTab ht;
    WalkCfg cfg;              // (thread 0 fills cfg and ctx before the first barrier)
    Ctx ctx;                  // the workgroup's contig, for the out-of-line functions
    Best edit_best;           // make_edit's argument
    // walker state (thread 0 writes)
    uint32_t hi, ti, hn, tn, nn;
    uint64_t fh, rh;
    uint32_t char_in;
    int32_t cur_tested;
    uint32_t min_tested_ti;   // k-mers whose tail is left of this are still being skipped
    uint32_t clean_from;      // k-mers starting at or right of this coordinate are untouched contig
    uint32_t bit_from;        // first k-mer start the bitmap search looks at
    uint32_t need_hash;       // the walker jumped: hash of the current k-mer must be rebuilt
    uint32_t fast_head;       // window can be read straight from the contig at S.hi
    uint32_t confirmed;       // the bitmap already did the confirm step for this site
    uint32_t nc_idx[2];       // tiny cache of the node list (thread 0 only): indices ...
    Node nc_node[2];          // ... and copies; dropped whenever the list is edited
    uint32_t nc_next;
    // site window as prefix XORs of rotated seeds: any k-mer hash of the window in O(1)
    uint64_t PF[2 * kMaxK + 16], QR[2 * kMaxK + 16];
    uint64_t wave_f[kWalkThreads / 64], wave_r[kWalkThreads / 64];
    int32_t req[2];           // this trip's request and the next one's (see THE REQUEST PROTOCOL in ntedit_walk)
    uint32_t nz;              // chars available in Z
    uint32_t scan_j0, scan_n; // first k-mer index / number of k-mers of the scan
    uint32_t draft;
    uint32_t n_sub;
    uint8_t subs[4];
    uint32_t sub_in_bf[4], sub_present[4];
    uint32_t confirm_missing, confirm_bad;
    uint32_t num_del;
    uint32_t best_key;        // (support << 16) | order, reduced with atomicMax
    unsigned long long st_scan, st_sites, st_probes, st_edits;
    uint8_t Z[kZMax];
    // what gather_plan leaves for every lane to copy: runs of untouched contig inside the window
    struct GSeg {
        uint32_t z0, pos, cnt;
    } gseg[kGSegMax];
    uint32_t n_gseg;
    // window of the contig the head is in, kept in LDS while everything right of the head is
    // untouched: the two bitmaps thread 0 scans and the bases a site's window is cut from
    unsigned long long c_site[kCacheWords], c_ok[kCacheWords];
    uint64_t cw0;             // first bitmap word in the cache (absolute index)
    uint32_t cwn;             // words in the cache
    uint32_t cp0, cpn;        // contig positions [cp0, cp0 + cpn) held in c_seq
    alignas(16) uint8_t c_seq[kCacheBases];
    // (make_edit's work arrays -- thread 0 only, 1.2 KB of scratch for every lane while they were
    // private -- share the scan's hash arrays: a scan's hashes are consumed before its branch ends,
    // make_edit runs at the end of a site's or a sweep's branch)
    union {
        struct {
            uint64_t scan_fh[kWalkThreads], scan_rh[kWalkThreads];
        };
        struct {
            int32_t me_lps[224];
            uint8_t me_prev[224];
            uint8_t me_kmer[kMaxK];
        };
    };
    uint32_t wave_min[kWalkThreads / 64];
    unsigned long long prof_t[8], prof_n[8], prof_last;
    int32_t prof_req;
    uint32_t seg_failed;      // a segment's walk worked on something too close to its end (1, 2) or gave up (3)
    uint32_t seg_busy;        // consecutive decisions of a segment's walker that were not "clean"
    uint32_t seg_arrived;     // the segment's sweep reached its end
    uint32_t seg_refused;     // make_edit refused a repeat insertion (and dropped nodes of the list)
};

// The workgroup's LDS block lives at namespace scope, so that the out-of-line device functions below
// address it as LDS (ds_read / ds_write with constant offsets) instead of through a generic reference
// (flat loads) -- ntedit_walk and ntedit_site_eval both work in it.
__shared__ Shared g_S;

enum { REQ_DONE = 0, REQ_SCAN, REQ_SITE, REQ_NEXT, REQ_BITSCAN, REQ_REFILL, REQ_SWEEP };

// What the trials at one confirmed site decide, worked out ahead of the walk for every site of
// the untouched contig (ntedit_site_eval); the walk only consumes it (16 bytes per slot position).
struct SiteRec {
    uint8_t type;      // Best::type: 0 nothing, 1 substitution, 2 insertion, 3 deletion
    uint8_t sub_base;
    uint8_t indel_len;
    uint8_t flags;     // bit 0: record valid; bit 1: some substitute's k-mer was in the filter (trial + revert happened)
    uint8_t indel[8];
    uint32_t probes;   // filter look-ups the trials made
};

__device__ inline uint64_t roll_f(const HashTab& t, uint64_t fh, uint8_t out, uint8_t in)
{
    return srol1(fh) ^ t.f[fcode(in)] ^ t.fk[fcode(out)];
}

__device__ inline uint64_t roll_r(const HashTab& t, uint64_t rh, uint8_t out, uint8_t in)
{
    return sror1(rh ^ t.rk[rcode(in)] ^ t.r[rcode(out)]);
}

__device__ inline void changelast(const HashTab& t, uint64_t& fh, uint64_t& rh, uint8_t out,
                                  uint8_t in)
{
    fh ^= t.f[fcode(out)] ^ t.f[fcode(in)];
    rh = sror1(srol1(rh) ^ t.rk[rcode(out)] ^ t.rk[rcode(in)]);
}

__device__ inline void hash_kmer(const HashTab& t, const uint8_t* z, uint32_t k, uint64_t& fh,
                                 uint64_t& rh)
{
    uint64_t f = 0, r = 0;
    for (uint32_t i = 0; i < k; i++) {
        f = srol1(f) ^ t.f[fcode(z[i])];
        r = srol1(r) ^ t.r[rcode(z[k - 1 - i])];
    }
    fh = f;
    rh = r;
}

template <typename A> // WalkArgs (kernel bodies: scalar loads from the argument segment) or WalkCfg (LDS)
__device__ inline bool bf_contains(const A& a, const uint32_t* bf, uint64_t base)
{
    bool all = true;
    for (uint32_t i = 0; i < a.hash_num; i++) {
        uint64_t h = i ? ext_hash(base, a.k, i) : base;
        uint64_t n = fastmod(h, a.bf_mod);
        all &= ((bf[n >> 5] >> (uint32_t)(n & 31)) & 1u) != 0;
    }
    return all;
}

__device__ inline bool is_accepted_u(uint8_t c)
{
    uint8_t u = (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c;
    switch (u) {
    case 'A': case 'T': case 'G': case 'C': case 'R': case 'Y': case 'S':
    case 'W': case 'K': case 'M': case 'B': case 'D': case 'H': case 'V':
        return true;
    default:
        return false;
    }
}

__device__ inline uint8_t to_upper(uint8_t c) { return (c >= 'a' && c <= 'z') ? (uint8_t)(c - 32) : c; }
__device__ inline uint8_t to_lower(uint8_t c) { return (c >= 'A' && c <= 'Z') ? (uint8_t)(c + 32) : c; }

// ------------------------------------------------ node list (thread 0 only)
__device__ inline Node nd_get(const Ctx& c, const Shared& S, uint32_t i)
{
    if (i < S.nn)
        return c.nd[i];
    Node u;
    u.type = -1;
    u.s = u.e = u.c = 0;
    return u;
}

__device__ inline void nd_put(const Ctx& c, Shared& S, uint32_t i, Node n)
{
    if (i < S.nn) {
        c.nd[i] = n;
    } else if (S.nn < c.node_cap) {
        c.nd[S.nn++] = n;
    } else {
        atomicOr(S.cfg.err, 1u); // node list overflow
    }
}

// thread 0's view of the node list through a two-entry cache (head node, tail node)
__device__ inline Node nd_cached(const Ctx& c, Shared& S, uint32_t i)
{
    if (S.nc_idx[0] == i)
        return S.nc_node[0];
    if (S.nc_idx[1] == i)
        return S.nc_node[1];
    Node n = nd_get(c, S, i);
    const uint32_t slot = S.nc_next & 1u;
    S.nc_next++;
    S.nc_idx[slot] = i;
    S.nc_node[slot] = n;
    return n;
}

__device__ inline void nd_cache_drop(Shared& S) { S.nc_idx[0] = S.nc_idx[1] = 0xFFFFFFFEu; }

// thread 0's writes into the contig go to memory and to the LDS copy of that stretch
__device__ inline void seq_store(const Ctx& c, Shared& S, uint32_t pos, uint8_t v)
{
    c.seq[pos] = v;
    if (pos - S.cp0 < S.cpn)
        S.c_seq[pos - S.cp0] = v;
}

// inverse of sroln
__device__ inline uint64_t srorn(uint64_t x, uint32_t n)
{
    uint64_t hi = x >> 33, lo = x & 0x1FFFFFFFFULL;
    const uint32_t a = n % 31u, b = n % 33u;
    hi = ((hi >> a) | (hi << (31u - a))) & 0x7FFFFFFFULL;
    lo = ((lo >> b) | (lo << (33u - b))) & 0x1FFFFFFFFULL;
    return (hi << 33) | lo;
}

// hashes of the k-mer Z[a .. a+k) from the prefix arrays
__device__ inline uint64_t win_f(const Shared& S, uint32_t a, uint32_t k)
{
    return sroln(S.PF[a + k] ^ S.PF[a], a + k - 1);
}

__device__ inline uint64_t win_r(const Shared& S, uint32_t a, uint32_t k)
{
    return srorn(S.QR[a + k] ^ S.QR[a], a);
}

__device__ inline uint8_t get_char(const Ctx& c, uint32_t pos, const Node& n)
{
    if (n.type == 0)
        return pos < c.len ? c.seq[pos] : 0;
    if (n.type == 1)
        return (uint8_t)n.c;
    return 0;
}

__device__ inline void increment(const Ctx& c, const Shared& S, uint32_t& pos, uint32_t& ni)
{
    Node n = nd_get(c, S, ni);
    if (n.type == 0) {
        pos++;
        if (pos > n.e) {
            ni++;
            if (ni < S.nn && c.nd[ni].type == 0)
                pos = c.nd[ni].s;
        }
    } else if (n.type == 1) {
        ni++;
        if (ni < S.nn && c.nd[ni].type == 0)
            pos = c.nd[ni].s;
    }
}

__device__ inline bool roll_cursor(const Ctx& c, const Shared& S, uint32_t& hi, uint32_t& ti,
                                   uint32_t& hn, uint32_t& tn, uint8_t& out, uint8_t& in)
{
    if (hi >= c.len || hn >= S.nn)
        return false;
    out = get_char(c, hi, nd_get(c, S, hn));
    increment(c, S, hi, hn);
    if (ti >= c.len || tn >= S.nn)
        return false;
    increment(c, S, ti, tn);
    if (ti >= c.len || tn >= S.nn)
        return false;
    in = get_char(c, ti, nd_get(c, S, tn));
    return true;
}

__device__ void make_insertion(const Ctx& c, Shared& S, uint32_t& tn, int insert_pos,
                               const uint8_t* ins, uint32_t nins)
{
    Node orig = c.nd[tn];
    if (orig.type == 0 && !(insert_pos <= (int)orig.s)) {
        Node after;
        after.type = 0;
        after.s = (uint32_t)insert_pos;
        after.e = orig.e;
        after.c = 0;
        c.nd[tn].e = (uint32_t)insert_pos - 1;
        for (uint32_t i = 0; i < nins; i++) {
            Node x;
            x.type = 1;
            x.s = x.e = 0;
            x.c = ins[i];
            nd_put(c, S, tn + i + 1, x);
        }
        nd_put(c, S, tn + nins + 1, after);
        tn++;
        return;
    }
    if (orig.type == 0 || orig.type == 1) {
        // the live tail (a handful of nodes: edits happen at the right edge) moves right
        uint32_t nre = 0;
        while (tn + nre < S.nn && c.nd[tn + nre].type != -1)
            nre++;
        const uint32_t need = tn + nins + nre;
        if (need > c.node_cap) {
            atomicOr(S.cfg.err, 1u);
            return;
        }
        while (S.nn < need)
            c.nd[S.nn++].type = -1;
        for (uint32_t q = nre; q-- > 0;)
            c.nd[tn + nins + q] = c.nd[tn + q];
        for (uint32_t q = 0; q < nins; q++) {
            Node x;
            x.type = 1;
            x.s = x.e = 0;
            x.c = ins[q];
            c.nd[tn + q] = x;
        }
    }
}

__device__ void make_deletion(const Ctx& c, Shared& S, uint32_t& tn, uint32_t& pos,
                              uint32_t num_del)
{
    for (int depth = 0; depth < 64; depth++) { // the reference recurses; bounded here
        Node orig = nd_get(c, S, tn);
        uint32_t leftover = 0;
        if (orig.type == 0) {
            if (pos <= orig.s) {
                if ((uint32_t)(pos + num_del) <= orig.e) {
                    c.nd[tn].s = pos + num_del;
                    pos = c.nd[tn].s;
                    return;
                }
                leftover = (uint32_t)((uint64_t)(uint32_t)(pos + num_del) - (uint64_t)orig.e);
                pos = orig.e + 1;
                uint32_t i = tn + 1;
                while (i < S.nn && c.nd[i].type != -1) {
                    c.nd[i - 1] = c.nd[i];
                    c.nd[i].type = -1;
                    i++;
                }
            } else {
                if ((uint32_t)(pos + num_del) <= orig.e) {
                    Node split;
                    split.type = 0;
                    split.s = pos + num_del;
                    split.e = orig.e;
                    split.c = 0;
                    c.nd[tn].e = pos - 1;
                    pos = split.s;
                    tn++;
                    nd_put(c, S, tn, split);
                    return;
                }
                leftover = (uint32_t)((uint64_t)(uint32_t)(pos + num_del) - (uint64_t)orig.e);
                c.nd[tn].e = pos - 1;
                pos = orig.e + 1;
                tn++;
            }
        } else if (orig.type == 1) {
            uint32_t i = tn;
            leftover = num_del;
            while (i < S.nn && c.nd[i].type == 1 && leftover > 0) {
                c.nd[i].type = -1;
                leftover--;
                i++;
            }
            uint32_t j = tn;
            while (i < S.nn && c.nd[i].type != -1) {
                c.nd[j] = c.nd[i];
                c.nd[i].type = -1;
                i++;
                j++;
            }
        } else {
            return;
        }
        if (leftover > 0 && tn < S.nn && c.nd[tn].type != -1) {
            if (c.nd[tn].type == 0)
                pos = c.nd[tn].s;
            num_del = leftover;
            continue;
        }
        return;
    }
}

__device__ bool find_accepted_kmer(const Ctx& c, Shared& S, uint32_t& hi, uint32_t& ti,
                                   uint32_t& hn, uint32_t& tn, uint8_t* kmer)
{
    const uint32_t k = c.k;
    Node cur = nd_get(c, S, tn);
    uint32_t ttn = tn, thn = 0;
    uint32_t i = ti;
    while (i < c.len && ttn < S.nn && c.nd[ttn].type != -1) {
        uint8_t ch = get_char(c, i, cur);
        if (is_accepted_u(ch)) {
            uint32_t n = 0;
            kmer[n++] = ch;
            thn = ttn;
            uint32_t j = i;
            increment(c, S, j, ttn);
            while (j < c.len && ttn < S.nn && c.nd[ttn].type != -1) {
                cur = c.nd[ttn];
                ch = get_char(c, j, cur);
                if (!is_accepted_u(ch)) {
                    i = j;
                    break;
                }
                kmer[n++] = ch;
                if (n == k)
                    break;
                increment(c, S, j, ttn);
            }
            if (n == k) {
                hi = i;
                ti = j;
                hn = thn;
                tn = ttn;
                return true;
            }
        }
        increment(c, S, i, ttn);
    }
    hi = c.len;
    ti = c.len;
    return false;
}

__device__ inline uint8_t rc_char(uint8_t ch)
{
    switch (ch) {
    case 'A': case 'a': return 'T';
    case 'T': case 't': return 'A';
    case 'G': case 'g': return 'C';
    case 'C': case 'c': return 'G';
    default: return 'N';
    }
}

__device__ bool is_repeat_insertion(const uint8_t* s, int n, int* lps)
{
    if (n <= 0)
        return false;
    int len = 0, i = 1;
    lps[0] = 0;
    while (i < n) {
        if (s[i] == s[len]) {
            lps[i++] = ++len;
        } else if (len != 0) {
            len = lps[len - 1];
        } else {
            lps[i++] = 0;
        }
    }
    len = lps[n - 1];
    return len > 0 && n % (n - len) == 0;
}

__device__ void drop_repeat_nodes(const Ctx& c, Shared& S, uint32_t tn, uint32_t ti,
                                  uint32_t count)
{
    uint32_t j = 1;
    if (c.nd[tn].type == 0 && ti == c.nd[tn].s)
        j = 0;
    for (uint32_t i = count; i > 0; i--) {
        if (i > tn)
            continue;
        if (tn + j < S.nn && c.nd[tn + j].type != -1) {
            c.nd[tn - i] = c.nd[tn + j];
            c.nd[tn + j].type = -1;
            j++;
        } else {
            c.nd[tn - i].type = -1;
        }
    }
}

// multi_possible_bases[first][i]: shortest first, then A<C<G<T
__device__ inline uint32_t indel_string(uint8_t first, uint32_t i, uint8_t* out)
{
    uint32_t L = 1, start = 0, next = 1; // counts 1,5,21,85,341
    while (i >= next) {
        L++;
        start = next;
        next = next * 4 + 1;
    }
    uint32_t j = i - start;
    out[0] = first;
    for (uint32_t q = L - 1; q >= 1; q--) {
        out[q] = (uint8_t)("ACGT"[j & 3]);
        j >>= 2;
    }
    return L;
}

// thread 0 only; the edit to make is S.edit_best.  Out of line on purpose: it is the rare end of a
// site's or a sweep's branch, and inlined its node-list surgery sat in the trip loop's register budget.
__device__ __attribute__((noinline)) void make_edit(uint8_t draft)
{
    Shared& S = g_S;
    const Ctx& c = S.ctx;
    Best& b = S.edit_best;
    const HashTab& ht = S.ht;
    const uint32_t k = c.k;
    Node tnode = nd_cached(c, S, S.tn);
    if (b.type == 2 || b.type == 3 || tnode.type != 0)
        nd_cache_drop(S); // the node list is about to change
    switch (b.type) {
    case 1:
        if (tnode.type == 0)
            seq_store(c, S, S.ti, b.sub_base);
        else if (tnode.type == 1)
            c.nd[S.tn].c = b.sub_base;
        changelast(ht, S.fh, S.rh, draft, b.sub_base);
        S.st_edits++;
        S.clean_from = max(S.clean_from, tnode.type == 0 ? S.ti + 1 : S.ti);
        break;
    case 2: {
        bool skipped = false;
        // run of inserted chars just left of the tail, nearest first, complemented
        uint8_t* const prev = S.me_prev;
        int* const lps = S.me_lps;
        uint32_t np = 0;
        {
            uint32_t tn = S.tn;
            if ((tn < S.nn && c.nd[tn].type == 0 && S.ti == c.nd[tn].s) || c.nd[tn].type == 1)
                tn--;
            while (tn < S.nn && c.nd[tn].type == 1) {
                if (np < 200)
                    prev[np] = rc_char((uint8_t)c.nd[tn].c);
                np++;
                tn--;
            }
            if (np > 200) {
                atomicOr(S.cfg.err, 2u);
                np = 200;
            }
        }
        if (np + b.indel_len >= k) {
            uint8_t* const kmer = S.me_kmer;
            if (is_repeat_insertion(prev, (int)np, lps) ||
                np + b.indel_len >= S.cfg.insertion_cap) {
                drop_repeat_nodes(c, S, S.tn, S.ti, np);
                if (find_accepted_kmer(c, S, S.hi, S.ti, S.hn, S.tn, kmer))
                    hash_kmer(ht, kmer, k, S.fh, S.rh);
                skipped = true;
            } else {
                for (uint32_t w = 0; w < b.indel_len; w++) {
                    for (uint32_t q = np; q > 0; q--)
                        prev[q] = prev[q - 1];
                    prev[0] = rc_char(b.indel[w]);
                    np++;
                    if (is_repeat_insertion(prev, (int)np, lps)) {
                        drop_repeat_nodes(c, S, S.tn, S.ti, np - w);
                        if (find_accepted_kmer(c, S, S.hi, S.ti, S.hn, S.tn, kmer))
                            hash_kmer(ht, kmer, k, S.fh, S.rh);
                        skipped = true;
                    }
                }
            }
        }
        if (skipped)
            S.seg_refused = 1; // (see the segment walk: a refused insertion sends the contig to the one-workgroup walk)
        if (!skipped) {
            make_insertion(c, S, S.tn, (int)S.ti, b.indel, b.indel_len);
            changelast(ht, S.fh, S.rh, draft, b.indel[0]);
            S.st_edits++;
            S.clean_from = max(S.clean_from, S.ti); // k-mers from the draft base on are untouched
        } else {
            // The insertion was refused (it would have repeated what was inserted just before) and
            // the walker moved on to the next accepted k-mer, [hi, ti], which it does not test
            // (the reference's loop rolls on from it).  That k-mer starts at or right of the old
            // tail, the refused insertion wrote nothing there, and drop_repeat_nodes only took
            // nodes left of it: k-mers from hi + 1 on are untouched contig again, the records hold.
            // (Rounds 1-2 gave up here -- clean_from = len -- and worked every later site of the
            // contig out on the spot: on a near-saturated filter, where such refusals are common,
            // that was most of a long contig's walk.)
            S.clean_from = max(S.clean_from, S.hi < c.len ? S.hi + 1 : c.len);
        }
        break;
    }
    case 3:
        make_deletion(c, S, S.tn, S.ti, b.indel_len);
        changelast(ht, S.fh, S.rh, draft, get_char(c, S.ti, nd_get(c, S, S.tn)));
        S.st_edits++;
        // the base pulled in as the new last one was never looked at by the skip logic
        S.clean_from = max(S.clean_from, S.ti + 1);
        break;
    default:
        if (S.cfg.soft_mask) {
            uint8_t lo = to_lower(draft);
            if (tnode.type == 0)
                seq_store(c, S, S.ti, lo);
            else if (tnode.type == 1)
                c.nd[S.tn].c = lo;
            changelast(ht, S.fh, S.rh, draft, lo);
        }
        break;
    }
}

// Up to `want` chars of the edited sequence starting at the head cursor, a NODE at a time
// (thread 0): inserted bases go straight into z, runs of untouched contig are left as segments for
// every lane to copy (gather_fill, after a barrier).  Char by char -- a node look-up and a
// sequence byte from memory, one behind the other, for each of 2k + 12 chars -- this was most of
// the 89 us of a site worked out on the spot (head among edited nodes) and of the walk.  Returns
// the number of chars, exactly as the char-by-char loop of get_char / increment would.
__device__ uint32_t gather_plan(const Ctx& c, Shared& S, uint8_t* z, uint32_t want)
{
    uint32_t pos = S.hi, ni = S.hn, n = 0, ns = 0;
    while (n < want) {
        if (pos >= c.len || ni >= S.nn)
            break;
        const Node nd = c.nd[ni];
        if (nd.type == 0) {
            // as many chars as the run still holds (one if the cursor stands past its end: the
            // reference reads the base under the cursor before it looks at the run)
            uint32_t take = pos <= nd.e ? nd.e - pos + 1 : 1u;
            take = min(take, min(want - n, c.len - pos));
            if (ns < (uint32_t)kGSegMax) {
                S.gseg[ns].z0 = n;
                S.gseg[ns].pos = pos;
                S.gseg[ns].cnt = take;
                ns++;
            } else {
                for (uint32_t i = 0; i < take; i++)
                    z[n + i] = c.seq[pos + i];
            }
            n += take;
            pos += take;
            if (pos > nd.e) { // the cursor left the run: where increment() would have put it
                ni++;
                if (ni < S.nn && c.nd[ni].type == 0)
                    pos = c.nd[ni].s;
            }
        } else if (nd.type == 1) {
            z[n++] = (uint8_t)nd.c;
            ni++;
            if (ni < S.nn && c.nd[ni].type == 0)
                pos = c.nd[ni].s;
        } else {
            // the reference keeps reading a 0 byte from an unset node without advancing
            while (n < want)
                z[n++] = 0;
            break;
        }
    }
    S.n_gseg = ns;
    return n;
}

// every lane, after the barrier behind gather_plan; a barrier must follow
__device__ inline void gather_fill(const Ctx& c, Shared& S, uint8_t* z)
{
    const uint32_t ns = S.n_gseg;
    for (uint32_t g = 0; g < ns; g++) {
        const uint32_t z0 = S.gseg[g].z0, pos = S.gseg[g].pos, cnt = S.gseg[g].cnt;
        for (uint32_t i = threadIdx.x; i < cnt; i += blockDim.x)
            z[z0 + i] = c.seq[pos + i];
    }
}

// `steps` times increment(), a node at a time
__device__ void advance(const Ctx& c, const Shared& S, uint32_t& pos, uint32_t& ni, uint32_t steps)
{
    while (steps) {
        const Node n = nd_get(c, S, ni);
        if (n.type == 0) {
            if (pos < n.e) { // room inside the run
                const uint32_t m = min(steps, n.e - pos);
                pos += m;
                steps -= m;
                if (!steps)
                    break;
            }
            pos++; // this step leaves the run
            steps--;
            ni++;
            if (ni < S.nn && c.nd[ni].type == 0)
                pos = c.nd[ni].s;
        } else if (n.type == 1) {
            ni++;
            steps--;
            if (ni < S.nn && c.nd[ni].type == 0)
                pos = c.nd[ni].s;
        } else {
            break; // an unset node: increment() stands still however often it is called
        }
    }
}

// ------------------------------------------------------------- the kernel
__device__ inline uint32_t block_min(Shared& S, uint32_t v)
{
    // v != ~0u marks a hit at index v; returns the smallest, ~0u if none
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t m = __ballot(v != 0xFFFFFFFFu);
    uint32_t first = m ? (uint32_t)__shfl((int)v, __ffsll((long long)m) - 1) : 0xFFFFFFFFu;
    // hits are index-ordered by lane, so the first set lane of a wave is its minimum
    if (lane == 0)
        S.wave_min[wave] = first;
    __syncthreads();
    uint32_t r = 0xFFFFFFFFu;
    for (int w = 0; w < (int)(blockDim.x / 64); w++)
        r = min(r, S.wave_min[w]);
    __syncthreads();
    return r;
}

__device__ inline uint32_t block_sum(Shared& S, uint32_t v)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = 32; o > 0; o >>= 1)
        v += __shfl_down((int)v, o);
    if (lane == 0)
        S.wave_min[wave] = v;
    __syncthreads();
    uint32_t r = 0;
    for (int w = 0; w < (int)(blockDim.x / 64); w++)
        r += S.wave_min[w];
    __syncthreads();
    return r;
}

// insertion / deletion sweep for one index char; every thread takes candidates
// t, t+256, ...; result: S.best_key (0 = nothing passed)
__device__ __forceinline__ void indel_sweep_body(uint8_t index_char, uint32_t del_from)
{
    Shared& S = g_S;
    const Ctx& c = S.ctx;
    const WalkCfg& a = S.cfg;
    const HashTab& ht = S.ht;
    const uint32_t k = c.k, jump = a.jump;
    const uint8_t draft = (uint8_t)S.draft;
    const uint32_t n_ins = a.max_ins_tries;
    // deletions num_del = del_from .. max_del are tried at iterations 0.. of the loop
    const uint32_t n_del = (n_ins > 0 && del_from <= a.max_del)
                               ? min(a.max_del - del_from + 1, n_ins)
                               : 0u;
    uint32_t probes = 0;
    uint32_t my_best = 0;
    for (uint32_t cand = threadIdx.x; cand < n_ins + n_del; cand += blockDim.x) {
        uint64_t fh = S.fh, rh = S.rh;
        uint32_t present = 0, order;
        bool pass;
        if (cand < n_ins) {
            uint8_t ins[8];
            const uint32_t L = indel_string(index_char, cand, ins);
            ins[L] = draft;
            changelast(ht, fh, rh, draft, index_char);
            // X = Z[0..k-1) ++ ins[0..L] ; then Z[k..): k-mer q ends at X[q+k]
            for (uint32_t q = 0; q + 1 < k; q++) {
                uint8_t in;
                if (q < L) {
                    in = ins[q + 1];
                } else {
                    uint32_t zi = k + (q - L);
                    if (zi >= S.nz)
                        break;
                    in = S.Z[zi];
                }
                fh = roll_f(ht, fh, S.Z[q], in);
                rh = roll_r(ht, rh, S.Z[q], in);
                if (q % jump == 0) {
                    probes++;
                    present += bf_contains(a, c.bf, fh + rh) ? 1u : 0u;
                }
            }
            pass = present >= a.need_edit;
            order = 2 * cand;
        } else {
            const uint32_t it = cand - n_ins;
            const uint32_t n = del_from + it;
            // X' = Z[0..k-1) ++ Z[k-1+n ..)
            uint8_t last = (k - 1 + n) < S.nz ? S.Z[k - 1 + n] : 0;
            changelast(ht, fh, rh, draft, last);
            probes++;
            present += bf_contains(a, c.bf, fh + rh) ? 1u : 0u;
            for (uint32_t q = 1; q + 2 <= k; q++) {
                uint32_t zi = k - 1 + n + q;
                if (zi >= S.nz)
                    break;
                // chars leaving: Z[q-1]; but once q-1 reaches k-1 the window has no
                // original prefix left (q <= k-2, so it never does)
                fh = roll_f(ht, fh, S.Z[q - 1], S.Z[zi]);
                rh = roll_r(ht, rh, S.Z[q - 1], S.Z[zi]);
                if (q % jump == 0) {
                    probes++;
                    present += bf_contains(a, c.bf, fh + rh) ? 1u : 0u;
                }
            }
            pass = present >= a.need_del;
            order = 2 * it + 1;
        }
        if (pass) {
            uint32_t key = (present << 16) | order;
            my_best = max(my_best, key);
        }
    }
    if (my_best)
        atomicMax(&S.best_key, my_best);
    atomicAdd(&S.st_probes, (unsigned long long)probes);
}

// thread 0, before the workgroup's first barrier
__device__ inline void fill_cfg(Shared& S, const WalkArgs& a, const Ctx& c)
{
    S.cfg.k = a.k;
    S.cfg.hash_num = a.hash_num;
    S.cfg.jump = a.jump;
    S.cfg.max_ins_tries = a.max_ins_tries;
    S.cfg.max_del = a.max_del;
    S.cfg.need_missing = a.need_missing;
    S.cfg.need_edit = a.need_edit;
    S.cfg.need_del = a.need_del;
    S.cfg.insertion_cap = a.insertion_cap;
    S.cfg.soft_mask = a.soft_mask;
    S.cfg.bf_mod = a.bf_mod;
    S.cfg.err = a.err;
    S.ctx = c;
}

// out of line for the walk (walk_site calls it up to three times); ntedit_site_eval inlines the body
__device__ __attribute__((noinline)) void indel_sweep(uint8_t index_char, uint32_t del_from)
{
    indel_sweep_body(index_char, del_from);
}

// ---- every site of the untouched contig, ahead of the walk.
// While everything right of the walker's head is untouched contig ("clean"), what happens at a
// confirmed site -- which substitute k-mers are in the filter, which substitution / insertion /
// deletion wins -- depends on nothing but the 2k+12 bases from the site on and the filter.  Only
// about one site in a hundred ends in an edit, so instead of making the serial walk stop at each
// one (ten barrier-separated phases, three dependent memory round trips), every workgroup of
// this kernel takes 256 positions, finds the confirmed sites among them from the two bitmaps
// exactly as the walk does, runs the trials of each (same code path as REQ_SITE: prefix-XOR
// window hashes, one substitute per wave, the insertion / deletion sweep) and leaves a 16-byte
// record.  The walk then only reads records; sites it reaches in a stretch that an edit has
// touched are still worked out on the spot.  Case changes that trials leave behind (a reverted
// trial writes the draft base back in upper case, soft masking lower-cases it) do not change
// any hash, so records stay valid.
constexpr int kEvalTile = 256;

__global__ __launch_bounds__(kWalkThreads, 6) void ntedit_site_eval(WalkArgs a, const uint32_t* tile_contig,
                                                                const uint32_t* tile_first)
{
    Shared& S = g_S;
    __shared__ uint32_t s_cand[kEvalTile], s_n;
    const uint32_t tid = threadIdx.x;
    // the tile tables are per 4096-position tile (kMatTile); this kernel cuts them in 16
    const uint32_t unit = xcd_contiguous(blockIdx.x, gridDim.x); // a contig's tiles probe one filter: one XCD, one L2
    const uint32_t eval_tile = blockDim.x; // positions per workgroup = its threads (256, or 128: see stage_b_pass)
    const uint32_t big = unit / (kMatTile / eval_tile), sub = unit % (kMatTile / eval_tile);
    const uint32_t cid = tile_contig[big];
    if (a.dropped[cid])
        return;
    const uint32_t len = a.len_cur[cid];
    const uint32_t k = a.k, jump = a.jump;
    if (len < a.min_contig_len || len < 2 * k)
        return;
    const uint32_t p0 = (big - tile_first[cid]) * kMatTile + sub * eval_tile;
    if (p0 + k > len)
        return;
    const uint64_t base_bit = a.off[cid];
    Ctx c;
    c.seq = a.seq + a.off[cid];
    c.nd = nullptr;
    c.node_cap = 0;
    c.bf = a.bf + ((uint64_t)a.contig_batch[cid] * a.n_k + a.ki) * a.bf_words;
    c.len = len;
    c.k = k;
    if (tid == 0) {
        fill_cfg(S, a, c);
        for (uint32_t code = 0; code < 5; code++) {
            uint64_t sd = code < 4 ? seed_of_code(code) : 0ULL;
            S.ht.f[code] = sd;
            S.ht.r[code] = sd;
            S.ht.fk[code] = sroln(sd, k);
            S.ht.rk[code] = sroln(sd, k);
        }
        s_n = 0;
        S.st_probes = 0;
    }
    __syncthreads();
    {
        // confirmed site at p (ntedit.cpp:1487-1523, from the bitmaps as the walk does it)
        const uint32_t p = p0 + tid;
        bool cand = false;
        if (p + k <= len) {
            const uint64_t bit = base_bit + p;
            cand = ((a.bm_site[bit >> 6] & a.bm_ok[bit >> 6]) >> (bit & 63)) & 1ULL;
            cand = cand && (uint64_t)p + 2ull * k <= len;
            if (cand) {
                const uint64_t b2 = bit + k;
                cand = (a.bm_ok[b2 >> 6] >> (b2 & 63)) & 1ULL;
            }
            if (cand) {
                uint32_t missing = 0;
                for (uint32_t q = 0; q < k; q += jump) {
                    const uint64_t b3 = bit + 1 + q;
                    missing += (uint32_t)((a.bm_site[b3 >> 6] >> (b3 & 63)) & 1ULL);
                }
                cand = missing >= a.need_missing;
            }
        }
        if (cand) {
            // first look, one lane per site: is any substitute's k-mer in the filter at all?  If
            // not, the trials end right there (no look-up is counted, nothing is edited, no trial
            // is reverted) -- the fate of nearly every site when the filter is not crowded.
            uint64_t fh, rh;
            hash_kmer(S.ht, c.seq + p, k, fh, rh);
            const uint8_t draft = to_upper(c.seq[p + k - 1]);
            const char* sb;
            switch (draft) {
            case 'A': sb = "TCG"; break;
            case 'T': sb = "ACG"; break;
            case 'C': sb = "ATG"; break;
            case 'G': sb = "ATC"; break;
            case 'R': sb = "TC"; break;
            case 'Y': sb = "AG"; break;
            case 'S': sb = "AT"; break;
            case 'W': sb = "CG"; break;
            case 'K': sb = "AC"; break;
            case 'M': sb = "TG"; break;
            case 'B': sb = "A"; break;
            case 'D': sb = "C"; break;
            case 'H': sb = "G"; break;
            case 'V': sb = "T"; break;
            default: sb = ""; break;
            }
            bool any = false;
            for (uint32_t i = 0; sb[i]; i++) {
                uint64_t f2 = fh, r2 = rh;
                changelast(S.ht, f2, r2, draft, (uint8_t)sb[i]);
                any |= bf_contains(a, c.bf, f2 + r2);
            }
            if (any) {
                s_cand[atomicAdd(&s_n, 1u)] = p; // the full trials below
            } else {
                SiteRec r;
                r.type = 0;
                r.sub_base = 0;
                r.indel_len = 0;
                r.flags = 1;
                for (int i = 0; i < 8; i++)
                    r.indel[i] = 0;
                r.probes = 0;
                a.site_rec[base_bit + p] = r;
            }
        }
    }
    __syncthreads();
    const uint32_t n_cand = s_n;
    for (uint32_t ci = 0; ci < n_cand; ci++) {
        const uint32_t cand = s_cand[ci];
        __syncthreads();
        if (tid == 0) {
            S.hi = cand;
            S.ti = cand + k - 1;
            S.nz = min(2 * k + 12, len - cand);
            S.best_key = 0;
            S.st_probes = 0;
        }
        __syncthreads();
        for (uint32_t i = tid; i < S.nz; i += blockDim.x)
            S.Z[i] = c.seq[cand + i];
        __syncthreads();
        {
            const uint32_t nzz = S.nz;
            uint64_t tf = 0, tr = 0;
            if (tid < nzz) {
                tf = srorn(S.ht.f[fcode(S.Z[tid])], tid);
                tr = sroln(S.ht.r[rcode(S.Z[tid])], tid);
            }
            const int lane = tid & 63, wave = tid >> 6;
            for (int o = 1; o < 64; o <<= 1) {
                const uint64_t uf = __shfl_up(tf, o), ur = __shfl_up(tr, o);
                if (lane >= o) {
                    tf ^= uf;
                    tr ^= ur;
                }
            }
            if (lane == 63) {
                S.wave_f[wave] = tf;
                S.wave_r[wave] = tr;
            }
            __syncthreads();
            uint64_t pf = 0, pr = 0;
            for (int w = 0; w < wave; w++) {
                pf ^= S.wave_f[w];
                pr ^= S.wave_r[w];
            }
            if (tid < nzz) {
                S.PF[tid + 1] = tf ^ pf;
                S.QR[tid + 1] = tr ^ pr;
            }
            if (tid == 0) {
                S.PF[0] = 0;
                S.QR[0] = 0;
            }
            __syncthreads();
            if (tid == 0) {
                S.fh = win_f(S, 0, k);
                S.rh = win_r(S, 0, k);
                S.char_in = S.Z[k - 1];
                S.draft = to_upper(S.Z[k - 1]);
                const char* sb;
                switch ((uint8_t)S.draft) {
                case 'A': sb = "TCG"; break;
                case 'T': sb = "ACG"; break;
                case 'C': sb = "ATG"; break;
                case 'G': sb = "ATC"; break;
                case 'R': sb = "TC"; break;
                case 'Y': sb = "AG"; break;
                case 'S': sb = "AT"; break;
                case 'W': sb = "CG"; break;
                case 'K': sb = "AC"; break;
                case 'M': sb = "TG"; break;
                case 'B': sb = "A"; break;
                case 'D': sb = "C"; break;
                case 'H': sb = "G"; break;
                case 'V': sb = "T"; break;
                default: sb = ""; break;
                }
                uint32_t n = 0;
                for (; sb[n]; n++)
                    S.subs[n] = (uint8_t)sb[n];
                S.n_sub = n;
                for (uint32_t i = 0; i < 4; i++)
                    S.sub_in_bf[i] = S.sub_present[i] = 0;
                S.num_del = 1;
            }
            __syncthreads();
        }
        {
            const uint32_t q = tid & 63;
            for (uint32_t si = tid >> 6; si < S.n_sub; si += blockDim.x / 64) { // (a wave per substitute; two waves take two turns)
                const uint8_t sb = S.subs[si];
                if (q == 63) {
                    uint64_t fh = S.fh, rh = S.rh;
                    changelast(S.ht, fh, rh, (uint8_t)S.draft, sb);
                    if (bf_contains(a, c.bf, fh + rh))
                        S.sub_in_bf[si] = 1;
                } else if (q < k && q % jump == 0 && k + q < S.nz) {
                    const uint32_t a0 = q + 1, site = k - 1;
                    uint64_t f = win_f(S, a0, k), r = win_r(S, a0, k);
                    if (a0 <= site) {
                        const uint8_t old_c = S.Z[site];
                        f ^= sroln(S.ht.f[fcode(old_c)] ^ S.ht.f[fcode(sb)], (a0 + k - 1) - site);
                        r ^= sroln(S.ht.r[rcode(old_c)] ^ S.ht.r[rcode(sb)], site - a0);
                    }
                    if (bf_contains(a, c.bf, f + r))
                        atomicAdd(&S.sub_present[si], 1u);
                }
            }
        }
        __syncthreads();
        uint32_t best_type = 0, best_sup = 0, best_len = 0, best_order = 0; // (scalars: see walk_site)
        uint8_t best_sub = 0, best_char = 0;
        uint32_t any_in_bf = 0;
        for (uint32_t si = 0; si < S.n_sub; si++) {
            if (!S.sub_in_bf[si])
                continue;
            any_in_bf = 1;
            if (tid == 0)
                S.st_probes += 1 + (k + jump - 1) / jump;
            const uint32_t present = S.sub_present[si];
            if (present >= a.need_edit) {
                if (present >= best_sup) {
                    best_type = 1;
                    best_sub = S.subs[si];
                    best_sup = present;
                }
                continue;
            }
            if (best_type != 1) {
                __syncthreads();
                if (tid == 0)
                    S.best_key = 0;
                __syncthreads();
                indel_sweep_body(S.subs[si], S.num_del);
                __syncthreads();
                const uint32_t key = S.best_key;
                uint32_t n_del = (a.max_ins_tries > 0 && S.num_del <= a.max_del)
                                     ? min(a.max_del - S.num_del + 1, a.max_ins_tries)
                                     : 0u;
                const uint32_t del_from = S.num_del;
                __syncthreads();
                if (tid == 0)
                    S.num_del = del_from + n_del;
                if (key) {
                    const uint32_t order = key & 0xFFFF;
                    best_sup = key >> 16;
                    if (order & 1) {
                        best_type = 3;
                        best_len = del_from + (order >> 1);
                    } else {
                        best_type = 2;
                        best_order = order >> 1;
                        best_char = S.subs[si];
                    }
                    break;
                }
            }
        }
        __syncthreads();
        if (tid == 0) {
            // (the record is put together in LDS -- indel_string writes its bytes one by one -- and
            // leaves as one 16-byte store)
            Best& b = S.edit_best;
            for (int i = 0; i < 8; i++)
                b.indel[i] = 0;
            const uint32_t ilen = best_type == 2 ? indel_string(best_char, best_order, b.indel) : best_len;
            SiteRec r;
            r.type = (uint8_t)best_type;
            r.sub_base = best_sub;
            r.indel_len = (uint8_t)ilen;
            r.flags = (uint8_t)(1u | (any_in_bf << 1));
            for (int i = 0; i < 8; i++)
                r.indel[i] = b.indel[i];
            r.probes = (uint32_t)S.st_probes;
            a.site_rec[base_bit + cand] = r;
        }
    }
}

// REQ_SITE: the trials of a site worked out on the spot (the head among edited nodes, or right
// after an edit) -- gather the 2k + 12 window, confirm, substitutes, insertion / deletion sweep,
// make_edit.  Every thread of the workgroup calls it.  Out of line: it is the rare request, and
// inlined its dozen phases set the register budget of the whole trip loop (188 VGPRs unspilled; cut to
// the 128 of four waves per SIMD it spilled 50 and reloaded them on every trip).
__device__ __attribute__((noinline)) void walk_site(uint32_t nxt)
{
    Shared& S = g_S;
    const Ctx& c = S.ctx;
    const WalkCfg& cfg = S.cfg;
    const uint32_t tid = threadIdx.x, k = c.k, jump = cfg.jump, len = c.len;
    // ------------------------------------------------------------ REQ_SITE
    if (tid == 0) {
        const uint32_t want = 2 * k + 12;
        Node hnode = nd_cached(c, S, S.hn);
        const bool in_final = hnode.type == 0 && hnode.e == len - 1 && S.hi <= hnode.e;
        const bool inside = hnode.type == 0 && S.hi + want - 1 <= hnode.e;
        if (in_final || inside) {
            S.fast_head = 1;
            S.nz = min(want, hnode.e - S.hi + 1);
        } else {
            S.fast_head = 0;
            S.nz = gather_plan(c, S, S.Z, want);
        }
        S.draft = to_upper((uint8_t)S.char_in);
        S.best_key = 0;
    }
    __syncthreads();
    // flags are read once, here, by everybody; thread 0 only rewrites them behind barriers
    const uint32_t f_fast = S.fast_head, f_hash = S.need_hash, f_confirmed = S.confirmed;
    if (f_fast) {
        for (uint32_t i = tid; i < S.nz; i += blockDim.x) {
            const uint32_t pos = S.hi + i;
            S.Z[i] = pos - S.cp0 < S.cpn ? S.c_seq[pos - S.cp0] : c.seq[pos];
        }
        __syncthreads();
    } else {
        gather_fill(c, S, S.Z);
        __syncthreads();
    }
    {
        // prefix XORs of the rotated seeds of the whole window (one element per lane):
        // PF[i] = xor_{j<i} ror^j(fwd(Z_j)), QR[i] = xor_{j<i} rol^j(rev(Z_j))
        const uint32_t nzz = S.nz;
        uint64_t tf = 0, tr = 0;
        if (tid < nzz) {
            tf = srorn(S.ht.f[fcode(S.Z[tid])], tid);
            tr = sroln(S.ht.r[rcode(S.Z[tid])], tid);
        }
        const int lane = tid & 63, wave = tid >> 6;
        for (int o = 1; o < 64; o <<= 1) {
            const uint64_t uf = __shfl_up(tf, o), ur = __shfl_up(tr, o);
            if (lane >= o) {
                tf ^= uf;
                tr ^= ur;
            }
        }
        if (lane == 63) {
            S.wave_f[wave] = tf;
            S.wave_r[wave] = tr;
        }
        __syncthreads();
        uint64_t pf = 0, pr = 0;
        for (int w = 0; w < wave; w++) {
            pf ^= S.wave_f[w];
            pr ^= S.wave_r[w];
        }
        if (tid < nzz) {
            S.PF[tid + 1] = tf ^ pf;
            S.QR[tid + 1] = tr ^ pr;
        }
        if (tid == 0) {
            S.PF[0] = 0;
            S.QR[0] = 0;
        }
        __syncthreads();
        if (f_hash && tid == 0) {
            // the walker jumped here: rebuild its rolling state from the window
            S.fh = win_f(S, 0, k);
            S.rh = win_r(S, 0, k);
            S.char_in = S.Z[k - 1];
            S.draft = to_upper(S.Z[k - 1]);
        }
        __syncthreads();
    }
    if (!f_confirmed) {
        // confirm: k more k-mers, every jump-th one probed
        uint32_t miss = 0, bad = 0;
        if (tid < k) {
            const uint32_t q = tid;
            if (k + q >= S.nz)
                bad = 1;
            else if (!is_accepted_u(S.Z[k + q]))
                bad = 1;
        }
        bad = block_sum(S, bad);
        if (!bad && tid < k && tid % jump == 0)
            miss = bf_contains(cfg, c.bf, win_f(S, tid + 1, k) + win_r(S, tid + 1, k)) ? 0u : 1u;
        miss = block_sum(S, miss);
        if (tid == 0) {
            S.confirm_bad = bad;
            S.confirm_missing = miss;
            S.st_probes += bad ? 0 : (k + jump - 1) / jump;
        }
        __syncthreads();
    }
    if (!f_confirmed && (S.confirm_bad || S.confirm_missing < cfg.need_missing)) {
        if (tid == 0) {
            S.req[nxt] = REQ_NEXT;
            S.need_hash = 0;
        }
        __syncthreads();
        return;
    }

    // substitution trials: one wave-sized slice of work per candidate base
    if (tid == 0) {
        S.st_sites++;
        S.confirmed = 0;
        S.need_hash = 0;
        const char* s;
        switch ((uint8_t)S.draft) {
        case 'A': s = "TCG"; break;
        case 'T': s = "ACG"; break;
        case 'C': s = "ATG"; break;
        case 'G': s = "ATC"; break;
        case 'R': s = "TC"; break;
        case 'Y': s = "AG"; break;
        case 'S': s = "AT"; break;
        case 'W': s = "CG"; break;
        case 'K': s = "AC"; break;
        case 'M': s = "TG"; break;
        case 'B': s = "A"; break;
        case 'D': s = "C"; break;
        case 'H': s = "G"; break;
        case 'V': s = "T"; break;
        default: s = ""; break;
        }
        uint32_t n = 0;
        for (; s[n]; n++)
            S.subs[n] = (uint8_t)s[n];
        S.n_sub = n;
        for (uint32_t i = 0; i < 4; i++)
            S.sub_in_bf[i] = S.sub_present[i] = 0;
        S.num_del = 1;
    }
    __syncthreads();
    {
        // thread (si*64 + q): q == 63 probes the changed k-mer itself,
        // q < k probes the k-mer after q+1 rolls when q % jump == 0
        const uint32_t q = tid & 63;
        for (uint32_t si = tid >> 6; si < S.n_sub; si += blockDim.x / 64) { // (128-thread workgroups: two waves for up to three substitutes)
            const uint8_t sb = S.subs[si];
            if (q == 63) {
                uint64_t fh = S.fh, rh = S.rh;
                changelast(S.ht, fh, rh, (uint8_t)S.draft, sb);
                if (bf_contains(cfg, c.bf, fh + rh))
                    S.sub_in_bf[si] = 1;
            } else if (q < k && q % jump == 0 && k + q < S.nz) {
                // k-mer = Z[q+1 .. q+1+k) with Z[k-1] replaced by sb: O(1) from the prefixes
                const uint32_t a0 = q + 1, site = k - 1;
                uint64_t f = win_f(S, a0, k), r = win_r(S, a0, k);
                if (a0 <= site) {
                    const uint8_t old_c = S.Z[site];
                    f ^= sroln(S.ht.f[fcode(old_c)] ^ S.ht.f[fcode(sb)], (a0 + k - 1) - site);
                    r ^= sroln(S.ht.r[rcode(old_c)] ^ S.ht.r[rcode(sb)], site - a0);
                }
                if (bf_contains(cfg, c.bf, f + r))
                    atomicAdd(&S.sub_present[si], 1u);
            }
        }
    }
    __syncthreads();

    // replay the reference's control flow over the trial results (uniform)
    // (the winner as scalars: a private Best -- its indel[] is indexed at run time -- would be promoted to
    // 24 bytes of LDS per lane)
    uint32_t best_type = 0, best_sup = 0, best_len = 0, best_order = 0;
    uint8_t best_sub = 0, best_char = 0;
    for (uint32_t si = 0; si < S.n_sub; si++) {
        if (!S.sub_in_bf[si])
            continue;
        if (tid == 0) {
            // trial + revert leaves the UPPER-case draft base behind (:1608-1615)
            Node tnode = nd_cached(c, S, S.tn);
            if (tnode.type == 0) {
                seq_store(c, S, S.ti, (uint8_t)S.draft);
            } else if (tnode.type == 1) {
                c.nd[S.tn].c = S.draft;
                nd_cache_drop(S);
            }
            S.st_probes += 1 + (k + jump - 1) / jump;
        }
        const uint32_t present = S.sub_present[si];
        if (present >= cfg.need_edit) {
            if (present >= best_sup) {
                best_type = 1;
                best_sub = S.subs[si];
                best_sup = present;
            }
            continue; // modes 0/1: a good substitution ends this base's turn
        }
        if (best_type != 1) {
            __syncthreads();
            if (tid == 0)
                S.best_key = 0;
            __syncthreads();
            indel_sweep(S.subs[si], S.num_del);
            __syncthreads();
            const uint32_t key = S.best_key;
            // deletions del_from.. were consumed by this call
            uint32_t n_del = (cfg.max_ins_tries > 0 && S.num_del <= cfg.max_del)
                                 ? min(cfg.max_del - S.num_del + 1, cfg.max_ins_tries)
                                 : 0u;
            const uint32_t del_from = S.num_del;
            __syncthreads();
            if (tid == 0)
                S.num_del = del_from + n_del;
            if (key) {
                const uint32_t order = key & 0xFFFF;
                best_sup = key >> 16;
                if (order & 1) {
                    best_type = 3;
                    best_len = del_from + (order >> 1);
                } else {
                    best_type = 2;
                    best_order = order >> 1;
                    best_char = S.subs[si];
                }
                break; // tryIndels returned true (modes 0/1)
            }
        }
    }
    __syncthreads();
    if (tid == 0) {
        Best& b = S.edit_best;
        b.type = best_type;
        b.sub_base = best_sub;
        b.sup = best_sup;
        b.indel_len = best_type == 2 ? indel_string(best_char, best_order, b.indel) : best_len;
        make_edit((uint8_t)S.draft);
        S.req[nxt] = REQ_NEXT;
    }
    __syncthreads();
}

// kMinWaves: 1 = as many registers as the compiler likes (136: three workgroups per CU, the
// fastest single walk); 6 = 80 registers (six per CU; some spills).  With thousands of contigs
// the walk is throughput-bound on resident workgroups and the tight build is 17 % faster; with a
// few hundred it is latency-bound and the roomy one is 8 % faster (stage_b_pass picks).
template <int kMinWaves, bool SEG>
__global__ __launch_bounds__(kWalkThreads, kMinWaves) void ntedit_walk(WalkArgs a)
{
    Shared& S = g_S;
    const uint32_t tid = threadIdx.x;
    uint32_t cid = blockIdx.x;
    uint32_t seg_begin = 0, seg_end = 0xFFFFFFFFu, seg_index = 0; // k-mer starts this workgroup walks
    if (SEG) {
        seg_index = a.seg_base + blockIdx.x;
        const SegDesc sd = a.segs[seg_index];
        if (sd.p_begin == sd.p_end)
            return; // unused slot
        cid = sd.cid;
        seg_begin = sd.p_begin;
        seg_end = sd.p_end;
    } else if (a.contig_mode) {
        const uint32_t mode = a.contig_mode[cid];
        if (a.walk_phase == 0 ? mode != 0u : mode != 2u)
            return; // walked as segments / nothing to do again
    }
    if (a.dropped[cid])
        return;
    const uint32_t len = a.len_cur[cid];
    if (len < a.min_contig_len) {
        // ntEdit silently leaves out contigs shorter than -z (ntedit.cpp:1850-1853)
        if (tid == 0) {
            a.dropped[cid] = 1;
            a.len_next[cid] = 0;
        }
        return;
    }
    Ctx c;
    c.seq = a.seq + a.off[cid];
    c.nd = a.nodes + a.node_off[cid];
    c.node_cap = (uint32_t)(a.node_off[cid + 1] - a.node_off[cid]);
    if (SEG) {
        c.nd += a.segs[seg_index].node_first;
        c.node_cap = a.segs[seg_index].node_cap;
    }
    c.bf = a.bf + ((uint64_t)a.contig_batch[cid] * a.n_k + a.ki) * a.bf_words;
    c.len = len;
    c.k = a.k;
    const uint32_t k = a.k, jump = a.jump;
    // the last k-mer start the sweep of this workgroup settles (a whole contig: every one)
    const bool seg_open = SEG && seg_end != 0xFFFFFFFFu; // a right neighbour takes over at seg_end

    if (tid == 0) {
        fill_cfg(S, a, c);
        S.seg_failed = 0;
        S.seg_busy = 0;
        S.seg_arrived = 0;
        S.seg_refused = 0;
    }
    if (SEG && a.segs[seg_index].resume_from != 0xFFFFFFFFu) {
        // go on where descriptor resume_from stopped (its node list is this descriptor's, untouched)
        if (tid == 0) {
            for (uint32_t code = 0; code < 5; code++) {
                uint64_t sd0 = code < 4 ? seed_of_code(code) : 0ULL;
                S.ht.f[code] = sd0;
                S.ht.r[code] = sd0;
                S.ht.fk[code] = sroln(sd0, k);
                S.ht.rk[code] = sroln(sd0, k);
            }
            const SegState st = a.seg_state[a.segs[seg_index].resume_from];
            S.hi = st.hi;
            S.ti = st.ti;
            S.hn = st.hn;
            S.tn = st.tn;
            S.nn = st.nn;
            S.char_in = st.char_in;
            S.min_tested_ti = st.min_tested_ti;
            S.clean_from = st.clean_from;
            S.need_hash = st.need_hash;
            S.fast_head = st.fast_head;
            S.confirmed = st.confirmed;
            S.cur_tested = st.cur_tested;
            S.fh = st.fh;
            S.rh = st.rh;
            S.st_scan = st.st[0];
            S.st_sites = st.st[1];
            S.st_probes = st.st[2];
            S.st_edits = st.st[3];
            S.nc_next = 0;
            S.cw0 = 0;
            S.cwn = 0;
            S.cp0 = 0;
            S.cpn = 0;
            nd_cache_drop(S);
            S.req[0] = REQ_NEXT;
        }
    } else if (SEG && seg_begin != 0) {
        // start clean at seg_begin: the head parked just left of it, everything from it on untouched
        if (tid == 0) {
            for (uint32_t code = 0; code < 5; code++) {
                uint64_t s = code < 4 ? seed_of_code(code) : 0ULL;
                S.ht.f[code] = s;
                S.ht.r[code] = s;
                S.ht.fk[code] = sroln(s, k);
                S.ht.rk[code] = sroln(s, k);
            }
            S.st_scan = S.st_sites = S.st_probes = S.st_edits = 0;
            S.hi = seg_begin - 1;
            S.ti = S.hi + k - 1;
            S.hn = S.tn = 0;
            S.nn = 1;
            Node root;
            root.type = 0;
            root.s = seg_begin - 1;
            root.e = len - 1;
            root.c = 0;
            c.nd[0] = root;
            S.fh = S.rh = 0;
            S.char_in = 0;
            S.cur_tested = 1;
            S.min_tested_ti = 0;
            S.clean_from = 0;
            S.need_hash = 0;
            S.fast_head = 0;
            S.confirmed = 0;
            S.nc_next = 0;
            S.cw0 = 0;
            S.cwn = 0;
            S.cp0 = 0;
            S.cpn = 0;
            nd_cache_drop(S);
            S.req[0] = REQ_NEXT;
        }
    } else if (tid == 0) {
        for (uint32_t code = 0; code < 5; code++) {
            uint64_t s = code < 4 ? seed_of_code(code) : 0ULL;
            S.ht.f[code] = s;
            S.ht.r[code] = s;
            S.ht.fk[code] = sroln(s, k);
            S.ht.rk[code] = sroln(s, k);
        }
        S.st_scan = S.st_sites = S.st_probes = S.st_edits = 0;
        // findFirstAcceptedKmer(0)
        uint32_t hi = len - 1;
        for (uint32_t i = 0; i + k < len;) {
            if (is_accepted_u(c.seq[i])) {
                bool good = true;
                for (uint32_t j = i + 1; j < i + k; j++)
                    if (!is_accepted_u(c.seq[j])) {
                        good = false;
                        i = j + 1;
                        break;
                    }
                if (good) {
                    hi = i;
                    break;
                }
            } else {
                i++;
            }
        }
        S.hi = hi;
        S.ti = hi + k - 1;
        S.hn = S.tn = 0;
        S.nn = 1;
        Node root;
        root.type = 0;
        root.s = 0;
        root.e = len - 1;
        root.c = 0;
        c.nd[0] = root;
        S.fh = S.rh = 0;
        S.char_in = 0;
        S.cur_tested = 0;
        S.min_tested_ti = 0;
        S.clean_from = 0;
        S.need_hash = 0;
        S.fast_head = 0;
        S.confirmed = 0;
        S.nc_next = 0;
        S.cw0 = 0;
        S.cwn = 0;
        S.cp0 = 0;
        S.cpn = 0;
        nd_cache_drop(S);
        if (hi + k - 1 < len) {
            hash_kmer(S.ht, c.seq + hi, k, S.fh, S.rh);
            S.char_in = c.seq[S.ti];
            S.req[0] = REQ_NEXT;
        } else {
            S.req[0] = REQ_DONE;
        }
    }
    __syncthreads();

    // Each pass of this loop is one "request" decided by thread 0.  Every request moves the
    // walker forward over the sequence as edited so far -- at most len positions plus one per
    // node (an inserted base is a node) -- so this many passes is a bug trap, never a limit.
    const uint64_t trip_trap = 8ull * ((uint64_t)len + c.node_cap) + 4096ull;
    if (a.prof && tid == 0) {
        for (int i = 0; i < 8; i++)
            S.prof_t[i] = S.prof_n[i] = 0;
        S.prof_last = wall_clock64();
        S.prof_req = REQ_NEXT;
    }
    for (uint64_t trip = 0;; trip++) {
        const uint32_t cur = (uint32_t)trip & 1u, nxt = cur ^ 1u; // the request slot of this trip / of the next one
        if (a.prof && tid == 0) {
            const unsigned long long now = wall_clock64();
            S.prof_t[S.prof_req & 7] += now - S.prof_last;
            S.prof_n[S.prof_req & 7]++;
            S.prof_last = now;
            S.prof_req = S.req[cur]; // what the trip that starts here works on (thread 0's decision below belongs to it)
        }
        if (trip > trip_trap) {
            if (tid == 0) {
                atomicOr(a.err, 4u);
                if (atomicCAS(a.err + 1, 0u, cid + 1) == 0u) { // the first one leaves a note (words 1..3 of err)
                    a.err[2] = (uint32_t)S.req[cur];
                    a.err[3] = S.hi;
                }
            }
            break;
        }
        // ---------------- thread 0: decide how to reach the next untested k-mer
        if (tid == 0 && S.req[cur] == REQ_NEXT) {
            Node tnode = nd_cached(c, S, S.tn);
            const bool plain = tnode.type == 0 && tnode.e == len - 1 && S.ti <= tnode.e;
            // (the first k-mer looked at starts at hi, or at hi + 1 when the one at hi has been tested)
            const bool clean = plain && S.hn == S.tn && S.hi + (S.cur_tested ? 1u : 0u) >= S.clean_from &&
                               a.bm_site != nullptr;
            if (SEG) {
                // a walker that has not been clean for thousands of decisions is in a stretch where
                // every few bases are edited (ntEdit inserting without end on a near-saturated
                // filter): that stretch is serial whatever is done, and walking it here, then in a
                // repair, then whole would cost it three times -- give the contig up at once
                S.seg_busy = (clean && a.site_rec != nullptr) ? 0u : S.seg_busy + 1u;
                if (seg_open && S.seg_busy > 3000u) {
                    S.seg_failed = 3;
                    S.req[cur] = REQ_DONE;
                }
                // A REFUSED INSERTION (make_edit: it would have repeated what was inserted just
                // before) rearranges the node list -- drop_repeat_nodes moves later nodes over
                // earlier ones and leaves dead ones behind, at which the reference's writer ENDS
                // the contig (ntedit.cpp: the output loop stops at the first node of type -1) -- and
                // may end the walk (findAcceptedKmer finding nothing).  What the output of such a
                // contig is depends on the whole list, not on a segment's part of it (found by
                // tools/fuzz_parity.py with segments of 2048 starts, k = 11 on an 8 KiB filter):
                // the contig goes to the one-workgroup walk.  Any segment, the last one included.
                if (S.seg_refused) {
                    S.seg_failed = 3;
                    S.req[cur] = REQ_DONE;
                }
            }
            if (SEG && S.seg_failed) {
                // (given up just above)
            } else if (SEG && seg_open && S.hi < len &&
                ((clean && a.site_rec != nullptr) ? S.hi + (S.cur_tested ? 1u : 0u) > seg_end
                                                  : (uint64_t)S.hi + seg_guard(k) >= seg_end)) {
                // the walker is still among what an edit touched this close to the segment's end (or
                // has jumped past the end): it does not arrive there the way the next segment assumed
                S.seg_failed = 1;
                S.req[cur] = REQ_DONE;
            } else if (clean && a.site_rec != nullptr) {
                // everything from the head on is untouched contig and every site there has its
                // record: all lanes sweep forward to the first site that ends in an edit
                S.bit_from = S.hi + (S.cur_tested ? 1u : 0u);
                S.req[cur] = REQ_SWEEP;
            } else if (clean) {
                // everything from the head on is untouched contig: the precomputed bitmaps
                // say where the next testable, absent k-mer starts AND whether the k k-mers
                // after it confirm it, so unconfirmed candidates cost no barrier at all
                uint32_t p = S.hi + (S.cur_tested ? 1u : 0u);
                const uint32_t last_p = len - k;
                const uint64_t base_bit = a.off[cid];
                uint32_t words_left = 16;
                int32_t next_req = REQ_DONE;
                // bitmap words from the LDS window when they are there
                auto site_w = [&](uint64_t w) -> unsigned long long {
                    const uint64_t i = w - S.cw0;
                    return i < S.cwn ? S.c_site[i] : a.bm_site[w];
                };
                auto ok_w = [&](uint64_t w) -> unsigned long long {
                    const uint64_t i = w - S.cw0;
                    return i < S.cwn ? S.c_ok[i] : a.bm_ok[w];
                };
                if (p <= last_p && ((base_bit + p) >> 6) - S.cw0 + 3 >= (uint64_t)S.cwn &&
                    S.cw0 + S.cwn <= ((base_bit + len - 1) >> 6)) {
                    // the head is leaving the window (or there is none yet): every lane fetches the next
                    S.bit_from = p;
                    S.req[cur] = REQ_REFILL;
                    p = last_p + 1; // skip the scan below; it resumes after the refill
                    next_req = REQ_REFILL;
                }
                while (p <= last_p) {
                    const uint64_t bit = base_bit + p;
                    unsigned long long w = site_w(bit >> 6) & ok_w(bit >> 6);
                    w &= ~0ULL << (bit & 63);
                    if (!w) {
                        const uint32_t np = (uint32_t)((((bit >> 6) + 1) << 6) - base_bit);
                        S.st_scan += min(np, last_p + 1) - p;
                        p = np;
                        if (--words_left == 0) {
                            next_req = REQ_BITSCAN; // long quiet stretch: search with every lane
                            break;
                        }
                        continue;
                    }
                    const uint32_t cand = (uint32_t)(((bit >> 6) << 6) + (uint64_t)(__ffsll((long long)w) - 1) - base_bit);
                    if (cand > last_p) {
                        S.st_scan += last_p + 1 - p;
                        p = last_p + 1;
                        break;
                    }
                    S.st_scan += cand - p + 1;
                    // confirm (ntedit.cpp:1487-1523): the next k bases must exist and be
                    // accepted, and enough of every jump-th following k-mer must be absent
                    bool ok = (uint64_t)cand + 2ull * k <= len;
                    if (ok) {
                        const uint64_t b2 = base_bit + cand + k;
                        ok = (ok_w(b2 >> 6) >> (b2 & 63)) & 1ULL;
                    }
                    uint32_t missing = 0;
                    if (ok) {
                        for (uint32_t q = 0; q < k; q += jump) {
                            const uint64_t b3 = base_bit + cand + 1 + q;
                            missing += (uint32_t)((site_w(b3 >> 6) >> (b3 & 63)) & 1ULL);
                        }
                        S.st_probes += (k + jump - 1) / jump;
                    }
                    if (ok && missing >= a.need_missing) {
                        S.hi = cand;
                        S.ti = cand + k - 1;
                        S.cur_tested = 1;
                        S.need_hash = 1;
                        S.confirmed = 1;
                        S.min_tested_ti = 0;
                        next_req = REQ_SITE;
                        break;
                    }
                    p = cand + 1;
                }
                if (next_req == REQ_BITSCAN)
                    S.bit_from = p;
                // REQ_DONE here means the scan ran off the end of the contig: nothing is left
                // that the reference could act on
                S.req[cur] = next_req;
            } else if (plain) {
                // scan: current k-mer chars via the cursor, lookahead straight from the contig
                Node hnode = nd_cached(c, S, S.hn);
                S.fast_head = (hnode.type == 0 && S.hi + k - 1 <= hnode.e) ? 1u : 0u;
                uint32_t got = S.fast_head ? k : gather_plan(c, S, S.Z, k);
                if (got < k) {
                    S.req[cur] = REQ_DONE; // cannot happen while the tail is in range
                } else {
                    uint32_t ahead = len - 1 - S.ti;
                    uint32_t j0 = S.cur_tested ? 1u : 0u;
                    uint32_t n = min((uint32_t)blockDim.x, ahead + 1 - j0);
                    // head and tail in the final run: k-mer j starts at contig position hi + j, and
                    // from clean_from on the records answer -- hash only what an edit has touched
                    if (S.hn == S.tn && S.clean_from > S.hi + j0 && a.site_rec != nullptr)
                        n = min(n, S.clean_from - (S.hi + j0));
                    if (SEG && seg_open) // a site this scan finds must be workable inside the segment (hi + guard < seg_end holds here)
                        n = min(n, seg_end - (k + 13) - (S.hi + j0));
                    if (n == 0) {
                        S.req[cur] = REQ_DONE; // the next roll would fail: end of contig
                    } else {
                        S.scan_j0 = j0;
                        S.scan_n = n;
                        S.nz = k + min(ahead, j0 + n - 1);
                        S.req[cur] = REQ_SCAN;
                    }
                }
            } else {
                // literal: one trip through the bottom of the reference loop, then its top
                bool go = true;
                if (S.cur_tested) {
                    int target = -1;
                    do {
                        uint8_t out = 0, in = 0;
                        if (roll_cursor(c, S, S.hi, S.ti, S.hn, S.tn, out, in)) {
                            if (!is_accepted_u(in))
                                target = (int)S.ti + (int)k;
                            S.fh = roll_f(S.ht, S.fh, out, in);
                            S.rh = roll_r(S.ht, S.rh, out, in);
                            S.char_in = in;
                        } else {
                            go = false;
                            break;
                        }
                    } while (target >= 0 && (int)S.ti != target);
                    S.min_tested_ti = 0;
                }
                if (!go || S.hi + k - 1 >= len) {
                    S.req[cur] = REQ_DONE;
                } else {
                    S.cur_tested = 1;
                    S.st_probes++;
                    if (!bf_contains(a, c.bf, S.fh + S.rh))
                        S.req[cur] = REQ_SITE;
                    else
                        S.req[cur] = REQ_NEXT;
                }
            }
        }
        if (a.prof && tid == 0)
            S.prof_req = S.req[cur]; // (what a REQ_NEXT decision turned into -- sweep, scan, bit scan, refill -- is clocked as that)
        __syncthreads();
        // THE REQUEST PROTOCOL.  Thread 0 alone writes requests; every thread reads this trip's
        // exactly once, here, behind the barrier that makes thread 0's decision (and everything the
        // previous trip wrote) visible.  Requests live in TWO slots: a branch below leaves the next
        // trip's request in the OTHER slot, so nothing a branch does -- at any point, with or without
        // a barrier of its own -- can change what a slower wave is still about to read; this slot is
        // written again only after the next trip's barrier, which every thread passes after its read
        // here.  (Round 2 had one slot and relied on each branch happening to contain a barrier ahead
        // of its write; REQ_NEXT did not, and a wave that fell behind while another job's kernel held
        // its CU took the next trip's branch.)
        const int req = S.req[cur];
        if (req == REQ_DONE)
            break;

        if (req == REQ_SWEEP) {
            // 256 positions at a time: which are confirmed sites (bitmaps), what do their records
            // say?  Sites where the trials change nothing only leave counters and a case change
            // behind (a reverted trial writes the draft base back in upper case, soft masking
            // lower-cases it), all independent of each other, so every lane settles its own; the
            // first site that ends in an edit stops the sweep and thread 0 applies it.
            const uint32_t p_start = S.bit_from;
            const uint32_t last_p = (SEG && seg_open) ? min(len - k, seg_end - 1) : len - k; // last k-mer start this workgroup settles
            const uint64_t base_bit = a.off[cid];
            uint32_t p = p_start, first = 0xFFFFFFFFu;
            uint32_t my_sites = 0;
            unsigned long long my_probes = 0;
            while (p <= last_p) {
                {
                    // nothing absent and testable in this chunk's words (every lane reads the same
                    // few words, so this is uniform)?  Then fast-forward: one word per lane,
                    // 16 k positions per step, to the next word that holds a bit.
                    const uint64_t w_lo = (base_bit + p) >> 6, w_hi = (base_bit + min(p + (uint32_t)blockDim.x - 1, last_p)) >> 6;
                    unsigned long long any = 0;
                    for (uint64_t w = w_lo; w <= w_hi; w++)
                        any |= a.bm_site[w] & a.bm_ok[w];
                    if (!any) {
                        const uint64_t w_last = (base_bit + last_p) >> 6;
                        uint64_t w0 = w_hi + 1;
                        uint32_t found = 0xFFFFFFFFu;
                        while (w0 <= w_last && found == 0xFFFFFFFFu) {
                            const uint64_t w = w0 + tid;
                            uint32_t hit = 0xFFFFFFFFu;
                            if (w <= w_last && (a.bm_site[w] & a.bm_ok[w]))
                                hit = (uint32_t)(w - w_lo);
                            found = block_min(S, hit);
                            w0 += blockDim.x;
                        }
                        if (found == 0xFFFFFFFFu) {
                            p = last_p + 1; // ran off the end
                            break;
                        }
                        p = (uint32_t)(((w_lo + found) << 6) - base_bit);
                        continue;
                    }
                }
                const uint32_t pos = p + tid;
                bool cand = false, looked = false;
                if (pos <= last_p && (uint64_t)pos + 2ull * k <= len) {
                    const uint64_t bit = base_bit + pos;
                    cand = ((a.bm_site[bit >> 6] & a.bm_ok[bit >> 6]) >> (bit & 63)) & 1ULL;
                    if (cand) {
                        const uint64_t b2 = bit + k;
                        cand = (a.bm_ok[b2 >> 6] >> (b2 & 63)) & 1ULL;
                    }
                    if (cand) {
                        looked = true; // the reference makes its confirm look-ups here
                        uint32_t missing = 0;
                        for (uint32_t q = 0; q < k; q += jump) {
                            const uint64_t b3 = bit + 1 + q;
                            missing += (uint32_t)((a.bm_site[b3 >> 6] >> (b3 & 63)) & 1ULL);
                        }
                        cand = missing >= a.need_missing;
                    }
                }
                SiteRec r;
                r.flags = 0;
                r.type = 0;
                r.probes = 0;
                if (cand)
                    r = a.site_rec[base_bit + pos];
                const bool stop = cand && (!(r.flags & 1u) || r.type != 0);
                first = block_min(S, stop ? pos : 0xFFFFFFFFu);
                if (looked && pos <= first)
                    my_probes += (k + jump - 1) / jump;
                if (cand && pos < first) {
                    my_sites++;
                    my_probes += r.probes;
                    const uint32_t ti = pos + k - 1;
                    const uint8_t draft = to_upper(c.seq[ti]);
                    if (a.soft_mask)
                        seq_store(c, S, ti, to_lower(draft));
                    else if (r.flags & 2u)
                        seq_store(c, S, ti, draft);
                }
                if (first != 0xFFFFFFFFu)
                    break;
                p += blockDim.x;
            }
            if (my_sites)
                atomicAdd(&S.st_sites, (unsigned long long)my_sites);
            if (my_probes)
                atomicAdd(&S.st_probes, my_probes);
            __syncthreads();
            if (tid == 0) {
                if (first == 0xFFFFFFFFu) {
                    if (p_start <= last_p)
                        S.st_scan += last_p + 1 - p_start;
                    S.req[nxt] = REQ_DONE; // nothing left that the reference could act on (a segment: arrived clean at its end)
                    if (SEG)
                        S.seg_arrived = 1;
                } else if (SEG && seg_open && (uint64_t)first + seg_guard(k) >= seg_end) {
                    // an edit this close to the segment's end: park the (clean) walker just left of
                    // the site, as the bit scan does, so that whoever goes on sweeps into it first thing
                    S.seg_failed = 2;
                    S.st_probes -= (k + jump - 1) / jump; // the site's confirm look-ups are counted by the sweep that goes on
                    S.st_scan += first - p_start;
                    S.hi = first - 1;
                    S.ti = S.hi + k - 1;
                    S.cur_tested = 1;
                    S.min_tested_ti = 0;
                    S.req[nxt] = REQ_DONE;
                } else {
                    S.st_scan += first - p_start + 1;
                    const SiteRec r = a.site_rec[base_bit + first];
                    const uint32_t ti = first + k - 1;
                    S.hi = first;
                    S.ti = ti;
                    S.cur_tested = 1;
                    S.min_tested_ti = 0;
                    if (!(r.flags & 1u)) {
                        // no record (cannot happen while the contig is as the pass found it):
                        // work the site out on the spot
                        S.need_hash = 1;
                        S.confirmed = 1;
                        S.req[nxt] = REQ_SITE;
                    } else {
                        const uint8_t raw = c.seq[ti], draft = to_upper(raw);
                        S.st_sites++;
                        S.st_probes += r.probes;
                        if (r.flags & 2u)
                            seq_store(c, S, ti, draft); // trial + revert leaves the upper-case draft base
                        S.need_hash = 0;
                        S.confirmed = 0;
                        hash_kmer(S.ht, c.seq + first, k, S.fh, S.rh);
                        S.char_in = raw;
                        S.draft = draft;
                        Best& b = S.edit_best;
                        b.type = r.type;
                        b.sub_base = r.sub_base;
                        b.indel_len = r.indel_len;
                        b.sup = 0;
                        for (int i = 0; i < 8; i++)
                            b.indel[i] = r.indel[i];
                        make_edit(draft);
                        S.req[nxt] = REQ_NEXT;
                    }
                }
            }
            __syncthreads();
            continue;
        }

        if (req == REQ_REFILL) {
            // slide the LDS window to the word that holds position bit_from
            const uint64_t base_bit = a.off[cid];
            const uint64_t w0 = (base_bit + S.bit_from) >> 6, w_last = (base_bit + len - 1) >> 6;
            const uint32_t nw = (uint32_t)min((uint64_t)kCacheWords, w_last - w0 + 1);
            const uint32_t p0 = (uint32_t)((w0 << 6) - base_bit); // slots start on 1024-byte boundaries
            const uint32_t nb = min((uint32_t)kCacheBases, len - p0);
            for (uint32_t i = tid; i < nw; i += blockDim.x) {
                S.c_site[i] = a.bm_site[w0 + i];
                S.c_ok[i] = a.bm_ok[w0 + i];
            }
            {
                const uint4* src4 = reinterpret_cast<const uint4*>(c.seq + p0); // p0 is a multiple of 64
                uint4* dst4 = reinterpret_cast<uint4*>(S.c_seq);
                for (uint32_t i = tid; i < nb / 16; i += blockDim.x)
                    dst4[i] = src4[i];
                for (uint32_t i = (nb / 16) * 16 + tid; i < nb; i += blockDim.x)
                    S.c_seq[i] = c.seq[p0 + i];
            }
            __syncthreads();
            if (tid == 0) {
                S.cw0 = w0;
                S.cwn = nw;
                S.cp0 = p0;
                S.cpn = nb;
                S.req[nxt] = REQ_NEXT;
            }
            __syncthreads();
            continue;
        }

        if (req == REQ_BITSCAN) {
            const uint32_t from = S.bit_from, last_p = len - k;
            const uint64_t base_bit = a.off[cid];
            const uint64_t w_first = (base_bit + from) >> 6, w_last = (base_bit + last_p) >> 6;
            uint32_t found = 0xFFFFFFFFu;
            for (uint64_t w0 = w_first; w0 <= w_last && found == 0xFFFFFFFFu; w0 += blockDim.x) {
                const uint64_t w = w0 + tid;
                uint32_t hit = 0xFFFFFFFFu;
                if (w <= w_last) {
                    unsigned long long bits = a.bm_site[w] & a.bm_ok[w];
                    if (w == w_first)
                        bits &= ~0ULL << ((base_bit + from) & 63);
                    if (w == w_last && ((base_bit + last_p) & 63) != 63)
                        bits &= (1ULL << (((base_bit + last_p) & 63) + 1)) - 1;
                    if (bits)
                        hit = (uint32_t)((w << 6) + (uint64_t)(__ffsll((long long)bits) - 1) - base_bit);
                }
                found = block_min(S, hit);
            }
            if (tid == 0) {
                if (found == 0xFFFFFFFFu) {
                    S.st_scan += last_p - from + 1;
                    S.req[nxt] = REQ_DONE; // nothing absent up to the end of the contig
                } else {
                    // park the head just left of the candidate; thread 0 takes it from there
                    S.st_scan += found - from;
                    if (found == 0) {
                        S.hi = 0;
                        S.cur_tested = 0;
                    } else {
                        S.hi = found - 1;
                        S.cur_tested = 1;
                    }
                    S.ti = S.hi + k - 1;
                    S.min_tested_ti = 0;
                    S.req[nxt] = REQ_NEXT;
                }
            }
            __syncthreads();
            continue;
        }

        if (req == REQ_SCAN) {
            // lookahead chars Z[k..nz) come straight from the (untouched) contig
            const uint32_t j0 = S.scan_j0, n = S.scan_n, ti0 = S.ti;
            if (S.fast_head)
                for (uint32_t i = tid; i < k; i += blockDim.x)
                    S.Z[i] = c.seq[S.hi + i];
            else
                gather_fill(c, S, S.Z); // the runs gather_plan left
            for (uint32_t i = k + tid; i < S.nz; i += blockDim.x)
                S.Z[i] = c.seq[ti0 + 1 + (i - k)];
            __syncthreads();
            uint32_t hit = 0xFFFFFFFFu;
            if (tid < n) {
                const uint32_t jj = j0 + tid;
                uint64_t fh, rh;
                hash_kmer(S.ht, S.Z + jj, k, fh, rh);
                S.scan_fh[tid] = fh;
                S.scan_rh[tid] = rh;
                bool tested = (ti0 + jj) >= S.min_tested_ti;
                for (uint32_t i = max(jj, k); i < jj + k; i++)
                    tested &= is_accepted_u(S.Z[i]);
                if (jj == 0)
                    tested = true;
                if (tested && !bf_contains(a, c.bf, fh + rh))
                    hit = jj;
            }
            uint32_t first = block_min(S, hit);
            if (tid == 0) {
                S.st_scan += n;
                const uint32_t dest = first != 0xFFFFFFFFu ? first : j0 + n - 1;
                // a non-accepted base that came in on the way keeps blocking k-mers that hold it
                for (uint32_t i = k; i < k + dest && i < S.nz; i++)
                    if (!is_accepted_u(S.Z[i]))
                        S.min_tested_ti = max(S.min_tested_ti, ti0 + 1 + (i - k) + k);
                {
                    Node hnode = nd_cached(c, S, S.hn);
                    if (hnode.type == 0 && S.hi + dest <= hnode.e) {
                        S.hi += dest;
                    } else {
                        advance(c, S, S.hi, S.hn, dest);
                    }
                }
                S.ti = ti0 + dest;
                // tn stays: the tail never leaves the final node
                S.fh = S.scan_fh[dest - j0];
                S.rh = S.scan_rh[dest - j0];
                S.char_in = S.Z[dest + k - 1];
                S.cur_tested = 1;
                S.req[nxt] = first != 0xFFFFFFFFu ? REQ_SITE : REQ_NEXT;
            }
            __syncthreads();
            continue;
        }

        if (req == REQ_NEXT) {
            if (tid == 0)
                S.req[nxt] = REQ_NEXT; // thread 0 takes it from the top of the loop
            continue;
        }

        // ------------------------------------------------------------ REQ_SITE (out of line: walk_site)
        walk_site(nxt);
    }

    // ---- lay the edited contig out: every live node learns where its bytes go
    // (materialize_kernel then copies with the whole chip)
    __syncthreads();
    if (SEG) {
        // a segment's nodes with offsets from the segment's own start; merge_segments puts the
        // segments of a contig behind one another.  Bases [b_lo, b_hi) are this segment's.
        if (tid == 0) {
            // The walk of a contig can END inside it: after a refused insertion the reference looks for
            // the next accepted k-mer along its node list and gives the contig up when it finds none
            // (findAcceptedKmer returning "", ntedit.cpp:704-760: hi = len).  A segment whose walker
            // ended like that is valid, everything right of it is not to be touched: code 4, its
            // last node runs to the end of the contig, the host drops the segments behind it.
            if (seg_open && !S.seg_failed && !S.seg_arrived)
                S.seg_failed = 4;
            const bool ended = S.seg_failed == 4;
            const uint32_t b_lo = seg_begin ? seg_begin + k - 1 : 0u, b_hi = (seg_open && !ended) ? seg_end + k - 1 : len;
            uint32_t o = 0, live = 0;
            const uint32_t nn = (S.seg_failed && !ended) ? 0u : S.nn; // (a failed segment's node list stays as the walker left it: it is gone on from)
            for (uint32_t ni = 0; ni < nn; ni++) {
                Node n = c.nd[ni];
                if (n.type == -1)
                    break;
                live++;
                if (n.type == 0) {
                    uint32_t lo = max(n.s, b_lo), hi_ex = 0;
                    if (n.s <= len) {
                        uint32_t cnt = n.e - n.s + 1;
                        if (cnt > len - n.s)
                            cnt = len - n.s;
                        hi_ex = min(n.s + cnt, b_hi);
                    }
                    if (hi_ex < lo)
                        hi_ex = lo;
                    c.nd[ni].s = lo;
                    c.nd[ni].c = o;
                    c.nd[ni].e = hi_ex; // exclusive
                    o += hi_ex - lo;
                } else {
                    c.nd[ni].s = o;
                    o++;
                }
            }
            SegOut so;
            so.failed = S.seg_failed;
            so.n_nodes = live;
            so.out_len = o;
            so.pad = 0;
            so.st[0] = S.st_scan;
            so.st[1] = S.st_sites;
            so.st[2] = S.st_probes;
            so.st[3] = S.st_edits;
            a.seg_out[seg_index] = so;
            if (S.seg_failed && !ended) {
                SegState st;
                st.hi = S.hi;
                st.ti = S.ti;
                st.hn = S.hn;
                st.tn = S.tn;
                st.nn = S.nn;
                st.char_in = S.char_in;
                st.min_tested_ti = S.min_tested_ti;
                st.clean_from = S.clean_from;
                st.need_hash = S.need_hash;
                st.fast_head = S.fast_head;
                st.confirmed = S.confirmed;
                st.cur_tested = S.cur_tested;
                st.fh = S.fh;
                st.rh = S.rh;
                st.st[0] = S.st_scan;
                st.st[1] = S.st_sites;
                st.st[2] = S.st_probes;
                st.st[3] = S.st_edits;
                a.seg_state[seg_index] = st;
            }
            if (a.prof && seg_index < 8192u)
                for (int i = 0; i < 8; i++) {
                    a.prof[((uint64_t)a.n_prof_contigs + seg_index) * 16 + i] = S.prof_t[i];
                    a.prof[((uint64_t)a.n_prof_contigs + seg_index) * 16 + 8 + i] = S.prof_n[i];
                }
        }
        return;
    }
    if (tid == 0) {
        uint32_t o = 0, live = 0;
        const uint32_t nn = S.nn;
        for (uint32_t ni = 0; ni < nn; ni++) {
            Node n = c.nd[ni];
            if (n.type == -1)
                break;
            live++;
            if (n.type == 0) {
                uint32_t cnt = 0;
                if (n.s <= len) {
                    cnt = n.e - n.s + 1;
                    if (cnt > len - n.s)
                        cnt = len - n.s;
                }
                c.nd[ni].c = o; // output offset of a position run
                c.nd[ni].e = n.s + cnt; // exclusive end, clamped like substr()
                o += cnt;
            } else {
                c.nd[ni].s = o; // output offset of an inserted base
                o++;
            }
        }
        a.n_nodes[cid] = live;
        if ((uint64_t)o > a.off[cid + 1] - a.off[cid]) {
            // the edited contig no longer fits its slot: nothing is laid out, the host grows
            // the slots and runs the cascade again (stage_b: capacity error)
            atomicOr(a.err, 1024u);
            o = 0;
            a.n_nodes[cid] = 0;
        }
        a.len_next[cid] = o;
        if (a.prof)
            for (int i = 0; i < 8; i++) {
                a.prof[(uint64_t)cid * 16 + i] = S.prof_t[i];
                a.prof[(uint64_t)cid * 16 + 8 + i] = S.prof_n[i];
            }
        atomicAdd(&a.stats[0], S.st_scan);
        atomicAdd(&a.stats[1], S.st_sites);
        atomicAdd(&a.stats[2], S.st_probes);
        atomicAdd(&a.stats[3], S.st_edits);
    }
}


// ---- long contigs as segments (SegDesc): where to cut, and how the pieces come back together
struct BoundaryReq {
    uint32_t cid, nominal, half_window, pad;
};

// One workgroup per wanted cut: the longest stretch of k-mer starts within half_window of `nominal`
// whose site records all say "no edit" (the sweep passes such sites without stopping); the cut goes
// to the END of that stretch, so that the whole stretch lies between the left segment's last edit
// and the cut.  0xFFFFFFFF: no stretch of seg_guard(k) + 16 quiet starts there -- the two segments stay one.
__global__ __launch_bounds__(256) void pick_boundaries(WalkArgs a, const BoundaryReq* req, uint32_t* cut)
{
    __shared__ unsigned long long stops[256]; // a bit per k-mer start of the window: its record says "edit"
    const BoundaryReq r = req[blockIdx.x];
    const uint32_t len = a.len_cur[r.cid], k = a.k;
    const uint64_t base = a.off[r.cid];
    const uint32_t lo = r.nominal > r.half_window ? r.nominal - r.half_window : 1u;
    const uint32_t span = min(2u * r.half_window, 256u * 64u);
    const uint32_t hi = min(lo + span, len > 2 * k ? len - 2 * k : 0u); // exclusive
    unsigned long long m = 0;
    for (uint32_t i = 0; i < 64; i++) {
        const uint32_t p = lo + threadIdx.x * 64 + i;
        if (p < hi) {
            const SiteRec rec = a.site_rec[base + p];
            if ((rec.flags & 1u) && rec.type != 0)
                m |= 1ull << i;
        } else {
            m |= 1ull << i; // outside the window: ends a stretch
        }
    }
    stops[threadIdx.x] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t best_len = 0, best_end = 0, run = 0;
        for (uint32_t i = 0; i < 256u * 64u; i++) {
            const bool stop = (stops[i >> 6] >> (i & 63)) & 1ull;
            if (stop) {
                if (run > best_len) {
                    best_len = run;
                    best_end = lo + i; // the first start past the stretch
                }
                run = 0;
            } else {
                run++;
            }
        }
        if (run > best_len) {
            best_len = run;
            best_end = lo + 256u * 64u;
        }
        cut[2 * blockIdx.x] = best_len >= seg_guard(k) + 16 ? best_end : 0xFFFFFFFFu;
        // how many records of the window say "edit": the host leaves contigs that are edited every
        // few dozen bases (a near-saturated filter) to the one-workgroup walk -- their segments
        // would rarely arrive clean, and every failed one is walked twice
        uint32_t n_stop = 0;
        for (uint32_t w = 0; w < 256; w++)
            n_stop += (uint32_t)__popcll(stops[w]);
        const uint32_t outside = 256u * 64u - (hi > lo ? hi - lo : 0u);
        cut[2 * blockIdx.x + 1] = n_stop - min(n_stop, outside);
    }
}

// bases [lo, hi) of a contig back from the pass's copy in the output slot (a failed segment is walked again, with its right neighbour)
struct RestoreReq {
    uint32_t cid, lo, hi, pad;
};
__global__ __launch_bounds__(256) void restore_ranges(WalkArgs a, const RestoreReq* req)
{
    const RestoreReq r = req[blockIdx.y];
    const uint8_t* bak = a.out + a.off[r.cid];
    uint8_t* seq = a.seq + a.off[r.cid];
    for (uint64_t i = (uint64_t)r.lo + (uint64_t)blockIdx.x * 256 + threadIdx.x; i < r.hi; i += (uint64_t)gridDim.x * 256)
        seq[i] = bak[i];
}

struct SegContig {
    uint32_t cid, seg_first, n_seg, pad;
};

// One workgroup per segmented contig, after its segments' walks.  All valid: the segments' node
// lists move to the head of the contig's node region, one behind the other, with their output
// offsets shifted, and the contig's length, node count and counters are set as one walk would have
// set them.  Any failed: the contig's bases come back from the copy in its output slot and the
// contig is marked for the whole-contig walk that follows.
__global__ __launch_bounds__(256) void merge_segments(WalkArgs a, const SegContig* list, uint32_t* contig_mode)
{
    __shared__ uint32_t s_fail, s_nodes, s_len;
    const SegContig sc = list[blockIdx.x];
    const uint32_t cid = sc.cid, tid = threadIdx.x;
    if (a.dropped[cid])
        return;
    if (tid == 0) {
        uint32_t f = 0, nn = 0, ol = 0;
        for (uint32_t j = 0; j < sc.n_seg; j++) {
            const SegOut& so = a.seg_out[a.seg_order[sc.seg_first + j]];
            f |= (so.failed != 0 && so.failed != 4) ? 1u : 0u;
            nn += so.n_nodes;
            ol += so.out_len;
        }
        s_fail = f;
        s_nodes = nn;
        s_len = ol;
    }
    __syncthreads();
    const uint32_t len = a.len_cur[cid];
    if (s_fail) {
        const uint8_t* bak = a.out + a.off[cid];
        uint8_t* seq = a.seq + a.off[cid];
        for (uint32_t i = tid; i < len; i += 256)
            seq[i] = bak[i];
        if (tid == 0)
            contig_mode[cid] = 2u;
        return;
    }
    Node* const nd = a.nodes + a.node_off[cid];
    uint32_t dst0 = 0, out0 = 0;
    for (uint32_t j = 0; j < sc.n_seg; j++) {
        const SegOut so = a.seg_out[a.seg_order[sc.seg_first + j]];
        const uint32_t src0 = a.segs[a.seg_order[sc.seg_first + j]].node_first;
        for (uint32_t i0 = 0; i0 < so.n_nodes; i0 += 256) {
            const uint32_t i = i0 + tid;
            Node n{};
            if (i < so.n_nodes)
                n = nd[src0 + i];
            __syncthreads(); // dst <= src: the chunk is read before any of it is overwritten
            if (i < so.n_nodes) {
                if (n.type == 0)
                    n.c += out0;
                else
                    n.s += out0;
                nd[dst0 + i] = n;
            }
            __syncthreads();
        }
        dst0 += so.n_nodes;
        out0 += so.out_len;
    }
    if (tid == 0) {
        uint32_t o = s_len;
        a.n_nodes[cid] = s_nodes;
        if ((uint64_t)o > a.off[cid + 1] - a.off[cid]) {
            atomicOr(a.err, 1024u); // the edited contig no longer fits its slot (as in the walk's own layout)
            o = 0;
            a.n_nodes[cid] = 0;
        }
        a.len_next[cid] = o;
        unsigned long long st[4] = { 0, 0, 0, 0 };
        for (uint32_t j = 0; j < sc.n_seg; j++)
            for (int q = 0; q < 4; q++)
                st[q] += a.seg_out[a.seg_order[sc.seg_first + j]].st[q];
        for (int q = 0; q < 4; q++)
            atomicAdd(&a.stats[q], st[q]);
    }
}

// copies output tile blockIdx.x of contig blockIdx.y
__global__ __launch_bounds__(256) void materialize_kernel(const uint8_t* seq, uint8_t* out,
                                                          const uint64_t* off,
                                                          const uint32_t* len_next,
                                                          const uint8_t* dropped, const Node* nodes,
                                                          const uint64_t* node_off,
                                                          const uint32_t* n_nodes,
                                                          const uint32_t* tile_contig,
                                                          const uint32_t* tile_first)
{
    const uint32_t cid = tile_contig[blockIdx.x];
    if (dropped[cid])
        return;
    const uint32_t total = len_next[cid];
    const uint32_t t0 = (blockIdx.x - tile_first[cid]) * kMatTile;
    if (t0 >= total)
        return;
    const uint32_t t1 = min(total, t0 + kMatTile);
    const Node* nd = nodes + node_off[cid];
    const uint32_t nn = n_nodes[cid];
    const uint8_t* src = seq + off[cid];
    uint8_t* dst = out + off[cid];
    // last node whose output offset is <= t0
    uint32_t lo = 0, hi = nn;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        const Node m = nd[mid];
        const uint32_t o = m.type == 0 ? m.c : m.s;
        if (o <= t0)
            lo = mid;
        else
            hi = mid;
    }
    for (uint32_t ni = lo; ni < nn; ni++) {
        const Node n = nd[ni];
        const uint32_t o = n.type == 0 ? n.c : n.s;
        if (o >= t1)
            break;
        if (n.type == 0) {
            const uint32_t cnt = n.e - n.s;
            const uint32_t a0 = max(o, t0), a1 = min(o + cnt, t1);
            for (uint32_t i = a0 + threadIdx.x; i < a1; i += blockDim.x)
                dst[i] = src[n.s + (i - o)];
        } else if (threadIdx.x == 0) {
            dst[o] = (uint8_t)n.c;
        }
    }
}

// bit (off[c] + p) of bm_site: the k-mer starting at p has only accepted bases and is not
// in the filter.  One workgroup covers 1024 consecutive slot positions (4 x 256 lanes; a
// wave's ballot is one 64-bit word of the bitmap).
constexpr int kScanTile = 1024;

__global__ __launch_bounds__(256) void ntedit_scan(WalkArgs a, uint32_t n_contigs,
                                                   unsigned long long* bm_site,
                                                   unsigned long long* bm_ok)
{
    __shared__ uint8_t z[kScanTile + kMaxK];
    __shared__ HashTab ht;
    const uint64_t tile0 = (uint64_t)xcd_contiguous(blockIdx.x, gridDim.x) * kScanTile; // (as in ntedit_site_eval)
    // contig whose slot holds tile0
    uint32_t lo = 0, hi = n_contigs;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) >> 1;
        if (a.off[mid] <= tile0)
            lo = mid;
        else
            hi = mid;
    }
    const uint32_t cid = lo;
    const uint32_t len = a.dropped[cid] ? 0u : a.len_cur[cid];
    const uint64_t p0 = tile0 - a.off[cid];
    const uint32_t k = a.k;
    if (len < a.min_contig_len || p0 + k > len)
        return; // the walker never looks here
    if (threadIdx.x < 5) {
        const uint32_t code = threadIdx.x;
        const uint64_t s = code < 4 ? seed_of_code(code) : 0ULL;
        ht.f[code] = s;
        ht.r[code] = s;
        ht.fk[code] = sroln(s, k);
        ht.rk[code] = sroln(s, k);
    }
    const uint8_t* seq = a.seq + a.off[cid];
    for (uint32_t i = threadIdx.x; i < kScanTile + k - 1; i += 256)
        z[i] = (p0 + i < len) ? seq[p0 + i] : (uint8_t)'N';
    __syncthreads();
    const uint32_t* bf = a.bf + ((uint64_t)a.contig_batch[cid] * a.n_k + a.ki) * a.bf_words;
    for (uint32_t sub = 0; sub < kScanTile / 256; sub++) {
        const uint32_t i = sub * 256 + threadIdx.x;
        const uint64_t p = p0 + i;
        bool site = false, okb = false;
        if (p + k <= len) {
            uint64_t f = 0, r = 0;
            bool ok = true;
            for (uint32_t j = 0; j < k; j++) {
                const uint8_t cf = z[i + j];
                ok &= is_accepted_u(cf);
                f = srol1(f) ^ ht.f[fcode(cf)];
                r = srol1(r) ^ ht.r[rcode(z[i + k - 1 - j])];
            }
            okb = ok;
            site = !bf_contains(a, bf, f + r);
        }
        const unsigned long long word = __ballot(site), word_ok = __ballot(okb);
        if ((threadIdx.x & 63) == 0) {
            bm_site[(tile0 + i) >> 6] = word;
            bm_ok[(tile0 + i) >> 6] = word_ok;
        }
    }
}

// gp_debug_nthash, variants 2 and 4: stage B's hashing over one sequence of raw bytes (HashTab with
// fcode / rcode: IUPAC letters and N hash as the reference's seed tables have them, nthash.hpp:14-46).
// Variant 2: a lane starts four positions from scratch (hash_kmer) and rolls (roll_f / roll_r);
// variant 4: the first k-mer with its last base replaced by A, C, G, T, a, N, Y in turn (changelast,
// nthash.hpp:146-152,171-178): entries 0..6.
__global__ __launch_bounds__(256) void debug_nthash_b(const uint8_t* seq, uint64_t len, uint32_t k, uint32_t hash_num,
                                                      int variant, uint64_t* out_f, uint64_t* out_r, uint64_t* out_h,
                                                      uint8_t* valid)
{
    __shared__ HashTab ht;
    if (threadIdx.x < 5) {
        const uint32_t code = threadIdx.x;
        const uint64_t s = code < 4 ? seed_of_code(code) : 0ULL;
        ht.f[code] = s;
        ht.r[code] = s;
        ht.fk[code] = sroln(s, k);
        ht.rk[code] = sroln(s, k);
    }
    __syncthreads();
    auto emit = [&](uint64_t at, uint64_t fh, uint64_t rh, bool ok) {
        out_f[at] = fh;
        out_r[at] = rh;
        valid[at] = ok ? 1 : 0;
        const uint64_t base = fh + rh;
        for (uint32_t h = 0; h < hash_num; h++)
            out_h[at * hash_num + h] = h ? ext_hash(base, k, h) : base;
    };
    if (variant == 4) {
        if (blockIdx.x == 0 && threadIdx.x < 7 && len >= k) {
            uint64_t fh, rh;
            hash_kmer(ht, seq, k, fh, rh);
            changelast(ht, fh, rh, seq[k - 1], "ACGTaNY"[threadIdx.x]);
            emit(threadIdx.x, fh, rh, true);
        }
        return;
    }
    const uint64_t p0 = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    uint64_t fh = 0, rh = 0;
    for (int j = 0; j < 4; j++) {
        const uint64_t p = p0 + j;
        if (p + k > len)
            break;
        if (j == 0)
            hash_kmer(ht, seq + p, k, fh, rh);
        else {
            fh = roll_f(ht, fh, seq[p - 1], seq[p + k - 1]);
            rh = roll_r(ht, rh, seq[p - 1], seq[p + k - 1]);
        }
        bool ok = true;
        for (uint32_t i = 0; i < k; i++)
            ok &= fcode(seq[p + i]) < 4;
        emit(p, fh, rh, ok);
    }
}

int debug_nthash_stage_b(hipStream_t s, const uint8_t* d_seq, uint64_t len, uint32_t k, uint32_t hash_num, int variant,
                         uint64_t* d_f, uint64_t* d_r, uint64_t* d_h, uint8_t* d_valid)
{
    const uint32_t grid = (uint32_t)((len + 1023) / 1024);
    hipLaunchKernelGGL(debug_nthash_b, dim3(grid), dim3(256), 0, s, d_seq, len, k, hash_num, variant, d_f, d_r, d_h,
                       d_valid);
    GP_HIP(hipGetLastError());
    return GP_OK;
}

// ------------------------------------------------------------- small kernels
__global__ __launch_bounds__(256) void copy_contigs(const uint8_t* src, const uint64_t* src_off,
                                                    const uint32_t* src_len, uint8_t* dst,
                                                    const uint64_t* dst_off, uint32_t* dst_len,
                                                    uint8_t* dropped,
                                                    const uint8_t* only_if_batch_skip,
                                                    const uint32_t* contig_batch,
                                                    const uint32_t* tile_contig,
                                                    const uint32_t* tile_first)
{
    const uint32_t cid = tile_contig[blockIdx.x];
    if (only_if_batch_skip && !only_if_batch_skip[contig_batch[cid]])
        return;
    const uint32_t n = src_len[cid];
    const uint32_t t0 = (blockIdx.x - tile_first[cid]) * kMatTile;
    if (t0 == 0 && threadIdx.x == 0) {
        dst_len[cid] = n;
        dropped[cid] = 0;
    }
    if (t0 >= n)
        return;
    const uint32_t t1 = min(n, t0 + kMatTile);
    const uint8_t* s = src + src_off[cid];
    uint8_t* d = dst + dst_off[cid];
    for (uint32_t i = t0 + threadIdx.x; i < t1; i += blockDim.x)
        d[i] = s[i];
}

// scripts/goldpolish-ntedit:31-40: bytes(out)/bytes(in) truncated to 4 decimals < 0.75
__global__ void guard_kernel(const uint32_t* batch_begin, uint32_t n_batches,
                             const uint32_t* name_len, const uint32_t* in_len,
                             const uint32_t* cur_len, const uint8_t* dropped, uint8_t* batch_skip)
{
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= n_batches)
        return;
    unsigned long long in_bytes = 0, out_bytes = 0;
    for (uint32_t c = batch_begin[b]; c < batch_begin[b + 1]; c++) {
        in_bytes += 1ull + name_len[c] + 1 + in_len[c] + 1;
        if (!dropped[c])
            out_bytes += 1ull + name_len[c] + 1 + cur_len[c] + 1;
    }
    batch_skip[b] = (in_bytes > 0 && (out_bytes * 10000ull) / in_bytes < 7500ull) ? 1 : 0;
}

// One pass over every tile of every live contig: bit 0 of flags[c] = some byte is not
// upper-case ACGT (the mask rule has work to do), bit 1 = some byte is a Sealer gap
// character.  Lets the per-contig kernels below skip clean contigs, which are then
// copied by copy_flagged with the whole chip.
__device__ inline bool gap_char_b(uint8_t c)
{
    switch (c) {
    case 'N': case 'n': case 'a': case 't': case 'c': case 'g': case 'u': case 'r': case 'y':
    case 'k': case 'm': case 's': case 'w': case 'b': case 'd': case 'h': case 'v': case 'x':
        return true;
    default:
        return false;
    }
}

__global__ __launch_bounds__(256) void classify_kernel(const uint8_t* seq, const uint64_t* off,
                                                       const uint32_t* len, const uint8_t* dropped,
                                                       const uint32_t* tile_contig,
                                                       const uint32_t* tile_first, uint32_t* flags)
{
    const uint32_t cid = tile_contig[blockIdx.x];
    if (dropped[cid])
        return;
    const uint32_t n = len[cid];
    const uint32_t t0 = (blockIdx.x - tile_first[cid]) * kMatTile;
    if (t0 >= n)
        return;
    const uint32_t t1 = min(n, t0 + kMatTile);
    const uint8_t* s = seq + off[cid];
    uint32_t f = 0;
    // kMatTile = 256 lanes x 16 bytes; slots are 1024-byte aligned
    const uint32_t i0 = t0 + threadIdx.x * 16;
    if (i0 + 16 <= t1) {
        const uint4 v = *reinterpret_cast<const uint4*>(s + i0);
        const uint32_t w[4] = { v.x, v.y, v.z, v.w };
        for (int q = 0; q < 4; q++)
            for (int r = 0; r < 4; r++) {
                const uint8_t ch = (uint8_t)(w[q] >> (8 * r));
                f |= (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') ? 0u : 1u;
                f |= gap_char_b(ch) ? 2u : 0u;
            }
    } else {
        for (uint32_t i = i0; i < t1; i++) {
            const uint8_t ch = s[i];
            f |= (ch == 'A' || ch == 'C' || ch == 'G' || ch == 'T') ? 0u : 1u;
            f |= gap_char_b(ch) ? 2u : 0u;
        }
    }
    // one atomic per wave at most
    const uint32_t wf = __reduce_or_sync_compat(f);
    if ((threadIdx.x & 63) == 0 && wf)
        atomicOr(&flags[cid], wf);
}

// dst := src for every contig whose flags & reject == 0 (and len >= min_len); per tile
__global__ __launch_bounds__(256) void copy_flagged(const uint8_t* src, uint8_t* dst,
                                                    const uint64_t* off, const uint32_t* len,
                                                    uint32_t* len_next, const uint8_t* dropped,
                                                    const uint32_t* tile_contig,
                                                    const uint32_t* tile_first,
                                                    const uint32_t* flags, uint32_t reject,
                                                    uint32_t min_len)
{
    const uint32_t cid = tile_contig[blockIdx.x];
    if (dropped[cid] || (flags[cid] & reject))
        return;
    const uint32_t n = len[cid];
    if (n < min_len)
        return;
    const uint32_t t0 = (blockIdx.x - tile_first[cid]) * kMatTile;
    if (t0 == 0 && threadIdx.x == 0)
        len_next[cid] = n;
    if (t0 >= n)
        return;
    const uint32_t t1 = min(n, t0 + kMatTile);
    const uint8_t* s = src + off[cid];
    uint8_t* d = dst + off[cid];
    const uint32_t i0 = t0 + threadIdx.x * 16;
    if (i0 + 16 <= t1) {
        *reinterpret_cast<uint4*>(d + i0) = *reinterpret_cast<const uint4*>(s + i0);
    } else {
        for (uint32_t i = i0; i < t1; i++)
            d[i] = s[i];
    }
}

// ---- soft mask (scripts/goldpolish-mask:44-72), one workgroup per contig.
// The regex ([ACTG]+|[Nn]+|[actgUNMRWSYKVHDBunmrwsykvhdb]+) is a 4-state
// machine over the bytes; each thread runs it over its own chunk for every
// possible entry state, the chunk summaries are chained, and a second sweep
// applies the run-length rule.  Run lengths are carried across chunks.
__device__ inline int cls_acgt(uint8_t c) { return c == 'A' || c == 'C' || c == 'T' || c == 'G'; }
__device__ inline int cls_n(uint8_t c) { return c == 'N' || c == 'n'; }
__device__ inline int cls_other(uint8_t c)
{
    switch (c) {
    case 'a': case 'c': case 't': case 'g': case 'U': case 'N': case 'M': case 'R': case 'W':
    case 'S': case 'Y': case 'K': case 'V': case 'H': case 'D': case 'B': case 'u': case 'n':
    case 'm': case 'r': case 'w': case 's': case 'y': case 'k': case 'v': case 'h': case 'd':
    case 'b':
        return 1;
    default:
        return 0;
    }
}
// state: 0 = between runs, 1 = in [ACTG]+, 2 = in [Nn]+, 3 = in the third class
__device__ inline int mask_next(int st, uint8_t c, bool& starts)
{
    starts = false;
    if (st == 1 && cls_acgt(c))
        return 1;
    if (st == 2 && cls_n(c))
        return 2;
    if (st == 3 && cls_other(c))
        return 3;
    starts = true;
    if (cls_acgt(c))
        return 1;
    if (cls_n(c))
        return 2;
    if (cls_other(c))
        return 3;
    starts = false;
    return 0;
}

__device__ inline uint8_t mask_view(const uint8_t* s, uint32_t len, uint32_t k, uint32_t i)
{
    // the input with its first and last k bases upper-cased (all of it if len < 2k)
    uint8_t ch = s[i];
    if (len < 2 * k || i < k || i >= len - k)
        ch = to_upper(ch);
    return ch;
}

constexpr int kMaskThreads = 256;

__global__ __launch_bounds__(kMaskThreads) void mask_kernel(const uint8_t* src,
                                                            const uint64_t* off,
                                                            const uint32_t* len_cur, uint8_t* dst,
                                                            uint32_t* len_next,
                                                            const uint8_t* dropped, uint32_t k,
                                                            const uint32_t* flags)
{
    __shared__ uint8_t exit_state[kMaskThreads][4];
    __shared__ uint32_t run_open_len[kMaskThreads][4]; // length of the run still open at chunk end
    __shared__ uint8_t run_open_first[kMaskThreads][4];
    __shared__ uint32_t kept[kMaskThreads];
    __shared__ uint8_t entry_state[kMaskThreads + 1];
    __shared__ uint32_t entry_len[kMaskThreads + 1];
    __shared__ uint8_t entry_first[kMaskThreads + 1];
    __shared__ uint32_t tail_len[kMaskThreads + 1]; // how far the run open at chunk start continues
    __shared__ uint32_t out_base[kMaskThreads + 1];
    __shared__ uint32_t first_keep, last_keep;
    const uint32_t cid = blockIdx.x, tid = threadIdx.x;
    if (dropped[cid])
        return;
    const uint32_t len = len_cur[cid];
    const uint8_t* s = src + off[cid];
    uint8_t* d = dst + off[cid];
    // a contig of nothing but upper-case ACGT, at least k long, is one run that the rule
    // leaves alone: copy_flagged has already copied it
    if (!(flags[cid] & 1u) && len >= k && len > 0)
        return;
    const uint32_t chunk = (len + kMaskThreads - 1) / kMaskThreads;
    const uint32_t b = min(len, tid * chunk), e = min(len, b + chunk);

    // pass 1: chunk summary for each entry state
    for (int st0 = 0; st0 < 4; st0++) {
        int st = st0;
        uint32_t open_len = 0; // chars of the currently open run inside this chunk
        uint8_t open_first = 0;
        bool from_entry = true; // open run began before the chunk
        for (uint32_t i = b; i < e; i++) {
            bool starts;
            uint8_t ch = mask_view(s, len, k, i);
            st = mask_next(st, ch, starts);
            if (st == 0) {
                open_len = 0;
                from_entry = false;
            } else if (starts) {
                open_len = 1;
                open_first = ch;
                from_entry = false;
            } else {
                open_len++;
            }
        }
        exit_state[tid][st0] = (uint8_t)st;
        // if the run reaches back to the chunk start, mark with the high bit
        run_open_len[tid][st0] = open_len | (from_entry && st != 0 ? 0x80000000u : 0u);
        run_open_first[tid][st0] = open_first;
    }
    __syncthreads();
    if (tid == 0) {
        // chain: state / open run length / first char of the open run at each chunk start
        uint8_t st = 0, first = 0;
        uint32_t rl = 0;
        for (int t = 0; t < kMaskThreads; t++) {
            entry_state[t] = st;
            entry_len[t] = rl;
            entry_first[t] = first;
            uint32_t ol = run_open_len[t][st];
            uint8_t nst = exit_state[t][st];
            if (ol & 0x80000000u) {
                rl += ol & 0x7FFFFFFFu;
            } else {
                rl = ol;
                first = run_open_first[t][st];
            }
            if (nst == 0)
                rl = 0;
            st = nst;
        }
        entry_state[kMaskThreads] = st;
    }
    __syncthreads();
    // pass 2: how far does the run that is open at my chunk start extend to the right?
    {
        int st = entry_state[tid];
        uint32_t ext = 0;
        bool whole = true;
        if (st != 0) {
            for (uint32_t i = b; i < e; i++) {
                bool starts;
                uint8_t ch = mask_view(s, len, k, i);
                int nst = mask_next(st, ch, starts);
                if (nst == 0 || starts) {
                    whole = false;
                    break;
                }
                st = nst;
                ext++;
            }
        } else {
            whole = false;
        }
        kept[tid] = ext | (whole && e > b ? 0x80000000u : (e == b && st != 0 ? 0x80000000u : 0u));
    }
    __syncthreads();
    if (tid == 0) {
        // tail_len[t] = length of the entry run's continuation from chunk t's start onwards
        uint32_t acc = 0;
        tail_len[kMaskThreads] = 0;
        for (int t = kMaskThreads - 1; t >= 0; t--) {
            uint32_t v = kept[t];
            if (v & 0x80000000u)
                acc = (v & 0x7FFFFFFFu) + acc;
            else
                acc = v;
            tail_len[t] = acc;
        }
    }
    __syncthreads();
    // pass 3: count kept chars per chunk (chars outside every class vanish)
    {
        int st = entry_state[tid];
        uint32_t n = 0;
        for (uint32_t i = b; i < e; i++) {
            bool starts;
            st = mask_next(st, mask_view(s, len, k, i), starts);
            n += st != 0;
        }
        kept[tid] = n;
    }
    __syncthreads();
    if (tid == 0) {
        uint32_t acc = 0;
        for (int t = 0; t < kMaskThreads; t++) {
            out_base[t] = acc;
            acc += kept[t];
        }
        out_base[kMaskThreads] = acc;
        first_keep = 0xFFFFFFFFu;
        last_keep = 0;
    }
    __syncthreads();
    // pass 4: emit.  A run is lower-cased when it is shorter than k and does not start
    // with 'N'; its full length = part before the chunk + part inside + continuation.
    {
        int st = entry_state[tid];
        uint32_t o = out_base[tid];
        uint32_t i = b;
        uint32_t lo_first = 0xFFFFFFFFu, hi_last = 0; // first / last+1 non-[Nn] output index
        while (i < e) {
            bool starts;
            uint8_t ch = mask_view(s, len, k, i);
            int nst = mask_next(st, ch, starts);
            if (nst == 0) {
                st = 0;
                i++;
                continue;
            }
            // [i, j) = the part of this run inside my chunk
            uint32_t before = 0;
            uint8_t first = ch;
            if (!starts) { // continuing the entry run (only possible at i == b)
                before = entry_len[tid];
                first = entry_first[tid];
            }
            uint32_t j = i + 1;
            int rs = nst;
            while (j < e) {
                bool st2;
                int n2 = mask_next(rs, mask_view(s, len, k, j), st2);
                if (n2 == 0 || st2)
                    break;
                rs = n2;
                j++;
            }
            uint32_t after = (j == e) ? tail_len[tid + 1] : 0;
            // does the run really continue into the next chunk?  tail_len[t+1] counts the
            // continuation of whatever run is open at that chunk's start, which is this
            // run exactly when we reached the chunk end inside it.
            uint32_t total = before + (j - i) + after;
            bool lower = (first != 'N') && total < k;
            for (uint32_t q = i; q < j; q++) {
                uint8_t c2 = mask_view(s, len, k, q);
                if (lower)
                    c2 = to_lower(c2);
                d[o] = c2;
                if (!cls_n(c2)) {
                    lo_first = min(lo_first, o);
                    hi_last = max(hi_last, o + 1);
                }
                o++;
            }
            st = rs;
            i = j;
        }
        if (lo_first != 0xFFFFFFFFu) {
            atomicMin(&first_keep, lo_first);
            atomicMax(&last_keep, hi_last);
        }
    }
    __syncthreads();
    // strip("Nn") from both ends; empty -> "N"
    const uint32_t fk = first_keep, lk = last_keep;
    if (fk == 0xFFFFFFFFu) {
        if (tid == 0) {
            d[0] = 'N';
            len_next[cid] = 1;
        }
        return;
    }
    // shift left in chunks so that no thread overwrites bytes another still has to read
    if (fk > 0) {
        const uint32_t n = lk - fk;
        for (uint32_t base = 0; base < n; base += kMaskThreads) {
            uint8_t v = 0;
            if (base + tid < n)
                v = d[fk + base + tid];
            __syncthreads();
            if (base + tid < n)
                d[base + tid] = v;
            __syncthreads();
        }
    }
    if (tid == 0)
        len_next[cid] = lk - fk;
}

__global__ void upper_kernel(uint8_t* seq, const uint64_t* off, const uint32_t* len,
                             const uint8_t* dropped)
{
    const uint32_t cid = blockIdx.y;
    if (dropped[cid])
        return;
    uint8_t* s = seq + off[cid];
    const uint32_t n = len[cid];
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        s[i] = to_upper(s[i]);
}

// ------------------------------------------------------------------ host side
static uint32_t smallest_count_passing(float threshold)
{
    // smallest unsigned c with float(c) >= threshold, the reference's comparison
    uint32_t c = 0;
    while (!((float)c >= threshold) && c < 100000)
        c++;
    return c;
}

int stage_bc_prepare(gp_job* job)
{
    StageBC* b = new (std::nothrow) StageBC();
    if (!b)
        return GP_ERR_OOM;
    job->bc = b;
    const uint32_t n = job->n_contigs;
    b->h_off.resize((size_t)n + 1);
    b->h_node_off.resize((size_t)n + 1);
    uint64_t off = 0, noff = 0;
    std::vector<uint32_t> in_len(n);
    for (uint32_t c = 0; c < n; c++) {
        uint64_t len = job->h_contig_off[c + 1] - job->h_contig_off[c];
        if (len >= (1ull << 31)) {
            set_error("contig longer than 2^31 bases");
            return GP_ERR_UNSUPPORTED;
        }
        in_len[c] = (uint32_t)len;
        b->h_off[c] = off;
        b->h_node_off[c] = noff;
        // room for net insertions (ntEdit) and gap fills (Sealer)
        // (doubled by gp_job_polish when a contig outgrows it: filters that call nearly every
        // k-mer present make ntEdit insert without end)
        uint64_t cap = len + (len / 2 + 4096) * job->bc_slack;
        cap = (cap + 1023) & ~1023ull; // slots start on ntedit_scan tile boundaries
        off += cap;
        noff += len + 256 + (len / 2 + 4096) * (job->bc_slack - 1);
    }
    b->h_off[n] = off;
    b->h_node_off[n] = noff;
    b->total_slots = off;
    int rc;
    for (int i = 0; i < 2; i++) {
        if ((rc = b->seq[i].alloc(off + 64)) != GP_OK)
            return rc;
        if ((rc = b->len[i].alloc(n)) != GP_OK)
            return rc;
    }
    if ((rc = b->off.alloc((size_t)n + 1)) != GP_OK || (rc = b->dropped.alloc(n)) != GP_OK ||
        (rc = b->nodes.alloc(noff + 1)) != GP_OK ||
        (rc = b->node_off.alloc((size_t)n + 1)) != GP_OK ||
        (rc = b->contig_batch.alloc(n)) != GP_OK || (rc = b->name_len.alloc(n)) != GP_OK ||
        (rc = b->in_len.alloc(n)) != GP_OK || (rc = b->in_off.alloc((size_t)n + 1)) != GP_OK ||
        (rc = b->batch_begin.alloc((size_t)job->n_batches + 1)) != GP_OK ||
        (rc = b->err.alloc(4)) != GP_OK || (rc = b->stats.alloc(16)) != GP_OK ||
        (rc = b->batch_skip.alloc(job->n_batches)) != GP_OK ||
        (rc = b->bm_site.alloc(off / 64 + 64)) != GP_OK ||
        (rc = b->bm_ok.alloc(off / 64 + 64)) != GP_OK || (rc = b->n_nodes.alloc(n)) != GP_OK ||
        (getenv("GP_NTEDIT_SPECULATE") && getenv("GP_NTEDIT_SPECULATE")[0] == '0'
             ? GP_OK : (rc = b->site_rec.alloc(off + 64))) != GP_OK ||
        (rc = b->tile_first.alloc((size_t)n + 1)) != GP_OK || (rc = b->flags.alloc(n)) != GP_OK)
        return rc;
    // (copies and clears on the context's stream, never the null stream: that one waits for every blocking
    // stream of the device -- CU-masked streams are -- i.e. for another job's kernels)
    hipStream_t sq = job->ctx->stream;
    {
        std::vector<uint32_t> tf((size_t)n + 1), tc;
        for (uint32_t c = 0; c < n; c++) {
            tf[c] = (uint32_t)tc.size();
            uint64_t cap = b->h_off[c + 1] - b->h_off[c];
            for (uint64_t t = 0; t < (cap + kMatTile - 1) / kMatTile; t++)
                tc.push_back(c);
        }
        tf[n] = (uint32_t)tc.size();
        b->n_tiles = (uint32_t)tc.size();
        if ((rc = b->tile_contig.alloc(tc.size())) != GP_OK)
            return rc;
        if (!tc.empty())
            GP_HIP(hipMemcpyAsync(b->tile_contig.p, tc.data(), tc.size() * sizeof(uint32_t), hipMemcpyHostToDevice, sq));
        GP_HIP(hipMemcpyAsync(b->tile_first.p, tf.data(), tf.size() * sizeof(uint32_t), hipMemcpyHostToDevice, sq));
    }
    GP_HIP(hipMemcpyAsync(b->off.p, b->h_off.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, sq));
    GP_HIP(hipMemcpyAsync(b->node_off.p, b->h_node_off.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, sq));
    if (n) {
        GP_HIP(hipMemcpyAsync(b->contig_batch.p, job->h_contig_batch.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, sq));
        GP_HIP(hipMemcpyAsync(b->name_len.p, job->h_contig_name_len.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, sq));
        GP_HIP(hipMemcpyAsync(b->in_len.p, in_len.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice, sq));
    }
    GP_HIP(hipMemcpyAsync(b->in_off.p, job->h_contig_off.data(), (n + 1) * sizeof(uint64_t), hipMemcpyHostToDevice, sq));
    GP_HIP(hipMemcpyAsync(b->batch_begin.p, job->h_batch_contig_begin.data(),
                     (job->n_batches + 1) * sizeof(uint32_t), hipMemcpyHostToDevice, sq));
    GP_HIP(hipMemsetAsync(b->err.p, 0, 4 * sizeof(uint32_t), sq));
    GP_HIP(hipMemsetAsync(b->stats.p, 0, 16 * sizeof(unsigned long long), sq));
    GP_HIP(hipStreamSynchronize(sq)); // (in_len is a local)
    return stage_c_prepare(job);
}

void stage_bc_release(gp_job* job)
{
    if (!job->bc)
        return;
    stage_c_release(job);
    delete job->bc;
    job->bc = nullptr;
}

int stage_b_reset(gp_job* job)
{
    StageBC* b = job->bc;
    hipStream_t s = job->ctx->stream;
    b->cur = 0;
    GP_HIP(hipMemsetAsync(b->dropped.p, 0, std::max<size_t>(job->n_contigs, 1), s));
    GP_HIP(hipMemsetAsync(b->stats.p, 0, 16 * sizeof(unsigned long long), s));
    GP_HIP(hipMemsetAsync(b->err.p, 0, 4 * sizeof(uint32_t), s));
    if (job->n_contigs) {
        hipLaunchKernelGGL(copy_contigs, dim3(b->n_tiles), dim3(256), 0, s,
                           (const uint8_t*)job->d_contigs_in.p, (const uint64_t*)b->in_off.p,
                           (const uint32_t*)b->in_len.p, b->seq[0].p, (const uint64_t*)b->off.p,
                           b->len[0].p, b->dropped.p, (const uint8_t*)nullptr,
                           (const uint32_t*)b->contig_batch.p, (const uint32_t*)b->tile_contig.p,
                           (const uint32_t*)b->tile_first.p);
        GP_HIP(hipGetLastError());
    }
    job->counters.b_scan_kmers = job->counters.b_sites = job->counters.b_probes =
        job->counters.b_edits = 0;
    job->t_walk.reset();
    job->t_scan.reset();
    job->t_mask.reset();
    job->t_upper.reset();
    job->t_bfs.reset();
    return GP_OK;
}

static int pull_stats(gp_job* job)
{
    StageBC* b = job->bc;
    unsigned long long* const st = job->h_readback;   // pinned: [0..15] statistics, [16..17] the four error words
    GP_HIP(hipMemcpyAsync(st, b->stats.p, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost, job->ctx->stream));
    GP_HIP(hipMemcpyAsync(st + 16, b->err.p, 4 * sizeof(uint32_t), hipMemcpyDeviceToHost, job->ctx->stream));
    GP_HIP(hipStreamSynchronize(job->ctx->stream));
    const uint32_t* const errw = reinterpret_cast<const uint32_t*>(st + 16);
    const uint32_t err = errw[0];
    job->counters.b_scan_kmers = st[0];
    job->counters.b_sites = st[1];
    job->counters.b_probes = st[2];
    job->counters.b_edits = st[3];
    job->counters.c_gaps = st[4];
    job->counters.c_closed = st[5];
    job->counters.c_vertex_tests = st[6];
    job->counters.c_no_start_goal = st[7];
    job->counters.c_no_path = st[8];
    job->counters.c_unique_path = st[9];
    job->counters.c_multiple_paths = st[10];
    job->counters.c_too_many_paths = st[11];
    job->counters.c_too_many_branches = st[12];
    job->counters.c_max_cost = st[13];
    int rc;
    if ((rc = job->t_walk.collect()) != GP_OK || (rc = job->t_scan.collect()) != GP_OK ||
        (rc = job->t_mask.collect()) != GP_OK || (rc = job->t_upper.collect()) != GP_OK ||
        (rc = job->t_bfs.collect()) != GP_OK)
        return rc;
    // everything sized from the contig slots, which gp_job_polish doubles before it runs the
    // cascade again -- 1: node list full, 8: more gaps than a slot of this size is given room for
    // (only reachable through the step-wise calls on unmasked input), 32: Sealer's pool of
    // connecting sequences full, 512 / 1024: filled / edited contig longer than its slot
    if (err & (1u | 8u | 32u | 1024u | 512u)) {
        set_error("a contig outgrew its device slot, or Sealer the arenas sized from the slots (code " +
                  std::to_string(err) + "): gp_job_polish retries with larger slots, the step-wise calls do not");
        return GP_ERR_CAPACITY;
    }
    if (err) {
        // bug traps; the walker's (4) leaves which contig it was and where it stood
        set_error("device work buffer overflow (code " + std::to_string(err) + "; walker trap: contig " +
                  std::to_string((long long)errw[1] - 1) + ", request " + std::to_string(errw[2]) + ", head " +
                  std::to_string(errw[3]) + ")");
        return GP_ERR_STATE;
    }
    return GP_OK;
}

int stage_b_pass(gp_job* job, uint32_t ki)
{
    StageBC* b = job->bc;
    const gp_params& p = job->prm;
    if (job->n_contigs == 0)
        return GP_OK;
    WalkArgs a{};
    a.seq = b->seq[b->cur].p;
    a.out = b->seq[b->cur ^ 1].p;
    a.off = b->off.p;
    a.len_cur = b->len[b->cur].p;
    a.len_next = b->len[b->cur ^ 1].p;
    a.dropped = b->dropped.p;
    a.nodes = b->nodes.p;
    a.node_off = b->node_off.p;
    a.bf = job->d_bf.p;
    a.contig_batch = b->contig_batch.p;
    a.n_k = p.n_k;
    a.ki = ki;
    a.k = p.k[ki];
    a.hash_num = p.hash_num;
    a.bf_words = (uint32_t)(p.bf_bytes / 4);
    a.bf_mod = make_fastmod(p.bf_bytes * 8);
    static const uint32_t tries[6] = { 0, 1, 5, 21, 85, 341 };
    a.max_ins_tries = tries[p.max_insertions];
    a.max_del = p.max_deletions;
    a.jump = p.jump;
    a.min_contig_len = p.min_contig_len;
    a.soft_mask = p.soft_mask ? 1 : 0;
    // the reference's float expressions (ntedit.cpp:1517-1523, 1333-1337, 1226-1230)
    const float kf = (float)a.k;
    a.need_missing = smallest_count_passing((kf / (float)p.jump) * p.missing_ratio);
    a.need_edit = smallest_count_passing((kf / (float)p.jump) * p.edit_ratio);
    a.need_del = smallest_count_passing((1 + (kf / (float)p.jump)) * p.edit_ratio);
    a.insertion_cap = (uint32_t)(kf * 1.5f);
    a.err = b->err.p;
    a.stats = b->stats.p;
    a.bm_site = b->bm_site.p;
    a.bm_ok = b->bm_ok.p;
    a.n_nodes = b->n_nodes.p;
    a.site_rec = b->site_rec.n ? reinterpret_cast<SiteRec*>(b->site_rec.p) : nullptr;
    hipStream_t s = job->ctx->stream;
    // 1. membership bitmap of the pass's input, every CU
    int rc = job->t_scan.begin(s, b->total_slots);
    if (rc != GP_OK)
        return rc;
    hipLaunchKernelGGL(ntedit_scan, dim3((unsigned)((b->total_slots + kScanTile - 1) / kScanTile)),
                       dim3(256), 0, s, a, job->n_contigs, b->bm_site.p, b->bm_ok.p);
    GP_HIP(hipGetLastError());
    if (a.site_rec) {
        // 1b. what the trials decide at every confirmed site of the untouched contigs
        GP_HIP(hipMemsetAsync(b->site_rec.p, 0, b->site_rec.n * sizeof(uint4), s));
        // (128-thread workgroups: a site's full trials take the whole workgroup, one site after the
        // other, so more and smaller workgroups wait less -- 82.6 -> 77.0 ms of scan + evaluation per
        // configs[2] step; k <= 56 so that a site's window fits; GP_EVAL_THREADS=256 for the old shape)
        const unsigned eval_threads = env_u64("GP_EVAL_THREADS", 128) == 128 && a.k <= 56 ? 128u : (unsigned)kEvalTile;
        hipLaunchKernelGGL(ntedit_site_eval, dim3(b->n_tiles * (kMatTile / eval_threads)), dim3(eval_threads), 0, s,
                           a, (const uint32_t*)b->tile_contig.p, (const uint32_t*)b->tile_first.p);
        GP_HIP(hipGetLastError());
    }
    if ((rc = job->t_scan.end(s)) != GP_OK)
        return rc;
    // 2. the ordered walk, one workgroup per contig; 3. copy the edited contigs out
    if ((rc = job->t_walk.begin(s, job->h_contig_off[job->n_contigs])) != GP_OK)
        return rc;
    struct TimerScope { // every way out of the walk section closes the timer's interval
        KernelTimer& t;
        hipStream_t s;
        bool open;
        ~TimerScope()
        {
            if (open)
                (void)t.end(s);
        }
    } walk_timer{ job->t_walk, s, true };
    // 2a. contigs of several segment lengths are cut into segments, each walked by a workgroup of
    // its own (SegDesc above).  GP_WALK_SEG: k-mer starts per segment (0: never cut), default 4096.  SURVEY 8d
    // config 3's shape, walk per step: 37 s uncut, 4.2 s with 65536, 2.5 s with 32768, 2.33 s with 16384 (the
    // default up to round 4), 2.02 s with 8192, 1.93 s with 4096, 1.77 s with 2048 (there 44 segments of 48 000 fail
    // and are repaired, 12 of 24 000 at 4096).  The gain is small because a launch lasts as long as its slowest
    // segment, and the slowest are stretches edited every other base where no cut is found at any spacing
    // (9 000 sites in 18 000 starts at 22-30 us each: 0.55 s at 16384, 0.42 s at 4096), plus one contig per
    // pass walked whole after a refused insertion (0.8 s at k = 20, whatever the spacing).
    const uint64_t seg_len = env_u64("GP_WALK_SEG", 4096);
    std::vector<SegDesc> h_segs;
    std::vector<SegContig> h_segc;
    std::vector<uint32_t> h_len; // the contigs' lengths going into this pass (filled when a contig is cut)
    {
        // only contigs of GP_WALK_SEG_MIN starts and more (default 131072; two segment lengths at least)
        // are cut: a job of thousands of 30-100 kbp contigs fills the chip with one workgroup per
        // contig, and cutting them would only add descriptors, copies and host round trips
        const uint64_t seg_min = std::max<uint64_t>(2 * seg_len, env_u64("GP_WALK_SEG_MIN", 131072));
        bool any_long = false;
        for (uint32_t c = 0; seg_len && a.site_rec && c < job->n_contigs && !any_long; c++)
            any_long = job->h_contig_off[c + 1] - job->h_contig_off[c] >= seg_min;
        if (any_long) {
            // the contigs' current lengths (edits of the earlier passes have changed them)
            h_len.assign(job->n_contigs, 0);
            std::vector<uint8_t> h_drop(job->n_contigs);
            GP_HIP(hipMemcpyAsync(h_len.data(), a.len_cur, (size_t)job->n_contigs * 4, hipMemcpyDeviceToHost, s));
            GP_HIP(hipMemcpyAsync(h_drop.data(), a.dropped, job->n_contigs, hipMemcpyDeviceToHost, s));
            GP_HIP(hipStreamSynchronize(s));
            const uint32_t half_window = (uint32_t)std::min<uint64_t>(8192, std::max<uint64_t>(1024, seg_len / 8));
            std::vector<BoundaryReq> h_req;
            std::vector<uint32_t> req_first(job->n_contigs + 1, 0);
            for (uint32_t c = 0; c < job->n_contigs; c++) {
                req_first[c] = (uint32_t)h_req.size();
                if (h_drop[c] || h_len[c] < seg_min)
                    continue;
                for (uint64_t p = seg_len; p + seg_len <= h_len[c]; p += seg_len)
                    h_req.push_back({ c, (uint32_t)p, half_window, 0 });
            }
            req_first[job->n_contigs] = (uint32_t)h_req.size();
            if (!h_req.empty()) {
                std::vector<uint32_t> h_cut(2 * h_req.size());
                if ((rc = b->seg_req.ensure(h_req.size() * sizeof(BoundaryReq))) != GP_OK ||
                    (rc = b->seg_cut.ensure(2 * h_req.size())) != GP_OK)
                    return rc;
                GP_HIP(hipMemcpyAsync(b->seg_req.p, h_req.data(), h_req.size() * sizeof(BoundaryReq), hipMemcpyHostToDevice, s));
                hipLaunchKernelGGL(pick_boundaries, dim3((unsigned)h_req.size()), dim3(256), 0, s, a,
                                   reinterpret_cast<const BoundaryReq*>(b->seg_req.p), b->seg_cut.p);
                GP_HIP(hipGetLastError());
                GP_HIP(hipMemcpyAsync(h_cut.data(), b->seg_cut.p, h_cut.size() * 4, hipMemcpyDeviceToHost, s));
                GP_HIP(hipStreamSynchronize(s));
                if (getenv("GP_WALK_SEG_DEBUG")) {
                    size_t found = 0;
                    for (size_t i = 0; i < h_req.size(); i++)
                        found += h_cut[2 * i] != 0xFFFFFFFFu;
                    fprintf(stderr, "WALKSEG k=%u: %zu cuts wanted, %zu found\n", a.k, h_req.size(), found);
                }
                const uint32_t force = (uint32_t)env_u64("GP_WALK_SEG_FORCE", 0); // tests: cut at the nominal positions whatever the records say
                // records saying "edit" per window position above which a contig is not cut (GP_WALK_SEG_DENSE: one in so many)
                // (off by default: on SURVEY 8d's 100 Mbp of 0.3-12 Mbp contigs the contigs that cost most ARE
                // the densely edited ones, and cutting them took the walk from 37 s to 6 s per step
                // in spite of their failed segments; leaving them whole gave the 37 s back)
                const uint64_t dense = env_u64("GP_WALK_SEG_DENSE", 0);
                for (uint32_t c = 0; c < job->n_contigs; c++) {
                    std::vector<uint32_t> cuts;
                    uint64_t stops = 0;
                    for (uint32_t i = req_first[c]; i < req_first[c + 1]; i++)
                        stops += h_cut[2 * i + 1];
                    if (!force && dense && stops * dense > (uint64_t)(req_first[c + 1] - req_first[c]) * 2 * half_window)
                        continue;
                    for (uint32_t i = req_first[c]; i < req_first[c + 1]; i++) {
                        const uint32_t at = force ? h_req[i].nominal : h_cut[2 * i];
                        const uint32_t prev = cuts.empty() ? 0u : cuts.back();
                        if (at != 0xFFFFFFFFu && at >= prev + 4 * kSegGuard && (uint64_t)at + 4 * kSegGuard < h_len[c])
                            cuts.push_back(at);
                    }
                    if (cuts.empty())
                        continue;
                    const uint64_t region = b->h_node_off[c + 1] - b->h_node_off[c];
                    SegContig sc{ c, (uint32_t)h_segs.size(), (uint32_t)cuts.size() + 1, 0 };
                    auto node_at = [&](uint32_t p) { return (uint32_t)(region * (uint64_t)p / h_len[c]); };
                    for (size_t j = 0; j <= cuts.size(); j++) {
                        SegDesc d;
                        d.cid = c;
                        d.p_begin = j ? cuts[j - 1] : 0u;
                        d.p_end = j < cuts.size() ? cuts[j] : 0xFFFFFFFFu;
                        d.node_first = node_at(d.p_begin);
                        d.node_cap = (j < cuts.size() ? node_at(cuts[j]) : (uint32_t)region) - d.node_first;
                        d.resume_from = 0xFFFFFFFFu;
                        d.pad = 0;
                        h_segs.push_back(d);
                    }
                    h_segc.push_back(sc);
                }
            }
        }
    }
    if (!h_segs.empty()) {
        std::vector<uint32_t> h_mode(job->n_contigs, 0);
        for (const SegContig& sc : h_segc)
            h_mode[sc.cid] = 1;
        if ((rc = b->seg_desc.ensure(h_segs.size() * sizeof(SegDesc))) != GP_OK ||
            (rc = b->seg_out.ensure(h_segs.size() * sizeof(SegOut))) != GP_OK ||
            (rc = b->seg_state.ensure(2 * h_segs.size() * sizeof(SegState))) != GP_OK ||
            (rc = b->seg_contigs.ensure(h_segc.size() * sizeof(SegContig))) != GP_OK ||
            (rc = b->contig_mode.ensure(job->n_contigs)) != GP_OK)
            return rc;
        GP_HIP(hipMemcpyAsync(b->seg_desc.p, h_segs.data(), h_segs.size() * sizeof(SegDesc), hipMemcpyHostToDevice, s));
        GP_HIP(hipMemcpyAsync(b->seg_contigs.p, h_segc.data(), h_segc.size() * sizeof(SegContig), hipMemcpyHostToDevice, s));
        GP_HIP(hipMemcpyAsync(b->contig_mode.p, h_mode.data(), h_mode.size() * 4, hipMemcpyHostToDevice, s));
        GP_HIP(hipMemsetAsync(b->seg_out.p, 0, h_segs.size() * sizeof(SegOut), s));
        // the pass's input of every segmented contig, kept in its (still unused) output slot: what a
        // failed segment's contig is walked from again
        for (const SegContig& sc : h_segc) {
            const uint64_t o = b->h_off[sc.cid];
            GP_HIP(hipMemcpyAsync(a.out + o, a.seq + o, b->h_off[sc.cid + 1] - o, hipMemcpyDeviceToDevice, s));
        }
        GP_HIP(hipStreamSynchronize(s)); // the host vectors above are gone when this function returns
        a.segs = reinterpret_cast<const SegDesc*>(b->seg_desc.p);
        a.seg_out = reinterpret_cast<SegOut*>(b->seg_out.p);
        a.seg_state = reinterpret_cast<SegState*>(b->seg_state.p);
        a.contig_mode = b->contig_mode.p;
        a.walk_phase = 0;
    }
    DevBuf<unsigned long long> d_prof;
    if (getenv("GP_WALK_PROFILE")) {
        if ((rc = d_prof.alloc(((uint64_t)job->n_contigs + 8192) * 16)) != GP_OK)
            return rc;
        GP_HIP(hipMemsetAsync(d_prof.p, 0, ((uint64_t)job->n_contigs + 8192) * 16 * 8, s));
        a.prof = d_prof.p;
        a.n_prof_contigs = job->n_contigs;
    }
    // waves per SIMD the walk is compiled for: many contigs -> more workgroups per CU at fewer
    // registers (GP_WALK_WAVES overrides: 1 or 4; compiled for 6 the kernel no longer fits 80 registers and was slower: 170 against 149 ms on 4000 x 25 kbp)
    const uint64_t walk_waves = env_u64("GP_WALK_WAVES", job->n_contigs > 3 * 256 ? 4 : 1); // (measured on 1000 x 25 kbp: 6 -> 52 ms, 5 -> 50, 4 -> 45, 1 -> 63; on 4000 x 25 kbp: 4 -> 150, 3 (168 registers, 14 spilled) -> 174, 1 -> 216)
    // Threads per walking workgroup: the walk is thread 0's chain of decisions with all lanes helping in
    // bursts (a scan after an edit covers k positions, a site's window 2k + 12 characters), and what
    // bounds a launch of thousands of contigs is how many of them a CU holds -- at 128 threads twice as
    // many for the same registers.  (A site's window must fit the workgroup: k <= 56.)
    const unsigned walk_threads =
        (unsigned)env_u64("GP_WALK_THREADS", (walk_waves >= 4 && a.k <= 56) ? 128 : kWalkThreads) == 128 && a.k <= 56 ? 128u : (unsigned)kWalkThreads;
    // The segments of long contigs are thousands of workgroups as well (SURVEY 8d config 3's shape: 6 100 per k),
    // but twice as many of them resident (GP_WALK_SEG_WAVES=4: 128 threads each, eight per CU instead of four)
    // buys nothing there -- walk 2.33 s per step against 2.22: a launch lasts as long as its slowest segments
    // and the contigs walked whole after a failed repair, not as long as the queue of the others
    const bool seg_tight = h_segs.size() > 3 * 256 && env_u64("GP_WALK_SEG_WAVES", 1) >= 4;
    const unsigned walk_threads_seg = (env_u64("GP_WALK_SEG_THREADS", 128) == 128 && a.k <= 56) ? 128u : (unsigned)kWalkThreads;
    auto launch_walk = [&](unsigned grid, bool seg) {
        if (seg && seg_tight)
            hipLaunchKernelGGL((ntedit_walk<4, true>), dim3(grid), dim3(walk_threads_seg), 0, s, a);
        else if (seg)
            hipLaunchKernelGGL((ntedit_walk<1, true>), dim3(grid), dim3(kWalkThreads), 0, s, a);
        else if (walk_waves >= 6)
            hipLaunchKernelGGL((ntedit_walk<6, false>), dim3(grid), dim3(walk_threads), 0, s, a);
        else if (walk_waves == 5)
            hipLaunchKernelGGL((ntedit_walk<5, false>), dim3(grid), dim3(walk_threads), 0, s, a);
        else if (walk_waves >= 4)
            hipLaunchKernelGGL((ntedit_walk<4, false>), dim3(grid), dim3(walk_threads), 0, s, a);
        else
            hipLaunchKernelGGL((ntedit_walk<1, false>), dim3(grid), dim3(kWalkThreads), 0, s, a);
    };
    const bool seg_dbg = getenv("GP_WALK_SEG_DEBUG") != nullptr;
    auto lap = [&](const char* what) { // development aid: wall time of a step of the segmented walk
        static thread_local std::chrono::steady_clock::time_point t0;
        if (!seg_dbg)
            return;
        (void)hipStreamSynchronize(s);
        const auto now = std::chrono::steady_clock::now();
        if (what)
            fprintf(stderr, "WALKSEG k=%u: %s %.2f ms\n", a.k, what, std::chrono::duration<double, std::milli>(now - t0).count());
        t0 = now;
    };
    lap(nullptr);
    launch_walk(job->n_contigs, false);
    lap("whole contigs");
    uint32_t n_repairs = 0;
    if (!h_segs.empty()) {
        a.seg_base = 0;
        launch_walk((unsigned)h_segs.size(), true);
        lap("segments");
        // A failed segment goes ON, from the state its walker was left in, through what follows it up
        // to and including the first segment that did not fail (whose end the right neighbours
        // were built on); the bases of that stretch come back from the pass's copy first.  One that
        // fails again sends the whole contig to the one-workgroup walk (merge_segments).
        std::vector<SegOut> so(h_segs.size());
        GP_HIP(hipMemcpyAsync(so.data(), b->seg_out.p, so.size() * sizeof(SegOut), hipMemcpyDeviceToHost, s));
        GP_HIP(hipStreamSynchronize(s));
        std::vector<uint32_t> order;
        std::vector<RestoreReq> restores;
        // (len_cur is this pass's INPUT length: the walk writes len_next, so the copy made for the
        // cut planning still holds)
        auto contig_len = [&](uint32_t cid) -> uint32_t { return h_len[cid]; };
        const size_t n_first = h_segs.size();
        for (SegContig& sc : h_segc) {
            const uint32_t first = sc.seg_first, n = sc.n_seg;
            sc.seg_first = (uint32_t)order.size();
            bool gave_up = false;
            for (uint32_t j = 0; j < n; j++)
                gave_up |= so[first + j].failed == 3;
            for (uint32_t j = 0; j < n;) {
                const uint32_t f = so[first + j].failed;
                if (f == 0 || f == 4 || gave_up) { // (a contig with a segment that gave up is walked whole: merge_segments sees the flag)
                    order.push_back(first + j);
                    j++;
                    if (f == 4 && !gave_up)
                        break; // the contig's walk ended inside this segment: what lies behind it is dropped below
                    continue;
                }
                uint32_t m = j + 1;
                while (m < n && (so[first + m].failed == 1 || so[first + m].failed == 2))
                    m++;
                if (m >= n)
                    m = n - 1; // (the last segment cannot fail; kept for safety)
                SegDesc d = h_segs[first + j];
                // the bases of the stretch it goes on through come back from the pass's copy: what
                // segments j + 1 .. m wrote there is void (its own bases stay as its walker left them)
                const uint32_t b_lo = d.p_end + a.k - 1;
                d.p_end = h_segs[first + m].p_end;
                d.node_cap = h_segs[first + m].node_first + h_segs[first + m].node_cap - d.node_first;
                d.resume_from = first + j;
                const uint32_t len_c = contig_len(d.cid);
                const uint32_t b_hi = d.p_end != 0xFFFFFFFFu ? std::min<uint32_t>(len_c, d.p_end + a.k - 1) : len_c;
                restores.push_back({ d.cid, std::min(b_lo, b_hi), b_hi, 0 });
                order.push_back((uint32_t)h_segs.size());
                h_segs.push_back(d);
                j = m + 1;
            }
            sc.n_seg = (uint32_t)order.size() - sc.seg_first;
        }
        n_repairs = (uint32_t)(h_segs.size() - n_first);
        if ((rc = b->seg_order.ensure(order.size())) != GP_OK)
            return rc;
        if (n_repairs) {
            // (the descriptor and result arrays grow: what round one left in them is kept)
            DevBuf<uint8_t> nd, no;
            if ((rc = nd.alloc(h_segs.size() * sizeof(SegDesc))) != GP_OK || (rc = no.alloc(h_segs.size() * sizeof(SegOut))) != GP_OK ||
                (rc = b->seg_req.ensure(restores.size() * sizeof(RestoreReq))) != GP_OK)
                return rc;
            GP_HIP(hipMemcpyAsync(nd.p, h_segs.data(), h_segs.size() * sizeof(SegDesc), hipMemcpyHostToDevice, s));
            GP_HIP(hipMemsetAsync(no.p, 0, h_segs.size() * sizeof(SegOut), s));
            GP_HIP(hipMemcpyAsync(no.p, b->seg_out.p, n_first * sizeof(SegOut), hipMemcpyDeviceToDevice, s));
            GP_HIP(hipMemcpyAsync(b->seg_req.p, restores.data(), restores.size() * sizeof(RestoreReq), hipMemcpyHostToDevice, s));
            GP_HIP(hipStreamSynchronize(s));
            std::swap(b->seg_desc.p, nd.p);
            std::swap(b->seg_desc.n, nd.n);
            std::swap(b->seg_desc.cap_bytes, nd.cap_bytes);
            std::swap(b->seg_out.p, no.p);
            std::swap(b->seg_out.n, no.n);
            std::swap(b->seg_out.cap_bytes, no.cap_bytes);
            a.segs = reinterpret_cast<const SegDesc*>(b->seg_desc.p);
            a.seg_out = reinterpret_cast<SegOut*>(b->seg_out.p);
            hipLaunchKernelGGL(restore_ranges, dim3(64, (unsigned)restores.size()), dim3(256), 0, s, a,
                               reinterpret_cast<const RestoreReq*>(b->seg_req.p));
            a.seg_base = (uint32_t)n_first;
            launch_walk(n_repairs, true);
            lap("resumes");
            so.resize(h_segs.size());
            GP_HIP(hipMemcpyAsync(so.data(), b->seg_out.p, so.size() * sizeof(SegOut), hipMemcpyDeviceToHost, s));
            GP_HIP(hipStreamSynchronize(s));
        }
        // A walk that ENDED inside a segment (code 4) ended its contig: the descriptors behind it are
        // dropped and the bases they edited come back from the pass's copy.
        {
            std::vector<RestoreReq> tails;
            std::vector<uint32_t> kept;
            for (SegContig& sc : h_segc) {
                const uint32_t o0 = sc.seg_first, n = sc.n_seg;
                sc.seg_first = (uint32_t)kept.size();
                for (uint32_t j = 0; j < n; j++) {
                    const uint32_t di = order[o0 + j];
                    kept.push_back(di);
                    if (so[di].failed == 4) {
                        const uint32_t len_c = contig_len(sc.cid);
                        if (h_segs[di].p_end != 0xFFFFFFFFu) // (segments lay behind it: they ran, whether they are still listed or not)
                            tails.push_back({ sc.cid, std::min<uint32_t>(len_c, h_segs[di].p_end + a.k - 1), len_c, 0 });
                        break;
                    }
                }
                sc.n_seg = (uint32_t)kept.size() - sc.seg_first;
            }
            order.swap(kept);
            if (!tails.empty()) {
                // (seg_req may still be read by the restore of the resumes: a buffer of its own)
                if ((rc = b->seg_req2.ensure(tails.size() * sizeof(RestoreReq))) != GP_OK)
                    return rc;
                GP_HIP(hipMemcpyAsync(b->seg_req2.p, tails.data(), tails.size() * sizeof(RestoreReq), hipMemcpyHostToDevice, s));
                GP_HIP(hipStreamSynchronize(s));
                hipLaunchKernelGGL(restore_ranges, dim3(64, (unsigned)tails.size()), dim3(256), 0, s, a,
                                   reinterpret_cast<const RestoreReq*>(b->seg_req2.p));
            }
        }
        GP_HIP(hipMemcpyAsync(b->seg_order.p, order.data(), order.size() * 4, hipMemcpyHostToDevice, s));
        GP_HIP(hipMemcpyAsync(b->seg_contigs.p, h_segc.data(), h_segc.size() * sizeof(SegContig), hipMemcpyHostToDevice, s));
        GP_HIP(hipStreamSynchronize(s));
        a.seg_order = b->seg_order.p;
        hipLaunchKernelGGL(merge_segments, dim3((unsigned)h_segc.size()), dim3(256), 0, s, a,
                           reinterpret_cast<const SegContig*>(b->seg_contigs.p), b->contig_mode.p);
        lap("merge");
        a.walk_phase = 1; // contigs a failed repair sent back: whole, from the restored bases
        launch_walk(job->n_contigs, false);
        lap("rerun");
        if (seg_dbg) {
            GP_HIP(hipStreamSynchronize(s));
            std::vector<SegOut> so2(h_segs.size());
            std::vector<uint32_t> mode(job->n_contigs);
            GP_HIP(hipMemcpy(so2.data(), b->seg_out.p, so2.size() * sizeof(SegOut), hipMemcpyDeviceToHost));
            GP_HIP(hipMemcpy(mode.data(), b->contig_mode.p, mode.size() * 4, hipMemcpyDeviceToHost));
            uint32_t f1 = 0, f2 = 0, fr = 0, whole = 0;
            for (size_t i = 0; i < so2.size(); i++) {
                if (i < h_segs.size() - n_repairs) {
                    f1 += so2[i].failed == 1;
                    f2 += so2[i].failed == 2;
                } else {
                    fr += so2[i].failed != 0;
                }
            }
            for (uint32_t m : mode)
                whole += m == 2;
            {
                std::vector<SegState> stv(so2.size());
                GP_HIP(hipMemcpy(stv.data(), b->seg_state.p, stv.size() * sizeof(SegState), hipMemcpyDeviceToHost));
                for (size_t i = h_segs.size() - n_repairs; i < so2.size(); i++)
                    if (so2[i].failed)
                        fprintf(stderr, "WALKSEG k=%u: resume %zu (contig %u, starts %u..%u, from %u whose head stood at %u) failed with %u, head now %u\n",
                                a.k, i, h_segs[i].cid, h_segs[i].p_begin, h_segs[i].p_end, h_segs[i].resume_from,
                                stv[h_segs[i].resume_from].hi, so2[i].failed, stv[i].hi);
            }
            fprintf(stderr, "WALKSEG k=%u: %zu contigs in %zu segments; failed: %u (walker not clean near the end) + %u (edit near "
                            "the end); %u repairs, %u of them failed; contigs walked whole again: %u\n",
                    a.k, h_segc.size(), h_segs.size() - n_repairs, f1, f2, n_repairs, fr, whole);
        }
    }
    hipLaunchKernelGGL(materialize_kernel, dim3(b->n_tiles), dim3(256), 0, s,
                       (const uint8_t*)a.seq, a.out, a.off, (const uint32_t*)a.len_next,
                       (const uint8_t*)a.dropped, (const Node*)a.nodes, a.node_off,
                       (const uint32_t*)a.n_nodes, (const uint32_t*)b->tile_contig.p,
                       (const uint32_t*)b->tile_first.p);
    GP_HIP(hipGetLastError());
    walk_timer.open = false;
    if ((rc = job->t_walk.end(s)) != GP_OK)
        return rc;
    if (a.prof) {
        // development aid: the contigs whose walk took longest, and on which requests
        GP_HIP(hipStreamSynchronize(s));
        std::vector<unsigned long long> h((size_t)job->n_contigs * 16);
        GP_HIP(hipMemcpy(h.data(), d_prof.p, h.size() * 8, hipMemcpyDeviceToHost));
        if (!h_segs.empty()) {
            const size_t ns = std::min<size_t>(h_segs.size(), 8192);
            std::vector<unsigned long long> hs(ns * 16);
            GP_HIP(hipMemcpy(hs.data(), d_prof.p + (size_t)job->n_contigs * 16, hs.size() * 8, hipMemcpyDeviceToHost));
            std::vector<std::pair<double, size_t>> t;
            for (size_t i = 0; i < ns; i++) {
                double ms = 0;
                for (int q = 0; q < 8; q++)
                    ms += (double)hs[i * 16 + q] / 1e5;
                t.push_back({ ms, i });
            }
            std::sort(t.rbegin(), t.rend());
            for (size_t r = 0; r < std::min<size_t>(5, t.size()); r++) {
                const size_t i = t[r].second;
                fprintf(stderr, "WALKPROF k=%u segment %zu (contig %u, starts %u..%u)%s: %.1f ms = site %.1f ms / %llu, next %.1f ms / %llu\n", a.k, i,
                        h_segs[i].cid, h_segs[i].p_begin, h_segs[i].p_end, i >= h_segs.size() - n_repairs ? " [repair]" : "", t[r].first,
                        (double)hs[i * 16 + 2] / 1e5, hs[i * 16 + 8 + 2], (double)hs[i * 16 + 3] / 1e5, hs[i * 16 + 8 + 3]);
            }
        }
        std::vector<uint32_t> order(job->n_contigs);
        std::vector<double> tot(job->n_contigs, 0);
        for (uint32_t c = 0; c < job->n_contigs; c++) {
            order[c] = c;
            for (int i = 0; i < 8; i++)
                tot[c] += (double)h[(size_t)c * 16 + i] / 1e5; // ms at 100 MHz
        }
        std::sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return tot[x] > tot[y]; });
        static const char* names[8] = { "done", "scan", "site", "next", "bitscan", "refill", "sweep", "-" };
        {
            double sum_t[8] = {}, all = 0;
            unsigned long long sum_n[8] = {};
            for (uint32_t c = 0; c < job->n_contigs; c++)
                for (int i = 0; i < 8; i++) {
                    sum_t[i] += (double)h[(size_t)c * 16 + i] / 1e5;
                    sum_n[i] += h[(size_t)c * 16 + 8 + i];
                    all += (double)h[(size_t)c * 16 + i] / 1e5;
                }
            fprintf(stderr, "WALKPROF k=%u all %u contigs: %.1f ms of workgroup time =", a.k, job->n_contigs, all);
            for (int i = 1; i < 7; i++)
                fprintf(stderr, " %s %.1f ms / %llu (%.1f us each)", names[i], sum_t[i], sum_n[i], sum_n[i] ? sum_t[i] * 1e3 / sum_n[i] : 0.0);
            fprintf(stderr, "\n");
        }
        for (uint32_t r = 0; r < std::min<uint32_t>(6, job->n_contigs); r++) {
            const uint32_t c = order[r];
            fprintf(stderr, "WALKPROF k=%u contig %u len %llu: %.1f ms =", a.k, c,
                    (unsigned long long)(job->h_contig_off[c + 1] - job->h_contig_off[c]), tot[c]);
            for (int i = 1; i < 7; i++)
                fprintf(stderr, " %s %.1f ms / %llu", names[i], (double)h[(size_t)c * 16 + i] / 1e5, h[(size_t)c * 16 + 8 + i]);
            fprintf(stderr, "\n");
        }
    }
    b->cur ^= 1;
    return GP_OK;
}

int stage_b_guard(gp_job* job)
{
    StageBC* b = job->bc;
    if (job->n_contigs == 0)
        return GP_OK;
    hipStream_t s = job->ctx->stream;
    hipLaunchKernelGGL(guard_kernel, dim3((job->n_batches + 255) / 256), dim3(256), 0, s,
                       (const uint32_t*)b->batch_begin.p, job->n_batches,
                       (const uint32_t*)b->name_len.p, (const uint32_t*)b->in_len.p,
                       (const uint32_t*)b->len[b->cur].p, (const uint8_t*)b->dropped.p,
                       b->batch_skip.p);
    hipLaunchKernelGGL(copy_contigs, dim3(b->n_tiles), dim3(256), 0, s,
                       (const uint8_t*)job->d_contigs_in.p, (const uint64_t*)b->in_off.p,
                       (const uint32_t*)b->in_len.p, b->seq[b->cur].p, (const uint64_t*)b->off.p,
                       b->len[b->cur].p, b->dropped.p, (const uint8_t*)b->batch_skip.p,
                       (const uint32_t*)b->contig_batch.p, (const uint32_t*)b->tile_contig.p,
                       (const uint32_t*)b->tile_first.p);
    GP_HIP(hipGetLastError());
    return GP_OK;
}

int stage_mask(gp_job* job)
{
    StageBC* b = job->bc;
    if (job->n_contigs == 0)
        return GP_OK;
    hipStream_t s = job->ctx->stream;
    int rc = job->t_mask.begin(s, job->n_contigs);
    if (rc != GP_OK)
        return rc;
    // contigs without anything to mask are copied tile-parallel; the rest go through the FSM
    if ((rc = launch_classify(job)) != GP_OK || (rc = launch_copy_flagged(job, 1u, job->prm.k[0])) != GP_OK)
        return rc;
    hipLaunchKernelGGL(mask_kernel, dim3(job->n_contigs), dim3(kMaskThreads), 0, s,
                       (const uint8_t*)b->seq[b->cur].p, (const uint64_t*)b->off.p,
                       (const uint32_t*)b->len[b->cur].p, b->seq[b->cur ^ 1].p,
                       b->len[b->cur ^ 1].p, (const uint8_t*)b->dropped.p, job->prm.k[0],
                       (const uint32_t*)b->flags.p);
    GP_HIP(hipGetLastError());
    if ((rc = job->t_mask.end(s)) != GP_OK)
        return rc;
    b->cur ^= 1;
    return GP_OK;
}

int launch_classify(gp_job* job)
{
    StageBC* b = job->bc;
    hipStream_t s = job->ctx->stream;
    GP_HIP(hipMemsetAsync(b->flags.p, 0, job->n_contigs * sizeof(uint32_t), s));
    hipLaunchKernelGGL(classify_kernel, dim3(b->n_tiles), dim3(256), 0, s,
                       (const uint8_t*)b->seq[b->cur].p, (const uint64_t*)b->off.p,
                       (const uint32_t*)b->len[b->cur].p, (const uint8_t*)b->dropped.p,
                       (const uint32_t*)b->tile_contig.p, (const uint32_t*)b->tile_first.p,
                       b->flags.p);
    GP_HIP(hipGetLastError());
    return GP_OK;
}

int launch_copy_flagged(gp_job* job, uint32_t reject, uint32_t min_len)
{
    StageBC* b = job->bc;
    hipStream_t s = job->ctx->stream;
    hipLaunchKernelGGL(copy_flagged, dim3(b->n_tiles), dim3(256), 0, s,
                       (const uint8_t*)b->seq[b->cur].p, b->seq[b->cur ^ 1].p,
                       (const uint64_t*)b->off.p, (const uint32_t*)b->len[b->cur].p,
                       b->len[b->cur ^ 1].p, (const uint8_t*)b->dropped.p,
                       (const uint32_t*)b->tile_contig.p, (const uint32_t*)b->tile_first.p,
                       (const uint32_t*)b->flags.p, reject, min_len);
    GP_HIP(hipGetLastError());
    return GP_OK;
}

int stage_upper(gp_job* job)
{
    StageBC* b = job->bc;
    if (job->n_contigs == 0)
        return GP_OK;
    hipStream_t s = job->ctx->stream;
    int rc0 = job->t_upper.begin(s, job->n_contigs);
    if (rc0 != GP_OK)
        return rc0;
    hipLaunchKernelGGL(upper_kernel, dim3(64, job->n_contigs), dim3(256), 0, s, b->seq[b->cur].p,
                       (const uint64_t*)b->off.p, (const uint32_t*)b->len[b->cur].p,
                       (const uint8_t*)b->dropped.p);
    GP_HIP(hipGetLastError());
    if ((rc0 = job->t_upper.end(s)) != GP_OK)
        return rc0;
    return pull_stats(job);
}

int stage_lengths(gp_job* job, uint64_t* lengths, uint8_t* dropped, uint64_t* total)
{
    StageBC* b = job->bc;
    const uint32_t n = job->n_contigs;
    GP_HIP(hipStreamSynchronize(job->ctx->stream));
    int rc = pull_stats(job);
    if (rc != GP_OK)
        return rc;
    std::vector<uint32_t> len(n ? n : 1);
    std::vector<uint8_t> dr(n ? n : 1);
    if (n) {
        GP_HIP(hipMemcpyAsync(len.data(), b->len[b->cur].p, n * sizeof(uint32_t), hipMemcpyDeviceToHost, job->ctx->stream));
        GP_HIP(hipMemcpyAsync(dr.data(), b->dropped.p, n, hipMemcpyDeviceToHost, job->ctx->stream));
        GP_HIP(hipStreamSynchronize(job->ctx->stream));
    }
    uint64_t t = 0;
    for (uint32_t c = 0; c < n; c++) {
        uint64_t l = dr[c] ? 0 : len[c];
        if (lengths)
            lengths[c] = l;
        if (dropped)
            dropped[c] = dr[c];
        t += l;
    }
    if (total)
        *total = t;
    return GP_OK;
}

// Concatenates the surviving records (input order, no separators) at `dst`, a device pointer.
__global__ void pack_kernel(uint8_t* __restrict__ dst, const uint8_t* __restrict__ seq,
                            const uint64_t* __restrict__ off, const uint64_t* __restrict__ out_off)
{
    const uint32_t cid = blockIdx.y;
    const uint64_t n = out_off[cid + 1] - out_off[cid];
    const uint8_t* s = seq + off[cid];
    uint8_t* d = dst + out_off[cid];
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
         i += (uint64_t)gridDim.x * blockDim.x)
        d[i] = s[i];
}

static int pack_to_device(gp_job* job, uint8_t* d_dst, uint64_t dst_bytes, const char* who)
{
    StageBC* b = job->bc;
    const uint32_t n = job->n_contigs;
    std::vector<uint64_t> lens(n ? n : 1);
    uint64_t total = 0;
    int rc = stage_lengths(job, lens.data(), nullptr, &total);
    if (rc != GP_OK)
        return rc;
    if (total > dst_bytes) {
        set_error(std::string(who) + ": buffer too small");
        return GP_ERR_INVALID_ARG;
    }
    if (!n || !total)
        return GP_OK;
    std::vector<uint64_t> out_off(n + 1, 0);
    for (uint32_t c = 0; c < n; c++)
        out_off[c + 1] = out_off[c] + lens[c];
    if (b->pack_off.n < n + 1 && (rc = b->pack_off.alloc(n + 1)) != GP_OK)
        return rc;
    hipStream_t s = job->ctx->stream;
    GP_HIP(hipMemcpyAsync(b->pack_off.p, out_off.data(), (n + 1) * sizeof(uint64_t),
                          hipMemcpyHostToDevice, s));
    const uint32_t gx = n >= 1024 ? 4 : (n >= 64 ? 64 : 1024);
    hipLaunchKernelGGL(pack_kernel, dim3(gx, n), dim3(256), 0, s, d_dst,
                       (const uint8_t*)b->seq[b->cur].p, (const uint64_t*)b->off.p,
                       (const uint64_t*)b->pack_off.p);
    GP_HIP(hipGetLastError());
    GP_HIP(hipStreamSynchronize(s));
    return GP_OK;
}

int stage_sequences_device(gp_job* job, void* d_buf, uint64_t buf_bytes)
{
    return pack_to_device(job, (uint8_t*)d_buf, buf_bytes, "gp_job_get_sequences_device");
}

int stage_sequences(gp_job* job, uint8_t* buf, uint64_t buf_bytes)
{
    StageBC* b = job->bc;
    uint64_t total = 0;
    int rc = stage_lengths(job, nullptr, nullptr, &total);
    if (rc != GP_OK)
        return rc;
    if (total > buf_bytes) {
        set_error("gp_job_get_sequences: buffer too small");
        return GP_ERR_INVALID_ARG;
    }
    if (!total)
        return GP_OK;
    if (b->pack_buf.n < total && (rc = b->pack_buf.alloc(total)) != GP_OK)
        return rc;
    if ((rc = pack_to_device(job, b->pack_buf.p, total, "gp_job_get_sequences")) != GP_OK)
        return rc;
    GP_HIP(hipMemcpyAsync(buf, b->pack_buf.p, total, hipMemcpyDeviceToHost, job->ctx->stream));
    GP_HIP(hipStreamSynchronize(job->ctx->stream));
    return GP_OK;
}

} // namespace gp
