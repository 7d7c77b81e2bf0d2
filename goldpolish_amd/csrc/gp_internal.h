// gp_internal.h -- host-side state behind the C ABI (include/gpcore.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/gpcore.h"
#include "gp_device.h"

namespace gp {

void set_error(const std::string& msg);
void track_device_bytes(long long delta); // bytes this library holds on devices (gp_device_bytes_in_use)
int hip_fail(hipError_t e, const char* what, const char* file, int line);
int on_exception(const char* where) noexcept; // inside a catch: records the message, gives the status

// every entry point's body is a function-try-block closed by one of these: nothing is thrown
// across the C boundary (std::bad_alloc from a host-side table becomes GP_ERR_OOM)
#define GP_CATCH(where) catch (...) { return gp::on_exception(where); }
#define GP_CATCH_SIZE(where) catch (...) { (void)gp::on_exception(where); return 0; }

#define GP_HIP(call)                                                    \
    do {                                                                \
        hipError_t e__ = (call);                                        \
        if (e__ != hipSuccess)                                          \
            return gp::hip_fail(e__, #call, __FILE__, __LINE__);        \
    } while (0)

// tuning knobs read from the environment (INTEGRATION.md lists them)
inline uint64_t env_u64(const char* name, uint64_t dflt)
{
    const char* v = getenv(name);
    if (!v || !*v)
        return dflt;
    return strtoull(v, nullptr, 10);
}

// Device memory comes from a per-device pool of blocks this library has freed before (api.hip):
// a job of the same shape as an earlier one costs no hipMalloc (page tables for tens of GiB
// take about a second) and no hipFree (which waits for the whole device), which is what a
// stream of jobs through gp_pipeline_* needs.  Recycled blocks hold old bytes -- as hipMalloc'd
// memory may -- so nothing may rely on fresh memory being zero.
int dev_alloc(size_t bytes, void** p, size_t* cap_bytes);
void dev_free(void* p, size_t cap_bytes);

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    size_t cap_bytes = 0; // size of the block behind p (>= n * sizeof(T): a pooled block may be larger)
    int alloc(size_t count)
    {
        release();
        if (count == 0)
            count = 1;
        void* q = nullptr;
        const int rc = dev_alloc(count * sizeof(T), &q, &cap_bytes);
        if (rc != GP_OK)
            return rc;
        p = (T*)q;
        n = count;
        return GP_OK;
    }
    // at least `count` elements, keeping what is there when it is enough (work buffers that
    // are sized per run: a job polished again does not allocate again)
    int ensure(size_t count) { return (p && n >= count) ? GP_OK : alloc(count); }
    void release()
    {
        if (p)
            dev_free(p, cap_bytes);
        p = nullptr;
        n = 0;
        cap_bytes = 0;
    }
    ~DevBuf() { release(); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
};

struct KernelTimer {
    // pairs of events recorded around launches of one kernel family
    std::vector<hipEvent_t> start, stop;
    std::vector<uint64_t> units;
    size_t used = 0;
    double total_ms = 0;
    uint64_t launches = 0, total_units = 0;
    void reset() { used = 0; total_ms = 0; launches = 0; total_units = 0; }
    int begin(hipStream_t s, uint64_t n_units);
    int end(hipStream_t s);
    int collect(); // after a stream sync
    void destroy();
};

} // namespace gp

namespace gp {
struct StageBC;
}

struct gp_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipDeviceProp_t prop{};
    // CUs the context's streams may use (gp_ctx_create_masked; empty: all of them).  Streams the
    // stages create for themselves (the filter build's second and third) take the same mask.
    std::vector<uint32_t> cu_mask;
    // Two pinned staging chunks for uploads (gp_job_create), allocated on first use and kept for
    // the context's life: the host fills one while the DMA engine drains the other.
    uint8_t* stage[2] = { nullptr, nullptr };
    hipEvent_t stage_ev[2] = { nullptr, nullptr };
    size_t stage_bytes = 0;
    // What the filter build needs beside the context's stream -- two more streams, the events that join them,
    // the pinned feedback ring -- made by the first build on the context (stage_a_prepare) and kept for the
    // context's life: a job borrows them.  (Up to round 4 every job made and destroyed its own: 30 creations
    // and as many destructions per job, and hipHostFree / hipStreamDestroy wait for the DEVICE -- in
    // gp_pipeline that made the filter build of one set end when the back half of the set before did.)
    struct StageA {
        hipStream_t stream2 = nullptr, stream3 = nullptr;
        hipEvent_t ev[16] = {}, ev_b[16] = {}, ev_init = nullptr, ev_census[9] = {};
        uint32_t* h_count_pinned = nullptr;
        bool ready = false;
    } a;
};

// Per-k constants of the rolling hash, prepared on the host.
struct KConsts {
    uint32_t k;
    uint64_t out_fwd[4]; // sroln(seed(code), k): what leaves the forward hash
    uint64_t in_rev[4];  // sroln(seed(code ^ 2), k): what enters the reverse hash
};

struct gp_job {
    gp_ctx* ctx = nullptr;
    gp_params prm{};
    uint32_t n_batches = 0, n_contigs = 0, n_reads = 0;
    std::vector<std::string> headers;     // "id[ comment]" per contig when the job came from files

    // ---- inputs resident in HBM
    // reads: every read followed by one '\n' separator, in (batch, contig, rank)
    // order, so that a k-mer never spans two reads and "position in this buffer"
    // is the serial order the reference feeds its counting filter in.
    gp::DevBuf<uint8_t> d_reads;
    uint64_t read_positions = 0;          // bytes in d_reads
    gp::DevBuf<uint64_t> d_seg_start;     // [n_contigs+1] first position of each contig's reads
    gp::DevBuf<uint32_t> d_seg_batch;     // [n_contigs]
    gp::DevBuf<uint32_t> d_seg_thr;       // [n_contigs] kmer_threshold of the contig
    gp::DevBuf<uint32_t> d_batch_seg;     // [n_batches+1] first contig of each batch
    gp::DevBuf<uint32_t> d_seg_smax;      // [n_contigs] largest threshold of the contig and the later ones of its batch
    std::vector<uint64_t> h_seg_start;
    std::vector<uint32_t> h_batch_contig_begin, h_contig_batch, h_contig_thr, h_contig_name_len;

    gp::DevBuf<uint8_t> d_contigs_in;     // original contigs, concatenated
    std::vector<uint64_t> h_contig_off;

    // ---- filters
    gp::DevBuf<uint8_t> d_cbf;            // arena of cbf_slots counting filters (scratch of stage A, see stage_a.hip)
    uint32_t cbf_slots = 0;
    gp::DevBuf<uint32_t> d_bf;            // [n_batches][n_k][bf_bytes/4]

    // ---- stage A work buffers
    gp::DevBuf<unsigned long long> d_resv; // reservation slots (hashed), (tag << 32) | position
    uint32_t resv_bits = 0;
    gp::DevBuf<uint4> d_pending[2];       // two lists of {hash lo, hash hi, pos, meta},{idx x4}; steps alternate
    gp::DevBuf<uint32_t> d_sat, d_batch_thr_max; // saturation flags per (batch, k); max threshold per batch
    gp::DevBuf<uint64_t> d_batch_pos;     // [n_batches+1] first read position of each batch
    gp::DevBuf<uint32_t> d_batch_order;   // [n_batches] batches as stage A takes them
    gp::DevBuf<uint32_t> d_batch_slot;    // [n_batches] first arena slot of a batch of the grid-wide schedule
    gp::DevBuf<uint32_t> d_queue;         // cbf_local's pair counter
    // census of first touches (stage_a.hip): work buffers of a cohort, results for the whole job
    gp::DevBuf<unsigned long long> d_cand, d_cand2, d_kset;
    gp::DevBuf<uint8_t> d_census_cnt;
    gp::DevBuf<uint32_t> d_census_rec, d_census_over, d_census_tab32, d_pair_mode;
    gp::DevBuf<uint64_t> d_census_tab64, d_pair_bit_off;
    gp::DevBuf<uint32_t> d_counts;        // ring of per-step list lengths, one 128-B line each
    gp::DevBuf<unsigned long long> d_stats; // device-side gp_counters
    uint64_t window = 0;
    uint32_t* h_count_pinned = nullptr;
    hipEvent_t a_ev[16] = {}, a_ev_b[16] = {}; // per-step joins of the two stage-A streams
    hipEvent_t a_ev_init = nullptr;
    hipEvent_t a_ev_census[9] = {}; // scattered[3], marked[3], packed[3]: the census's buffer sets (stage_a_run)
    hipStream_t a_stream2 = nullptr, a_stream3 = nullptr;

    // ---- stage B/C state (see stage_b.hip / stage_c.hip)
    gp::StageBC* bc = nullptr;
    uint32_t bc_slack = 1; // sequence slots hold len + (len/2 + 4 KiB) * bc_slack bytes

    // small pinned host block the per-stage statistics are read back into: with several jobs in
    // flight, copies to pageable memory from two host threads went through the runtime's staging
    // path at the same time; pinned memory is written by the device directly
    unsigned long long* h_readback = nullptr; // 64 words

    gp::KernelTimer t_first_pass, t_round, t_local, t_census, t_ckc, t_scan, t_walk, t_bfs, t_mask, t_upper;
    gp_counters counters{};
    bool filters_ready = false;
    // 0: only the inputs are on the device (gp_job_create); 1: filters and work buffers allocated
    // (the first call that needs them); 2: lost -- an allocation failed while they were (re)built
    int work_state = 0;
};

namespace gp {
int ctx_stream_create(const gp_ctx* ctx, hipStream_t* out); // a non-blocking stream with the context's CU mask
// gp_pipeline's two halves of gp_job_polish (api.hip): the filter build, after which the build's own
// work buffers (counting-filter arena, census records, reservation tables) go back to the pool for the
// next set's build; and everything behind it.  Both wait for the job's stream before they return.
int job_polish_front(gp_job* job);
int job_polish_back(gp_job* job);
int stage_a_prepare(gp_job* job);            // allocate work buffers
int stage_a_run(gp_job* job);                // enqueue + drive the filter build
void stage_a_release(gp_job* job);
void stage_a_ctx_release(gp_ctx* ctx);      // what stage_a_prepare keeps in the context (gp_ctx::StageA)
int stage_bc_prepare(gp_job* job);
void stage_bc_release(gp_job* job);
int stage_b_reset(gp_job* job);              // current sequences := inputs
int stage_b_pass(gp_job* job, uint32_t ki);
int stage_b_guard(gp_job* job);
int stage_mask(gp_job* job);
int stage_c_run(gp_job* job);
int stage_upper(gp_job* job);
int sealer_consensus_sets(gp_ctx* ctx, uint32_t n_sets, const uint32_t* set_seq_begin, const uint8_t* seq_buf,
                          const uint64_t* seq_off, uint8_t* out_buf, uint64_t out_stride, uint32_t* out_len);
// gp_debug_nthash: the hashing of each stage's kernels over one sequence (device pointers)
int debug_nthash_stage_a(hipStream_t s, const uint8_t* d_seq, uint64_t len, uint32_t k, uint32_t hash_num, int variant,
                         uint64_t* d_f, uint64_t* d_r, uint64_t* d_h, uint8_t* d_valid);
int debug_nthash_stage_b(hipStream_t s, const uint8_t* d_seq, uint64_t len, uint32_t k, uint32_t hash_num, int variant,
                         uint64_t* d_f, uint64_t* d_r, uint64_t* d_h, uint8_t* d_valid);
int debug_nthash_stage_c(hipStream_t s, const uint8_t* d_seq, uint64_t len, uint32_t k, uint32_t hash_num,
                         uint64_t* d_f, uint64_t* d_r, uint64_t* d_h, uint8_t* d_valid);
int stage_lengths(gp_job* job, uint64_t* lengths, uint8_t* dropped, uint64_t* total);
int stage_sequences(gp_job* job, uint8_t* buf, uint64_t buf_bytes);
int stage_sequences_device(gp_job* job, void* d_buf, uint64_t buf_bytes);
} // namespace gp
