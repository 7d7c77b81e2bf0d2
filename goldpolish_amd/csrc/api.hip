// api.hip -- the C ABI of libgpcore (include/gpcore.h): contexts, jobs, uploads,
// stage sequencing and result copies.  Host C++; the kernels live in stage_*.hip.
#include "gp_internal.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <iterator>
#include <map>
#include <mutex>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <thread>
#include <vector>

// Hardware queues.  The filter build keeps three streams busy (census_scatter | census_mark + census_pack |
// the ckc_* kernels), gp_pipeline adds an upload stream and the streams of its other contexts, and the HIP
// runtime deals its streams onto FOUR hardware queues by default: two streams on one queue run one after the
// other.  Seen in the kernel trace of BASELINE configs[2]: the mark/pack stream and the ckc stream on the same
// queue in every job, 966 ms per step against 926 with eight queues.  The runtime reads GPU_MAX_HW_QUEUES
// when it initialises, so the library asks for eight when it is loaded -- unless the host has set the variable
// itself; a host that has initialised HIP before loading this library sets it itself.
__attribute__((constructor)) static void gp_runtime_defaults()
{
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
}

namespace gp {

static thread_local std::string g_err;

static std::atomic<long long> g_device_bytes{0};
void track_device_bytes(long long delta) { g_device_bytes += delta; }

void set_error(const std::string& msg) { g_err = msg; }

// ---- device memory pool (see gp_internal.h) ---------------------------------------------------
// Free blocks PER DEVICE, keyed by size.  A request takes the smallest free block that holds it
// and is at most an eighth larger; otherwise hipMalloc (after which, on failure, that device's
// pool is emptied and hipMalloc tried once more).  What a device's pool may hold is capped by the
// most this library has ever had in use on that device at one time (a stream of equal jobs
// recycles one job's worth of blocks; nothing more is worth keeping) and by GP_POOL_GIB (default
// 200; 0 turns pooling off); beyond the cap the largest free blocks go back to the driver.  The
// pool of a device is emptied when the last context on it is destroyed and by
// gp_device_pool_trim(): other allocators in the process (torch's, RCCL's) cannot make this
// library give memory back, so a caller that mixes them in trims at its own quiet points.
namespace {
constexpr int kMaxDev = 16;
struct DevPool {
    std::mutex mu;
    std::multimap<size_t, void*> free_blocks;
    size_t pooled = 0;
    size_t in_use = 0, high_water = 0;
    int contexts = 0;
};
DevPool g_pool[kMaxDev];
std::atomic<long long> g_pool_bytes{0};
std::atomic<long long> g_fail_countdown{0};
std::atomic<unsigned long long> g_alloc_calls{0};

size_t pool_cap_env()
{
    static const size_t cap = (size_t)env_u64("GP_POOL_GIB", 200) << 30;
    return cap;
}

size_t pool_cap_locked(const DevPool& P) { return std::min(pool_cap_env(), P.high_water); }

void pool_trim_locked(DevPool& P, size_t keep)
{
    while (!P.free_blocks.empty() && P.pooled > keep) {
        auto it = std::prev(P.free_blocks.end());
        (void)hipFree(it->second);
        P.pooled -= it->first;
        g_pool_bytes -= (long long)it->first;
        P.free_blocks.erase(it);
    }
}
} // namespace

void pool_context_born(int dev)
{
    if (dev < 0 || dev >= kMaxDev)
        return;
    std::lock_guard<std::mutex> g(g_pool[dev].mu);
    g_pool[dev].contexts++;
}

void pool_context_gone(int dev)
{
    if (dev < 0 || dev >= kMaxDev)
        return;
    std::lock_guard<std::mutex> g(g_pool[dev].mu);
    if (--g_pool[dev].contexts <= 0) {
        g_pool[dev].contexts = 0;
        pool_trim_locked(g_pool[dev], 0);
    }
}

int dev_alloc(size_t bytes, void** out, size_t* cap_bytes)
{
    int dev = 0;
    (void)hipGetDevice(&dev);
    bytes = (bytes + 255) & ~(size_t)255;
    g_alloc_calls++;
    // test hook (gp_debug_fail_alloc): the n-th allocation from now on fails
    if (g_fail_countdown.load() > 0 && g_fail_countdown.fetch_sub(1) == 1) {
        set_error("hipMalloc of " + std::to_string(bytes) + " bytes failed: injected by gp_debug_fail_alloc");
        return GP_ERR_OOM;
    }
    const bool pooled_dev = dev >= 0 && dev < kMaxDev && pool_cap_env();
    if (pooled_dev) {
        DevPool& P = g_pool[dev];
        std::lock_guard<std::mutex> g(P.mu);
        auto it = P.free_blocks.lower_bound(bytes);
        if (it != P.free_blocks.end() && it->first <= bytes + bytes / 8) {
            *out = it->second;
            *cap_bytes = it->first;
            P.pooled -= it->first;
            g_pool_bytes -= (long long)it->first;
            P.free_blocks.erase(it);
            P.in_use += *cap_bytes;
            P.high_water = std::max(P.high_water, P.in_use);
            track_device_bytes((long long)*cap_bytes);
            return GP_OK;
        }
    }
    void* q = nullptr;
    hipError_t e = hipMalloc(&q, bytes);
    if (e != hipSuccess && pooled_dev) {
        (void)hipGetLastError();
        {
            std::lock_guard<std::mutex> g(g_pool[dev].mu);
            pool_trim_locked(g_pool[dev], 0);
        }
        e = hipMalloc(&q, bytes);
    }
    if (e != hipSuccess) {
        (void)hipGetLastError();
        set_error(std::string("hipMalloc of ") + std::to_string(bytes) + " bytes failed: " + hipGetErrorString(e));
        return GP_ERR_OOM;
    }
    *out = q;
    *cap_bytes = bytes;
    if (dev >= 0 && dev < kMaxDev) {
        std::lock_guard<std::mutex> g(g_pool[dev].mu);
        g_pool[dev].in_use += bytes;
        g_pool[dev].high_water = std::max(g_pool[dev].high_water, g_pool[dev].in_use);
    }
    track_device_bytes((long long)bytes);
    return GP_OK;
}

void dev_free(void* q, size_t cap_bytes)
{
    if (!q)
        return;
    track_device_bytes(-(long long)cap_bytes);
    hipPointerAttribute_t attr{};
    int dev = -1;
    if (hipPointerGetAttributes(&attr, q) == hipSuccess)
        dev = attr.device;
    else
        (void)hipGetLastError();
    if (dev >= 0 && dev < kMaxDev) {
        DevPool& P = g_pool[dev];
        std::lock_guard<std::mutex> g(P.mu);
        P.in_use -= std::min(P.in_use, cap_bytes);
        const size_t cap = pool_cap_locked(P);
        if (cap && cap_bytes <= cap && P.contexts > 0) {
            P.free_blocks.emplace(cap_bytes, q);
            P.pooled += cap_bytes;
            g_pool_bytes += (long long)cap_bytes;
            if (P.pooled > cap)
                pool_trim_locked(P, cap);
            return;
        }
    }
    (void)hipFree(q);
}

int hip_fail(hipError_t e, const char* what, const char* file, int line)
{
    g_err = std::string(what) + " failed at " + file + ":" + std::to_string(line) + ": " +
            hipGetErrorString(e);
    return e == hipErrorOutOfMemory ? GP_ERR_OOM : GP_ERR_HIP;
}

int on_exception(const char* where) noexcept
{
    int rc = GP_ERR_INVALID_ARG;
    try {
        try {
            throw;
        } catch (const std::bad_alloc&) {
            rc = GP_ERR_OOM;
            g_err = std::string(where) + ": out of host memory";
        } catch (const std::exception& e) {
            g_err = std::string(where) + ": " + e.what();
        } catch (...) {
            g_err = std::string(where) + ": unknown exception";
        }
    } catch (...) {
        // the message itself could not be built; the status still says what happened
    }
    return rc;
}

} // namespace gp

using namespace gp;

extern "C" {

const char* gp_last_error(void) { return g_err.c_str(); }

int gp_version(void) { return 100; }

uint64_t gp_device_bytes_in_use(void)
{
    const long long v = g_device_bytes.load();
    return v > 0 ? (uint64_t)v : 0;
}

uint64_t gp_device_pool_bytes(void)
{
    const long long v = g_pool_bytes.load();
    return v > 0 ? (uint64_t)v : 0;
}

uint64_t gp_debug_fail_alloc(uint64_t nth)
{
    g_fail_countdown = (long long)nth;
    return g_alloc_calls.load();
}

int64_t gp_debug_nthash(gp_ctx* ctx, int variant, const uint8_t* seq, uint64_t len, uint32_t k, uint32_t hash_num,
                        uint64_t* out_fwd, uint64_t* out_rev, uint64_t* out_hashes, uint8_t* valid)
try {
    if (!ctx || !seq || !out_fwd || !out_rev || !out_hashes || !valid || k < 4 || k > 63 || hash_num == 0 ||
        hash_num > (uint32_t)kMaxHash || variant < 0 || variant > 4) {
        set_error("gp_debug_nthash: bad argument (k must be 4..63, hash_num 1..8, variant 0..4)");
        return GP_ERR_INVALID_ARG;
    }
    if (len < k)
        return 0;
    GP_HIP(hipSetDevice(ctx->device));
    const uint64_t n = variant == 4 ? 7 : len - k + 1;
    DevBuf<uint8_t> d_seq, d_valid;
    DevBuf<uint64_t> d_f, d_r, d_h;
    int rc;
    if ((rc = d_seq.alloc(len + 64)) != GP_OK || (rc = d_valid.alloc(len + 8)) != GP_OK || (rc = d_f.alloc(len + 8)) != GP_OK ||
        (rc = d_r.alloc(len + 8)) != GP_OK || (rc = d_h.alloc((len + 8) * hash_num)) != GP_OK)
        return rc;
    hipStream_t s = ctx->stream;
    GP_HIP(hipMemcpyAsync(d_seq.p, seq, len, hipMemcpyHostToDevice, s));
    GP_HIP(hipMemsetAsync(d_valid.p, 0, len, s));
    if (variant <= 1)
        rc = debug_nthash_stage_a(s, d_seq.p, len, k, hash_num, variant, d_f.p, d_r.p, d_h.p, d_valid.p);
    else if (variant == 3)
        rc = debug_nthash_stage_c(s, d_seq.p, len, k, hash_num, d_f.p, d_r.p, d_h.p, d_valid.p);
    else
        rc = debug_nthash_stage_b(s, d_seq.p, len, k, hash_num, variant, d_f.p, d_r.p, d_h.p, d_valid.p);
    if (rc != GP_OK)
        return rc;
    GP_HIP(hipMemcpyAsync(out_fwd, d_f.p, n * 8, hipMemcpyDeviceToHost, s));
    GP_HIP(hipMemcpyAsync(out_rev, d_r.p, n * 8, hipMemcpyDeviceToHost, s));
    GP_HIP(hipMemcpyAsync(out_hashes, d_h.p, n * 8 * hash_num, hipMemcpyDeviceToHost, s));
    GP_HIP(hipMemcpyAsync(valid, d_valid.p, n, hipMemcpyDeviceToHost, s));
    GP_HIP(hipStreamSynchronize(s));
    return (int64_t)n;
} GP_CATCH("gp_debug_nthash")

void gp_device_pool_trim(void)
{
    try {
        for (DevPool& P : g_pool) {
            std::lock_guard<std::mutex> g(P.mu);
            pool_trim_locked(P, 0);
        }
    } catch (...) {
    }
}

void gp_default_params(gp_params* p)
{
    if (!p)
        return;
    std::memset(p, 0, sizeof *p);
    p->n_k = 4;
    p->k[0] = 32;
    p->k[1] = 28;
    p->k[2] = 24;
    p->k[3] = 20;
    p->hash_num = 4;
    p->cbf_bytes = 10ull * 1024 * 1024;
    p->bf_bytes = 512ull * 1024;
    p->max_insertions = 5;
    p->max_deletions = 5;
    p->mode = 1;
    p->missing_ratio = 0.5f;
    p->edit_ratio = 0.5f;
    p->jump = 3;
    p->min_contig_len = 100;
    p->soft_mask = 1;
    p->sealer_max_gap = 800;
    p->sealer_max_paths = 10;
    p->sealer_max_branches = 1000;
    p->sealer_max_cost = 100000;
    p->run_sealer = 1;
}

} // extern "C"

namespace gp {
int ctx_stream_create(const gp_ctx* ctx, hipStream_t* out)
{
    if (ctx->cu_mask.empty())
        GP_HIP(hipStreamCreateWithFlags(out, hipStreamNonBlocking));
    else
        GP_HIP(hipExtStreamCreateWithCUMask(out, (uint32_t)ctx->cu_mask.size(), ctx->cu_mask.data()));
    return GP_OK;
}
} // namespace gp

static int ctx_create(int device, const uint32_t* cu_mask, uint32_t mask_words, gp_ctx** out);

extern "C" {

int gp_ctx_create(int device, gp_ctx** out)
try {
    return ctx_create(device, nullptr, 0, out);
} GP_CATCH("gp_ctx_create")

int gp_ctx_create_masked(int device, const uint32_t* cu_mask, uint32_t mask_words, gp_ctx** out)
try {
    if (!cu_mask || mask_words == 0) {
        set_error("gp_ctx_create_masked: no mask");
        return GP_ERR_INVALID_ARG;
    }
    return ctx_create(device, cu_mask, mask_words, out);
} GP_CATCH("gp_ctx_create_masked")

} // extern "C"

static int ctx_create(int device, const uint32_t* cu_mask, uint32_t mask_words, gp_ctx** out)
{
    if (!out) {
        set_error("gp_ctx_create: out is NULL");
        return GP_ERR_INVALID_ARG;
    }
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        set_error("gp_ctx_create: no HIP device available (libgpcore has no CPU fallback)");
        return GP_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        set_error("gp_ctx_create: device index out of range");
        return GP_ERR_INVALID_ARG;
    }
    gp_ctx* c = new (std::nothrow) gp_ctx();
    if (!c)
        return GP_ERR_OOM;
    c->device = device;
    if (cu_mask)
        c->cu_mask.assign(cu_mask, cu_mask + mask_words);
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&c->prop, device) != hipSuccess ||
        ctx_stream_create(c, &c->stream) != GP_OK) {
        set_error("gp_ctx_create: could not initialise the device");
        delete c;
        return GP_ERR_HIP;
    }
    pool_context_born(device);
    *out = c;
    return GP_OK;
}

extern "C" {


void gp_ctx_destroy(gp_ctx* ctx)
{
    if (!ctx)
        return;
    (void)hipSetDevice(ctx->device);
    for (int i = 0; i < 2; i++) {
        if (ctx->stage[i])
            (void)hipHostFree(ctx->stage[i]);
        if (ctx->stage_ev[i])
            (void)hipEventDestroy(ctx->stage_ev[i]);
    }
    stage_a_ctx_release(ctx);
    if (ctx->stream)
        (void)hipStreamDestroy(ctx->stream);
    pool_context_gone(ctx->device); // the last context on a device takes its pool along
    delete ctx;
}

void* gp_ctx_stream(gp_ctx* ctx) { return ctx ? (void*)ctx->stream : nullptr; }

/* src/goldpolish_targeted_bfs.cpp:45-53 */
int gp_kmer_threshold(uint64_t mapped_bases)
{
    const double a = 4.66943, b = 2.11391e-07;
    int t = (int)std::round(a + (double)mapped_bases * b);
    return std::min(t, 13);
}

/* src/goldpolish_targeted_bfs.cpp:96-123 */
size_t gp_rank_reads(size_t n, const char* const* ids, const double* phred_avg,
                     const uint64_t* lens, uint64_t target_len, double max_reads_per_10kbp,
                     uint32_t* order, uint64_t* bases)
try {
    if (bases)
        *bases = 0;
    if (n == 0 || !ids || !phred_avg || !lens || !order)
        return 0;
    struct Item {
        uint32_t idx;
        size_t phred;
    };
    std::vector<Item> v(n);
    for (size_t i = 0; i < n; i++)
        v[i] = { (uint32_t)i, (size_t)phred_avg[i] };
    std::sort(v.begin(), v.end(), [&](const Item& x, const Item& y) {
        if (x.phred != y.phred)
            return x.phred > y.phred;
        return std::strcmp(ids[x.idx], ids[y.idx]) < 0;
    });
    size_t cap = (size_t)((double)target_len * max_reads_per_10kbp / 10000.0);
    size_t use = std::min(n, cap);
    uint64_t sum = 0;
    for (size_t i = 0; i < use; i++) {
        order[i] = v[i].idx;
        sum += lens[v[i].idx];
    }
    if (bases)
        *bases = sum;
    return use;
} GP_CATCH_SIZE("gp_rank_reads")

/* SURVEY.md 8e */
int gp_partition_batches(uint32_t n_batches, const uint64_t* cost, uint32_t n_ranks,
                         uint32_t* rank_of, uint64_t* rank_load)
try {
    if (n_ranks == 0 || (n_batches && (!cost || !rank_of))) {
        set_error("gp_partition_batches: bad argument");
        return GP_ERR_INVALID_ARG;
    }
    std::vector<uint32_t> order(n_batches);
    for (uint32_t b = 0; b < n_batches; b++)
        order[b] = b;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return cost[x] > cost[y]; });
    std::vector<uint64_t> load(n_ranks, 0);
    for (uint32_t b : order) {
        uint32_t best = 0;
        for (uint32_t r = 1; r < n_ranks; r++) // a node has 8 ranks: no heap needed
            if (load[r] < load[best])
                best = r;
        rank_of[b] = best;
        load[best] += cost[b];
    }
    if (rank_load)
        std::copy(load.begin(), load.end(), rank_load);
    return GP_OK;
} GP_CATCH("gp_partition_batches")

// the context's two pinned staging chunks (GP_STAGE_MIB each, default 64), made on first use
static int ensure_staging(gp_ctx* ctx)
{
    if (ctx->stage[0] && ctx->stage[1])
        return GP_OK;
    const size_t bytes = (size_t)std::max<uint64_t>(1, env_u64("GP_STAGE_MIB", 64)) << 20;
    for (int i = 0; i < 2; i++) {
        if (!ctx->stage[i] && hipHostMalloc((void**)&ctx->stage[i], bytes) != hipSuccess) {
            ctx->stage[i] = nullptr;
            (void)hipGetLastError();
            set_error("pinned staging allocation failed");
            return GP_ERR_OOM;
        }
        if (!ctx->stage_ev[i])
            GP_HIP(hipEventCreateWithFlags(&ctx->stage_ev[i], hipEventDisableTiming));
    }
    ctx->stage_bytes = bytes;
    return GP_OK;
}

static int check_params(const gp_params& p)
{
    if (p.n_k == 0 || p.n_k > GP_MAX_K_VALUES) {
        set_error("params: n_k must be 1..8");
        return GP_ERR_INVALID_ARG;
    }
    for (uint32_t i = 0; i < p.n_k; i++)
        if (p.k[i] < 4 || p.k[i] > 63) {
            set_error("params: every k must be 4..63");
            return GP_ERR_UNSUPPORTED;
        }
    if (p.hash_num == 0 || p.hash_num > (uint32_t)kMaxHash) {
        set_error("params: hash_num must be 1..8");
        return GP_ERR_UNSUPPORTED;
    }
    if (p.cbf_bytes == 0 || p.cbf_bytes >= (1ull << 32) || p.bf_bytes == 0 || p.bf_bytes % 4 ||
        p.bf_bytes >= (1ull << 29)) {
        set_error("params: cbf_bytes must be < 4 GiB, bf_bytes a multiple of 4 below 512 MiB");
        return GP_ERR_UNSUPPORTED;
    }
    if (p.mode != 1) {
        set_error("params: only ntEdit mode 1 (GoldPolish's) is implemented");
        return GP_ERR_UNSUPPORTED;
    }
    if (p.max_insertions > 5 || p.max_deletions > 10 || p.jump == 0) {
        set_error("params: max_insertions <= 5, max_deletions <= 10, jump >= 1");
        return GP_ERR_INVALID_ARG;
    }
    return GP_OK;
}

int gp_job_create(gp_ctx* ctx, const gp_params* params, const gp_batch_set* set, gp_job** out)
try {
    if (!ctx || !params || !set || !out) {
        set_error("gp_job_create: NULL argument");
        return GP_ERR_INVALID_ARG;
    }
    *out = nullptr;
    int rc = check_params(*params);
    if (rc != GP_OK)
        return rc;
    // every array is checked before anything reads it: a job with no contigs or no reads at all
    // is legal (it polishes nothing), NULL arrays that would have to be read are not
    if (set->n_batches == 0 || !set->batch_contig_begin) {
        set_error("gp_job_create: no batches, or batch_contig_begin is NULL");
        return GP_ERR_INVALID_ARG;
    }
    if (set->n_contigs && (!set->contig_off || !set->contig_read_begin || !set->contig_kmer_threshold)) {
        set_error("gp_job_create: batch set has NULL arrays");
        return GP_ERR_INVALID_ARG;
    }
    if (set->n_reads && (!set->read_off || !set->read_buf)) {
        set_error("gp_job_create: n_reads > 0 but read_off / read_buf is NULL");
        return GP_ERR_INVALID_ARG;
    }
    if (set->batch_contig_begin[0] != 0 || set->batch_contig_begin[set->n_batches] != set->n_contigs) {
        set_error("gp_job_create: batch_contig_begin does not cover the contigs");
        return GP_ERR_INVALID_ARG;
    }
    for (uint32_t b = 0; b < set->n_batches; b++)
        if (set->batch_contig_begin[b + 1] < set->batch_contig_begin[b]) {
            set_error("gp_job_create: batch_contig_begin must be non-decreasing");
            return GP_ERR_INVALID_ARG;
        }
    if (set->n_contigs >= (1u << 28)) {
        set_error("gp_job_create: too many contigs in one job");
        return GP_ERR_UNSUPPORTED;
    }
    if (set->n_contigs && set->contig_read_begin[set->n_contigs] > set->n_reads) {
        set_error("gp_job_create: contig_read_begin exceeds n_reads");
        return GP_ERR_INVALID_ARG;
    }
    for (uint32_t r = 0; r < set->n_reads; r++)
        if (set->read_off[r + 1] < set->read_off[r]) {
            set_error("gp_job_create: read_off must be non-decreasing");
            return GP_ERR_INVALID_ARG;
        }
    for (uint32_t c = 0; c < set->n_contigs; c++) {
        if (set->contig_read_begin[c + 1] < set->contig_read_begin[c] ||
            set->contig_off[c + 1] < set->contig_off[c]) {
            set_error("gp_job_create: offsets must be non-decreasing");
            return GP_ERR_INVALID_ARG;
        }
        // fill_bfs refuses thresholds below 4 (src/utils.cpp:105-107) when it is reached
        if (set->contig_read_begin[c + 1] > set->contig_read_begin[c] &&
            set->contig_kmer_threshold[c] < 4) {
            set_error("kmer_threshold must be greater than or equal to 4");
            return GP_ERR_THRESHOLD;
        }
        if (set->contig_kmer_threshold[c] + params->n_k > 250) {
            set_error("kmer_threshold too large for 8-bit counters");
            return GP_ERR_INVALID_ARG;
        }
    }
    if (set->n_contigs && set->contig_off[set->n_contigs] > set->contig_off[0] && !set->contig_buf) {
        set_error("gp_job_create: contig_buf is NULL");
        return GP_ERR_INVALID_ARG;
    }
    GP_HIP(hipSetDevice(ctx->device));

    gp_job* j = new (std::nothrow) gp_job();
    if (!j)
        return GP_ERR_OOM;
    if (hipHostMalloc((void**)&j->h_readback, 64 * sizeof(unsigned long long)) != hipSuccess) {
        j->h_readback = nullptr;
        delete j;
        set_error("gp_job_create: pinned read-back block allocation failed");
        return GP_ERR_OOM;
    }
    j->ctx = ctx;
    j->prm = *params;
    j->n_batches = set->n_batches;
    j->n_contigs = set->n_contigs;
    j->n_reads = set->n_reads;
    struct Owner { // frees the half-built job on every early return and on an exception
        gp_job* j;
        ~Owner() { gp_job_destroy(j); }
    } owner{ j };
    auto fail = [&](int code) { return code; };

    // ---- host-side tables
    j->h_batch_contig_begin.assign(set->batch_contig_begin,
                                   set->batch_contig_begin + set->n_batches + 1);
    j->h_contig_batch.resize(set->n_contigs);
    for (uint32_t b = 0; b < set->n_batches; b++)
        for (uint32_t c = set->batch_contig_begin[b]; c < set->batch_contig_begin[b + 1]; c++)
            j->h_contig_batch[c] = b;
    j->h_contig_thr.assign(set->contig_kmer_threshold, set->contig_kmer_threshold + set->n_contigs);
    j->h_contig_name_len.resize(set->n_contigs);
    for (uint32_t c = 0; c < set->n_contigs; c++)
        j->h_contig_name_len[c] = set->contig_name_len ? set->contig_name_len[c] : 0;
    if (set->n_contigs)
        j->h_contig_off.assign(set->contig_off, set->contig_off + set->n_contigs + 1);
    else
        j->h_contig_off.assign(1, 0);

    // ---- reads: concatenate with one separator after each read
    j->h_seg_start.resize((size_t)set->n_contigs + 1);
    uint64_t total = 0;
    for (uint32_t c = 0; c < set->n_contigs; c++) {
        j->h_seg_start[c] = total;
        for (uint32_t r = set->contig_read_begin[c]; r < set->contig_read_begin[c + 1]; r++)
            total += set->read_off[r + 1] - set->read_off[r] + 1;
    }
    j->h_seg_start[set->n_contigs] = total;
    j->read_positions = total;
    // Everything below goes through the context's stream (copies from the context's two pinned
    // staging chunks, the host filling one while the other drains) and is waited for once, at the
    // end: nothing here touches the default stream or allocates pinned memory per job, so a job
    // can be created by one thread while another polishes on the same device.
    hipStream_t s = ctx->stream;
    if ((rc = ensure_staging(ctx)) != GP_OK)
        return fail(rc);
    if ((rc = j->d_reads.alloc(total + 64)) != GP_OK)
        return fail(rc);
    {
        int slot = 0;
        size_t fill = 0;       // bytes in the chunk being filled
        uint64_t sent = 0;     // bytes of d_reads already on their way
        auto flush = [&]() -> int {
            if (!fill)
                return GP_OK;
            GP_HIP(hipMemcpyAsync(j->d_reads.p + sent, ctx->stage[slot], fill, hipMemcpyHostToDevice, s));
            GP_HIP(hipEventRecord(ctx->stage_ev[slot], s));
            sent += fill;
            fill = 0;
            slot ^= 1;
            GP_HIP(hipEventSynchronize(ctx->stage_ev[slot])); // the other chunk has drained
            return GP_OK;
        };
        for (uint32_t c = 0; c < set->n_contigs; c++)
            for (uint32_t r = set->contig_read_begin[c]; r < set->contig_read_begin[c + 1]; r++) {
                const uint8_t* src = set->read_buf + set->read_off[r];
                uint64_t left = set->read_off[r + 1] - set->read_off[r];
                while (left) {
                    const size_t n = (size_t)std::min<uint64_t>(left, ctx->stage_bytes - fill);
                    std::memcpy(ctx->stage[slot] + fill, src, n);
                    fill += n;
                    src += n;
                    left -= n;
                    if (fill == ctx->stage_bytes && (rc = flush()) != GP_OK)
                        return fail(rc);
                }
                ctx->stage[slot][fill++] = '\n';
                if (fill == ctx->stage_bytes && (rc = flush()) != GP_OK)
                    return fail(rc);
            }
        // the tail of the buffer (kernels stage a halo past the last position)
        for (int i = 0; i < 64; i++) {
            ctx->stage[slot][fill++] = '\n';
            if (fill == ctx->stage_bytes && (rc = flush()) != GP_OK)
                return fail(rc);
        }
        if ((rc = flush()) != GP_OK)
            return fail(rc);
    }
    if ((rc = j->d_seg_start.alloc((size_t)set->n_contigs + 1)) != GP_OK ||
        (rc = j->d_seg_batch.alloc(set->n_contigs)) != GP_OK ||
        (rc = j->d_seg_thr.alloc(set->n_contigs)) != GP_OK)
        return fail(rc);
    // (the sources of these copies are vectors of the job: they outlive the copies)
    GP_HIP(hipMemcpyAsync(j->d_seg_start.p, j->h_seg_start.data(), j->h_seg_start.size() * sizeof(uint64_t),
                          hipMemcpyHostToDevice, s));
    if (set->n_contigs) {
        GP_HIP(hipMemcpyAsync(j->d_seg_batch.p, j->h_contig_batch.data(), set->n_contigs * sizeof(uint32_t),
                              hipMemcpyHostToDevice, s));
        GP_HIP(hipMemcpyAsync(j->d_seg_thr.p, j->h_contig_thr.data(), set->n_contigs * sizeof(uint32_t),
                              hipMemcpyHostToDevice, s));
    }

    // ---- contigs
    {
        const uint64_t base = set->n_contigs ? set->contig_off[0] : 0;
        const uint64_t cb = j->h_contig_off[set->n_contigs] - base;
        if ((rc = j->d_contigs_in.alloc(cb + 64)) != GP_OK)
            return fail(rc);
        GP_HIP(hipEventSynchronize(ctx->stage_ev[0]));
        GP_HIP(hipEventSynchronize(ctx->stage_ev[1]));
        int slot = 0;
        for (uint64_t at = 0; at < cb;) {
            const size_t n = (size_t)std::min<uint64_t>(cb - at, ctx->stage_bytes);
            GP_HIP(hipEventSynchronize(ctx->stage_ev[slot]));
            std::memcpy(ctx->stage[slot], set->contig_buf + base + at, n);
            GP_HIP(hipMemcpyAsync(j->d_contigs_in.p + at, ctx->stage[slot], n, hipMemcpyHostToDevice, s));
            GP_HIP(hipEventRecord(ctx->stage_ev[slot], s));
            at += n;
            slot ^= 1;
        }
        if (base)
            for (auto& o : j->h_contig_off)
                o -= base;
    }
    GP_HIP(hipStreamSynchronize(s));
    // Filters and work buffers are allocated by the first call that needs them (ensure_work): a
    // job that has only been created holds its inputs and nothing else, so the next job of a
    // stream can be uploaded while the one before it is polished without both holding tens of
    // GiB of work buffers.
    owner.j = nullptr;
    *out = j;
    return GP_OK;
} GP_CATCH("gp_job_create")

void gp_job_destroy(gp_job* job)
{
    if (!job)
        return;
    if (job->ctx) {
        (void)hipSetDevice(job->ctx->device);
        (void)hipStreamSynchronize(job->ctx->stream);
    }
    stage_a_release(job);
    stage_bc_release(job);
    if (job->h_readback)
        (void)hipHostFree(job->h_readback);
    job->t_first_pass.destroy();
    job->t_round.destroy();
    job->t_local.destroy();
    job->t_census.destroy();
    job->t_ckc.destroy();
    job->t_scan.destroy();
    job->t_walk.destroy();
    job->t_bfs.destroy();
    job->t_mask.destroy();
    job->t_upper.destroy();
    delete job;
}

int gp_job_sync(gp_job* job)
try {
    if (!job)
        return GP_ERR_INVALID_ARG;
    GP_HIP(hipSetDevice(job->ctx->device)); // a caller may hold contexts on several GPUs
    GP_HIP(hipStreamSynchronize(job->ctx->stream));
    return GP_OK;
} GP_CATCH("gp_job_sync")

static int finish(gp_job* job, int rc, int sync)
{
    if (rc != GP_OK)
        return rc;
    if (sync)
        return gp_job_sync(job);
    return GP_OK;
}

// Filters and work buffers, allocated by the first call that needs them (gp_job_create uploads the
// inputs and nothing else).  They are gone for good only if growing the contig slots ran out of
// memory half way (polish_back): later calls then get GP_ERR_STATE.
static int ensure_work(gp_job* j, const char* who)
{
    if (j->work_state == 1 && j->bc)
        return GP_OK;
    if (j->work_state != 0) {
        set_error(std::string(who) + ": the job lost its polishing state (allocation failed while growing the contig slots)");
        return GP_ERR_STATE;
    }
    const gp_params& p = j->prm;
    // Bloom filters: one per (batch, k), they are results.  Counting filters: an arena of slots,
    // scratch of the filter build (512 slots = 5 GiB at GoldPolish's 10 MiB; GP_STAGE_A_SLOTS)
    const uint64_t streams = (uint64_t)j->n_batches * p.n_k;
    {
        uint64_t want = env_u64("GP_STAGE_A_SLOTS", 512);
        want = std::max<uint64_t>(want, p.n_k); // one batch at least
        j->cbf_slots = (uint32_t)std::min<uint64_t>(streams, want);
    }
    int rc;
    if ((rc = j->d_cbf.alloc((uint64_t)j->cbf_slots * ((p.cbf_bytes + 15) / 16 * 16) + 64)) == GP_OK &&
        (rc = j->d_bf.alloc(streams * (p.bf_bytes / 4))) == GP_OK && (rc = stage_a_prepare(j)) == GP_OK &&
        (rc = stage_bc_prepare(j)) == GP_OK) {
        j->work_state = 1;
        return GP_OK;
    }
    // nothing half-built stays behind: the call may be tried again when memory has been freed
    const std::string msg = gp_last_error();
    stage_a_release(j);
    stage_bc_release(j);
    j->d_cbf.release();
    j->d_bf.release();
    set_error(msg);
    return rc;
}

int gp_job_build_filters(gp_job* job, int sync)
try {
    if (!job)
        return GP_ERR_INVALID_ARG;
    GP_HIP(hipSetDevice(job->ctx->device));
    if (int rcw = ensure_work(job, "gp_job_build_filters"))
        return rcw;
    return finish(job, stage_a_run(job), sync);
} GP_CATCH("gp_job_build_filters")

int gp_job_ntedit_pass(gp_job* job, uint32_t k_index, int sync)
try {
    if (!job || k_index >= job->prm.n_k)
        return GP_ERR_INVALID_ARG;
    if (!job->filters_ready) {
        set_error("gp_job_ntedit_pass: filters have not been built or set");
        return GP_ERR_STATE;
    }
    GP_HIP(hipSetDevice(job->ctx->device));
    if (int rcb = ensure_work(job, "gp_job_ntedit_pass"))
        return rcb;
    int rc = GP_OK;
    if (k_index == 0)
        rc = stage_b_reset(job);
    if (rc == GP_OK)
        rc = stage_b_pass(job, k_index);
    return finish(job, rc, sync);
} GP_CATCH("gp_job_ntedit_pass")

int gp_job_reset(gp_job* job, int sync)
try {
    if (!job)
        return GP_ERR_INVALID_ARG;
    GP_HIP(hipSetDevice(job->ctx->device));
    if (int rcb = ensure_work(job, "gp_job_reset"))
        return rcb;
    return finish(job, stage_b_reset(job), sync);
} GP_CATCH("gp_job_reset")

int gp_job_ntedit_guard(gp_job* job, int sync)
try {
    if (!job)
        return GP_ERR_INVALID_ARG;
    GP_HIP(hipSetDevice(job->ctx->device));
    if (int rcb = ensure_work(job, "gp_job_ntedit_guard"))
        return rcb;
    return finish(job, stage_b_guard(job), sync);
} GP_CATCH("gp_job_ntedit_guard")

int gp_job_mask(gp_job* job, int sync)
try {
    if (!job)
        return GP_ERR_INVALID_ARG;
    GP_HIP(hipSetDevice(job->ctx->device));
    if (int rcb = ensure_work(job, "gp_job_mask"))
        return rcb;
    return finish(job, stage_mask(job), sync);
} GP_CATCH("gp_job_mask")

int gp_job_sealer(gp_job* job, int sync)
try {
    if (!job)
        return GP_ERR_INVALID_ARG;
    if (!job->filters_ready) {
        set_error("gp_job_sealer: filters have not been built or set");
        return GP_ERR_STATE;
    }
    GP_HIP(hipSetDevice(job->ctx->device));
    if (int rcb = ensure_work(job, "gp_job_sealer"))
        return rcb;
    return finish(job, stage_c_run(job), sync);
} GP_CATCH("gp_job_sealer")

int gp_job_upper(gp_job* job, int sync)
try {
    if (!job)
        return GP_ERR_INVALID_ARG;
    GP_HIP(hipSetDevice(job->ctx->device));
    if (int rcb = ensure_work(job, "gp_job_upper"))
        return rcb;
    return finish(job, stage_upper(job), sync);
} GP_CATCH("gp_job_upper")

int gp_sealer_consensus(gp_ctx* ctx, uint32_t n_sets, const uint32_t* set_seq_begin, const uint8_t* seq_buf,
                        const uint64_t* seq_off, uint8_t* out_buf, uint64_t out_stride, uint32_t* out_len)
try {
    if (!ctx || !set_seq_begin || !seq_buf || !seq_off || !out_buf || !out_len) {
        set_error("gp_sealer_consensus: NULL argument");
        return GP_ERR_INVALID_ARG;
    }
    if (n_sets == 0)
        return GP_OK;
    for (uint32_t i = 0; i < n_sets; i++)
        if (set_seq_begin[i + 1] < set_seq_begin[i]) {
            set_error("gp_sealer_consensus: set_seq_begin must not decrease");
            return GP_ERR_INVALID_ARG;
        }
    for (uint32_t q = set_seq_begin[0]; q < set_seq_begin[n_sets]; q++)
        if (seq_off[q + 1] < seq_off[q]) {
            set_error("gp_sealer_consensus: seq_off must not decrease");
            return GP_ERR_INVALID_ARG;
        }
    if (set_seq_begin[0] != 0) {
        set_error("gp_sealer_consensus: set_seq_begin[0] must be 0");
        return GP_ERR_INVALID_ARG;
    }
    GP_HIP(hipSetDevice(ctx->device));
    return sealer_consensus_sets(ctx, n_sets, set_seq_begin, seq_buf, seq_off, out_buf, out_stride, out_len);
} GP_CATCH("gp_sealer_consensus")

// everything behind the filter build: ntEdit cascade + guard, mask, Sealer, upper
static int polish_back(gp_job* job)
{
    int rc = GP_OK;
    for (int attempt = 0;; attempt++) {
        for (uint32_t ki = 0; rc == GP_OK && ki < job->prm.n_k; ki++) {
            if (ki == 0)
                rc = stage_b_reset(job);
            if (rc == GP_OK)
                rc = stage_b_pass(job, ki);
        }
        if (rc == GP_OK)
            rc = stage_b_guard(job);
        if (rc == GP_OK)
            rc = stage_mask(job);
        if (rc == GP_OK && job->prm.run_sealer)
            rc = stage_c_run(job);
        if (rc == GP_OK)
            rc = stage_upper(job);
        // errors raised inside kernels surface when stage_upper reads the statistics back
        if (rc != GP_ERR_CAPACITY || attempt == 4)
            break;
        // a contig grew past its slot (ntEdit inserting on a near-saturated filter): twice the
        // room, and stages B and C again from the unedited contigs; the filters stay
        job->bc_slack *= 2;
        stage_bc_release(job);
        if ((rc = stage_bc_prepare(job)) != GP_OK) {
            stage_bc_release(job); // not a half-built state: later calls get GP_ERR_STATE
            job->work_state = 2;
            break;
        }
    }
    return rc;
}

} // extern "C"

namespace gp {
int job_polish_front(gp_job* job)
{
    GP_HIP(hipSetDevice(job->ctx->device));
    if (int rcb = ensure_work(job, "gp_pipeline (filter build)"))
        return rcb;
    int rc = stage_a_run(job);
    if (rc == GP_OK)
        rc = gp_job_sync(job);
    // the filters are results; what built them is scratch: the next set's build takes it from the pool
    // (the job is destroyed after its back half and never builds filters again)
    stage_a_release(job);
    job->d_cbf.release();
    return rc;
}

int job_polish_back(gp_job* job)
{
    GP_HIP(hipSetDevice(job->ctx->device));
    if (!job->bc || !job->filters_ready) {
        set_error("gp_pipeline (back half): the filters have not been built");
        return GP_ERR_STATE;
    }
    const int rc = polish_back(job);
    return rc == GP_OK ? gp_job_sync(job) : rc;
}
} // namespace gp

extern "C" {

int gp_job_polish(gp_job* job, int sync)
try {
    if (!job)
        return GP_ERR_INVALID_ARG;
    GP_HIP(hipSetDevice(job->ctx->device));
    if (int rcb = ensure_work(job, "gp_job_polish"))
        return rcb;
    int rc = stage_a_run(job);
    if (rc == GP_OK)
        rc = polish_back(job);
    return finish(job, rc, sync);
} GP_CATCH("gp_job_polish")

/* Several jobs (SURVEY.md 8e has batches independent by construction) as a two-stage pipeline.
 * The stages of the path lean on different parts of the chip -- the filter build on random
 * memory transactions (it runs at the same speed with two, three or four workgroups per CU), the
 * ntEdit walk and the Sealer search on latency -- so the filter build of job i + 1 runs beside
 * the rest of job i. */
int gp_jobs_polish(gp_job* const* jobs, uint32_t n_jobs)
try {
    if (!jobs || n_jobs == 0) {
        set_error("gp_jobs_polish: no jobs");
        return GP_ERR_INVALID_ARG;
    }
    for (uint32_t i = 0; i < n_jobs; i++) {
        if (!jobs[i]) {
            set_error("gp_jobs_polish: NULL job");
            return GP_ERR_INVALID_ARG;
        }
        for (uint32_t j = 0; j < i; j++)
            if (jobs[j] == jobs[i] || jobs[j]->ctx == jobs[i]->ctx) {
                // a context is one stream and may be used by one thread at a time
                set_error("gp_jobs_polish: every job needs a context of its own");
                return GP_ERR_INVALID_ARG;
            }
        if (jobs[i]->ctx->device != jobs[0]->ctx->device) {
            set_error("gp_jobs_polish: the jobs must be on one device");
            return GP_ERR_INVALID_ARG;
        }
    }
    if (n_jobs == 1)
        return gp_job_polish(jobs[0], 1);
    std::vector<int> rc(n_jobs, GP_OK);
    std::vector<std::string> msg(n_jobs);
    {
        // Two-stage pipeline over the jobs: one thread builds the filters of job i + 1 while this
        // one runs everything behind them for job i.  Two streams for any number of jobs: every
        // filter build goes to the first job's stream, everything else to the second job's (a
        // job's own stream is put back before this returns).
        const hipStream_t front_stream = jobs[0]->ctx->stream, back_stream = jobs[1]->ctx->stream;
        std::vector<hipStream_t> own(n_jobs);
        for (uint32_t i = 0; i < n_jobs; i++)
            own[i] = jobs[i]->ctx->stream;
        std::mutex mu;
        std::condition_variable cv;
        uint32_t built = 0;
        std::thread front([&] {
            for (uint32_t i = 0; i < n_jobs; i++) {
                // nothing may leave this thread function (std::terminate would take the host
                // process down) and `built` must be published whatever happens, or the other
                // thread waits for ever
                int r;
                try {
                    // a new thread's current device is 0: the job's device first, then the lazy
                    // allocations of ensure_work (dev_alloc allocates on the current device)
                    if (hipSetDevice(jobs[i]->ctx->device) != hipSuccess) {
                        set_error("gp_jobs_polish: hipSetDevice failed");
                        r = GP_ERR_HIP;
                    } else
                        r = ensure_work(jobs[i], "gp_jobs_polish");
                    if (r == GP_OK) {
                        jobs[i]->ctx->stream = front_stream;
                        r = stage_a_run(jobs[i]); // returns with the filters complete (both its streams drained)
                    }
                } catch (...) {
                    r = on_exception("gp_jobs_polish (filter build)");
                }
                if (r != GP_OK) {
                    rc[i] = r;
                    try {
                        msg[i] = gp_last_error();
                    } catch (...) {
                    }
                }
                {
                    std::lock_guard<std::mutex> g(mu);
                    built = i + 1;
                }
                cv.notify_all();
            }
        });
        // joined on every way out of this scope, a throw from the loop below included (a joinable
        // std::thread's destructor terminates the process)
        struct Joiner {
            std::thread& t;
            ~Joiner()
            {
                if (t.joinable())
                    t.join();
            }
        } joiner{ front };
        for (uint32_t i = 0; i < n_jobs; i++) {
            {
                std::unique_lock<std::mutex> g(mu);
                cv.wait(g, [&] { return built > i; });
            }
            if (rc[i] == GP_OK) {
                jobs[i]->ctx->stream = back_stream;
                int r = hipSetDevice(jobs[i]->ctx->device) == hipSuccess ? polish_back(jobs[i]) : GP_ERR_HIP;
                if (r == GP_OK)
                    r = gp_job_sync(jobs[i]);
                if (r != GP_OK) {
                    rc[i] = r;
                    msg[i] = gp_last_error();
                }
            }
            jobs[i]->ctx->stream = own[i];
        }
    }
    for (uint32_t i = 0; i < n_jobs; i++)
        if (rc[i] != GP_OK) {
            set_error("job " + std::to_string(i) + ": " + msg[i]);
            return rc[i];
        }
    return GP_OK;
} GP_CATCH("gp_jobs_polish")

int gp_job_get_filter(gp_job* job, uint32_t batch, uint32_t k_index, uint8_t* out)
try {
    if (!job || !out || batch >= job->n_batches || k_index >= job->prm.n_k) {
        set_error("gp_job_get_filter: bad argument");
        return GP_ERR_INVALID_ARG;
    }
    GP_HIP(hipSetDevice(job->ctx->device));
    if (!job->filters_ready) {
        set_error("gp_job_get_filter: filters have not been built or set");
        return GP_ERR_STATE;
    }
    GP_HIP(hipStreamSynchronize(job->ctx->stream));
    const uint64_t words = job->prm.bf_bytes / 4;
    GP_HIP(hipMemcpyAsync(out, job->d_bf.p + ((uint64_t)batch * job->prm.n_k + k_index) * words,
                          job->prm.bf_bytes, hipMemcpyDeviceToHost, job->ctx->stream));
    GP_HIP(hipStreamSynchronize(job->ctx->stream));
    return GP_OK;
} GP_CATCH("gp_job_get_filter")

int gp_job_set_filters(gp_job* job, const uint8_t* filters)
try {
    if (!job || !filters)
        return GP_ERR_INVALID_ARG;
    GP_HIP(hipSetDevice(job->ctx->device));
    if (int rcw = ensure_work(job, "gp_job_set_filters"))
        return rcw;
    GP_HIP(hipStreamSynchronize(job->ctx->stream));
    GP_HIP(hipMemcpyAsync(job->d_bf.p, filters, (uint64_t)job->n_batches * job->prm.n_k * job->prm.bf_bytes,
                          hipMemcpyHostToDevice, job->ctx->stream));
    GP_HIP(hipStreamSynchronize(job->ctx->stream));
    job->filters_ready = true;
    return GP_OK;
} GP_CATCH("gp_job_set_filters")

int gp_job_get_lengths(gp_job* job, uint64_t* lengths, uint8_t* dropped, uint64_t* total)
try {
    if (!job)
        return GP_ERR_INVALID_ARG;
    GP_HIP(hipSetDevice(job->ctx->device));
    if (int rcb = ensure_work(job, "gp_job_get_lengths"))
        return rcb;
    return stage_lengths(job, lengths, dropped, total);
} GP_CATCH("gp_job_get_lengths")

int gp_job_get_sequences(gp_job* job, uint8_t* buf, uint64_t buf_bytes)
try {
    if (!job || !buf)
        return GP_ERR_INVALID_ARG;
    GP_HIP(hipSetDevice(job->ctx->device));
    if (int rcb = ensure_work(job, "gp_job_get_sequences"))
        return rcb;
    return stage_sequences(job, buf, buf_bytes);
} GP_CATCH("gp_job_get_sequences")

int gp_job_get_sequences_device(gp_job* job, void* device_buf, uint64_t buf_bytes)
try {
    if (!job || !device_buf)
        return GP_ERR_INVALID_ARG;
    GP_HIP(hipSetDevice(job->ctx->device));
    if (int rcb = ensure_work(job, "gp_job_get_sequences_device"))
        return rcb;
    return stage_sequences_device(job, device_buf, buf_bytes);
} GP_CATCH("gp_job_get_sequences_device")

int gp_job_get_counters(gp_job* job, gp_counters* out)
try {
    if (!job || !out)
        return GP_ERR_INVALID_ARG;
    *out = job->counters;
    return GP_OK;
} GP_CATCH("gp_job_get_counters")

int gp_job_kernel_time(gp_job* job, const char* kernel, double* total_ms, uint64_t* launches,
                       uint64_t* units)
try {
    if (!job || !kernel)
        return GP_ERR_INVALID_ARG;
    const KernelTimer* t = nullptr;
    if (!std::strcmp(kernel, "cbf_first_pass"))
        t = &job->t_first_pass;
    else if (!std::strcmp(kernel, "cbf_round"))
        t = &job->t_round;
    else if (!std::strcmp(kernel, "cbf_local"))
        t = &job->t_local;
    else if (!std::strcmp(kernel, "cbf_census"))
        t = &job->t_census;
    else if (!std::strcmp(kernel, "cbf_ckc"))
        t = &job->t_ckc; // (part of cbf_census's span: the candidates-by-k-mer kernels behind each cohort's pack, on the pack's stream)
    else if (!std::strcmp(kernel, "ntedit_scan"))
        t = &job->t_scan;
    else if (!std::strcmp(kernel, "ntedit_walk"))
        t = &job->t_walk;
    else if (!std::strcmp(kernel, "sealer_bfs"))
        t = &job->t_bfs;
    else if (!std::strcmp(kernel, "mask"))
        t = &job->t_mask;
    else if (!std::strcmp(kernel, "upper"))
        t = &job->t_upper;
    if (!t) {
        set_error("gp_job_kernel_time: unknown kernel name");
        return GP_ERR_INVALID_ARG;
    }
    if (total_ms)
        *total_ms = t->total_ms;
    if (launches)
        *launches = t->launches;
    if (units)
        *units = t->total_units;
    return GP_OK;
} GP_CATCH("gp_job_kernel_time")

} // extern "C"
