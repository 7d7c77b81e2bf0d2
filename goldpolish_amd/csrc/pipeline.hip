// pipeline.hip -- gp_pipeline_*: a stream of batch sets through one GPU, host memory to host
// memory (SURVEY.md 8d's timed region: reads + contigs resident in host memory -> polished bytes
// in host memory).  Host C++ only; the device work is gp_job_create / gp_job_polish.
//
// Three stages run side by side, each in a thread of its own, connected by bounded queues:
//   upload   gp_job_create of set i + 1 on the pipeline's upload context (pinned staging, DMA on
//            that context's stream) ...
//   polish   ... while set i runs the whole path on the polishing context; its polished records
//            are packed into an HBM buffer, copied to pinned host memory, and the job is destroyed
//            (its device blocks go to the pool and are what the next job's first use takes) ...
//   collect  ... while the caller still reads the result of set i - 1.
// Results come back in submission order -- the ordered append of scripts/goldpolish-reaper:51-73.
// The reference overlaps the same things with processes: its filter server builds the next
// batches' filters while polishing workers run (scripts/goldpolish:496-514).
//
// A set that holds a LONG contig (GP_PIPELINE_SPLIT_FROM bases, default 200 000) is split (round 4): a BUILD
// thread builds its filters on a context whose streams are restricted to 7 CUs in 8 (gp_ctx_create_masked)
// while TWO polishing threads run everything behind the filters of the two sets before it on the eighth
// (a masked context each).  GP_PIPELINE_OVERLAP: 3 (default) as said; 0 never; 1 every set; 2 every set, no
// masks; GP_PIPELINE_BACK_SHARE: CUs in 8 for the back halves (default 1); GP_PIPELINE_BACK_THREADS (2).
// Measured, ms per step:
//   BASELINE configs[2] (4000 x 25 kbp, 40x): three stages 920; every set split, masked 7 | 1: 909 (the filter
//   build's kernels take 862 ms beside the back half instead of 664 alone, the back half 313 instead of 263);
//   unmasked 906; device memory 62 -> 90 GiB.  The two halves do run side by side (kernel trace) and the sum
//   does not move: the build's streaming kernels fill every wave slot they are given, and what the back half
//   takes it takes from them.  So read-length sets stay whole.
//   SURVEY 8d config 3's own shape (48 contigs of 0.3-12 Mbp), on the round's last filter build (3.4 s alone):
//   whole 8 160; split without masks 6 930; split 6 | 2, one polishing thread 6 500 (2, 3 or 4 CUs in 8 for
//   the back half: the same -- it waits for its slowest segments and searches); two
//   polishing threads 5 700 in steady state (7 | 1 and 6 | 2 alike; the "5 450" a first run gave was the start
//   of the stream: time such a pipeline from its fourth set on).  GP_PIPELINE_TRACE=1 prints every stage of every set.
//   With the waits described next gone, read-length sets were tried again: every set split 877 ms per
//   configs[2] step against 916 whole (the census kernels 792 instead of 580 ms beside a back half, 88 instead
//   of 65 GiB): 4 %, still not the default.
//   What the trace showed first: build and back half ENDING TOGETHER, set after set, 6.3 s each -- every job
//   made and destroyed its filter build's streams, events and pinned ring, hipHostFree / hipStreamDestroy
//   wait for the whole device, and the back half never leaves it idle.  Those are the context's now
//   (gp_ctx::StageA), made once.
// A split set of fewer than GP_PIPELINE_MASK_FROM contigs (default 4) runs its halves on contexts WITHOUT masks
// (the build on front_plain, the back half on run_ctx): see gp_pipeline_create.
// (With CU-masked streams nothing in a job's path may touch the null stream: masked streams are created
// BLOCKING, and a synchronous hipMemcpy / hipMemset waits for every blocking stream of the device -- the
// first version of this ran the two halves strictly one after the other for that reason.  And a mask must
// be interleaved: build on CUs [0, 192) | back half on [192, 256) took 1139 ms for two jobs of 1000 contigs
// against 561 for 7-in-8 | 1-in-8 and 601 for one job after the other -- a range of CU bits is whole
// shader engines, and workgroups are still dealt to all of them.)
#include "gp_internal.h"

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

using namespace gp;

namespace {

const char* const kTimerNames[] = { "cbf_first_pass", "cbf_round", "cbf_local", "cbf_census", "ntedit_scan",
                                    "ntedit_walk",    "sealer_bfs", "mask",     "upper",      "cbf_ckc" };
constexpr int kTimers = 10;

struct Slot { // one finished set
    uint64_t tag = 0;
    uint64_t seq = 0; // its place in the order of submission
    int rc = GP_OK;
    std::string msg;
    std::vector<uint64_t> lengths;
    std::vector<uint8_t> dropped;
    uint64_t total = 0;
    uint8_t* h_bytes = nullptr; // pinned, grows
    uint64_t h_cap = 0;
    DevBuf<uint8_t> d_bytes;
    gp_counters counters{};
    double upload_ms = 0, polish_ms = 0, download_ms = 0;
    double t_ms[kTimers] = {};
    uint64_t t_launches[kTimers] = {}, t_units[kTimers] = {};
    uint64_t device_bytes = 0;
};

struct Item {
    gp_batch_set set{};
    uint64_t tag = 0;
    gp_job* job = nullptr;
    int rc = GP_OK;
    std::string msg;
    double upload_ms = 0, build_ms = 0;
    uint64_t seq = 0;   // order of submission = order of collection
    bool split = false; // filter build and back half in two threads on two contexts ...
    bool masked = false; // ... CU-masked ones
};

double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// development aid (GP_PIPELINE_TRACE=1): one line per stage of every set on stderr, wall-clock ms
void trace(const char* what, uint64_t seq, double t0)
{
    static const bool on = getenv("GP_PIPELINE_TRACE") != nullptr;
    static const double origin = now_ms();
    if (on)
        fprintf(stderr, "PIPE set %llu %-8s %10.1f .. %10.1f ms (%.1f)\n", (unsigned long long)seq, what, t0 - origin, now_ms() - origin, now_ms() - t0);
}

} // namespace

struct gp_pipeline {
    gp_ctx* up_ctx = nullptr;
    gp_ctx* front_ctx = nullptr; // the filter build's, for sets that are split (null: never; build and back half both on run_ctx, in the polishing thread)
    gp_ctx* back_ctx = nullptr;  // the back half's of a split set
    gp_ctx* run_ctx = nullptr;   // the whole path of a set that is not split
    uint64_t split_from = 0;     // sets with a contig of this many bases or more are split (0 with front_ctx: every set)
    gp_ctx* front_plain = nullptr; // the filter build's for split sets of fewer than mask_from contigs: no CU mask (their back half: run_ctx)
    uint32_t mask_from = 0;
    gp_params prm{};
    uint32_t depth = 2;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Item> submitted; // waiting for the upload thread
    std::deque<Item> uploaded;  // on the device, waiting for the build thread (one at most)
    std::deque<Item> built;     // filters built, waiting for the polishing thread (one at most)
    std::deque<int> done;       // slots ready for collect (in the order they finished; collect takes them by seq)
    std::vector<int> free_slots;
    Slot slots[4];
    uint64_t next_seq = 0, next_out = 0; // submitted so far / the set the next collect returns
    gp_ctx* back_ctx2 = nullptr; // the second polishing thread's (split, masked sets only)
    int held = -1;              // slot the caller is reading (released by the next collect)
    uint32_t in_flight = 0;     // submitted and not yet collected
    bool stop = false;
    std::thread uploader, builder, polisher, polisher2;
};

static void upload_loop(gp_pipeline* p)
{
    for (;;) {
        Item it;
        {
            std::unique_lock<std::mutex> g(p->mu);
            p->cv.wait(g, [&] { return p->stop || (!p->submitted.empty() && p->uploaded.empty()); });
            if (p->stop)
                return;
            it = std::move(p->submitted.front());
            p->submitted.pop_front();
        }
        const double t0 = now_ms();
        try {
            it.rc = gp_job_create(p->up_ctx, &p->prm, &it.set, &it.job);
            if (it.rc != GP_OK)
                it.msg = gp_last_error();
        } catch (...) {
            it.rc = on_exception("gp_pipeline (upload)");
        }
        it.upload_ms = now_ms() - t0;
        trace("upload", it.seq, t0);
        {
            std::lock_guard<std::mutex> g(p->mu);
            p->uploaded.push_back(std::move(it));
        }
        p->cv.notify_all();
    }
}

static void build_loop(gp_pipeline* p)
{
    for (;;) {
        Item it;
        {
            std::unique_lock<std::mutex> g(p->mu);
            p->cv.wait(g, [&] { return p->stop || (!p->uploaded.empty() && p->built.empty()); });
            if (p->stop)
                return;
            it = std::move(p->uploaded.front());
            p->uploaded.pop_front();
        }
        p->cv.notify_all(); // the upload thread may start on the next set
        if (it.rc == GP_OK && it.split) {
            const double t0 = now_ms();
            try {
                it.job->ctx = it.masked ? p->front_ctx : p->front_plain; // the upload is complete (gp_job_create waits for it)
                it.rc = job_polish_front(it.job);
                if (it.rc != GP_OK)
                    it.msg = gp_last_error();
            } catch (...) {
                it.rc = on_exception("gp_pipeline (filter build)");
                try {
                    it.msg = gp_last_error();
                } catch (...) {
                }
            }
            it.build_ms = now_ms() - t0;
            trace("build", it.seq, t0);
        }
        {
            std::lock_guard<std::mutex> g(p->mu);
            p->built.push_back(std::move(it));
        }
        p->cv.notify_all();
    }
}

static int polish_one(gp_pipeline* p, Item& it, Slot& s, gp_ctx* back)
{
    gp_job* job = it.job;
    gp_ctx* const ctx = (it.split && it.masked) ? back : p->run_ctx;
    job->ctx = ctx; // from here on the polishing context's stream
    double t0 = now_ms();
    int rc = it.split ? job_polish_back(job) : gp_job_polish(job, 1);
    if (rc != GP_OK)
        return rc;
    s.polish_ms = now_ms() - t0 + it.build_ms;
    t0 = now_ms();
    s.lengths.assign(job->n_contigs ? job->n_contigs : 1, 0);
    s.dropped.assign(s.lengths.size(), 0);
    if ((rc = gp_job_get_lengths(job, s.lengths.data(), s.dropped.data(), &s.total)) != GP_OK)
        return rc;
    if (s.h_cap < s.total || !s.h_bytes) {
        if (s.h_bytes)
            (void)hipHostFree(s.h_bytes);
        s.h_bytes = nullptr;
        s.h_cap = s.total + s.total / 8 + 4096;
        if (hipHostMalloc((void**)&s.h_bytes, s.h_cap) != hipSuccess) {
            s.h_bytes = nullptr;
            s.h_cap = 0;
            (void)hipGetLastError();
            set_error("gp_pipeline: pinned result buffer allocation failed");
            return GP_ERR_OOM;
        }
    }
    if ((rc = s.d_bytes.ensure(s.total + 64)) != GP_OK)
        return rc;
    if (s.total) {
        if ((rc = gp_job_get_sequences_device(job, s.d_bytes.p, s.total)) != GP_OK)
            return rc;
        GP_HIP(hipMemcpyAsync(s.h_bytes, s.d_bytes.p, s.total, hipMemcpyDeviceToHost, ctx->stream));
        GP_HIP(hipStreamSynchronize(ctx->stream));
    }
    s.download_ms = now_ms() - t0;
    s.counters = job->counters;
    for (int i = 0; i < kTimers; i++)
        (void)gp_job_kernel_time(job, kTimerNames[i], &s.t_ms[i], &s.t_launches[i], &s.t_units[i]);
    s.device_bytes = gp_device_bytes_in_use();
    return GP_OK;
}

// `second`: the second polishing thread.  The back half of a set with long contigs is as long as its slowest
// members -- a walk segment edited every other base, a contig walked whole after a refused insertion, the
// longest one-lane Sealer searches -- and leaves most of the chip idle while it waits for them (the same
// 6.5 s on 2, 3 or 4 CUs in 8: section 10 of DESIGN.md), so two such back halves run side by side, each on
// a masked context of its own.  The second thread takes nothing else: sets that run whole, and split sets
// without masks, use run_ctx, which is the first thread's.
static void polish_loop(gp_pipeline* p, bool second)
{
    gp_ctx* const back = second ? p->back_ctx2 : p->back_ctx;
    for (;;) {
        Item it;
        int si;
        {
            std::unique_lock<std::mutex> g(p->mu);
            std::deque<Item>& from = p->front_ctx ? p->built : p->uploaded;
            p->cv.wait(g, [&] {
                return p->stop || (!from.empty() && !p->free_slots.empty() && (!second || (from.front().split && from.front().masked)));
            });
            if (p->stop)
                return;
            it = std::move(from.front());
            from.pop_front();
            si = p->free_slots.back();
            p->free_slots.pop_back();
        }
        p->cv.notify_all(); // the stage before may start on the next set
        Slot& s = p->slots[si];
        s.tag = it.tag;
        s.seq = it.seq;
        s.rc = it.rc;
        s.msg = it.msg;
        s.upload_ms = it.upload_ms;
        s.total = 0;
        if (it.rc == GP_OK) {
            const double t_back = now_ms();
            try {
                (void)hipSetDevice(p->run_ctx->device);
                s.rc = polish_one(p, it, s, back);
                trace(second ? "back/2" : (it.split ? "back/1" : "whole"), it.seq, t_back);
                if (s.rc != GP_OK)
                    s.msg = gp_last_error();
            } catch (...) {
                s.rc = on_exception("gp_pipeline (polish)");
                try {
                    s.msg = gp_last_error();
                } catch (...) {
                }
            }
        }
        if (it.job) {
            it.job->ctx = (it.split && it.masked) ? back : p->run_ctx;
            gp_job_destroy(it.job); // its blocks go to the pool: the next job's work buffers
        }
        {
            std::lock_guard<std::mutex> g(p->mu);
            p->done.push_back(si);
        }
        p->cv.notify_all();
    }
}

extern "C" {

int gp_pipeline_create(int device, const gp_params* params, uint32_t depth, gp_pipeline** out)
try {
    if (!params || !out) {
        set_error("gp_pipeline_create: NULL argument");
        return GP_ERR_INVALID_ARG;
    }
    *out = nullptr;
    gp_pipeline* p = new (std::nothrow) gp_pipeline();
    if (!p)
        return GP_ERR_OOM;
    struct Owner {
        gp_pipeline* p;
        ~Owner()
        {
            if (p)
                gp_pipeline_destroy(p);
        }
    } owner{ p };
    int rc;
    if ((rc = gp_ctx_create(device, &p->up_ctx)) != GP_OK)
        return rc;
    // GP_PIPELINE_OVERLAP: 0 never split a set; 1 split every set (CU-masked contexts); 2 split every set,
    // no masks; 3 (default) split the sets that hold a long contig (GP_PIPELINE_SPLIT_FROM bases, default
    // 200 000): there the back half -- segments of long contigs walked by thousands of workgroups, searches
    // that branch -- is as long as the filter build and the two overlap for real (SURVEY 8d config 3's shape:
    // 11.3 -> ... s per step), where for read-length contigs the sum does not move (header of this file).
    const uint64_t overlap = env_u64("GP_PIPELINE_OVERLAP", 3);
    if ((rc = gp_ctx_create(device, &p->run_ctx)) != GP_OK)
        return rc;
    if (overlap != 0) {
        // the back half on `share` CUs of every eight (bit i of the mask = CU i; interleaved, so that each half
        // gets its share of every shader engine), the filter build on the others
        const uint32_t n_cu = (uint32_t)p->up_ctx->prop.multiProcessorCount;
        const uint32_t share = (uint32_t)std::min<uint64_t>(7, std::max<uint64_t>(1, env_u64("GP_PIPELINE_BACK_SHARE", 1)));
        std::vector<uint32_t> front((n_cu + 31) / 32, 0), back((n_cu + 31) / 32, 0);
        for (uint32_t cu = 0; cu < n_cu; cu++)
            (cu % 8 >= 8 - share ? back : front)[cu / 32] |= 1u << (cu % 32);
        if (overlap != 2 && n_cu >= 16) {
            if ((rc = gp_ctx_create_masked(device, front.data(), (uint32_t)front.size(), &p->front_ctx)) != GP_OK ||
                (rc = gp_ctx_create_masked(device, back.data(), (uint32_t)back.size(), &p->back_ctx)) != GP_OK)
                return rc;
        } else if ((rc = gp_ctx_create(device, &p->front_ctx)) != GP_OK || (rc = gp_ctx_create(device, &p->back_ctx)) != GP_OK) {
            return rc;
        }
        p->split_from = overlap == 3 ? env_u64("GP_PIPELINE_SPLIT_FROM", 200000) : 0;
        // A split set of FEWER than GP_PIPELINE_MASK_FROM contigs (default 4; 0: masks always) takes contexts
        // without masks: its filter build is the grid-wide schedule on a handful of filters -- thousands of
        // small launches steered by host feedback -- and on CU-masked streams the gaps between those launches
        // grow (BASELINE configs[1], one 10 Mbp contig: 53 ms per step whole, 88 split with masks, 48 split
        // without; SURVEY 8d config 3's 48 contigs: 8.2 s whole, 6.9 split without masks, 6.5 with).
        p->mask_from = overlap == 3 ? (uint32_t)env_u64("GP_PIPELINE_MASK_FROM", 4) : 0;
        if (p->mask_from && (rc = gp_ctx_create(device, &p->front_plain)) != GP_OK)
            return rc;
        // two back halves of split, masked sets at once (polish_loop; GP_PIPELINE_BACK_THREADS=1: one)
        if (overlap == 3 && n_cu >= 16 && env_u64("GP_PIPELINE_BACK_THREADS", 2) >= 2 &&
            (rc = gp_ctx_create_masked(device, back.data(), (uint32_t)back.size(), &p->back_ctx2)) != GP_OK)
            return rc;
    }
    p->prm = *params;
    p->depth = depth < 1 ? 1 : depth > 8 ? 8 : depth;
    for (int i = 0; i < 4; i++)
        p->free_slots.push_back(i);
    p->uploader = std::thread(upload_loop, p);
    if (p->front_ctx)
        p->builder = std::thread(build_loop, p);
    p->polisher = std::thread(polish_loop, p, false);
    if (p->back_ctx2)
        p->polisher2 = std::thread(polish_loop, p, true);
    owner.p = nullptr;
    *out = p;
    return GP_OK;
} GP_CATCH("gp_pipeline_create")

int gp_pipeline_submit(gp_pipeline* p, const gp_batch_set* set, uint64_t tag)
try {
    if (!p || !set) {
        set_error("gp_pipeline_submit: NULL argument");
        return GP_ERR_INVALID_ARG;
    }
    Item it;
    it.set = *set;
    it.tag = tag;
    if (p->front_ctx) {
        uint64_t longest = 0;
        if (p->split_from && set->contig_off)
            for (uint32_t c = 0; c < set->n_contigs; c++)
                longest = std::max<uint64_t>(longest, set->contig_off[c + 1] - set->contig_off[c]);
        it.split = p->split_from == 0 || longest >= p->split_from;
        it.masked = it.split && set->n_contigs >= p->mask_from;
    }
    {
        std::unique_lock<std::mutex> g(p->mu);
        if (p->in_flight >= p->depth) {
            // the caller is the one who collects: waiting here would never end
            set_error("gp_pipeline_submit: " + std::to_string(p->depth) + " sets are in flight; collect one first");
            return GP_ERR_STATE;
        }
        p->in_flight++;
        it.seq = p->next_seq++;
        p->submitted.push_back(std::move(it));
    }
    p->cv.notify_all();
    return GP_OK;
} GP_CATCH("gp_pipeline_submit")

int gp_pipeline_collect(gp_pipeline* p, gp_polished* out)
try {
    if (!p || !out) {
        set_error("gp_pipeline_collect: NULL argument");
        return GP_ERR_INVALID_ARG;
    }
    std::memset(out, 0, sizeof *out);
    int si;
    {
        std::unique_lock<std::mutex> g(p->mu);
        if (p->held >= 0) { // the result of the collect before this one is released
            p->free_slots.push_back(p->held);
            p->held = -1;
            p->cv.notify_all();
        }
        if (p->in_flight == 0) {
            set_error("gp_pipeline_collect: nothing has been submitted");
            return GP_ERR_STATE;
        }
        // results come back in submission order, whichever polishing thread finished first
        auto ready = [&] {
            for (auto d = p->done.begin(); d != p->done.end(); ++d)
                if (p->slots[*d].seq == p->next_out)
                    return d;
            return p->done.end();
        };
        p->cv.wait(g, [&] { return ready() != p->done.end(); });
        auto d = ready();
        si = *d;
        p->done.erase(d);
        p->next_out++;
        p->held = si;
        p->in_flight--;
    }
    const Slot& s = p->slots[si];
    out->tag = s.tag;
    out->upload_ms = s.upload_ms;
    if (s.rc != GP_OK) {
        set_error("set " + std::to_string(s.tag) + ": " + s.msg);
        return s.rc;
    }
    out->n_contigs = (uint32_t)s.lengths.size();
    out->lengths = s.lengths.data();
    out->dropped = s.dropped.data();
    out->host_bytes = s.h_bytes;
    out->device_bytes = s.d_bytes.p;
    out->total_bytes = s.total;
    out->counters = s.counters;
    out->polish_ms = s.polish_ms;
    out->download_ms = s.download_ms;
    out->device_bytes_in_use = s.device_bytes;
    return GP_OK;
} GP_CATCH("gp_pipeline_collect")

int gp_pipeline_kernel_time(gp_pipeline* p, const char* kernel, double* total_ms, uint64_t* launches, uint64_t* units)
try {
    if (!p || !kernel)
        return GP_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> g(p->mu);
    if (p->held < 0) {
        set_error("gp_pipeline_kernel_time: no collected set");
        return GP_ERR_STATE;
    }
    const Slot& s = p->slots[p->held];
    for (int i = 0; i < kTimers; i++)
        if (!std::strcmp(kernel, kTimerNames[i])) {
            if (total_ms)
                *total_ms = s.t_ms[i];
            if (launches)
                *launches = s.t_launches[i];
            if (units)
                *units = s.t_units[i];
            return GP_OK;
        }
    set_error("gp_pipeline_kernel_time: unknown kernel name");
    return GP_ERR_INVALID_ARG;
} GP_CATCH("gp_pipeline_kernel_time")

void gp_pipeline_destroy(gp_pipeline* p)
{
    if (!p)
        return;
    try {
        {
            std::lock_guard<std::mutex> g(p->mu);
            p->stop = true;
        }
        p->cv.notify_all();
        if (p->uploader.joinable())
            p->uploader.join();
        if (p->builder.joinable())
            p->builder.join();
        if (p->polisher.joinable())
            p->polisher.join();
        if (p->polisher2.joinable())
            p->polisher2.join();
        for (Item& it : p->uploaded)
            if (it.job)
                gp_job_destroy(it.job);
        for (Item& it : p->built)
            if (it.job)
                gp_job_destroy(it.job);
        if (p->run_ctx)
            (void)hipSetDevice(p->run_ctx->device);
        for (Slot& s : p->slots) {
            if (s.h_bytes)
                (void)hipHostFree(s.h_bytes);
            s.d_bytes.release();
        }
        gp_ctx_destroy(p->up_ctx);
        gp_ctx_destroy(p->front_ctx);
        gp_ctx_destroy(p->front_plain);
        gp_ctx_destroy(p->back_ctx);
        gp_ctx_destroy(p->back_ctx2);
        gp_ctx_destroy(p->run_ctx);
        delete p;
    } catch (...) {
    }
}

} // extern "C"
