// Micro-probe for an atomics-free "owner computes" schedule of stage A's ordered operations:
// every counter touch is routed to the workgroup that owns its counter range (two-level
// bucketing), ordered and applied there in LDS, and a per-operation kernel combines the four
// answers.  Random indices, no hashing: this measures the cost of the plumbing per step
// (N new operations + about as many carried over) against the 0.26 ms the reservation
// schedule needs for the same step.  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr uint32_t kCounters = 10u << 20;
constexpr uint32_t kR = 4096;                       // counters per fine partition
constexpr uint32_t kFine = kCounters / kR;          // 2560
constexpr uint32_t kFinePerCoarse = 64;
constexpr uint32_t kCoarse = kFine / kFinePerCoarse; // 40
constexpr uint32_t kSubs = 256;                     // sub-segments per fine bucket (one per PART lane)
constexpr uint32_t kSubCap = 48;
constexpr uint32_t kCoarseCap = 1u << 19;

__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

// K1: every new operation emits four touches into the coarse buckets (LDS-aggregated cursors)
__global__ __launch_bounds__(256) void emit(uint4* coarse, uint32_t* coarse_n, uint32_t n_ops, uint32_t id0, uint64_t seed,
                                            uint32_t* op_idx)
{
    __shared__ uint32_t hist[kCoarse], base[kCoarse];
    if (threadIdx.x < kCoarse) hist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t idx[4][4], rk[4][4];
    const uint32_t o0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    for (int j = 0; j < 4; j++)
        for (int h = 0; h < 4; h++) {
            idx[j][h] = (uint32_t)(mix((uint64_t)(id0 + o0 + j) * 4 + h + seed) % kCounters);
            rk[j][h] = o0 + j < n_ops ? atomicAdd(&hist[idx[j][h] / (kR * kFinePerCoarse)], 1u) : 0;
        }
    __syncthreads();
    if (threadIdx.x < kCoarse) base[threadIdx.x] = atomicAdd(&coarse_n[threadIdx.x], hist[threadIdx.x]);
    __syncthreads();
    for (int j = 0; j < 4; j++) {
        if (o0 + j >= n_ops) break;
        const uint32_t id = id0 + o0 + j;
        reinterpret_cast<uint4*>(op_idx)[id] = make_uint4(idx[j][0], idx[j][1], idx[j][2], idx[j][3]);
        for (int h = 0; h < 4; h++) {
            const uint32_t c = idx[j][h] / (kR * kFinePerCoarse);
            const uint32_t at = base[c] + rk[j][h];
            if (at < kCoarseCap)
                coarse[(uint64_t)c * kCoarseCap + at] = make_uint4(id, id /*pos*/, idx[j][h], h);
        }
    }
}

// K1b: one workgroup per coarse bucket spreads it over its 64 fine buckets x 256 sub-segments
__global__ __launch_bounds__(256) void split(const uint4* coarse, uint32_t* coarse_n, uint4* fine, uint16_t* sub_n)
{
    __shared__ uint16_t cur[kFinePerCoarse * kSubs]; // 32 KB
    const uint32_t c = blockIdx.x;
    uint16_t* g = sub_n + (uint64_t)c * kFinePerCoarse * kSubs;
    for (uint32_t i = threadIdx.x; i < kFinePerCoarse * kSubs; i += 256) cur[i] = g[i];
    __syncthreads();
    const uint32_t n = min(coarse_n[c], kCoarseCap);
    for (uint32_t i = threadIdx.x; i < n; i += 256) {
        const uint4 t = coarse[(uint64_t)c * kCoarseCap + i];
        const uint32_t f = (t.z / kR) % kFinePerCoarse, sub = (t.x * 2654435761u) >> 24;
        // 16-bit LDS counter: bump through its 32-bit word
        uint32_t* w = reinterpret_cast<uint32_t*>(cur) + ((f * kSubs + sub) >> 1);
        const uint32_t sh = ((f * kSubs + sub) & 1) * 16;
        const uint32_t old = (atomicAdd(w, 1u << sh) >> sh) & 0xFFFF;
        if (old < kSubCap)
            fine[(((uint64_t)c * kFinePerCoarse + f) * kSubs + sub) * kSubCap + old] = t;
    }
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < kFinePerCoarse * kSubs; i += 256) g[i] = min(cur[i], (uint16_t)kSubCap);
    if (threadIdx.x == 0) coarse_n[c] = 0;
}

// PART: one workgroup per fine partition, lane t owns sub-segment t
__global__ __launch_bounds__(256) void part(uint4* fine, uint16_t* sub_n, uint8_t* counters, const uint8_t* cmd, uint8_t* result)
{
    __shared__ uint8_t ctr[kR];
    __shared__ uint32_t min_pos[kR];
    const uint32_t f = blockIdx.x, t = threadIdx.x;
    for (uint32_t i = t; i < kR / 16; i += 256)
        reinterpret_cast<uint4*>(ctr)[i] = reinterpret_cast<const uint4*>(counters + (uint64_t)f * kR)[i];
    for (uint32_t i = t; i < kR; i += 256) min_pos[i] = 0xFFFFFFFFu;
    __syncthreads();
    uint4* seg = fine + ((uint64_t)f * kSubs + t) * kSubCap;
    const uint32_t n = sub_n[(uint64_t)f * kSubs + t];
    uint32_t kept = 0;
    for (uint32_t j0 = 0; j0 < n; j0 += 4) {
        uint4 tt[4]; uint8_t cc[4];
        for (int q = 0; q < 4; q++) if (j0 + q < n) tt[q] = seg[j0 + q];
        for (int q = 0; q < 4; q++) if (j0 + q < n) cc[q] = cmd[tt[q].x];
        for (int q = 0; q < 4; q++) {
            if (j0 + q >= n) break;
            const uint32_t local = tt[q].z % kR;
            if (cc[q] & 16) {
                if (cc[q] & (1u << tt[q].w)) ctr[local]++;
            } else {
                seg[kept++] = tt[q];
                atomicMin(&min_pos[local], tt[q].y);
            }
        }
    }
    sub_n[(uint64_t)f * kSubs + t] = (uint16_t)kept;
    __syncthreads();
    for (uint32_t j = 0; j < kept; j++) {
        const uint4 tt = seg[j];
        const uint32_t local = tt.z % kR;
        result[(uint64_t)tt.x * 4 + tt.w] = (uint8_t)((min_pos[local] == tt.y ? 128u : 0u) | ctr[local]);
    }
    __syncthreads();
    for (uint32_t i = t; i < kR / 16; i += 256)
        reinterpret_cast<uint4*>(counters + (uint64_t)f * kR)[i] = reinterpret_cast<const uint4*>(ctr)[i];
}

// OP: per waiting operation, combine the four answers
__global__ __launch_bounds__(256) void op(const uint32_t* pend_in, uint32_t n_in, uint32_t* pend_out, uint32_t* n_out,
                                          const uint32_t* result, uint8_t* cmd, const uint32_t* op_idx)
{
    __shared__ uint32_t s_n, s_base;
    if (threadIdx.x == 0) s_n = 0;
    __syncthreads();
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    uint32_t id = 0; bool waits = false; uint32_t rank = 0;
    if (i < n_in) {
        id = pend_in[i];
        const uint32_t r = result[id];
        const uint4 ix = reinterpret_cast<const uint4*>(op_idx)[id];
        if ((r & 0x80808080u) == 0x80808080u) {
            const uint32_t v0 = r & 127, v1 = (r >> 8) & 127, v2 = (r >> 16) & 127, v3 = (r >> 24) & 127;
            const uint32_t m = min(min(v0, v1), min(v2, v3));
            uint32_t c = 16;
            if (m < 11) c |= (v0 == m) | ((v1 == m && ix.y != ix.x) << 1) | ((v2 == m) << 2) | ((v3 == m) << 3);
            cmd[id] = (uint8_t)c;
        } else {
            waits = true;
            rank = atomicAdd(&s_n, 1u);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) s_base = s_n ? atomicAdd(n_out, s_n) : 0;
    __syncthreads();
    if (waits) pend_out[s_base + rank] = id;
}

__global__ void iota(uint32_t* p, uint32_t a, uint32_t n) { uint32_t i = blockIdx.x * 256 + threadIdx.x; if (i < n) p[i] = a + i; }

int main()
{
    const uint32_t N = 436224, STEPS = 60, MAXOPS = N * (STEPS + 2);
    uint4 *coarse, *fine; uint32_t *coarse_n, *op_idx, *result, *pend[2], *n_pend; uint16_t* sub_n; uint8_t *counters, *cmd;
    CK(hipMalloc(&coarse, (size_t)kCoarse * kCoarseCap * 16));
    CK(hipMalloc(&fine, (size_t)kFine * kSubs * kSubCap * 16));
    CK(hipMalloc(&coarse_n, kCoarse * 4)); CK(hipMemset(coarse_n, 0, kCoarse * 4));
    CK(hipMalloc(&sub_n, (size_t)kFine * kSubs * 2)); CK(hipMemset(sub_n, 0, (size_t)kFine * kSubs * 2));
    CK(hipMalloc(&counters, kCounters)); CK(hipMemset(counters, 0, kCounters));
    CK(hipMalloc(&cmd, MAXOPS)); CK(hipMemset(cmd, 0, MAXOPS));
    CK(hipMalloc(&result, (size_t)MAXOPS * 4)); CK(hipMemset(result, 0, (size_t)MAXOPS * 4));
    CK(hipMalloc(&op_idx, (size_t)MAXOPS * 16));
    CK(hipMalloc(&pend[0], (size_t)MAXOPS * 4)); CK(hipMalloc(&pend[1], (size_t)MAXOPS * 4));
    CK(hipMalloc(&n_pend, 8)); CK(hipMemset(n_pend, 0, 8));
    hipEvent_t ev[5]; for (auto& evt : ev) CK(hipEventCreate(&evt));
    uint32_t n_in = 0, id0 = 0; float acc[4] = { 0, 0, 0, 0 }; uint32_t measured = 0;
    for (uint32_t s = 0; s < STEPS; s++) {
        const bool hash = s % 2 == 0; // cadence 2: every other step takes a new slice
        uint32_t* in = pend[s & 1]; uint32_t* out = pend[(s + 1) & 1];
        CK(hipEventRecord(ev[0]));
        if (hash) {
            hipLaunchKernelGGL(emit, dim3((N + 1023) / 1024), dim3(256), 0, 0, coarse, coarse_n, N, id0, 12345ull, op_idx);
            hipLaunchKernelGGL(iota, dim3((N + 255) / 256), dim3(256), 0, 0, in + n_in, id0, N);
        }
        CK(hipEventRecord(ev[1]));
        if (hash) hipLaunchKernelGGL(split, dim3(kCoarse), dim3(256), 0, 0, coarse, coarse_n, fine, sub_n);
        CK(hipEventRecord(ev[2]));
        hipLaunchKernelGGL(part, dim3(kFine), dim3(256), 0, 0, fine, sub_n, counters, cmd, reinterpret_cast<uint8_t*>(result));
        CK(hipEventRecord(ev[3]));
        const uint32_t n_total = n_in + (hash ? N : 0);
        CK(hipMemsetAsync(n_pend, 0, 4));
        hipLaunchKernelGGL(op, dim3((n_total + 255) / 256), dim3(256), 0, 0, in, n_total, out, n_pend, result, cmd, op_idx);
        CK(hipEventRecord(ev[4]));
        CK(hipMemcpy(&n_in, n_pend, 4, hipMemcpyDeviceToHost));
        if (hash) id0 += N;
        if (s >= 20) {
            for (int i = 0; i < 4; i++) { float ms; CK(hipEventElapsedTime(&ms, ev[i], ev[i + 1])); acc[i] += ms; }
            measured++;
        }
        if (s % 10 == 9) printf("step %u: processed %u, still waiting %u\n", s, n_total, n_in);
    }
    printf("per step (avg of %u, half of them with a new slice of %u operations): emit %.4f  split %.4f  part %.4f  op %.4f  sum %.4f ms\n",
           measured, N, acc[0] / measured, acc[1] / measured, acc[2] / measured, acc[3] / measured,
           (acc[0] + acc[1] + acc[2] + acc[3]) / measured);
    printf("per slice (two steps): %.4f ms   [reservation schedule: 0.256 ms]\n", 2 * (acc[0] + acc[1] + acc[2] + acc[3]) / measured);
    return 0;
}
