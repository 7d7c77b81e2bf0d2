// Micro-probe: rate of scattered no-return atomics on MI355X as a function of table size,
// operand width and scope.  Not part of the product; informs stage A's reservation design.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

template <int MODE>
__global__ void probe(unsigned long long* t64, uint32_t* t32, uint8_t* t8, uint32_t bits, uint64_t n, uint64_t seed)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n; i += stride) {
        uint64_t h = mix(i + seed);
        uint64_t s = h >> (64 - bits);
        if (MODE == 0) atomicMin(&t64[s], h);
        else if (MODE == 1) atomicMin(&t32[s], (uint32_t)h);
        else if (MODE == 2) acc += (uint32_t)atomicMin(&t64[s], h);
        else if (MODE == 3) acc += t8[s];                       // random byte load
        else if (MODE == 4) t8[s] = (uint8_t)h;                 // random byte store
        else if (MODE == 5) atomicOr(&t32[s], 1u << (h & 31));
        else if (MODE == 6) __hip_atomic_fetch_min(&t64[s], h, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 7) acc += (uint32_t)__hip_atomic_load(&t64[s], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 8) acc += (uint32_t)t64[s];
    }
    if (acc == 0x12345678) t32[0] = acc;
}

template <int MODE>
void run(const char* name, void* buf, uint32_t bits, uint64_t n, int blocks = 4096, int threads = 256)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemset(buf, 0xFF, (size_t)8 << bits));
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, (unsigned long long*)buf, (uint32_t*)buf, (uint8_t*)buf, bits, n, (uint64_t)rep * 7919);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("%-28s slots 2^%-2u blocks %5d x %4d  %8.3f ms  %7.2f G/s\n", name, bits, blocks, threads, best, n / best / 1e6);
}

int main()
{
    void* buf; CK(hipMalloc(&buf, (size_t)8 << 27));
    const uint64_t n = 64ull << 20;
    for (int blocks : {32, 64, 128, 256, 512, 1024, 2048})
        for (int threads : {64, 256, 1024})
            run<0>("atomicMin u64 noret", buf, 22, n / 8, blocks, threads);
    for (uint32_t bits : {22u}) {
        run<0>("atomicMin u64 noret", buf, bits, n);
        run<1>("atomicMin u32 noret", buf, bits, n);
        run<2>("atomicMin u64 returning", buf, bits, n);
        run<6>("atomicMin u64 wg-scope", buf, bits, n);
        run<5>("atomicOr u32 noret", buf, bits, n);
        run<3>("byte load", buf, bits, n);
        run<4>("byte store", buf, bits, n);
        run<7>("u64 load sc1", buf, bits, n);
        run<8>("u64 load plain", buf, bits, n);
    }
    return 0;
}
