// Micro-probe: random 1-byte loads and load+store pairs over tables of 64 MiB .. 16 GiB
// (what cbf_local's counter traffic looks like).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
__device__ inline uint64_t mix(uint64_t x) { x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33; return x; }

template <int MODE>
__global__ void probe(uint8_t* t, uint32_t bits, uint64_t n, uint64_t seed)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    uint32_t acc = 0;
    for (; i < n; i += 4 * stride) {
        uint64_t s[4];
        uint8_t v[4];
        for (int j = 0; j < 4; j++)
            s[j] = mix(i + j * stride + seed) >> (64 - bits);
        for (int j = 0; j < 4; j++)
            v[j] = t[s[j]];
        for (int j = 0; j < 4; j++) {
            acc += v[j];
            if (MODE == 1)
                t[s[j]] = (uint8_t)(v[j] + 1);
        }
    }
    if (acc == 0x12345678) t[0] = (uint8_t)acc;
}

template <int MODE>
void run(const char* name, uint8_t* buf, uint32_t bits, uint64_t n)
{
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9;
    for (int rep = 0; rep < 3; rep++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(probe<MODE>, dim3(4096), dim3(256), 0, 0, buf, bits, n, (uint64_t)rep * 7919 + 1);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    printf("%-22s table 2^%-2u B  %8.3f ms  %7.2f G accesses/s\n", name, bits, best, n / best / 1e6);
}

int main(int argc, char** argv)
{
    const uint32_t only = argc > 1 ? (uint32_t)atoi(argv[1]) : 0; // one table size only (the PMC calibration pass uses 34)
    uint8_t* buf; CK(hipMalloc(&buf, (size_t)1 << 34));
    CK(hipMemset(buf, 0, (size_t)1 << 34));
    const uint64_t n = 256ull << 20;
    for (uint32_t bits : {26u, 28u, 30u, 32u, 34u}) {
        if (only && bits != only)
            continue;
        run<0>("byte load", buf, bits, n);
        run<1>("byte load + store", buf, bits, n);
    }
    return 0;
}
