// LDS probe: cycles per wave-instruction of random-address ds_add_rtn_u32 / ds_add_u32 / ds_read_b32 /
// ds_write_b32 on a 640-word table, at 4 / 8 / 12 / 16 waves per CU (all CUs busy).
// build: hipcc -O3 --offload-arch=gfx950 -o bin/lds_atomic_probe lds_atomic_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(256) void probe(unsigned long long* out, uint32_t iters, uint32_t words)
{
    extern __shared__ uint32_t t[];
    for (uint32_t i = threadIdx.x; i < words; i += 256)
        t[i] = 0;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u, acc = 0;
    const long long t0 = clock64();
    for (uint32_t i = 0; i < iters; i++) {
        x = x * 1664525u + 1013904223u;
        const uint32_t ix = (x >> 8) % words;
        if (MODE == 0)
            acc += atomicAdd(&t[ix], 1u); // returning: the next address does not depend on it, the sum does
        else if (MODE == 1)
            atomicAdd(&t[ix], 1u); // non-returning
        else if (MODE == 2)
            acc += t[ix];
        else
            t[ix] = x;
    }
    const long long t1 = clock64();
    __syncthreads();
    if (threadIdx.x == 0)
        out[blockIdx.x] = (unsigned long long)(t1 - t0);
    if (acc == 0x12345678u)
        out[blockIdx.x] = t[0];
}

template <int MODE>
static void run(const char* name, int wg_per_cu, uint32_t words)
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n = prop.multiProcessorCount * wg_per_cu;
    unsigned long long* d;
    hipMalloc(&d, n * sizeof(*d));
    const uint32_t iters = 4096;
    // pad dynamic LDS so that exactly wg_per_cu workgroups fit a CU
    const size_t lds = (size_t)(150 * 1024 / wg_per_cu);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(probe<MODE>, dim3(n), dim3(256), lds, 0, d, iters, words);
    hipLaunchKernelGGL(probe<MODE>, dim3(n), dim3(256), lds, 0, d, iters, words);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(n);
    hipMemcpy(h.data(), d, n * sizeof(*d), hipMemcpyDeviceToHost);
    double s = 0;
    for (auto v : h)
        s += (double)v;
    s /= n;
    // a CU ran wg_per_cu x 4 waves, each `iters` wave-instructions, in s cycles
    printf("%-22s %2d waves/CU  %7.1f cycles per wave-instruction per wave, %6.2f lanes per cycle per CU\n", name,
           wg_per_cu * 4, s / iters, (double)wg_per_cu * 4 * 64 * iters / s);
    hipFree(d);
}

int main()
{
    for (uint32_t words : { 640u, 16384u }) {
        printf("table of %u words\n", words);
        for (int w : { 1, 2, 3, 4 }) {
            run<0>("ds_add_rtn_u32", w, words);
            run<1>("ds_add_u32", w, words);
            run<2>("ds_read_b32", w, words);
            run<3>("ds_write_b32", w, words);
        }
    }
    return 0;
}
