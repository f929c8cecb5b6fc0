// ubench_ifetch.hip -- does the VALU issue rate of a straight-line instruction stream depend on its size?
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_ifetch tools/ubench/ubench_ifetch.hip && build/ubench_ifetch
// Every wave runs `iters` trips through a body of N v_pk_mad_u16 (8 independent accumulators, round robin); N sets the code size
// (8 bytes per instruction).  One workgroup per CU, all 256 CUs.  Cycles are s_memtime ticks (shader clock); the in-kernel
// clock is (s_memtime delta) / (s_memrealtime delta) x 100 MHz.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define M8 "v_pk_mad_u16 %0, %0, %8, %9\n v_pk_mad_u16 %1, %1, %8, %9\n v_pk_mad_u16 %2, %2, %8, %9\n v_pk_mad_u16 %3, %3, %8, %9\n" \
           "v_pk_mad_u16 %4, %4, %8, %9\n v_pk_mad_u16 %5, %5, %8, %9\n v_pk_mad_u16 %6, %6, %8, %9\n v_pk_mad_u16 %7, %7, %8, %9\n"
#define M32 M8 M8 M8 M8
#define M128 M32 M32 M32 M32
#define M512 M128 M128 M128 M128
#define M2048 M512 M512 M512 M512
#define M8192 M2048 M2048 M2048 M2048

#define DEFK(NAME, BODY)                                                                                                  \
    __global__ void __launch_bounds__(1024) NAME(unsigned long long *cyc, uint32_t *sink, int iters) {                    \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint32_t b = threadIdx.x * 3u + 7u, c = threadIdx.x ^ 0x55u;                                                      \
        __syncthreads();                                                                                                  \
        const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();                \
        for (int i = 0; i < iters; ++i)                                                                                   \
            asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c)); \
        asm volatile("s_nop 0" ::: "memory");                                                                             \
        const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();                \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;                              \
        if ((threadIdx.x & 63) == 0) { cyc[2 * (blockIdx.x * 16 + (threadIdx.x >> 6))] = t1 - t0; cyc[2 * (blockIdx.x * 16 + (threadIdx.x >> 6)) + 1] = r1 - r0; } \
    }
DEFK(k32, M32)
DEFK(k512, M512)
DEFK(k2048, M2048)
DEFK(k4096, M2048 M2048)
DEFK(k8192, M8192)
DEFK(k16384, M8192 M8192)

template <class K>
static void run(const char *name, K kern, int n, int threads, unsigned long long *d_cyc, uint32_t *d_sink) {
    const int blocks = 256, total = 1 << 22;          // ~4 M instructions per wave
    const int iters = total / n;
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_cyc, d_sink, iters / 8 + 1);
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, d_cyc, d_sink, iters);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(2 * blocks * 16);
    CK(hipMemcpy(h.data(), d_cyc, h.size() * 8, hipMemcpyDeviceToHost));
    const int waves = threads / 64;
    std::vector<double> cpi, clk;
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < waves; ++w) {
            const double t = (double)h[2 * (b * 16 + w)], r = (double)h[2 * (b * 16 + w) + 1];
            cpi.push_back(t / ((double)iters * n));      // ticks per instruction of this wave
            clk.push_back(t / r * 0.1);                  // GHz
        }
    std::sort(cpi.begin(), cpi.end());
    std::sort(clk.begin(), clk.end());
    // waves / 4 waves share a SIMD: cycles per wave-instruction per SIMD = ticks per instruction of one wave / (waves / 4)
    printf("%-8s code %7d B  waves/SIMD %d  cycles per instruction per SIMD: median %.2f (min %.2f)   in-kernel clock median %.3f GHz\n", name, n * 8,
           waves / 4, cpi[cpi.size() / 2] / (waves / 4.0), cpi[0] / (waves / 4.0), clk[clk.size() / 2]);
}

int main() {
    unsigned long long *d_cyc;
    uint32_t *d_sink;
    CK(hipMalloc(&d_cyc, 2 * 256 * 16 * 8));
    CK(hipMalloc(&d_sink, 256 * 1024 * 4));
    for (int threads : {1024, 512}) {
        run("k32", k32, 32, threads, d_cyc, d_sink);
        run("k512", k512, 512, threads, d_cyc, d_sink);
        run("k2048", k2048, 2048, threads, d_cyc, d_sink);
        run("k4096", k4096, 4096, threads, d_cyc, d_sink);
        run("k8192", k8192, 8192, threads, d_cyc, d_sink);
        run("k16384", k16384, 16384, threads, d_cyc, d_sink);
    }
    return 0;
}
