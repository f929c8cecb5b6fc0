// ubench_lds_atomic.hip -- what does an LDS float add cost on gfx950?  (tools only; never linked into the product)
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_lds_atomic tools/ubench/ubench_lds_atomic.hip && build/ubench_lds_atomic
// One 1024-thread workgroup per CU (4 waves per SIMD), each wave issues N LDS operations; cycles by s_memtime.
//   pattern 0: every lane its own dword            (no conflict)
//   pattern 1: 16-lane groups, lane = element, the 4 groups of a wave on 4 different rows
//   pattern 2: the same, all 4 groups of a wave on the same row (4 lanes per address)
//   pattern 3: as 2, and all 16 waves on the same row
//   pattern 4: all 64 lanes one address
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(1024) k(unsigned long long *out, int pattern, int n) {
    __shared__ float s[16384];
    for (int i = threadIdx.x; i < 16384; i += 1024) s[i] = 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, e = lane & 15, g = lane >> 4;
    int idx;
    if (pattern == 0) idx = threadIdx.x;
    else if (pattern == 1) idx = (wave * 4 + g) * 16 + e;
    else if (pattern == 2) idx = wave * 16 + e;
    else if (pattern == 3) idx = e;
    else idx = 0;
    const uint32_t addr = (uint32_t)(uintptr_t)&s[idx];
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    float v = 1.0f;
    for (int i = 0; i < n; ++i) {
        if (OP == 0) asm volatile("ds_add_f32 %0, %1" : : "v"(addr), "v"(v) : "memory");
        else if (OP == 1) asm volatile("ds_add_u32 %0, %1" : : "v"(addr), "v"(1) : "memory");
        else if (OP == 2) { float r; asm volatile("ds_add_rtn_f32 %0, %1, %2\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr), "v"(v) : "memory"); v += r * 1e-30f; }
        else if (OP == 3) { float r; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)\n\tv_add_f32 %0, %0, %2\n\tds_write_b32 %1, %0" : "=&v"(r) : "v"(addr), "v"(v) : "memory"); }
        else if (OP == 4) asm volatile("ds_write_b32 %0, %1" : : "v"(addr), "v"(v) : "memory");
        else if (OP == 5) asm volatile("ds_pk_add_f16 %0, %1" : : "v"(addr), "v"(0x3c003c00) : "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (s[threadIdx.x] == 12345.678f) out[0] = 0;
}

int main() {
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount, n = 2000;
    unsigned long long *d;
    CHECK(hipMalloc(&d, cus * 8));
    std::vector<unsigned long long> h(cus);
    const char *ops[] = {"ds_add_f32", "ds_add_u32", "ds_add_rtn_f32 + wait", "ds_read + v_add + ds_write", "ds_write_b32", "ds_pk_add_f16"};
    const char *pats[] = {"64 distinct dwords", "4 rows x 16 elements", "1 row x 16 elements per wave (4 lanes per address)", "1 row for the whole workgroup", "one address"};
    for (int op = 0; op < 6; ++op)
        for (int pat = 0; pat < 5; ++pat) {
            for (int rep = 0; rep < 2; ++rep) {
                switch (op) {
                    case 0: hipLaunchKernelGGL(k<0>, dim3(cus), dim3(1024), 0, 0, d, pat, n); break;
                    case 1: hipLaunchKernelGGL(k<1>, dim3(cus), dim3(1024), 0, 0, d, pat, n); break;
                    case 2: hipLaunchKernelGGL(k<2>, dim3(cus), dim3(1024), 0, 0, d, pat, n); break;
                    case 3: hipLaunchKernelGGL(k<3>, dim3(cus), dim3(1024), 0, 0, d, pat, n); break;
                    case 4: hipLaunchKernelGGL(k<4>, dim3(cus), dim3(1024), 0, 0, d, pat, n); break;
                    default: hipLaunchKernelGGL(k<5>, dim3(cus), dim3(1024), 0, 0, d, pat, n); break;
                }
                CHECK(hipDeviceSynchronize());
            }
            CHECK(hipMemcpy(h.data(), d, cus * 8, hipMemcpyDeviceToHost));
            double sum = 0;
            for (auto v : h) sum += (double)v;
            const double cyc = sum / cus / n;      // shader cycles (s_memtime) per operation of ONE wave; 16 waves issue concurrently on the CU
            printf("%-28s %-52s %8.1f cycles per op seen by a wave, %7.2f cycles per wave-instruction on the CU (16 waves)\n", ops[op], pats[pat], cyc, cyc / 16.0);
        }
    return 0;
}
