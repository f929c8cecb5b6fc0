// ubench_mix.hip -- how well do v_pk_mad_u16 and ds_read_b128 overlap on one gfx950 CU?
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_mix tools/ubench_mix.hip && build/ubench_mix
// The loop body is the expanded-band kernel's row loop (two passes of a pair: 5 row steps, each
// 4 ds_read_b128 + 16 v_pk_mad_u16) with synthetic row addresses.  One 1024-thread workgroup per CU.
//   MODE 0: loads + MACs   1: MACs only (rows come from registers)   2: loads only (rows are not used)
//   PAT  0: all lanes one row (broadcast)   1: lane-linear rows (conflict free)   2: random rows of the band
//        3: rows spread over a 64-row window (natural-image like)
// Output: cycles per pair (10 rows = 20 ds_read_b128 + 80 v_pk_mad_u16) per wave, and the same per CU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
static __device__ __forceinline__ uint32_t pkmad(uint32_t x, uint32_t w, uint32_t acc) {
    return __builtin_bit_cast(uint32_t, (u16x2)(__builtin_bit_cast(u16x2, x) * __builtin_bit_cast(u16x2, w) + __builtin_bit_cast(u16x2, acc)));
}

constexpr int kPlane = 34816;

template <int MODE, int PAT, int SYNC = 0, int IDX = 0>
__global__ void __launch_bounds__(1024) k_mix(unsigned long long *cyc, uint32_t *sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    for (int i = threadIdx.x; i < 2 * kPlane / 4; i += 1024) ((uint32_t *)smem)[i] = (uint32_t)i * 2654435761u & 0x00FF00FFu;
    __syncthreads();
    uint32_t lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
    uint32_t st = threadIdx.x * 747796405u + 2891336453u + blockIdx.x;
    const uint32_t lane = threadIdx.x & 63;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        uint32_t addr[5], w[5];
        st = st * 1664525u + 1013904223u;
        if (SYNC && it % SYNC == 0) __syncthreads();
        if (IDX) {   // stand-in for the index math: IDX dependent-ish slow-class VALU ops
            uint32_t q0 = st, q1 = st ^ 0x5555u, q2 = st + 77u, q3 = st * 3u;
#pragma unroll
            for (int k = 0; k < IDX / 4; ++k) {
                asm volatile("v_pk_max_u16 %0, %0, %1\n\tv_pk_min_u16 %1, %1, %2\n\tv_pk_add_u16 %2, %2, %3\n\tv_and_or_b32 %3, %3, %0, %1"
                             : "+v"(q0), "+v"(q1), "+v"(q2), "+v"(q3));
            }
            st ^= (q0 ^ q1 ^ q2 ^ q3) & 1u;
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            uint32_t ra, rb;
            if (PAT == 0) { ra = (uint32_t)(it * 5 + j) & 1023u; rb = ra + 7u; }
            else if (PAT == 1) { ra = lane + 64u * (uint32_t)j; rb = ra + 320u; }
            else if (PAT == 2) { ra = __umulhi(st * (2u * j + 3u), 2125u); rb = __umulhi(st * (2u * j + 5u) + 99u, 2125u); }
            else { ra = 1000u + ((st >> (3 * j)) & 63u); rb = 1000u + ((st >> (3 * j + 2)) & 63u); }
            addr[j] = (ra * 16u) | ((rb * 16u) << 16);
            w[j] = ((st >> j) & 15u) * 0x10001u;
        }
        uint4 a0, a1, b0, b1, na0, na1, nb0, nb1;
        auto ld = [&](int j, uint4 &x0, uint4 &x1, uint4 &y0, uint4 &y1) {
            if (MODE == 1) {
                x0 = make_uint4(addr[j], addr[j] + 1, addr[j] + 2, addr[j] + 3); x1 = x0; y0 = x0; y1 = x0;
                asm volatile("" : "+v"(x0.x), "+v"(x1.y), "+v"(y0.z), "+v"(y1.w));
            } else {
                const uint32_t oa = addr[j] & 0xFFFFu, ob = addr[j] >> 16;
                x0 = *(const uint4 *)(smem + oa);
                x1 = *(const uint4 *)(smem + oa + kPlane);
                y0 = *(const uint4 *)(smem + ob);
                y1 = *(const uint4 *)(smem + ob + kPlane);
            }
        };
        ld(0, a0, a1, b0, b1);
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            if (j < 4) ld(j + 1, na0, na1, nb0, nb1);
            if (MODE == 2) {
                asm volatile("" ::"v"(a0.x), "v"(a0.y), "v"(a0.z), "v"(a0.w), "v"(a1.x), "v"(a1.y), "v"(a1.z), "v"(a1.w));
                asm volatile("" ::"v"(b0.x), "v"(b0.y), "v"(b0.z), "v"(b0.w), "v"(b1.x), "v"(b1.y), "v"(b1.z), "v"(b1.w));
            } else {
                lo[0] = pkmad(a0.x, w[j], lo[0]); hi[0] = pkmad(a1.x, w[j], hi[0]);
                lo[1] = pkmad(a0.y, w[j], lo[1]); hi[1] = pkmad(a1.y, w[j], hi[1]);
                lo[2] = pkmad(a0.z, w[j], lo[2]); hi[2] = pkmad(a1.z, w[j], hi[2]);
                lo[3] = pkmad(a0.w, w[j], lo[3]); hi[3] = pkmad(a1.w, w[j], hi[3]);
                lo[3] = pkmad(b1.x, w[j], lo[3]); hi[3] = pkmad(b0.x, w[j], hi[3]);
                lo[2] = pkmad(b1.y, w[j], lo[2]); hi[2] = pkmad(b0.y, w[j], hi[2]);
                lo[1] = pkmad(b1.z, w[j], lo[1]); hi[1] = pkmad(b0.z, w[j], hi[1]);
                lo[0] = pkmad(b1.w, w[j], lo[0]); hi[0] = pkmad(b0.w, w[j], hi[0]);
            }
            a0 = na0; a1 = na1; b0 = nb0; b1 = nb1;
        }
    }
    asm volatile("s_nop 0" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    sink[blockIdx.x * blockDim.x + threadIdx.x] = lo[0] ^ lo[1] ^ lo[2] ^ lo[3] ^ hi[0] ^ hi[1] ^ hi[2] ^ hi[3] ^ st;
    if (lane == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE, int PAT, int SYNC = 0, int IDX = 0>
static void run(const char *name, int threads, unsigned long long *d_cyc, uint32_t *d_sink) {
    auto kern = k_mix<MODE, PAT, SYNC, IDX>;
    CK(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    const int iters = 8000, blocks = 256;
    const size_t lds = 150 * 1024;   // one workgroup per CU
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d_cyc, d_sink, 200);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, d_cyc, d_sink, iters);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    // wall time of the whole launch (all CUs run the same thing): ns per pair per CU, and that in 2.4 GHz cycles
    const double ns = (double)best * 1e6 / iters / (threads / 64);
    printf("%-28s waves/CU %2d   %7.3f ms   per pair per CU: %6.1f ns = %6.1f cycles @2.4GHz\n", name, threads / 64, best, ns, ns * 2.4);
}

int main() {
    unsigned long long *d_cyc;
    uint32_t *d_sink;
    CK(hipMalloc(&d_cyc, 256 * 16 * 8));
    CK(hipMalloc(&d_sink, 256 * 1024 * 4));
    printf("a pair = 20 ds_read_b128 + 80 v_pk_mad_u16 (+ ~25 VALU of address generation); event-timed launches\n");
    for (int threads : {1024, 512, 256}) {
        run<1, 1>("MAC only", threads, d_cyc, d_sink);
        run<2, 0>("loads only, broadcast", threads, d_cyc, d_sink);
        run<2, 1>("loads only, conflict-free", threads, d_cyc, d_sink);
        run<2, 3>("loads only, 64-row window", threads, d_cyc, d_sink);
        run<2, 2>("loads only, random", threads, d_cyc, d_sink);
        run<0, 0>("both, broadcast", threads, d_cyc, d_sink);
        run<0, 1>("both, conflict-free", threads, d_cyc, d_sink);
        run<0, 3>("both, 64-row window", threads, d_cyc, d_sink);
        run<0, 2>("both, random", threads, d_cyc, d_sink);
    }
    printf("-- with a 64-op VALU-only section per pair (index-math stand-in) and a workgroup barrier every S pairs, 16 waves/CU\n");
    run<1, 3, 0, 64>("MAC+idx only", 1024, d_cyc, d_sink);
    run<0, 3, 0, 64>("both+idx 64-row, no barrier", 1024, d_cyc, d_sink);
    run<0, 3, 6, 64>("both+idx 64-row, S=6", 1024, d_cyc, d_sink);
    run<0, 3, 3, 64>("both+idx 64-row, S=3", 1024, d_cyc, d_sink);
    run<0, 3, 1, 64>("both+idx 64-row, S=1", 1024, d_cyc, d_sink);
    run<0, 1, 0, 64>("both+idx conflict-free, none", 1024, d_cyc, d_sink);
    run<0, 1, 3, 64>("both+idx conflict-free, S=3", 1024, d_cyc, d_sink);
    run<0, 2, 0, 64>("both+idx random, none", 1024, d_cyc, d_sink);
    run<0, 2, 3, 64>("both+idx random, S=3", 1024, d_cyc, d_sink);
    return 0;
}
