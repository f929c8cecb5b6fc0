// ubench.hip -- instruction / gather micro-benchmarks on gfx950 that the kernel design decisions
// in DESIGN.md cite.   hipcc --offload-arch=gfx950 -O3 -o build/ubench tools/ubench.hip && build/ubench
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e = (x);                                                    \
        if (e != hipSuccess) {                                                 \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

enum Op { MAD24, MULLO, DOT4, PERM, PKMAD16, LSHLOR, MINMAX, CVTRND, ALIGNBIT, ADD3, BFE, NOPS };
static const char *opname[] = {"v_mad_u32_u24", "v_mul_lo_u32", "v_dot4_u32_u8", "v_perm_b32", "v_pk_mad_u16",
                               "v_lshl_or_b32", "v_min+v_max_u32", "v_cvt_f32_u32+v_rndne", "v_alignbit_b32",
                               "v_add3_u32", "v_bfe_u32"};

template <int OP>
__global__ void __launch_bounds__(256) valu_kernel(uint32_t *out, int iters, uint32_t seed) {
    uint32_t a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i * 40503u + seed;
    uint32_t b = seed | 3u, c = seed ^ 0x5bd1e995u;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if constexpr (OP == MAD24) a[i] = __umul24(a[i], b) + c;
                if constexpr (OP == MULLO) a[i] = a[i] * b + 1;
                if constexpr (OP == DOT4) a[i] = __builtin_amdgcn_udot4(a[i], b, c, false);
                if constexpr (OP == PERM) a[i] = __builtin_amdgcn_perm(a[i], b, c);
                if constexpr (OP == PKMAD16) {
                    us2 x = __builtin_bit_cast(us2, a[i]), y = __builtin_bit_cast(us2, b), z = __builtin_bit_cast(us2, c);
                    x = x * y + z;
                    a[i] = __builtin_bit_cast(uint32_t, x);
                }
                if constexpr (OP == LSHLOR) a[i] = (a[i] << 3) | b;
                if constexpr (OP == MINMAX) { uint32_t t = min(a[i], b); a[i] = max(t, c) + 0; }
                if constexpr (OP == CVTRND) { float f = (float)a[i]; f = __builtin_rintf(f * 0.021f); a[i] = __float_as_uint(f); }
                if constexpr (OP == ALIGNBIT) a[i] = __builtin_amdgcn_alignbit(a[i], b, 16);
                if constexpr (OP == ADD3) a[i] = a[i] + b + c;
                if constexpr (OP == BFE) a[i] = __builtin_amdgcn_ubfe(a[i] ^ b, 4, 20);
            }
    }
    uint32_t r = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) r ^= a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

// LDS gather: every lane reads WIDTH bytes at a pseudo-random (LCG) offset inside `span` bytes
template <int WIDTH>
__global__ void __launch_bounds__(1024) lds_gather(uint32_t *out, int iters, uint32_t span_mask, uint32_t seed) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    for (int i = threadIdx.x; i < 80 * 1024 / 4; i += blockDim.x) ((uint32_t *)smem)[i] = i * 2654435761u;
    __syncthreads();
    uint32_t s = threadIdx.x * 747796405u + seed, acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            s = s * 1664525u + 1013904223u;
            uint32_t off = (s >> 8) & span_mask;
            if constexpr (WIDTH == 1) acc += smem[off];
            if constexpr (WIDTH == 4) acc += *(const uint32_t *)(smem + (off & ~3u));
            if constexpr (WIDTH == 16) {
                uint4 v = *(const uint4 *)(smem + (off & ~15u));
                acc += v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

// global gather of 16-B rows from a table of `rows` rows; locality: lanes of a wave draw their row
// from a window of `window` rows around a per-wave random centre
// same gather with only `active_of_8` of every 8 lanes issuing the load (exec-masked): does TA cost scale with lanes?
__global__ void __launch_bounds__(256) glb_gather_masked(const uint4 *tab, uint32_t *out, int iters, uint32_t rows,
                                                         uint32_t window, uint32_t active_of_8) {
    uint32_t lane_s = (blockIdx.x * blockDim.x + threadIdx.x) * 747796405u + 99u;
    uint32_t wave_s = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 2891336453u + 99u;
    uint32_t acc = 0;
    const bool on = (threadIdx.x & 7) < active_of_8;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            lane_s = lane_s * 1664525u + 1013904223u;
            wave_s = wave_s * 22695477u + 1u;
            uint32_t centre = (wave_s >> 4) & (rows - 1);               // rows and window are powers of two:
            uint32_t idx = (centre + ((lane_s >> 10) & (window - 1))) & (rows - 1);   // no integer division in the loop
            if (on) {
                uint4 v = tab[idx];
                acc += v.x ^ v.y ^ v.z ^ v.w;
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ void __launch_bounds__(256) glb_gather(const uint4 *tab, uint32_t *out, int iters, uint32_t rows,
                                                  uint32_t window, uint32_t seed) {
    uint32_t lane_s = (blockIdx.x * blockDim.x + threadIdx.x) * 747796405u + seed;
    uint32_t wave_s = ((blockIdx.x * blockDim.x + threadIdx.x) >> 6) * 2891336453u + seed;
    uint32_t acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            lane_s = lane_s * 1664525u + 1013904223u;
            wave_s = wave_s * 22695477u + 1u;
            uint32_t centre = (wave_s >> 4) & (rows - 1);
            uint32_t idx = (centre + ((lane_s >> 10) & (window - 1))) & (rows - 1);
            uint4 v = tab[idx];
            acc += v.x ^ v.y ^ v.z ^ v.w;
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

static float time_ms(void (*launch)(void *), void *ctx, int reps) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    launch(ctx);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int i = 0; i < reps; ++i) launch(ctx);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

struct Ctx {
    uint32_t *out;
    const uint4 *tab;
    int op, iters;
    uint32_t p0, p1;
};

template <int OP>
static void launch_valu(void *c) {
    Ctx *x = (Ctx *)c;
    hipLaunchKernelGGL(valu_kernel<OP>, dim3(256 * 8), dim3(256), 0, 0, x->out, x->iters, 12345u);
}
template <int W>
static void launch_lds(void *c) {
    Ctx *x = (Ctx *)c;
    hipLaunchKernelGGL(lds_gather<W>, dim3(256), dim3(1024), 80 * 1024, 0, x->out, x->iters, x->p0, 777u);
}
static void launch_glb_masked(void *c) {
    Ctx *x = (Ctx *)c;
    hipLaunchKernelGGL(glb_gather_masked, dim3(256 * 8), dim3(256), 0, 0, x->tab, x->out, x->iters, x->p0, x->p1, (uint32_t)x->op);
}
static void launch_glb(void *c) {
    Ctx *x = (Ctx *)c;
    hipLaunchKernelGGL(glb_gather, dim3(256 * 8), dim3(256), 0, 0, x->tab, x->out, x->iters, x->p0, x->p1, 99u);
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const double ghz = prop.clockRate / 1e6;
    printf("device %s, %d CUs, clockRate %.2f GHz (cycle figures assume this clock)\n", prop.gcnArchName,
           prop.multiProcessorCount, ghz);
    Ctx c;
    CK(hipMalloc(&c.out, 256 * 8 * 1024 * 4));
    const uint32_t rows = 65536;   // power of two (1 MiB of 16-byte rows), close to the 83521-row tables
    std::vector<uint32_t> h(rows * 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (uint32_t)i * 2654435761u;
    uint4 *tab;
    CK(hipMalloc(&tab, rows * 16));
    CK(hipMemcpy(tab, h.data(), rows * 16, hipMemcpyHostToDevice));
    c.tab = tab;
    CK(hipFuncSetAttribute((const void *)lds_gather<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CK(hipFuncSetAttribute((const void *)lds_gather<4>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));
    CK(hipFuncSetAttribute((const void *)lds_gather<16>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024));

    // ---- VALU throughput: 2048 WGs x 4 waves, 8 independent chains per lane ----
    c.iters = 400;
    void (*vl[NOPS])(void *) = {launch_valu<MAD24>, launch_valu<MULLO>, launch_valu<DOT4>, launch_valu<PERM>,
                                launch_valu<PKMAD16>, launch_valu<LSHLOR>, launch_valu<MINMAX>, launch_valu<CVTRND>,
                                launch_valu<ALIGNBIT>, launch_valu<ADD3>, launch_valu<BFE>};
    const int per_iter[NOPS] = {32, 32, 32, 32, 32, 32, 64, 96, 32, 32, 64};   // VALU instrs per lane per iteration (approx)
    for (int op = 0; op < NOPS; ++op) {
        float ms = time_ms(vl[op], &c, 5);
        const double waves = 256.0 * 8 * 4;
        const double winst = waves * c.iters * per_iter[op];
        const double cyc_per_inst_per_simd = (ms * 1e-3 * ghz * 1e9) / (winst / (prop.multiProcessorCount * 4.0));
        printf("VALU %-24s %8.3f ms  -> %.2f cycles per wave-instruction per SIMD\n", opname[op], ms, cyc_per_inst_per_simd);
    }
    // ---- LDS gathers: 256 WGs x 1024 threads (16 waves per CU) ----
    c.iters = 200;
    struct { int w; uint32_t mask; const char *what; } lt[] = {
        {1, 0xFFFF, "ds_read_u8  random in 64 KB"}, {1, 0x3FF, "ds_read_u8  random in 1 KB"},
        {4, 0xFFFF, "ds_read_b32 random in 64 KB"}, {16, 0xFFFF, "ds_read_b128 random in 64 KB"},
        {16, 0x7FFF, "ds_read_b128 random in 32 KB"}, {16, 0x3FF, "ds_read_b128 random in 1 KB"},
        {16, 0xFF, "ds_read_b128 random in 256 B"}};
    for (auto &t : lt) {
        c.p0 = t.mask;
        float ms = t.w == 1 ? time_ms(launch_lds<1>, &c, 5) : t.w == 4 ? time_ms(launch_lds<4>, &c, 5) : time_ms(launch_lds<16>, &c, 5);
        const double winst_per_cu = 16.0 * c.iters * 8;   // wave-instructions per CU (one WG per CU)
        printf("LDS  %-32s %8.3f ms  -> %.1f cycles per wave-instruction per CU (%.1f B/clk/CU)\n", t.what, ms,
               ms * 1e-3 * ghz * 1e9 / winst_per_cu, 64.0 * t.w / (ms * 1e-3 * ghz * 1e9 / winst_per_cu));
    }
    // ---- global 16-B row gathers from a 1.3 MB table ----
    c.iters = 100;
    uint32_t wins[] = {1, 4, 16, 64, 256, 1024, 4096, 65536};
    for (uint32_t w : wins) {
        c.p0 = rows;
        c.p1 = w;
        float ms = time_ms(launch_glb, &c, 5);
        const double winst_per_cu = (256.0 * 8 * 4 / prop.multiProcessorCount) * c.iters * 8;
        printf("GLB  dwordx4 gather, lanes within %5u rows of a per-wave centre: %8.3f ms -> %.1f cycles per wave-instruction per CU\n",
               w, ms, ms * 1e-3 * ghz * 1e9 / winst_per_cu);
    }
    for (uint32_t w : {64u, 4096u})
        for (int act : {8, 4, 2, 1}) {
            c.p0 = rows; c.p1 = w; c.op = act;
            float ms = time_ms(launch_glb_masked, &c, 5);
            const double winst_per_cu = (256.0 * 8 * 4 / prop.multiProcessorCount) * c.iters * 8;
            printf("GLB  dwordx4 gather, %d of 8 lanes active, window %5u rows: %8.3f ms -> %.1f cycles per wave-instruction per CU\n",
                   act, w, ms, ms * 1e-3 * ghz * 1e9 / winst_per_cu);
        }
    return 0;
}
