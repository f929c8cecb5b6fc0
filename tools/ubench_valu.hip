// ubench_valu.hip -- issue cost of individual gfx950 VALU instructions, measured in shader cycles
// (s_memtime) with inline asm so the compiler can neither fold nor re-select them.
//   hipcc --offload-arch=gfx950 -O3 -o build/ubench_valu tools/ubench_valu.hip && build/ubench_valu
// One workgroup per CU; W waves per SIMD; each wave runs ITER x 32 independent instructions
// (8 destination registers, round robin).  Reported: cycles per wave-instruction per SIMD.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

#define DEFKERNEL(NAME, ASM3)                                                                         \
    __global__ void __launch_bounds__(1024) k_##NAME(unsigned long long *cyc, uint32_t *sink, int iters) { \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint32_t b = threadIdx.x * 3u + 7u, c = threadIdx.x ^ 0x55u;                                  \
        __syncthreads();                                                                              \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                         \
        for (int i = 0; i < iters; ++i) {                                                             \
            asm volatile(ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") ASM3("%4") ASM3("%5") ASM3("%6") ASM3("%7") \
                         ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") ASM3("%4") ASM3("%5") ASM3("%6") ASM3("%7") \
                         ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") ASM3("%4") ASM3("%5") ASM3("%6") ASM3("%7") \
                         ASM3("%0") ASM3("%1") ASM3("%2") ASM3("%3") ASM3("%4") ASM3("%5") ASM3("%6") ASM3("%7") \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
                         : "v"(b), "v"(c));                                                           \
        }                                                                                             \
        asm volatile("s_nop 0" ::: "memory");                                                         \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                         \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;          \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;             \
    }

// each ASM macro: D = op(D, b, c) on register D
#define A_ADD(D) "v_add_u32 " D ", " D ", %8\n"
#define A_AND(D) "v_and_b32 " D ", " D ", %8\n"
#define A_LSHR(D) "v_lshrrev_b32 " D ", 4, " D "\n"
#define A_MIN(D) "v_min_u32 " D ", " D ", %8\n"
#define A_MAX(D) "v_max_u32 " D ", " D ", %8\n"
#define A_MIN3(D) "v_min3_u32 " D ", " D ", %8, %9\n"
#define A_MED3(D) "v_med3_i32 " D ", " D ", %8, %9\n"
#define A_MAD24(D) "v_mad_u32_u24 " D ", " D ", %8, %9\n"
#define A_MUL24(D) "v_mul_u32_u24 " D ", " D ", %8\n"
#define A_MULLO(D) "v_mul_lo_u32 " D ", " D ", %8\n"
#define A_MULHI(D) "v_mul_hi_u32 " D ", " D ", %8\n"
#define A_PERM(D) "v_perm_b32 " D ", " D ", %8, %9\n"
#define A_DOT4(D) "v_dot4_u32_u8 " D ", " D ", %8, %9\n"
#define A_PKMAD(D) "v_pk_mad_u16 " D ", " D ", %8, %9\n"
#define A_PKMIN(D) "v_pk_min_u16 " D ", " D ", %8\n"
#define A_PKADD(D) "v_pk_add_u16 " D ", " D ", %8\n"
#define A_BFE(D) "v_bfe_u32 " D ", " D ", 4, 8\n"
#define A_LSHLOR(D) "v_lshl_or_b32 " D ", " D ", 3, %8\n"
#define A_ANDOR(D) "v_and_or_b32 " D ", " D ", %8, %9\n"
#define A_ADD3(D) "v_add3_u32 " D ", " D ", %8, %9\n"
#define A_LSHLADD(D) "v_lshl_add_u32 " D ", " D ", 2, %8\n"
#define A_SUBSDWA(D) "v_sub_u32_sdwa " D ", " D ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:WORD_1\n"
#define A_CVTF(D) "v_cvt_f32_u32 " D ", " D "\n"
#define A_CVTFSDWA(D) "v_cvt_f32_u32_sdwa " D ", " D " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n"
#define A_RNDNE(D) "v_rndne_f32 " D ", " D "\n"
#define A_MULF(D) "v_mul_f32 " D ", " D ", %8\n"
#define A_FMA(D) "v_fma_f32 " D ", " D ", %8, %9\n"
#define A_CVTPK(D) "v_cvt_pk_u8_f32 " D ", " D ", 1, %8\n"
#define A_ALIGN(D) "v_alignbit_b32 " D ", " D ", %8, 16\n"
#define A_CNDMASK(D) "v_cndmask_b32 " D ", " D ", %8, vcc\n"
#define A_MOV(D) "v_mov_b32 " D ", %8\n"
#define A_XAD(D) "v_xad_u32 " D ", " D ", %8, %9\n"
#define A_MADI24(D) "v_mad_i32_i24 " D ", " D ", %8, %9\n"
#define A_BFI(D) "v_bfi_b32 " D ", " D ", %8, %9\n"
#define A_SAD(D) "v_sad_u8 " D ", " D ", %8, %9\n"
#define A_MQSAD(D) "v_msad_u8 " D ", " D ", %8, %9\n"
#define A_MINF(D) "v_min_f32 " D ", " D ", %8\n"
#define A_MAXF(D) "v_max_f32 " D ", " D ", %8\n"
#define A_ADDF(D) "v_add_f32 " D ", " D ", %8\n"
#define A_SUBF(D) "v_sub_f32 " D ", " D ", %8\n"
#define A_MED3F(D) "v_med3_f32 " D ", " D ", %8, %9\n"
#define A_MIN3F(D) "v_min3_f32 " D ", " D ", %8, %9\n"
#define A_CVTU(D) "v_cvt_u32_f32 " D ", " D "\n"
#define A_CVTUB0(D) "v_cvt_f32_ubyte0 " D ", " D "\n"
#define A_FLOORF(D) "v_floor_f32 " D ", " D "\n"
#define A_FRACTF(D) "v_fract_f32 " D ", " D "\n"
#define A_OR(D) "v_or_b32 " D ", " D ", %8\n"
#define A_XOR(D) "v_xor_b32 " D ", " D ", %8\n"
#define A_LSHL(D) "v_lshlrev_b32 " D ", 4, " D "\n"
#define A_SUB(D) "v_sub_u32 " D ", " D ", %8\n"
#define A_ASHR(D) "v_ashrrev_i32 " D ", 4, " D "\n"
#define A_MINI(D) "v_min_i32 " D ", " D ", %8\n"
#define A_MINU16(D) "v_min_u16 " D ", " D ", %8\n"
#define A_ADDU16(D) "v_add_u16 " D ", " D ", %8\n"
#define A_MULF16(D) "v_mul_f16 " D ", " D ", %8\n"
#define A_MINF16(D) "v_min_f16 " D ", " D ", %8\n"
#define A_PKMINF16(D) "v_pk_min_f16 " D ", " D ", %8\n"
#define A_PKADDF16(D) "v_pk_add_f16 " D ", " D ", %8\n"
#define A_PKFMAF16(D) "v_pk_fma_f16 " D ", " D ", %8, %9\n"
#define A_BFIOR(D) "v_bitop3_b32 " D ", " D ", %8, %9 bitop3:0xc8\n"
#define A_DOT2C(D) "v_dot2c_i32_i16 " D ", %8, %9\n"
#define A_DOT2(D) "v_dot2_i32_i16 " D ", %8, %9, " D "\n"
#define A_MADI16(D) "v_mad_i32_i16 " D ", %8, %9, " D " op_sel:[1,1,0,0]\n"
#define A_MADU16(D) "v_mad_u32_u16 " D ", %8, 16, " D " op_sel:[1,0,0,0]\n"
#define A_SDWAB(D) "v_add_u32_sdwa " D ", " D ", %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2\n"
#define A_PKSHR(D) "v_pk_lshrrev_b16 " D ", 12, " D " op_sel_hi:[0,1]\n"
#define A_PKSUB(D) "v_pk_sub_u16 " D ", " D ", %8\n"
#define A_CVTI16(D) "v_cvt_f32_i32_sdwa " D ", sext(" D ") dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0\n"
#define A_PKMADSEL(D) "v_pk_mad_u16 " D ", %8, %9, " D " op_sel:[1,1,0] op_sel_hi:[0,1,1]\n"

#define OPS(X)                                                                                                       \
    X(ADD) X(AND) X(LSHR) X(MIN) X(MAX) X(MIN3) X(MED3) X(MAD24) X(MUL24) X(MULLO) X(MULHI) X(PERM) X(DOT4) X(PKMAD)   \
    X(PKMIN) X(PKADD) X(BFE) X(LSHLOR) X(ANDOR) X(ADD3) X(LSHLADD) X(SUBSDWA) X(CVTF) X(CVTFSDWA) X(RNDNE) X(MULF)    \
    X(FMA) X(CVTPK) X(ALIGN) X(CNDMASK) X(MOV) X(XAD) X(MADI24) X(BFI) X(SAD) X(MQSAD)              \
    X(MINF) X(MAXF) X(ADDF) X(SUBF) X(MED3F) X(MIN3F) X(CVTU) X(CVTUB0) X(FLOORF) X(FRACTF) X(OR) X(XOR) X(LSHL) X(SUB) X(ASHR)   \
    X(MINI) X(MINU16) X(ADDU16) X(MULF16) X(MINF16) X(PKMINF16) X(PKADDF16) X(PKFMAF16) X(BFIOR) \
    X(DOT2C) X(DOT2) X(MADI16) X(MADU16) X(SDWAB) X(PKSHR) X(PKSUB) X(CVTI16) X(PKMADSEL)

#define MK(N) DEFKERNEL(N, A_##N)
OPS(MK)

// dependent chains: every instruction consumes the previous result (latency, not throughput)
#define DEFDEP(NAME, ASM3)                                                                            \
    __global__ void __launch_bounds__(1024) d_##NAME(unsigned long long *cyc, uint32_t *sink, int iters) { \
        uint32_t a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;            \
        uint32_t b = threadIdx.x * 3u + 7u, c = threadIdx.x ^ 0x55u;                                  \
        __syncthreads();                                                                              \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                         \
        for (int i = 0; i < iters; ++i) {                                                             \
            asm volatile(ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") \
                         ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") \
                         ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") \
                         ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") ASM3("%0") \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
                         : "v"(b), "v"(c));                                                           \
        }                                                                                             \
        asm volatile("s_nop 0" ::: "memory");                                                         \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                         \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;          \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;             \
    }
// realistic mixes (one of each per 8 instructions, independent registers)
#define A_MIXA(D) "v_and_b32 " D ", " D ", %8\n v_lshrrev_b32 %1, 4, %1\n v_mad_u32_u24 %2, %2, %8, %9\n v_perm_b32 %3, %3, %8, %9\n v_pk_mad_u16 %4, %4, %8, %9\n v_min_u32 %5, %5, %8\n v_max_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
#define A_MIXB(D) "v_and_b32 " D ", " D ", %8\n v_lshrrev_b32 %1, 8, %1\n v_and_b32 %2, %2, %9\n v_pk_mad_u16 %3, %3, %8, %9\n v_pk_mad_u16 %4, %4, %8, %9\n v_and_b32 %5, %5, %8\n v_lshrrev_b32 %6, 8, %6\n v_pk_mad_u16 %7, %7, %8, %9\n"
#define A_MIXC(D) "v_pk_max_u16 " D ", " D ", %8\n v_pk_min_u16 %1, %1, %8\n v_pk_add_u16 %2, %2, %8\n v_pk_sub_u16 %3, %3, %8\n v_pk_lshrrev_b16 %4, 12, %4\n v_and_b32 %5, %5, %8\n v_pk_mad_u16 %6, %6, %8, %9\n v_add_u32_sdwa %7, %7, %8 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0\n"
// the instructions of MIXB (five full-rate, three half-rate), the full-rate ones first: does a RUN of full-rate instructions issue at their rate?
#define A_MIXD(D) "v_and_b32 " D ", " D ", %8\n v_lshrrev_b32 %1, 8, %1\n v_and_b32 %2, %2, %9\n v_and_b32 %5, %5, %8\n v_lshrrev_b32 %6, 8, %6\n v_pk_mad_u16 %3, %3, %8, %9\n v_pk_mad_u16 %4, %4, %8, %9\n v_pk_mad_u16 %7, %7, %8, %9\n"
// runs of sixteen: eight full-rate instructions twice, then eight half-rate ones twice (DEFMIX2 below)
#define A_FAST8(D) "v_and_b32 " D ", " D ", %8\n v_lshrrev_b32 %1, 8, %1\n v_and_b32 %2, %2, %9\n v_and_b32 %5, %5, %8\n v_lshrrev_b32 %6, 8, %6\n v_add_u32 %3, %3, %8\n v_add_u32 %4, %4, %8\n v_add_u32 %7, %7, %8\n"
#define A_SLOW8(D) "v_pk_mad_u16 " D ", " D ", %8, %9\n v_pk_mad_u16 %1, %1, %8, %9\n v_pk_mad_u16 %2, %2, %8, %9\n v_pk_mad_u16 %3, %3, %8, %9\n v_pk_mad_u16 %4, %4, %8, %9\n v_pk_mad_u16 %5, %5, %8, %9\n v_pk_mad_u16 %6, %6, %8, %9\n v_pk_mad_u16 %7, %7, %8, %9\n"
#define DEPOPS(X) X(ADD) X(AND) X(MIN) X(MAD24) X(PKMAD) X(PKMIN) X(PERM) X(LSHLOR) X(FMA) X(CVTF)
#define MKD(N) DEFDEP(N, A_##N)
DEPOPS(MKD)

#define DEFMIX(NAME, ASMG)                                                                            \
    __global__ void __launch_bounds__(1024) k_##NAME(unsigned long long *cyc, uint32_t *sink, int iters) { \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint32_t b = threadIdx.x * 3u + 7u, c = threadIdx.x ^ 0x55u;                                  \
        __syncthreads();                                                                              \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                         \
        for (int i = 0; i < iters; ++i) {                                                             \
            asm volatile(ASMG("%0") ASMG("%0") ASMG("%0") ASMG("%0")                                   \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
                         : "v"(b), "v"(c));                                                           \
        }                                                                                             \
        asm volatile("s_nop 0" ::: "memory");                                                         \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                         \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;          \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;             \
    }
DEFMIX(MIXA, A_MIXA)
DEFMIX(MIXB, A_MIXB)
DEFMIX(MIXC, A_MIXC)
DEFMIX(MIXD, A_MIXD)
#define DEFMIX2(NAME, GA, GB)                                                                        \
    __global__ void __launch_bounds__(1024) k_##NAME(unsigned long long *cyc, uint32_t *sink, int iters) { \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint32_t b = threadIdx.x * 3u + 7u, c = threadIdx.x ^ 0x55u;                                  \
        __syncthreads();                                                                              \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                         \
        for (int i = 0; i < iters; ++i) {                                                             \
            asm volatile(GA("%0") GA("%0") GB("%0") GB("%0")                                           \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)     \
                         : "v"(b), "v"(c));                                                           \
        }                                                                                             \
        asm volatile("s_nop 0" ::: "memory");                                                         \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                         \
        sink[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;          \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;             \
    }
DEFMIX2(MIXE, A_FAST8, A_SLOW8)

typedef void (*kfn)(unsigned long long *, uint32_t *, int);
struct Ent { const char *name; kfn fn; };
#define ENT(N) {#N, k_##N},
#define ENTD(N) {"dep_" #N, d_##N},
static Ent ents[] = {OPS(ENT) DEPOPS(ENTD) {"MIXA", k_MIXA}, {"MIXB", k_MIXB}, {"MIXC", k_MIXC}, {"MIXD", k_MIXD}, {"MIXE", k_MIXE}};

int main(int argc, char **argv) {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    unsigned long long *cyc;
    uint32_t *sink;
    const int ncu = prop.multiProcessorCount;
    CK(hipMalloc(&cyc, ncu * 2 * 16 * 8));
    CK(hipMalloc(&sink, ncu * 2 * 1024 * 4));
    const int iters = 2000;
    printf("%-10s", "op");
    const int wps[] = {1, 2, 4, 6, 8};   // 6 and 8: two workgroups per CU
    for (int w : wps) printf("  %dw/SIMD", w);
    printf("   (cycles per wave-instruction per SIMD, median over waves)\n");
    for (auto &e : ents) {
        if (argc > 1) {          // only the ops named on the command line
            bool want = false;
            for (int k = 1; k < argc; ++k) want = want || !strcmp(argv[k], e.name);
            if (!want) continue;
        }
        printf("%-10s", e.name);
        for (int w : wps) {
            const int nwg = w > 4 ? 2 : 1;
            const int threads = (w / nwg) * 4 * 64;
            hipLaunchKernelGGL(e.fn, dim3(ncu * nwg), dim3(threads), 0, 0, cyc, sink, iters);
            CK(hipDeviceSynchronize());
            hipLaunchKernelGGL(e.fn, dim3(ncu * nwg), dim3(threads), 0, 0, cyc, sink, iters);
            CK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(ncu * nwg * 16);
            CK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
            std::vector<double> v;
            for (int b = 0; b < ncu * nwg; ++b)
                for (int k = 0; k < (w / nwg) * 4; ++k) v.push_back((double)h[b * 16 + k]);
            std::sort(v.begin(), v.end());
            const double med = v[v.size() / 2];
            // a wave issues iters*32 instructions in `med` cycles while sharing its SIMD with w-1 others
            printf("  %7.2f", med / (iters * 32.0) / w);
        }
        printf("\n");
    }
    return 0;
}
