// mulut_k1.hip -- stages with 1-byte rows (non-final stages, and a final stage with u == 2 on the same kernel family):
//   stage_u1t_kernel     tube band of every mode + the tile as pixel codes in LDS; smooth tiles (the default)
//   stage_u1w_kernel     the whole 83.5 KB table of one mode in LDS, swapped per mode: detailed tiles (list mode) and the fallback
//   stage_u1_fix_kernel  sites the tube kernel flagged, recomputed from the full tables
#include <hip/hip_runtime.h>

#include "mulut_dev.h"

namespace mulut {

// ------------------------------------------------------------------------------------------
// K1-window: the same stage with the PIXEL reads taken out of the LDS instruction stream.  The LDS unit
// retires about one sub-dword read per 6.5 cycles per CU whatever the bank spread, and stage_u1_kernel issues
// 8 of them per pass (3 neighbours + 5 table bytes): it is bound by their count (tools/experiments/README.md).
// Here a thread owns FOUR horizontally adjacent pixels of a row; the 5 x 8-byte window around them (all 24
// neighbours of all four pixels, every mode, every rotation) is fetched as ten aligned ds_read_b32 per channel
// and mode, and every key nibble is a v_bfe_u32 at a compile-time bit position -- the mode pattern is a
// template parameter of the per-mode body, chosen by a scalar switch.  444 neighbour-byte reads per thread
// and tile become 90 dword reads; the 5 table-byte gathers per pass stay.
// ------------------------------------------------------------------------------------------
template <int ROW, int COL>
__device__ __forceinline__ int win_byte(const uint32_t (&win)[5][2]) {
    static_assert(ROW >= 0 && ROW < 5 && COL >= 0 && COL < 8, "window is 5 rows x 8 bytes");
    return (int)((win[ROW][COL >> 2] >> (8 * (COL & 3))) & 0xFFu);
}

// One mode over the thread's 3 x 4 sites.  Both loops are real loops (one pixel body per pattern in the binary,
// and nothing of a later pixel can be scheduled into an earlier one): the pixel loop shifts the window left by one
// byte per step so that the current pixel always sits at window column 2, and the accumulators rotate through
// fixed registers -- four steps per channel, three channel groups -- instead of being indexed.

// The same mode with rotations r / r + 2 of a pixel in packed 16-bit halves (simplex4_full_pair1): 33 VALU instructions per
// pass instead of 49.  A neighbour pair is one v_perm_b32 of two window registers; the row offsets are rebuilt per pass from
// the packed running sums (one SDWA add per row; the anchor's 13-bit stride rides as a marker bit that the pair math turns
// into the stride for both halves at once); the two passes' values of a row are packed by a v_perm_b32 and accumulated by one v_dot2_i32_i16.
template <int Q1, int J1, int Q2, int J2>
__device__ __forceinline__ uint32_t win_byte_pair(const uint32_t (&win)[5][2]) {      // byte (Q1, J1) | byte (Q2, J2) << 16
    constexpr uint32_t sel = 0x0C000C00u | ((uint32_t)(4 + (J2 & 3)) << 16) | (uint32_t)(J1 & 3);
    return __builtin_amdgcn_perm(win[Q2][J2 >> 2], win[Q1][J1 >> 2], sel);
}
template <int PAT, int R, int I>
__device__ __forceinline__ int u1p_pair(const int8_t *s_lut, const uint32_t (&win)[5][2], uint32_t k0, uint32_t ta, int sum) {
    constexpr int yb = rot_dy(R, kPatDi[PAT][0], kPatDj[PAT][0]), xb = rot_dx(R, kPatDi[PAT][0], kPatDj[PAT][0]);
    constexpr int yc = rot_dy(R, kPatDi[PAT][1], kPatDj[PAT][1]), xc = rot_dx(R, kPatDi[PAT][1], kPatDj[PAT][1]);
    constexpr int yd = rot_dy(R, kPatDi[PAT][2], kPatDj[PAT][2]), xd = rot_dx(R, kPatDi[PAT][2], kPatDj[PAT][2]);
    FullPair1 fp;      // pixel I of the thread's four: its window is columns I .. I + 4 of the 8 the registers hold
    simplex4_full_pair1(k0, win_byte_pair<2 + yb, I + 2 + xb, 2 - yb, I + 2 - xb>(win), win_byte_pair<2 + yc, I + 2 + xc, 2 - yc, I + 2 - xc>(win),
                        win_byte_pair<2 + yd, I + 2 + xd, 2 - yd, I + 2 - xd>(win), fp);
    uint32_t ra[4], rb[4];
    ra[0] = add_word<0>(ta, fp.base);
    rb[0] = add_word<1>(ta, fp.base);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        ra[j + 1] = add_word<0>(ra[0], fp.cum[j]);
        rb[j + 1] = add_word<1>(rb[0], fp.cum[j]);
    }
    int va[5], vb[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        va[j] = (int)s_lut[ra[j < 4 ? j : 0] + (uint32_t)(j < 4 ? 0 : kAllStrides)];
        vb[j] = (int)s_lut[rb[j < 4 ? j : 0] + (uint32_t)(j < 4 ? 0 : kAllStrides)];
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        const uint32_t t = __builtin_amdgcn_perm((uint32_t)vb[j], (uint32_t)va[j], 0x05040100u);      // value of pass A | value of pass B
        sum = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, t), __builtin_bit_cast(s16x2, fp.w[j]), sum, false);
    }
    return sum;
}
// acc[4 c + i]: pixel i of channel c.  The four pixels are unrolled with immediate window columns: no window shifting, no
// accumulator rotation (the rolled pixel loop of u1w_mode spends 13 of its 49 instructions per pass on those).
template <int PAT, int PW, int PH>
__device__ __forceinline__ void u1p_mode(const int8_t *s_lut, const uint8_t *s_img, int ty, int x4, int C, int (&acc)[12]) {
    static_for<0, 3>([&](auto CC) {
        constexpr int c = CC;
        if (c < C) {          // workgroup-uniform
            const uint32_t *row = (const uint32_t *)(s_img + c * (PH * PW) + ty * PW + x4);
            uint32_t win[5][2];
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                win[q][0] = row[q * (PW / 4)];
                win[q][1] = row[q * (PW / 4) + 1];
            }
            static_for<0, 4>([&](auto II) {
                constexpr int i = II;
                const uint32_t va = (i + 2 < 4 ? (win[2][0] >> (8 * ((i + 2) & 3))) : (win[2][1] >> (8 * ((i + 2) & 3)))) & 0xFFu;
                uint32_t k0 = full1_anchor_key(va);
                const uint32_t ta = (va >> 4) * (uint32_t)kStrideA;
                int sum = u1p_pair<PAT, 0, i>(s_lut, win, k0, ta, acc[4 * c + i]);
                asm volatile("" : "+v"(sum), "+v"(k0));      // one pair at a time (register budget)
                acc[4 * c + i] = u1p_pair<PAT, 1, i>(s_lut, win, k0, ta, sum);
            });
        }
    });
}

template <int TW, int TH, int NT, bool LIST>
__global__ void __launch_bounds__(NT) stage_u1w_kernel(StageArgs a) {
    constexpr int PW = TW + 2 * kHalo, PH = TH + 2 * kHalo;
    static_assert(TW * TH == 4 * NT && PW % 4 == 0, "four adjacent pixels per thread, dword-aligned tile rows");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    __shared__ int s_found;      // list mode: the next marked unit
    const int8_t *s_lut = (const int8_t *)smem;
    uint8_t *s_img = smem + kU1TableBytes;

    // list mode: a fixed grid of persistent workgroups; each walks an XCD-contiguous range of tiles (neighbouring tiles
    // share halo lines in one L2) and takes those the tube kernel marked in a.tile_list[tile]
    constexpr bool listed = LIST;
    // nothing marked (smooth content): leave at once -- a workgroup would otherwise walk its share of the marks one dependent global
    // load at a time (80 us per 32-frame launch for marks that are all zero)
    if (listed && a.tile_count && __builtin_amdgcn_readfirstlane((int)*a.tile_count) == 0) return;
    const int nt_all = a.N * a.tiles_x * a.tiles_y;
    const int G = (int)gridDim.x;
    // few marked tiles (fewer than half the workgroups): the unit of work is one CHANNEL of a tile, so that the launch does
    // not last as long as one whole tile (100 us) while most CUs have nothing to do
    const int nsub = (listed && a.tile_count && *a.tile_count * 2u < (uint32_t)G) ? a.C : 1;       // workgroup-uniform
    const int nu_all = nt_all * nsub;
    const bool by_xcd = (G & 7) == 0;
    const int per = (nu_all + 7) >> 3;
    int t_cur = !listed ? 0 : by_xcd ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int t_last = !listed ? 1 : by_xcd ? imin(((int)(blockIdx.x & 7) + 1) * per, nu_all) : nu_all;
    const int t_step = !listed ? 1 : by_xcd ? (G >> 3) : G;
    for (;; t_cur += t_step) {
    int n, y0, x0, c_lo = 0, c_n = a.C;
    if (listed) {
        // the next marked unit of this workgroup's run: NT candidates are looked at per round, one per thread (a serial walk is one
        // dependent global load per unit: 80 us per 32-frame launch when only a few tiles are marked)
        while (t_cur < t_last) {       // workgroup-uniform
            __syncthreads();           // the previous round's result has been read
            if (threadIdx.x == 0) s_found = 0x7fffffff;
            __syncthreads();
            const long long cand = (long long)t_cur + (long long)threadIdx.x * t_step;
            if (cand < t_last && a.tile_list[cand / nsub] != 0u) atomicMin(&s_found, (int)cand);
            __syncthreads();
            const int f = s_found;
            if (f != 0x7fffffff) { t_cur = f; break; }
            t_cur = (long long)t_cur + (long long)NT * t_step > (long long)t_last ? t_last : t_cur + NT * t_step;
        }
        if (t_cur >= t_last) break;
        decode_tile(a, t_cur / nsub, n, y0, x0, TW, TH);
        if (nsub > 1) { c_lo = t_cur % nsub; c_n = 1; }
    } else {
        decode_tile(a, xcd_remap(blockIdx.x, gridDim.x), n, y0, x0, TW, TH);
    }
    // The table of the NEXT mode travels through registers: fetched (6 x 16 B per thread) while the current mode is
    // being computed, written to LDS between the two barriers of the swap -- the swap then costs LDS stores only.
    static_assert((kU1TableBytes / 16 + NT - 1) / NT == 6, "six 16-byte chunks of the table per thread");
    constexpr int kVecs = kU1TableBytes / 16;
    const int c0 = (int)threadIdx.x, c5 = c0 + 5 * NT < kVecs ? c0 + 5 * NT : 0;   // chunk 5 exists for the first threads only
    uint4 n0, n1, n2, n3, n4, n5;   // named, not an array: they must live in registers across the mode body
#define MULUT_U1_FETCH(LUT)                                                                                         \
    do {                                                                                                            \
        const uint4 *src_ = (const uint4 *)(LUT);                                                                   \
        n0 = src_[c0]; n1 = src_[c0 + NT]; n2 = src_[c0 + 2 * NT]; n3 = src_[c0 + 3 * NT]; n4 = src_[c0 + 4 * NT];  \
        n5 = src_[c5];                                                                                              \
    } while (0)
    MULUT_U1_FETCH(a.lut[0]);
    load_tile_batched<TW, TH, NT>(a, n, y0, x0, s_img);
    const int x4 = (int)(threadIdx.x % (TW / 4)) * 4, ty = (int)(threadIdx.x / (TW / 4));
    int acc[12];   // [channel][pixel]
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = 0;

    for (int mv = 0; mv < a.M; ++mv) {
        const int m = __builtin_amdgcn_readfirstlane(mv);
        __syncthreads();  // everyone done with the previous table
        {
            uint4 *dst = (uint4 *)smem;
            dst[c0] = n0; dst[c0 + NT] = n1; dst[c0 + 2 * NT] = n2; dst[c0 + 3 * NT] = n3; dst[c0 + 4 * NT] = n4;
            if (c0 + 5 * NT < kVecs) dst[c0 + 5 * NT] = n5;
        }
        if (mv + 1 < a.M) MULUT_U1_FETCH(a.lut[__builtin_amdgcn_readfirstlane(mv + 1)]);
        __syncthreads();  // table and (m == 0) tile in place
        // pattern of this mode from its first key offset: s (0,1), d (0,2), y (1,1) -- scalar
        const int pat = a.dj[m][0] == 2 ? 1 : a.di[m][0] == 1 ? 2 : 0;
        const uint8_t *img_c = s_img + c_lo * (PH * PW);
        if (pat == 0) u1p_mode<0, PW, PH>(s_lut, img_c, ty, x4, c_n, acc);
        else if (pat == 1) u1p_mode<1, PW, PH>(s_lut, img_c, ty, x4, c_n, acc);
        else u1p_mode<2, PW, PH>(s_lut, img_c, ty, x4, c_n, acc);
    }
    const int y = y0 + ty;
    if (y < a.oy1) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            if (c < c_n) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int x = x0 + x4 + i;
                    if (x < a.W)
                        *const_cast<uint8_t *>(view_addr(a.out, n, c_lo + c, y, x)) = (uint8_t)rhe_clip_u8(acc[c * 4 + i] + a.bias_num, a.div);
                }
            }
    }
    if (!listed) break;
    __syncthreads();      // the next tile's image must not land while a wave still reads this one
    }
#undef MULUT_U1_FETCH
}

constexpr int K1_TW = 64, K1_TH = 64, K1_NT = 1024, K1_SPT = 12;  // 3 ch * 64*64 / 1024 = 12
static_assert(K1_SPT * K1_NT >= 3 * K1_TW * K1_TH, "SPT too small for 3 channels");

void stage_u1_tile(int &tw, int &th) { tw = K1_TW; th = K1_TH; }
const char *stage_u1_name(int variant) {
    return variant == 1 ? "stage_u1_kernel" : variant == 2 ? "stage_u1w_kernel" : variant == 3 ? "stage_u1t_kernel + stage_u1_fix_kernel"
                        : "stage_u1t_kernel (smooth tiles) + stage_u1w_kernel (detailed tiles) + stage_u1_fix_kernel";
}

hipError_t launch_stage_u1(const StageArgs &a, hipStream_t st, int variant) {
    if (a.C > 3) return hipErrorInvalidValue;
    auto kern = stage_u1w_kernel<K1_TW, K1_TH, K1_NT, false>;      // (variant: kept in the signature for the tuning knob; one kernel is left)
    const size_t lds = (size_t)kU1TableBytes + (size_t)a.C * (K1_TH + 2 * kHalo) * (K1_TW + 2 * kHalo);
    {
        const hipError_t e = raise_lds_limit((const void *)kern, 120 * 1024);
        if (e != hipSuccess) return e;
    }
    const long long nb = (long long)a.N * a.tiles_x * a.tiles_y;
    if (nb <= 0 || nb > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(K1_NT), lds, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// K1-tube: stage with 1-byte rows on the tube band (mulut_core.h).  The band of a mode is one dword per slot
// (the int8 value as int16 in both halves): 4,176 B, so the bands of all modes stay resident next to the image
// tile and two 1024-thread workgroups share a CU (8 waves per SIMD, <= 64 VGPRs).  A thread owns four horizontally
// adjacent pixels; per channel it reads its 5 x 8 window of pixel codes once (ten ds_read_b64) and every
// neighbour pair of every mode, rotation and pixel is one v_perm_b32 of two window registers.  Rotations r and
// r + 2 run in packed 16-bit halves (the index math of the final-stage tube kernel); the five rows of both passes
// are dword reads, combined per row by one v_bfi and accumulated by one v_dot2_i32_i16.
// Sites whose 5 x 5 neighbourhood spans more than one MSB step (some pass may leave the tube) are computed
// anyway -- their reads stay inside the band -- and appended to a work list that stage_u1_fix_kernel recomputes
// from the full tables; tiles with many such sites are not computed at all but handed to the full-table kernel
// (stage_u1w_kernel, list mode) through a tile list.  Both lists live in device memory; nothing syncs with the host.
// LDS: [ band s | band d | band y : 4,176 B each ][ image tile: C x 68 x 68 pixel codes ][ counters ]
// ------------------------------------------------------------------------------------------
constexpr int K1T_TW = 64, K1T_TH = 64;
// Threads per workgroup (a thread owns four adjacent pixels of a row: 1024 take the 64 x 64 tile at once, 512 in two halves of 32 rows) and
// waves per SIMD.  The instance with the shipped mode list compiled in needs 59 VGPRs: two 1024-thread workgroups share a CU at 8 waves per
// SIMD (the half-rate instruction class issues at 2.62 cycles per instruction and SIMD there, 2.83 at 6: profiles/r01_ubench_valu_issue_cost.txt).
// The u == 2 instance with the shipped list fits 62 VGPRs and 56 KB of LDS: the same two 1024-thread workgroups per CU (round 4; 67 VGPRs at
// three 512-thread workgroups before).  The run-time-list instances need 80: three 512-thread workgroups, 6 waves per SIMD.
// u == 3 (24,992-byte bands: one workgroup per CU): 1024 threads, 4 waves per SIMD.
__host__ __device__ constexpr int u1t_threads(int U, int pats) { return ((U <= 2 && pats != 0) || U == 3) ? 1024 : 512; }
__host__ __device__ constexpr int u1t_waves(int U, int pats) { return U == 3 ? 4 : (U <= 2 && pats != 0) ? 8 : 6; }
constexpr int K1T_PW = K1T_TW + 2 * kHalo, K1T_PH = K1T_TH + 2 * kHalo;
constexpr int kU1tTileBytes = 3 * K1T_PH * K1T_PW * 2;
constexpr int kU1tDirtyBytes = 3 * K1T_TH * (K1T_TW / 4);          // one byte per four-pixel group of the tile
static_assert(kU1tDirtyBytes % 1024 == 0, "the list pass gives every thread the same number of group bytes");

// accumulators of one site: u == 1 one int32; u == 2 two rotation-pair sets of four 16-bit fields (value + 128 rows, as
// the u == 4 kernels: a02 holds rotations 0 and 2 -- the latter added in reversed element order -- a13 rotations 1 and 3)
template <int U> struct U1tAcc;
template <> struct U1tAcc<1> { int v; __device__ __forceinline__ void clear() { v = 0; } };
template <> struct U1tAcc<2> {
    uint32_t a02[2], a13[2];
    __device__ __forceinline__ void clear() { a02[0] = a02[1] = a13[0] = a13[1] = 0; }
};
template <> struct U1tAcc<3> {      // ten fields per set (mulut_core.h: kTube3BandBytes)
    uint32_t a02[5], a13[5];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < 5; ++k) a02[k] = a13[k] = 0;
    }
};
__device__ __forceinline__ uint2 lds_u64(uint32_t addr) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = *(const __attribute__((address_space(3))) u32x2 *)(uintptr_t)addr;
    return make_uint2(v.x, v.y);
}
// (a & 0x0000FFFF) | (b & 0xFFFF0000) as ONE full-rate instruction: v_bitop3_b32 with the bit-field-insert truth table (mask, a, b -> 0xCA).
// The compiler's own choice for this expression is v_bfi_b32, which issues at the half rate of the packed / three-operand class
// (profiles/r01_ubench_valu_issue_cost.txt: BFIOR 1.49 vs 2.83 cycles per instruction and SIMD at 6 waves).
__device__ __forceinline__ uint32_t halves_lo_hi(uint32_t a, uint32_t b) {
    uint32_t r;
    const uint32_t m = 0x0000FFFFu;
    asm("v_bitop3_b32 %0, %1, %2, %3 bitop3:0xca" : "=v"(r) : "s"(m), "v"(a), "v"(b));
    return r;
}
// mode list as a compile-time constant (the encoding of stage_tube2_kernel: M | p0 << 2 | p1 << 4 | p2 << 6, M <= 3), 0 = read at run time
__host__ __device__ constexpr int u1t_modes(int pats) { return pats & 3; }
__host__ __device__ constexpr int u1t_pat(int pats, int m) { return (pats >> (2 + 2 * m)) & 3; }
constexpr int kU1tPatsSDY = 3 | (0 << 2) | (1 << 4) | (2 << 6);
template <int U> __host__ __device__ constexpr int u1t_band_bytes() { return U == 1 ? kTube1BandBytes : U == 2 ? kTube2BandBytes : kTube3BandBytes; }
template <int U> __host__ __device__ constexpr int u1t_slot_bytes() { return U == 1 ? 4 : U == 2 ? 8 : kTube3SlotBytes; }

// rotations R and R + 2 of the pixel at window column I + 2, pattern PAT
template <int U, int PAT, int R, int I, int NW>
__device__ __forceinline__ void u1t_pair(const uint32_t (&win)[5][NW], uint32_t k0, uint32_t base_a, U1tAcc<U> &acc) {
    constexpr int SLOT = u1t_slot_bytes<U>();
    constexpr int BAND = PAT * u1t_band_bytes<U>();      // LDS byte address of this pattern's band
    constexpr int yb = rot_dy(R, kPatDi[PAT][0], kPatDj[PAT][0]), xb = rot_dx(R, kPatDi[PAT][0], kPatDj[PAT][0]);
    constexpr int yc = rot_dy(R, kPatDi[PAT][1], kPatDj[PAT][1]), xc = rot_dx(R, kPatDi[PAT][1], kPatDj[PAT][1]);
    constexpr int yd = rot_dy(R, kPatDi[PAT][2], kPatDj[PAT][2]), xd = rot_dx(R, kPatDi[PAT][2], kPatDj[PAT][2]);
    const uint32_t pb = win_pair<2 + yb, I + 2 + xb, 2 - yb, I + 2 - xb, NW>(win);
    const uint32_t pc = win_pair<2 + yc, I + 2 + xc, 2 - yc, I + 2 - xc, NW>(win);
    const uint32_t pd = win_pair<2 + yd, I + 2 + xd, 2 - yd, I + 2 - xd, NW>(win);
    TubePair1 bp;
    simplex4_tube_pair1_slot<SLOT>(k0, base_a, pb, pc, pd, bp);
    // byte offsets of rows 0..3 of both passes, unpacked: row j + 1 = row j + stride of sorted key j (its low byte; twelve bits for u == 3)
    uint32_t aa[4], ab[4];
    aa[0] = bp.base & 0xFFFFu;
    ab[0] = bp.base >> 16;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if constexpr (U == 3) {
            aa[j + 1] = aa[j] + (bp.ks[j] & 0xFFFu);
            ab[j + 1] = ab[j] + ((bp.ks[j] >> 16) & 0xFFFu);
        } else {
            aa[j + 1] = add_byte<0>(aa[j], bp.ks[j]);
            ab[j + 1] = add_byte<2>(ab[j], bp.ks[j]);
        }
    }
    constexpr int kRow4 = kTubeAll * SLOT;
    if constexpr (U == 1) {
        uint32_t xa[5], xb2[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            // LDS addresses as plain integers (dynamic LDS starts at 0 -- checked at kernel entry): going through the
            // `smem` symbol would cost one v_add of a link-time zero per read
            xa[j] = lds_u32(aa[j < 4 ? j : 0] + (uint32_t)(BAND + (j < 4 ? 0 : kRow4)));
            xb2[j] = lds_u32(ab[j < 4 ? j : 0] + (uint32_t)(BAND + (j < 4 ? 0 : kRow4)));
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            typedef short s16x2 __attribute__((ext_vector_type(2)));
            const uint32_t t = halves_lo_hi(xa[j], xb2[j]);     // value of pass A | value of pass B
            acc.v = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, t), __builtin_bit_cast(s16x2, bp.w[j]), acc.v, false);
        }
    } else if constexpr (U == 3) {
        // pass A = rotation R (fields in place, weight = low half), pass B = rotation R + 2 (field p lands on 9 - p: dwords reversed,
        // halves swapped, weight = high half)
        uint32_t (&ac)[5] = R == 0 ? acc.a02 : acc.a13;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const uint32_t pa = aa[j < 4 ? j : 0] + (uint32_t)(BAND + (j < 4 ? 0 : kRow4)), pb2 = ab[j < 4 ? j : 0] + (uint32_t)(BAND + (j < 4 ? 0 : kRow4));
            const uint2 a01 = lds_u64(pa), a23 = lds_u64(pa + 8u), b01 = lds_u64(pb2), b23 = lds_u64(pb2 + 8u);
            const uint32_t a4 = lds_u32(pa + 16u), b4 = lds_u32(pb2 + 16u);
            pk_mac<0, false>(ac[0], a01.x, bp.w[j]);
            pk_mac<0, false>(ac[1], a01.y, bp.w[j]);
            pk_mac<0, false>(ac[2], a23.x, bp.w[j]);
            pk_mac<0, false>(ac[3], a23.y, bp.w[j]);
            pk_mac<0, false>(ac[4], a4, bp.w[j]);
            pk_mac<1, true>(ac[4], b01.x, bp.w[j]);
            pk_mac<1, true>(ac[3], b01.y, bp.w[j]);
            pk_mac<1, true>(ac[2], b23.x, bp.w[j]);
            pk_mac<1, true>(ac[1], b23.y, bp.w[j]);
            pk_mac<1, true>(ac[0], b4, bp.w[j]);
        }
    } else {
        uint2 xa[5], xb2[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            xa[j] = lds_u64(aa[j < 4 ? j : 0] + (uint32_t)(BAND + (j < 4 ? 0 : kRow4)));
            xb2[j] = lds_u64(ab[j < 4 ? j : 0] + (uint32_t)(BAND + (j < 4 ? 0 : kRow4)));
        }
        // pass A = rotation R (fields in place, weight = low half), pass B = rotation R + 2 (element e lands on 3 - e:
        // dwords and halves swapped, weight = high half)
        uint32_t (&ac)[2] = R == 0 ? acc.a02 : acc.a13;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            pk_mac<0, false>(ac[0], xa[j].x, bp.w[j]);
            pk_mac<0, false>(ac[1], xa[j].y, bp.w[j]);
            pk_mac<1, true>(ac[0], xb2[j].y, bp.w[j]);
            pk_mac<1, true>(ac[1], xb2[j].x, bp.w[j]);
        }
    }
}

// all four passes of one mode for the pixel at window column I + 2 (I = 0, 1: the pixel loop takes two pixels per step)
template <int U, int PAT, int I>
__device__ __forceinline__ void u1t_mode(const uint32_t (&win)[5][3], uint32_t &k0, uint32_t base_a, U1tAcc<U> &acc) {
    u1t_pair<U, PAT, 0, I, 3>(win, k0, base_a, acc);
    // one pair at a time: the second pair's index math must not be scheduled into the first (the window registers
    // leave room for one pair's temporaries under the VGPR budget); the empty asm ties the second pair's anchor key to
    // the first pair's sum
    if constexpr (U == 1) asm volatile("" : "+v"(acc.v), "+v"(k0));
    else asm volatile("" : "+v"(acc.a02[0]), "+v"(k0));
    u1t_pair<U, PAT, 1, I, 3>(win, k0, base_a, acc);
}

// one pixel: all modes, then the byte (u == 1) or the 2 x 2 block as four bytes, row-major (u == 2)
// PATS != 0: the mode list is a compile-time constant -- the passes of a pixel are straight-line code, no scalar loads of the pattern
// offsets, no branches (the run-time form reads a.di / a.dj from the kernel arguments per mode and pixel: two s_load + s_waitcnt
// lgkmcnt(0) in the hot loop, and its three pattern bodies share tails through extra address adds).  PATS == 0: `pats_rt` holds
// the patterns of the list, two bits per mode.
template <int U, int I, int PATS>
__device__ __forceinline__ uint32_t u1t_pixel(const StageArgs &a, uint32_t pats_rt, const uint32_t (&win)[5][3], uint32_t (&rows3)[3]) {
    U1tAcc<U> acc;
    acc.clear();
    // anchor terms, the same for every mode and rotation of the pixel
    const uint32_t ca_pk = win_pair<2, I + 2, 2, I + 2, 3>(win);
    uint32_t k0 = tube1_key(ca_pk, kTubeSA * u1t_slot_bytes<U>());
    const uint32_t base_a = pk_mad(ca_pk, pk_dup(16 * kTubeSA), 0u);
    if constexpr (PATS != 0) {
        static_for<0, u1t_modes(PATS)>([&](auto MI) {
            constexpr int m = MI;
            if constexpr (m > 0) {      // one mode at a time (register budget)
                if constexpr (U == 1) asm volatile("" : "+v"(acc.v), "+v"(k0));
                else asm volatile("" : "+v"(acc.a02[0]), "+v"(k0));
            }
            u1t_mode<U, u1t_pat(PATS, m), I>(win, k0, base_a, acc);
        });
    } else {
        for (int mv = 0; mv < a.M; ++mv) {
            const int pat = (int)((pats_rt >> (2 * mv)) & 3u);      // scalar
            if (pat == 0) u1t_mode<U, 0, I>(win, k0, base_a, acc);
            else if (pat == 1) u1t_mode<U, 1, I>(win, k0, base_a, acc);
            else u1t_mode<U, 2, I>(win, k0, base_a, acc);
        }
    }
    if constexpr (U == 1) {
        if (a.use_fma)       // wave-uniform: fused float epilogue proven exact; v_cvt_pk_u8_f32 rounds to nearest even and saturates
            return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)acc.v, a.inv_d, a.epi_c), 0u, 0u);
        return rhe_clip_u8(acc.v + a.bias_num, a.div);
    } else if constexpr (U == 3) {
        // block value (sy, sx) = field of element 3 sy + sx of the (0,2) set + field of element (2 - sx) 3 + sy of the (1,3) set, minus the
        // +128 bias of the rows; the three bytes of block row sy in rows3[sy]
        const int unbias = 128 * kQ * 4 * a.M - a.bias_num;
        auto fld = [](const uint32_t (&v)[5], int q) { return (int)((v[tube3_field(q) >> 1] >> (16 * (tube3_field(q) & 1))) & 0xFFFFu); };
#pragma unroll
        for (int sy = 0; sy < 3; ++sy) {
            uint32_t r = 0;
#pragma unroll
            for (int sx = 0; sx < 3; ++sx) r |= (uint32_t)rhe_clip_u8(fld(acc.a02, 3 * sy + sx) + fld(acc.a13, (2 - sx) * 3 + sy) - unbias, a.div) << (8 * sx);
            rows3[sy] = r;
        }
        return 0u;
    } else {
        // block value (sy, sx) = field 2 sy + sx of the (0,2) set + field (1 - sx) 2 + sy of the (1,3) set, minus the +128 bias of the rows
        const int unbias = 128 * kQ * 4 * a.M - a.bias_num;
        auto fld = [](const uint32_t (&v)[2], int e) { return (int)((v[e >> 1] >> (16 * (e & 1))) & 0xFFFFu); };
        const uint32_t o00 = rhe_clip_u8(fld(acc.a02, 0) + fld(acc.a13, 2) - unbias, a.div), o01 = rhe_clip_u8(fld(acc.a02, 1) + fld(acc.a13, 0) - unbias, a.div);
        const uint32_t o10 = rhe_clip_u8(fld(acc.a02, 2) + fld(acc.a13, 3) - unbias, a.div), o11 = rhe_clip_u8(fld(acc.a02, 3) + fld(acc.a13, 1) - unbias, a.div);
        return o00 | (o01 << 8) | (o10 << 16) | (o11 << 24);
    }
}

// bit i set <=> the 5 x 5 neighbourhood of the thread's pixel i spans more than one MSB step (then some pass of the
// site may leave the tube).  Column maxima / minima over the five rows first, then five adjacent columns per pixel,
// two pixels at a time in packed halves.
__device__ __forceinline__ uint32_t u1t_dirty(const uint32_t (&win)[5][4]) {
    uint32_t cx[4], cn[4], mx[4], mn[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        uint32_t hi = win[0][d] & 0x000F000Fu, lo = hi;
#pragma unroll
        for (int q = 1; q < 5; ++q) {
            const uint32_t h = win[q][d] & 0x000F000Fu;
            hi = pk_max(hi, h);
            lo = pk_min(lo, h);
        }
        cx[d] = hi; cn[d] = lo;
        mx[d] = pk_max(hi, __builtin_amdgcn_alignbit(hi, hi, 16));    // both halves: max of the dword's two columns
        mn[d] = pk_min(lo, __builtin_amdgcn_alignbit(lo, lo, 16));
    }
    // pixel 0: columns 0-4, pixel 1: columns 1-5 (low / high half); pixels 2, 3: columns 2-6, 3-7
    const uint32_t x01 = pk_max(pk_max((mx[0] & 0xFFFFu) | (cx[0] & 0xFFFF0000u), mx[1]), (cx[2] & 0xFFFFu) | (mx[2] & 0xFFFF0000u));
    const uint32_t n01 = pk_min(pk_min((mn[0] & 0xFFFFu) | (cn[0] & 0xFFFF0000u), mn[1]), (cn[2] & 0xFFFFu) | (mn[2] & 0xFFFF0000u));
    const uint32_t x23 = pk_max(pk_max((mx[1] & 0xFFFFu) | (cx[1] & 0xFFFF0000u), mx[2]), (cx[3] & 0xFFFFu) | (mx[3] & 0xFFFF0000u));
    const uint32_t n23 = pk_min(pk_min((mn[1] & 0xFFFFu) | (cn[1] & 0xFFFF0000u), mn[2]), (cn[3] & 0xFFFFu) | (mn[3] & 0xFFFF0000u));
    const uint32_t d01 = (x01 - n01) & 0xFFFEFFFEu, d23 = (x23 - n23) & 0xFFFEFFFEu;     // spread of the MSBs > 1
    return ((d01 & 0xFFFFu) ? 1u : 0u) | ((d01 >> 16) ? 2u : 0u) | ((d23 & 0xFFFFu) ? 4u : 0u) | ((d23 >> 16) ? 8u : 0u);
}

// (b, 0) pairs of bytes -> code1 pairs: b * 0x1001 = f << 12 | b per 16-bit lane; >> 4 moves the MSB nibble to bits 0-3
// and the LSB nibble to bits 8-11, where the mask drops it:  f << 12 | h
__device__ __forceinline__ uint32_t codes_of(uint32_t byte_pair) {
    const uint32_t x = pk_mad(byte_pair, pk_dup(0x1001u), 0u);
    return (x & 0xF000F000u) | ((x >> 4) & 0x000F000Fu);
}
// number of halves of a packed MSB pair... 1 if the two pixels of the pair differ by more than one MSB step
__device__ __forceinline__ uint32_t far_apart(uint32_t a, uint32_t b) {
    const uint32_t ha = a & 0x000F000Fu, hb = b & 0x000F000Fu;
    uint32_t hi = pk_max(ha, hb), lo = pk_min(ha, hb);
    hi = pk_max(hi, __builtin_amdgcn_alignbit(hi, hi, 16));
    lo = pk_min(lo, __builtin_amdgcn_alignbit(lo, lo, 16));
    return ((hi - lo) & 0xFFFEu) ? 1u : 0u;
}

template <int U, int PATS>
__global__ void __launch_bounds__(u1t_threads(U, PATS), u1t_waves(U, PATS)) stage_u1t_kernel(StageArgs a, BandArgs b, uint32_t detail_per_1024) {
    constexpr int TW = K1T_TW, TH = K1T_TH, NT = u1t_threads(U, PATS), PW = K1T_PW, PH = K1T_PH;
    constexpr int BB = u1t_band_bytes<U>();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *s_tile = smem + 3 * BB;
    uint32_t *s_cnt = (uint32_t *)(smem + 3 * BB + kU1tTileBytes);     // [0] detailed groups, [1] groups looked at
    uint8_t *s_dirty = smem + 3 * BB + kU1tTileBytes + 16;              // [channel][tile row][16 four-pixel groups]: dirty bit per pixel of the group
    uint32_t *s_scan = (uint32_t *)(s_dirty + kU1tDirtyBytes);          // [0..NT/64) flagged sites per wave, [16] where the tile's entries start in the fix-up list
    if (lds_addr_of(smem) != 0u) __builtin_trap();      // the band reads assume the dynamic LDS block starts at address 0 (no static LDS here): fail loudly, never skip the work

    uint32_t pats_rt = 0u;       // pattern of mode m in bits 2m, 2m + 1 (scalar; kMaxModes = 8 modes fit)
    for (int m = 0; m < a.M; ++m) {
        const int pat = a.dj[m][0] == 2 ? 1 : a.di[m][0] == 1 ? 2 : 0;
        pats_rt |= (uint32_t)pat << (2 * m);
        const uint32_t *src = (const uint32_t *)b.band[m];
        uint32_t *dst = (uint32_t *)(smem + pat * BB);
        for (int i = (int)threadIdx.x; i < BB / 4; i += NT) dst[i] = src[i];
    }
    pats_rt = (uint32_t)__builtin_amdgcn_readfirstlane((int)pats_rt);
    const int ntiles = a.N * a.tiles_x * a.tiles_y;
    const int G = gridDim.x;
    const bool by_xcd = (G & 7) == 0;
    const int per = (ntiles + 7) >> 3;
    int first = by_xcd ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    int last = by_xcd ? imin(((int)(blockIdx.x & 7) + 1) * per, ntiles) : ntiles;
    const int step = by_xcd ? (G >> 3) : G;
    if (G == ntiles) { first = xcd_remap(blockIdx.x, G); last = first + 1; }     // one workgroup per tile
    const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
    const bool al4 = ((a.W | a.in.sY) & 3) == 0 && (a.in.sN & 3) == 0 && (((uintptr_t)a.in.p) & 3) == 0;
    const bool hwc3 = al4 && a.C == 3 && a.in.sC == 1 && a.in.sX == 3;     // packed RGB rows: 12-byte groups of four pixels
    const bool planar = al4 && a.in.sX == 1 && (a.in.sC & 3) == 0;          // planar rows: dwords of four pixels
    // per-thread index terms are re-derived from an opaque copy of the thread id wherever a loop needs them: hoisted out of the tile
    // loop they would live across the pixel loop (which needs every register) -- in scratch, i.e. as HBM traffic
    auto opaque_tid = [&]() {
        int t = (int)threadIdx.x;
        asm volatile("" : "+v"(t));
        return t;
    };

#if defined(MULUT_VARIANT_k1prof) || defined(MULUT_VARIANT_k1prof2)   /* probe build: shader-clock ticks per phase, summed over all waves into the context's probe buffer (words 16..19) */
    const unsigned long long t_first = __builtin_amdgcn_s_memtime(), r_first = __builtin_amdgcn_s_memrealtime();
    uint32_t t_prev = (uint32_t)t_first, t_ph0 = 0, t_ph1 = 0, t_ph2 = 0, t_ph3 = 0, t_ph4 = 0, t_ph5 = 0, t_ph6 = 0, t_ph7 = 0;
#if defined(MULUT_VARIANT_k1prof2)      /* the sites phase in four parts (words 24..27) instead of as one */
#define K1_STAMP2(PH) K1_STAMP(PH)
#else
#define K1_STAMP2(PH) do { } while (0)
#endif
#define K1_STAMP(PH) do { const uint32_t t_now = (uint32_t)__builtin_amdgcn_s_memtime(); t_ph##PH += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define K1_STAMP(PH) do { } while (0)
#define K1_STAMP2(PH) do { } while (0)
#endif
    for (int tile = first; tile < last; tile += step) {
        int n, y0, x0;
        decode_tile(a, tile, n, y0, x0, TW, TH);
        __syncthreads();      // everyone is done with the previous tile (and, first trip, the bands are staged)
        K1_STAMP(0);          // band staging (first trip), barrier
        if (threadIdx.x == 0) {
            uint32_t z = 0;
            asm volatile("" : "+v"(z));      // made here: the compiler otherwise keeps a zero pair live across the whole kernel -- in scratch
            s_cnt[0] = z; s_cnt[1] = z;
        }
        for (int i = opaque_tid(); i < kU1tDirtyBytes / 4; i += NT) ((uint32_t *)s_dirty)[i] = 0u;      // (read again only after the tile's barriers)
        constexpr int GR = (TW + 8) / 4;            // 18 four-pixel groups cover image columns x0-4 .. x0+67
        const bool edge_tile = __builtin_amdgcn_readfirstlane((int)(x0 - 4 < 0 || x0 - 4 + 4 * (GR - 1) > a.W - 4)) != 0;
        // a group of four pixels of one image row, as two packed byte pairs per channel (edge columns replicated)
        auto group_hwc = [&](int row, int g, uint32_t (&bp)[6]) {
            const int gy = imin(imax(y0 + row - kHalo, ylo), yhi);
            const int gx = x0 - 4 + 4 * g, cgx = imin(imax(gx, 0), a.W - 4);
            const uint32_t *src = (const uint32_t *)view_addr(a.in, n, 0, gy, cgx);
            const uint32_t d0 = src[0], d1 = src[1], d2 = src[2];      // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
            bp[0] = __builtin_amdgcn_perm(0u, d0, 0x0C030C00u); bp[1] = __builtin_amdgcn_perm(d2, d1, 0x0C050C02u);
            bp[2] = __builtin_amdgcn_perm(d1, d0, 0x0C040C01u); bp[3] = __builtin_amdgcn_perm(d2, d1, 0x0C060C03u);
            bp[4] = __builtin_amdgcn_perm(d1, d0, 0x0C050C02u); bp[5] = __builtin_amdgcn_perm(0u, d2, 0x0C030C00u);
            if (edge_tile) {              // workgroup-uniform: interior tiles skip the per-lane choice (v_cndmask_b32 issues at a quarter of the rate of the others)
                if (gx < 0) {                 // left of the image: every column replicates column 0
                    bp[0] = bp[1] = pk_dup(bp[0] & 0xFFFFu); bp[2] = bp[3] = pk_dup(bp[2] & 0xFFFFu); bp[4] = bp[5] = pk_dup(bp[4] & 0xFFFFu);
                } else if (gx > a.W - 4) {    // right of it: column W-1
                    bp[0] = bp[1] = pk_dup(bp[1] >> 16); bp[2] = bp[3] = pk_dup(bp[3] >> 16); bp[4] = bp[5] = pk_dup(bp[5] >> 16);
                }
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) bp[k] = codes_of(bp[k]);
        };
        auto group_planar = [&](int c, int row, int g, uint32_t &p01, uint32_t &p23) {
            const int gy = imin(imax(y0 + row - kHalo, ylo), yhi);
            const int gx = x0 - 4 + 4 * g, cgx = imin(imax(gx, 0), a.W - 4);
            const uint32_t d = *(const uint32_t *)view_addr(a.in, n, c, gy, cgx);
            p01 = __builtin_amdgcn_perm(0u, d, 0x0C010C00u); p23 = __builtin_amdgcn_perm(0u, d, 0x0C030C02u);
            if (edge_tile) {
                if (gx < 0) p01 = p23 = pk_dup(p01 & 0xFFFFu);
                else if (gx > a.W - 4) p01 = p23 = pk_dup(p23 >> 16);
            }
            p01 = codes_of(p01); p23 = codes_of(p23);
        };
        if (a.verdict_take >= 0 && (hwc3 || planar)) {
            // routing statistic on every fourth row, before the tile is loaded: the share of four-pixel groups that span
            // more than one MSB step.  A detailed tile is handed to the full-table kernel without being staged here.
            if (tile == 0 && threadIdx.x == 0 && a.tile_count) a.tile_count[2] = 1u;      // "the marks mean something" (for the final stage's statistic)
            uint32_t far = 0, seen = 0;
            if (hwc3) {
                for (int i = opaque_tid(); i < (PH / 4) * GR; i += NT) {
                    uint32_t bp[6];
                    group_hwc(4 * (i / GR) + 1, i % GR, bp);
                    far += far_apart(bp[0], bp[1]) + far_apart(bp[2], bp[3]) + far_apart(bp[4], bp[5]);
                    seen += 3;
                }
            } else {
                for (int i = opaque_tid(); i < a.C * (PH / 4) * GR; i += NT) {
                    uint32_t p01, p23;
                    group_planar(i / (GR * (PH / 4)), 4 * ((i / GR) % (PH / 4)) + 1, i % GR, p01, p23);
                    far += far_apart(p01, p23);
                    seen += 1;
                }
            }
            far = wave_scan_add(far); seen = wave_scan_add(seen);      // the wave's totals in its last lane (DPP; as shuffles: twelve LDS permutes)
            __syncthreads();      // counters zeroed before anyone adds
            if ((threadIdx.x & 63) == 63 && seen) { atomicAdd(&s_cnt[0], far); atomicAdd(&s_cnt[1], seen); }
            __syncthreads();
            if (s_cnt[0] * 1024u > detail_per_1024 * s_cnt[1]) {       // workgroup-uniform
                if (threadIdx.x == 0) {      // verdict: left to the full-table kernel
                    a.tile_list[tile] = 1u;
                    // counted only while few: the list kernel asks "fewer than half the workgroups?", and on detailed content tens of
                    // thousands of atomics on one address would be a cost of their own
                    if (a.tile_count && __hip_atomic_load(a.tile_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 160u) atomicAdd(a.tile_count, 1u);
                }
                continue;
            }
        }
        K1_STAMP(1);          // routing statistic (two barriers)
        // store one group of four pixel codes (two packed pairs) of channel c: image columns gx .. gx + 3 -> tile columns gx - x0 + 2 ...
        auto put4 = [&](int c, int row, int g, uint32_t c01, uint32_t c23) {
            uint32_t *dst = (uint32_t *)(s_tile + 2 * ((c * PH + row) * PW + 4 * g - 2));
            if (g > 0) dst[0] = c01;                   // tile columns 4g-2, 4g-1
            if (4 * g + 1 < PW) dst[1] = c23;          // tile columns 4g, 4g+1
        };
        if (hwc3) {
            for (int i = opaque_tid(); i < PH * GR; i += NT) {
                const int g = i % GR, row = i / GR;
                uint32_t bp[6];
                group_hwc(row, g, bp);
                put4(0, row, g, bp[0], bp[1]); put4(1, row, g, bp[2], bp[3]); put4(2, row, g, bp[4], bp[5]);
            }
        } else if (planar) {
            for (int i = opaque_tid(); i < a.C * PH * GR; i += NT) {
                const int g = i % GR, row = (i / GR) % PH, c = i / (GR * PH);
                uint32_t p01, p23;
                group_planar(c, row, g, p01, p23);
                put4(c, row, g, p01, p23);
            }
        } else {
            for (int i = opaque_tid(); i < a.C * PH * PW; i += NT) {
                const int px = i % PW, row = (i / PW) % PH, c = i / (PW * PH);
                const int gy = imin(imax(y0 + row - kHalo, ylo), yhi);
                const int gx = imin(imax(x0 + px - kHalo, 0), a.W - 1);
                ((uint16_t *)s_tile)[i] = (uint16_t)pixel_code1(*view_addr(a.in, n, c, gy, gx));
            }
        }
        __syncthreads();      // tile in place
        K1_STAMP(2);          // tile load + barrier
#pragma clang loop unroll(disable)
        for (int half = 0; half < TH * (TW / 4) / NT; ++half) {
        {
            const int t = opaque_tid();
            if (y0 + t / (TW / 4) + half * (NT / (TW / 4)) >= a.oy1 || x0 + (t % (TW / 4)) * 4 >= a.W) continue;          // (no barrier below this point inside the trip)
        }
        // Thread coordinates are re-derived from an opaque copy of the thread id wherever they are needed: whatever is
        // computed from them before the pixel loop and used after it would otherwise be parked in scratch around the loop
        // (it needs every register), and scratch of 400k threads does not stay in L2 -- it was 0.7 GB of HBM writes per launch.
        auto coords = [&](int &ty_, int &tx_) {
            int t = (int)threadIdx.x;
            asm volatile("" : "+v"(t));
            tx_ = (t % (TW / 4)) * 4;
            ty_ = t / (TW / 4) + half * (NT / (TW / 4));
        };
#pragma clang loop unroll(disable)
        for (int c = 0; c < a.C; ++c) {
            uint32_t dirty;
            {   // the 5 x 8 window of the thread's four pixels, only for the neighbourhood test
                uint32_t win8[5][4];
                int ty, tx4;
                coords(ty, tx4);
                const uint2 *row = (const uint2 *)(s_tile + 2 * ((c * PH + ty) * PW + tx4));
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const uint2 lo = row[q * (PW / 4)], hi = row[q * (PW / 4) + 1];
                    win8[q][0] = lo.x; win8[q][1] = lo.y; win8[q][2] = hi.x; win8[q][3] = hi.y;
                }
                dirty = u1t_dirty(win8);
                asm volatile("" : "+v"(dirty));     // computed HERE: sunk below the pixel loop, its 20 window registers would be parked in scratch
            }
            K1_STAMP2(4);
            uint32_t packed = 0;
            // two pixels per step of a real loop: their 5 x 6 window is re-read (dword-aligned), nothing of a later step
            // can be scheduled into an earlier one
#pragma clang loop unroll(disable)
            for (int it = 0; it < 2; ++it) {
                int ty, tx4;
                coords(ty, tx4);
                const int y = y0 + ty, x = x0 + tx4;
                if (U >= 2 && x + 2 * it >= a.W) break;
                uint32_t win[5][3];
                const uint32_t *row = (const uint32_t *)(s_tile + 2 * ((c * PH + ty) * PW + tx4 + 2 * it));
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    win[q][0] = row[q * (PW / 2)]; win[q][1] = row[q * (PW / 2) + 1]; win[q][2] = row[q * (PW / 2) + 2];
                }
                uint32_t r0[3] = {0u, 0u, 0u}, r1[3] = {0u, 0u, 0u};      // u == 3: the pixels' three block rows, three bytes each
                uint32_t b0 = u1t_pixel<U, 0, PATS>(a, pats_rt, win, r0);
                if constexpr (U == 3) asm volatile("" : "+v"(r0[0]), "+v"(r0[1]), "+v"(r0[2]), "+v"(win[2][1]));
                else asm volatile("" : "+v"(b0), "+v"(win[2][1]));       // the second pixel starts after the first is done
                const uint32_t b1 = u1t_pixel<U, 1, PATS>(a, pats_rt, win, r1);
                if constexpr (U == 1) {
                    packed |= (b0 | (b1 << 8)) << (16 * it);
                } else if constexpr (U == 3) {
                    // two 3 x 3 blocks side by side: HR rows 3y .. 3y + 2, columns 3 (x + 2 it) .. + 5
                    const int xo = 3 * (x + 2 * it);
                    const bool both = x + 2 * it + 1 < a.W;
#pragma unroll
                    for (int sy = 0; sy < 3; ++sy) {
                        uint8_t *d = const_cast<uint8_t *>(view_addr(a.out, n, c, 3 * y + sy, xo));
                        const uint32_t lo = r0[sy] | (r1[sy] << 24), hi = r1[sy] >> 8;      // bytes 0..3, 4..5 of the six
                        if (a.out.sX == 1 && both && (((uintptr_t)d) & 1) == 0) {
                            ((uint16_t *)d)[0] = (uint16_t)lo;
                            ((uint16_t *)d)[1] = (uint16_t)(lo >> 16);
                            ((uint16_t *)d)[2] = (uint16_t)hi;
                        } else {
#pragma unroll
                            for (int i = 0; i < 6; ++i)
                                if (i < 3 || both) d[i * a.out.sX] = (uint8_t)((i < 4 ? lo >> (8 * i) : hi >> (8 * (i - 4))));
                        }
                    }
                } else {
                    // two 2 x 2 blocks side by side: HR rows 2y and 2y + 1, columns 2 (x + 2 it) .. + 3
                    const int xo = 2 * (x + 2 * it);
                    const uint32_t top = (b0 & 0xFFFFu) | (b1 << 16), bot = (b0 >> 16) | (b1 & 0xFFFF0000u);
                    uint8_t *d0 = const_cast<uint8_t *>(view_addr(a.out, n, c, 2 * y, xo));
                    uint8_t *d1 = const_cast<uint8_t *>(view_addr(a.out, n, c, 2 * y + 1, xo));
                    if (a.out.sX == 1 && x + 2 * it + 1 < a.W && ((((uintptr_t)d0) | ((uintptr_t)d1)) & 3) == 0) {
                        *(uint32_t *)d0 = top;
                        *(uint32_t *)d1 = bot;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (x + 2 * it + i / 2 < a.W) {
                                d0[i * a.out.sX] = (uint8_t)(top >> (8 * i));
                                d1[i * a.out.sX] = (uint8_t)(bot >> (8 * i));
                            }
                    }
                }
            }
            K1_STAMP2(5);
            int ty, tx4;
            coords(ty, tx4);
            const int y = y0 + ty, x = x0 + tx4;
            if constexpr (U == 1) {
                uint8_t *dst = const_cast<uint8_t *>(view_addr(a.out, n, c, y, x));
                if (a.out.sX == 1 && x + 3 < a.W && (((uintptr_t)dst) & 3) == 0) {
                    *(uint32_t *)dst = packed;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (x + i < a.W) dst[i * a.out.sX] = (uint8_t)(packed >> (8 * i));
                }
            }
            // sites that may have left the tube: noted per four-pixel group in LDS; the workgroup appends them to the fix-up list after the
            // tile with ONE memory-side atomic.  (One atomic per wave and pixel slot, as rounds 2-3 had it, is 60 k atomics on one
            // address for 8 x 270x480 frames of a steep field: they took 0.4 of the kernel's 0.48 ms, profiles/r04b_ab_fixlist_atomics_small.jsonl.)
            {
                const int rem = a.W - x;            // pixels of the group inside the image
                const uint32_t inside = rem >= 4 ? 0xFu : rem > 0 ? (1u << rem) - 1u : 0u;
                s_dirty[(c * TH + ty) * (TW / 4) + tx4 / 4] = (uint8_t)(dirty & inside);
            }
            K1_STAMP2(6);
        }
        }
#if !defined(MULUT_VARIANT_nofixlist)    /* (timing-only variant: nothing is listed, flagged sites stay wrong) */
        __syncthreads();      // every group byte of the tile is written
        {
            const int t = opaque_tid(), lane = t & 63, wave = t >> 6;
            constexpr int GB = kU1tDirtyBytes / NT;       // group bytes per thread: 6 (512 threads) or 3 (1024)
            uint32_t bytes = 0u, mine = 0u;               // this thread's group nibbles, four bits each
#pragma unroll
            for (int k = 0; k < GB; ++k) {
                const uint32_t v = (uint32_t)s_dirty[GB * t + k] & 0xFu;
                bytes |= v << (4 * k);
                mine += (uint32_t)__builtin_popcount(v);
            }
            // inclusive scan over the wave by DPP (four shifts inside the 16-lane rows, two row broadcasts): as __shfl_up steps it was six
            // LDS permutes and twelve v_cndmask_b32 (12.6 issue cycles each) per tile and thread
            const uint32_t inc = wave_scan_add(mine);
            if (lane == 63) s_scan[wave] = inc;
            __syncthreads();
            uint32_t wbase = 0, total = 0;
            const int wave_s = __builtin_amdgcn_readfirstlane(wave);      // scalar: the selects below are then scalar too
#pragma unroll
            for (int w = 0; w < NT / 64; ++w) {
                const uint32_t v = (uint32_t)__builtin_amdgcn_readfirstlane((int)s_scan[w]);
                wbase += w < wave_s ? v : 0u;
                total += v;
            }
            if (total != 0u) {        // workgroup-uniform
                if (t == 0) s_scan[16] = atomicAdd(a.fix_count, total);
                __syncthreads();
                uint32_t at = s_scan[16] + wbase + inc - mine;
#pragma unroll
                for (int k = 0; k < GB; ++k) {
                    const uint32_t bits = (bytes >> (4 * k)) & 0xFu;
                    if (bits == 0u) continue;
                    const int g = GB * t + k, cg = g % (TW / 4), row = (g / (TW / 4)) % TH, c = g / ((TW / 4) * TH);
                    const uint32_t id0 = (uint32_t)(((n * a.C + c) * a.H + y0 + row) * a.W + x0 + 4 * cg);
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if ((bits >> i) & 1u) a.fix_list[at++] = id0 + (uint32_t)i;
                }
            }
        }
#endif
        K1_STAMP2(7);
        K1_STAMP(3);          // the sites
    }
#if defined(MULUT_VARIANT_k1prof) || defined(MULUT_VARIANT_k1prof2)
    if (a.dbg && (threadIdx.x & 63) == 0) {
        atomicAdd(a.dbg + 16, (unsigned long long)t_ph0); atomicAdd(a.dbg + 17, (unsigned long long)t_ph1);
        atomicAdd(a.dbg + 18, (unsigned long long)t_ph2); atomicAdd(a.dbg + 19, (unsigned long long)t_ph3);
        atomicAdd(a.dbg + 20, __builtin_amdgcn_s_memtime() - t_first);          // wave lifetime in shader-clock ticks ...
        atomicAdd(a.dbg + 21, __builtin_amdgcn_s_memrealtime() - r_first);      // ... and in 100 MHz ticks: their ratio is the in-kernel clock
        atomicAdd(a.dbg + 22, 1ull);
        atomicAdd(a.dbg + 24, (unsigned long long)t_ph4); atomicAdd(a.dbg + 25, (unsigned long long)t_ph5);
        atomicAdd(a.dbg + 26, (unsigned long long)t_ph6); atomicAdd(a.dbg + 27, (unsigned long long)t_ph7);
    }
#endif
#undef K1_STAMP
}

// Fix-up of the 1-byte-row tube kernel: every listed site (id = ((n C + c) H + y) W + x) is recomputed from the full
// tables in global memory (the pass_kernel arithmetic over all modes and rotations) and its byte overwritten.
__global__ void __launch_bounds__(256) stage_u1_fix_kernel(StageArgs a) {
    const uint32_t count = *a.fix_count;
    const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
        uint32_t id = a.fix_list[i];
        const int x = (int)(id % (uint32_t)a.W); id /= (uint32_t)a.W;
        const int y = (int)(id % (uint32_t)a.H); id /= (uint32_t)a.H;
        const int c = (int)(id % (uint32_t)a.C), n = (int)(id / (uint32_t)a.C);
        if (n >= a.N || y < a.oy0 || y >= a.oy1) continue;      // never follow an entry outside the launch (a list bug must show as a wrong pixel, not as a memory fault)
        auto px = [&](int dy, int dx) {
            const int gy = imin(imax(y + dy, ylo), yhi), gx = imin(imax(x + dx, 0), a.W - 1);
            return (int)*view_addr(a.in, n, c, gy, gx);
        };
        const int va = px(0, 0);
        int acc = 0;
        for (int m = 0; m < a.M; ++m) {
            const int8_t *lut = (const int8_t *)a.lut[m];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int v[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    int dy, dx;
                    sample_offset(r, a.di[m][k], a.dj[m][k], dy, dx);
                    v[k] = px(dy, dx);
                }
                int idx[5], w[5];
                simplex4(va, v[0], v[1], v[2], idx, w);
#pragma unroll
                for (int j = 0; j < 5; ++j) acc += w[j] * (int)lut[idx[j]];
            }
        }
        *const_cast<uint8_t *>(view_addr(a.out, n, c, y, x)) = (uint8_t)rhe_clip_u8(acc + a.bias_num, a.div);
    }
}


void stage_u1t_tile(int &tw, int &th) { tw = K1T_TW; th = K1T_TH; }

template <int U>
static hipError_t launch_u1t_t(const StageArgs &a, const BandArgs &b, unsigned detail_per_1024, int num_cus, int persist_per_cu, hipStream_t st) {
    // the shipped mode list gets the instance with its passes in straight-line code
    const bool sdy = a.M == 3 && a.di[0][0] == 0 && a.dj[0][0] == 1 && a.dj[1][0] == 2 && a.di[2][0] == 1 && a.dj[2][0] == 1;
    auto kern = sdy ? stage_u1t_kernel<U, kU1tPatsSDY> : stage_u1t_kernel<U, 0>;
    {
        const hipError_t e = raise_lds_limit((const void *)kern, U == 3 ? 112 * 1024 : 80 * 1024);
        if (e != hipSuccess) return e;
    }
    const long long ntiles = (long long)a.N * a.tiles_x * a.tiles_y;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL) return hipErrorInvalidValue;
    // persist_per_cu > 0: that many persistent workgroups per CU walk XCD-contiguous tile ranges; 0: one workgroup per tile
    const int threads = sdy ? u1t_threads(U, kU1tPatsSDY) : u1t_threads(U, 0);
    const long long want = persist_per_cu > 0 ? (long long)persist_per_cu * num_cus : ntiles;
    const unsigned grid = (unsigned)(ntiles < want ? ntiles : want);
    const size_t lds = 3 * (size_t)u1t_band_bytes<U>() + kU1tTileBytes + 16 + kU1tDirtyBytes + 80;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, st, a, b, (uint32_t)detail_per_1024);
    return hipGetLastError();
}

hipError_t launch_stage_u1t(const StageArgs &a, const BandArgs &b, unsigned detail_per_1024, int num_cus, int persist_per_cu, hipStream_t st) {
    if (a.C > 3 || a.M > kMaxModes || !a.fix_list || !a.fix_count) return hipErrorInvalidValue;
    return launch_u1t_t<1>(a, b, detail_per_1024, num_cus, persist_per_cu, st);
}

hipError_t launch_stage_u1w_list(const StageArgs &a, int num_cus, hipStream_t st) {
    if (a.C > 3 || !a.tile_list || !a.tile_count) return hipErrorInvalidValue;
    auto kern = stage_u1w_kernel<K1_TW, K1_TH, K1_NT, true>;
    const size_t lds = (size_t)kU1TableBytes + (size_t)a.C * (K1_TH + 2 * kHalo) * (K1_TW + 2 * kHalo);
    {
        const hipError_t e = raise_lds_limit((const void *)kern, 120 * 1024);
        if (e != hipSuccess) return e;
    }
    const long long ntiles = (long long)a.N * a.tiles_x * a.tiles_y;
    const unsigned grid = (unsigned)(ntiles < num_cus ? ntiles : num_cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(K1_NT), lds, st, a);
    return hipGetLastError();
}

hipError_t launch_stage_u1_fix(const StageArgs &a, int num_cus, hipStream_t st) {
    if (!a.fix_list || !a.fix_count) return hipErrorInvalidValue;
    hipLaunchKernelGGL(stage_u1_fix_kernel, dim3((unsigned)(4 * num_cus)), dim3(256), 0, st, a);
    return hipGetLastError();
}

// Fix-up of the u == 2 tube kernel: every listed site (id = ((n C + c) H + y) W + x) recomputed from the full table
template <int U>
__global__ void __launch_bounds__(256) stage_up_fix_site_kernel(StageArgs a) {
    const uint32_t count = *a.fix_count;
    const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
        uint32_t id = a.fix_list[i];
        const int x = (int)(id % (uint32_t)a.W); id /= (uint32_t)a.W;
        const int y = (int)(id % (uint32_t)a.H); id /= (uint32_t)a.H;
        const int c = (int)(id % (uint32_t)a.C), n = (int)(id / (uint32_t)a.C);
        if (n >= a.N || y < a.oy0 || y >= a.oy1) continue;      // (as stage_u1_fix_kernel)
        auto px = [&](int dy, int dx) {
            const int gy = imin(imax(y + dy, ylo), yhi), gx = imin(imax(x + dx, 0), a.W - 1);
            return (int)*view_addr(a.in, n, c, gy, gx);
        };
        const int va = px(0, 0);
        RotAcc<U> acc;
        acc.clear();
        for (int mv = 0; mv < a.M; ++mv) {
            const int m = __builtin_amdgcn_readfirstlane(mv);
            const void *lut = a.lut[m];
            const int di0 = a.di[m][0], di1 = a.di[m][1], di2 = a.di[m][2];
            const int dj0 = a.dj[m][0], dj1 = a.dj[m][1], dj2 = a.dj[m][2];
            static_for<0, 4>([&](auto R) {
                constexpr int r = R;
                int dy, dx, v0, v1, v2;
                sample_offset(r, di0, dj0, dy, dx); v0 = px(dy, dx);
                sample_offset(r, di1, dj1, dy, dx); v1 = px(dy, dx);
                sample_offset(r, di2, dj2, dy, dx); v2 = px(dy, dx);
                pass_global<U, r>(lut, va, v0, v1, v2, a, acc);
            });
        }
        uint32_t o[U];
        finish_channel<U, kOutGeneric>(a, acc, n, c, y, x, o);
    }
}

// the same kernel family on a FINAL stage with u == 2 (4-value rows, 2 x 2 output blocks): b.band[m] = 8-byte-per-slot
// tube band; flagged sites go to stage_up_fix_site_kernel through a.fix_list; a.verdict_take >= 0: tiles whose local-detail statistic
// exceeds detail_per_1024 are marked in a.tile_list and left out (the caller runs the gather kernel on them)
hipError_t launch_stage_u2t(const StageArgs &a, const BandArgs &b, unsigned detail_per_1024, int num_cus, int persist_per_cu, hipStream_t st) {
    if (a.C > 3 || a.M > kMaxModes || !a.fix_list || !a.fix_count || (a.verdict_take >= 0 && !a.tile_list)) return hipErrorInvalidValue;      // (a merged pair of rotations sums to <= 8160 per mode in its unsigned 16-bit field: 8 modes fit)
    hipError_t e = launch_u1t_t<2>(a, b, detail_per_1024, num_cus, persist_per_cu, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(stage_up_fix_site_kernel<2>, dim3((unsigned)(4 * num_cus)), dim3(256), 0, st, a);
    return hipGetLastError();
}

// and with u == 3 (9-value rows as ten 16-bit fields, 24 bytes per slot; 3 x 3 output blocks)
hipError_t launch_stage_u3t(const StageArgs &a, const BandArgs &b, unsigned detail_per_1024, int num_cus, int persist_per_cu, hipStream_t st) {
    if (a.C > 3 || a.M > kMaxModes || !a.fix_list || !a.fix_count || (a.verdict_take >= 0 && !a.tile_list)) return hipErrorInvalidValue;      // (a merged pair of rotations sums to <= 8160 per mode in its unsigned 16-bit field: 8 modes fit)
    hipError_t e = launch_u1t_t<3>(a, b, detail_per_1024, num_cus, persist_per_cu, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(stage_up_fix_site_kernel<3>, dim3((unsigned)(4 * num_cus)), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace mulut
