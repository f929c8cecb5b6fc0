// mulut_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels for MuLUT LUT inference.
//
//   pass_kernel        one (table, mode, rotation) pass, q*out as int32 -- the unit-parity twin of
//                      FourSimplexInterpFaster (reference sr/4_test_lut.py:14-237)
//   stage_u1_kernel    K1: a whole stage with 1-byte rows (non-final stages): all modes x 4
//                      rotations + average/bias/round/clip fused; the active 83.5 KB table lives
//                      in LDS, swapped per mode; image tile + 2-px halo in LDS
//   stage_up_kernel    K2: the final stage (u*u-byte rows): 12 passes x 5 row gathers per site,
//                      16-bit SWAR accumulation per rotation, rotate-back + divide/round/clip
//                      fused, 4x4 (u x u) output block written per site
//
// No MFMA: this is gather + lerp, bounded by LDS/L1 gather rate and VALU, not by a contraction.
#include <hip/hip_runtime.h>

#include "mulut_kernels.h"

// MULUT_ABLATE selects timing-only variants (wrong results!) built by tools/ab_bench.py into
// build/ablate/; the shipped library is always built with MULUT_ABLATE == 0.
//   K2:  1 every row gather reads row 0   2 no gathers (rows synthesised)   3 gathers, no SWAR fma
//   K2:  4 SWAR fma without the byte unpack    5 epilogue without divide/round/clip
//   band K2 fast path:  6 xor instead of MAC   7 no LDS row gathers   8 no pixel reads / hoistable index math
//   K1: 11 every LUT read hits byte 0     12 no LUT reads                   13 no table staging
#ifndef MULUT_ABLATE
#define MULUT_ABLATE 0
#endif

namespace mulut {

// ------------------------------------------------------------------------------------------
// pass kernel (parity unit; not performance critical)
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) pass_kernel(PassArgs a) {
    const long long nsite = (long long)a.C * a.H * a.W;
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsite) return;
    const int x = (int)(s % a.W);
    const int y = (int)((s / a.W) % a.H);
    const int c = (int)(s / ((long long)a.W * a.H));
    const uint8_t *pl = a.in + (long long)c * a.H * a.W;
    int v[4];
    v[0] = pl[(long long)y * a.W + x];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int dy, dx;
        sample_offset(a.r, a.di[k], a.dj[k], dy, dx);
        const int yy = imin(imax(y + dy, 0), a.H - 1);
        const int xx = imin(imax(x + dx, 0), a.W - 1);
        v[k + 1] = pl[(long long)yy * a.W + xx];
    }
    int idx[5], w[5];
    simplex4(v[0], v[1], v[2], v[3], idx, w);
    const int u = a.u;
    const int Wo = a.W * u;
    int32_t *po = a.out + (long long)c * a.H * u * Wo;
    if (u == 1) {
        const int8_t *lut = (const int8_t *)a.lut;
        int acc = 0;
#pragma unroll
        for (int j = 0; j < 5; ++j) acc += w[j] * (int)lut[idx[j]];
        po[(long long)y * Wo + x] = acc;
    } else {
        const uint8_t *lut = (const uint8_t *)a.lut;
        const int rb = row_dwords(u) * 4;
        for (int sy = 0; sy < u; ++sy)
            for (int sx = 0; sx < u; ++sx) {
                const int e = row_elem(a.r, sy, sx, u);
                int acc = 0;
#pragma unroll
                for (int j = 0; j < 5; ++j) acc += w[j] * ((int)lut[(long long)idx[j] * rb + e] - 128);
                po[(long long)(y * u + sy) * Wo + (x * u + sx)] = acc;
            }
    }
}

hipError_t launch_pass(const PassArgs &a, hipStream_t st) {
    const long long nsite = (long long)a.C * a.H * a.W;
    const int nb = (int)((nsite + 255) / 256);
    hipLaunchKernelGGL(pass_kernel, dim3(nb), dim3(256), 0, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// shared tile helpers
// ------------------------------------------------------------------------------------------
constexpr int kHalo = 2;  // receptive field of one stage: +-2 px (d / y patterns over 4 rotations)

__device__ __forceinline__ const uint8_t *view_addr(const View &v, int n, int c, int y, int x) {
    return v.p + (long long)n * v.sN + (long long)c * v.sC + (long long)(y - v.row0) * v.sY + (long long)x * v.sX;
}

// Fill the LDS image tile [C][PH][PW] (TH x TW pixels + halo) with edge replication at the TRUE
// image borders only (clamp to [0,H-1] x [0,W-1]); rows outside the band held by `in` are never
// touched because the host checks halo coverage.
template <int TW, int TH, int NT>
__device__ __forceinline__ void load_tile(const StageArgs &a, int n, int y0, int x0, uint8_t *s_img) {
    constexpr int PW = TW + 2 * kHalo, PH = TH + 2 * kHalo;
    const int total = a.C * PH * PW;
    for (int i = threadIdx.x; i < total; i += NT) {
        const int px = i % PW;
        const int py = (i / PW) % PH;
        const int c = i / (PW * PH);
        // clamping to [oy0-2, oy1+1] as well is an identity for every row a valid site reads, and
        // keeps tiles that overhang the band from touching rows the caller's buffer does not hold
        const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
        const int gy = imin(imax(y0 + py - kHalo, ylo), yhi);
        const int gx = imin(imax(x0 + px - kHalo, 0), a.W - 1);
        s_img[i] = *view_addr(a.in, n, c, gy, gx);
    }
}

// The same copy with every byte load of a thread in flight before the first LDS store: a workgroup that owns
// the whole CU (K1) has nobody to hide a dependent load chain behind, so the chain must not exist.
template <int TW, int TH, int NT>
__device__ __forceinline__ void load_tile_batched(const StageArgs &a, int n, int y0, int x0, uint8_t *s_img) {
    constexpr int PW = TW + 2 * kHalo, PH = TH + 2 * kHalo;
    constexpr int PER = (3 * PH * PW + NT - 1) / NT;
    const int total = a.C * PH * PW;
    const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
    uint8_t v[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = (int)threadIdx.x + k * NT;
        const int px = i % PW, py = (i / PW) % PH, c = imin(i / (PW * PH), a.C - 1);   // past the end: a valid address, never stored
        const int gy = imin(imax(y0 + py - kHalo, ylo), yhi);
        const int gx = imin(imax(x0 + px - kHalo, 0), a.W - 1);
        v[k] = *view_addr(a.in, n, c, gy, gx);
    }
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int i = (int)threadIdx.x + k * NT;
        if (i < total) s_img[i] = v[k];
    }
}

// same tile, stored as 16-bit pixel codes (mulut_core.h pixel_code) for the expanded-band kernel
template <int TW, int TH, int NT>
__device__ __forceinline__ void load_tile_code(const StageArgs &a, int n, int y0, int x0, uint16_t *s_img) {
    constexpr int PW = TW + 2 * kHalo, PH = TH + 2 * kHalo;
    const int total = a.C * PH * PW;
    const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
    for (int i = threadIdx.x; i < total; i += NT) {
        const int px = i % PW;
        const int py = (i / PW) % PH;
        const int c = i / (PW * PH);
        const int gy = imin(imax(y0 + py - kHalo, ylo), yhi);
        const int gx = imin(imax(x0 + px - kHalo, 0), a.W - 1);
        s_img[i] = (uint16_t)pixel_code(*view_addr(a.in, n, c, gy, gx));
    }
}

__device__ __forceinline__ void decode_tile(const StageArgs &a, int tile, int &n, int &y0, int &x0, int TW, int TH) {
    int b = tile;
    const int tx = b % a.tiles_x;
    b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    n = b / a.tiles_y;
    y0 = a.oy0 + ty * TH;
    x0 = tx * TW;
}

// Workgroups are dealt round-robin over the 8 XCDs (ids b and b+8 share one L2), so give each XCD
// a contiguous range of tiles: neighbouring tiles then share the 128-B lines their halos straddle in
// ONE L2 instead of fetching them from HBM twice.  Bijective for any n (cdna guide T1).
__device__ __forceinline__ int xcd_remap(int id, int n) {
    const int q = n >> 3, r = n & 7, xcd = id & 7, idx = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// ------------------------------------------------------------------------------------------
// K1: stage with 1-byte rows.  One workgroup = one TH x TW pixel tile x all channels.
// LDS: [ table of the active mode : 83536 B ][ image tile C*(TH+4)*(TW+4) B ]
// ------------------------------------------------------------------------------------------
template <int TW, int TH, int NT, int SPT>
__global__ void __launch_bounds__(NT) stage_u1_kernel(StageArgs a) {
    constexpr int PW = TW + 2 * kHalo, PH = TH + 2 * kHalo;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int8_t *s_lut = (const int8_t *)smem;
    uint8_t *s_img = smem + kU1TableBytes;

    int n, y0, x0;
    decode_tile(a, xcd_remap(blockIdx.x, gridDim.x), n, y0, x0, TW, TH);
    load_tile<TW, TH, NT>(a, n, y0, x0, s_img);

    // Sites of this thread: s = tid + k*NT over [C][TH][TW].  With fewer than 3 channels the surplus
    // sites are folded back onto valid ones (recomputed, never stored), which keeps the loop body free
    // of per-site branches so that the compiler can interleave the SPT x 4 independent passes.
    const int nsamp = a.C * TH * TW;
    int ctr_off[SPT];
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        int s = threadIdx.x + k * NT;
        s = s < nsamp ? s : s % (TH * TW);
        const int tx = s % TW, ty = (s / TW) % TH, c = s / (TW * TH);
        ctr_off[k] = c * (PH * PW) + (ty + kHalo) * PW + (tx + kHalo);
    }
    int acc[SPT];
#pragma unroll
    for (int k = 0; k < SPT; ++k) acc[k] = 0;

    for (int mv = 0; mv < a.M; ++mv) {
        const int m = __builtin_amdgcn_readfirstlane(mv);   // SGPR: per-mode arguments by scalar loads
        __syncthreads();  // tile filled (m == 0) / everyone done with the previous table
#if MULUT_ABLATE != 13
        {
            const uint4 *src = (const uint4 *)a.lut[m];
            uint4 *dst = (uint4 *)smem;
            for (int i = threadIdx.x; i < kU1TableBytes / 16; i += NT) dst[i] = src[i];
        }
#endif
        __syncthreads();
        int off[4][3];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int dy, dx;
                sample_offset(r, a.di[m][k], a.dj[m][k], dy, dx);
                off[r][k] = dy * PW + dx;
            }
#pragma unroll
        for (int k = 0; k < SPT; ++k) {
            const uint8_t *ctr = s_img + ctr_off[k];
            const int va = ctr[0];
            // phases instead of four serial passes: 12 neighbour reads, 4 index computations, 20 table
            // reads, 20 MACs -- two LDS round trips per site and mode instead of eight
            int vb[4], vc[4], vd[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                vb[r] = ctr[off[r][0]];
                vc[r] = ctr[off[r][1]];
                vd[r] = ctr[off[r][2]];
            }
            int idx[4][5], w[4][5];
#pragma unroll
            for (int r = 0; r < 4; ++r) simplex4(va, vb[r], vc[r], vd[r], idx[r], w[r]);
#if MULUT_ABLATE == 11
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 5; ++j) idx[r][j] &= (a.N >> 30);
#endif
            int lv[4][5];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 5; ++j) {
#if MULUT_ABLATE == 12
                    lv[r][j] = idx[r][j];
#else
                    lv[r][j] = (int)s_lut[idx[r][j]];
#endif
                }
            int sum = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 5; ++j) sum += w[r][j] * lv[r][j];
            acc[k] += sum;
        }
    }
#pragma unroll
    for (int k = 0; k < SPT; ++k) {
        const int s = threadIdx.x + k * NT;
        if (s < nsamp) {
            const int tx = s % TW;
            const int ty = (s / TW) % TH;
            const int c = s / (TW * TH);
            const int y = y0 + ty, x = x0 + tx;
            if (y < a.oy1 && x < a.W) {
                const uint32_t v = rhe_clip_u8(acc[k] + a.bias_num, a.div);
                *const_cast<uint8_t *>(view_addr(a.out, n, c, y, x)) = (uint8_t)v;
            }
        }
    }
}

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
constexpr int kPatDi[3][3] = {{0, 1, 1}, {0, 2, 2}, {1, 1, 2}};   // s, d, y: row offsets of keys b, c, d (pattern_offsets)
constexpr int kPatDj[3][3] = {{1, 0, 1}, {2, 0, 2}, {1, 2, 1}};
constexpr int rot_dy(int r, int di, int dj) { return r == 0 ? di : r == 1 ? dj : r == 2 ? -di : -dj; }   // sample_offset
constexpr int rot_dx(int r, int di, int dj) { return r == 0 ? dj : r == 1 ? -di : r == 2 ? -dj : di; }

template <int ROW, int COL>
__device__ __forceinline__ int win_byte(const uint32_t (&win)[5][2]) {
    static_assert(ROW >= 0 && ROW < 5 && COL >= 0 && COL < 8, "window is 5 rows x 8 bytes");
    return (int)((win[ROW][COL >> 2] >> (8 * (COL & 3))) & 0xFFu);
}

// One mode over the thread's 3 x 4 sites.  Both loops are real loops (one pixel body per pattern in the binary,
// and nothing of a later pixel can be scheduled into an earlier one): the pixel loop shifts the window left by one
// byte per step so that the current pixel always sits at window column 2, and the accumulators rotate through
// fixed registers -- four steps per channel, three channel groups -- instead of being indexed.
template <int PAT, int PW, int PH>
__device__ __forceinline__ void u1w_mode(const int8_t *s_lut, const uint8_t *s_img, int ty, int x4, int C, int (&acc)[12]) {
    int c = 0;
#pragma clang loop unroll(disable)
    for (; c < C; ++c) {
        // window row q = image row y - 2 + q = tile row ty + q; byte j = pixel x4 - 2 + j = tile column x4 + j
        const uint32_t *row = (const uint32_t *)(s_img + c * (PH * PW) + ty * PW + x4);
        uint32_t win[5][2];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            win[q][0] = row[q * (PW / 4)];
            win[q][1] = row[q * (PW / 4) + 1];
        }
#pragma clang loop unroll(disable)
        for (int i = 0; i < 4; ++i) {
            const int va = win_byte<2, 2>(win);
            int idx[4][5], w[4][5];
            static_for<0, 4>([&](auto RR) {
                constexpr int r = RR;
                const int vb = win_byte<2 + rot_dy(r, kPatDi[PAT][0], kPatDj[PAT][0]), 2 + rot_dx(r, kPatDi[PAT][0], kPatDj[PAT][0])>(win);
                const int vc = win_byte<2 + rot_dy(r, kPatDi[PAT][1], kPatDj[PAT][1]), 2 + rot_dx(r, kPatDi[PAT][1], kPatDj[PAT][1])>(win);
                const int vd = win_byte<2 + rot_dy(r, kPatDi[PAT][2], kPatDj[PAT][2]), 2 + rot_dx(r, kPatDi[PAT][2], kPatDj[PAT][2])>(win);
                simplex4(va, vb, vc, vd, idx[r], w[r]);
            });
            int lv[4][5];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 5; ++j) lv[r][j] = (int)s_lut[idx[r][j]];
            int sum = acc[0];
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 5; ++j) sum += w[r][j] * lv[r][j];
            acc[0] = acc[1]; acc[1] = acc[2]; acc[2] = acc[3]; acc[3] = sum;       // next pixel's accumulator to slot 0
#pragma unroll
            for (int q = 0; q < 5; ++q) {                                            // window one pixel to the left
                win[q][0] = __builtin_amdgcn_alignbit(win[q][1], win[q][0], 8);
                win[q][1] >>= 8;
            }
        }
        // next channel's four accumulators to slots 0..3
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int t = acc[k]; acc[k] = acc[4 + k]; acc[4 + k] = acc[8 + k]; acc[8 + k] = t; }
    }
#pragma clang loop unroll(disable)
    for (; c < 3; ++c) {   // fewer than three channels: finish the cycle so that slot order is restored
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int t = acc[k]; acc[k] = acc[4 + k]; acc[4 + k] = acc[8 + k]; acc[8 + k] = t; }
    }
}

// dst = a + (16-bit half SEL of b): one SDWA add extracts and adds
template <int SEL>
__device__ __forceinline__ uint32_t add_word(uint32_t a, uint32_t b) {
    uint32_t r;
    if constexpr (SEL == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(a), "v"(b));
    else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

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
    const int8_t *s_lut = (const int8_t *)smem;
    uint8_t *s_img = smem + kU1TableBytes;

    // list mode: a fixed grid of persistent workgroups; each walks an XCD-contiguous range of tiles (neighbouring tiles
    // share halo lines in one L2) and takes those the tube kernel marked in a.tile_list[tile]
    constexpr bool listed = LIST;
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
        while (t_cur < t_last && a.tile_list[t_cur / nsub] == 0u) t_cur += t_step;       // workgroup-uniform
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
#if defined(MULUT_VARIANT_u1wold)
        if (pat == 0) u1w_mode<0, PW, PH>(s_lut, img_c, ty, x4, c_n, acc);
        else if (pat == 1) u1w_mode<1, PW, PH>(s_lut, img_c, ty, x4, c_n, acc);
        else u1w_mode<2, PW, PH>(s_lut, img_c, ty, x4, c_n, acc);
#else
        if (pat == 0) u1p_mode<0, PW, PH>(s_lut, img_c, ty, x4, c_n, acc);
        else if (pat == 1) u1p_mode<1, PW, PH>(s_lut, img_c, ty, x4, c_n, acc);
        else u1p_mode<2, PW, PH>(s_lut, img_c, ty, x4, c_n, acc);
#endif
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
    auto kern = variant == 1 ? stage_u1_kernel<K1_TW, K1_TH, K1_NT, K1_SPT> : stage_u1w_kernel<K1_TW, K1_TH, K1_NT, false>;
    const size_t lds = (size_t)kU1TableBytes + (size_t)a.C * (K1_TH + 2 * kHalo) * (K1_TW + 2 * kHalo);
    static bool attr_set[64][2] = {};  // per device and variant: >64 KB of dynamic LDS has to be opted into
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev][variant == 1]) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
        if (e != hipSuccess) return e;
        attr_set[dev][variant == 1] = true;
    }
    const long long nb = (long long)a.N * a.tiles_x * a.tiles_y;
    if (nb <= 0 || nb > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(K1_NT), lds, st, a);
    return hipGetLastError();
}

// acc += x * w[WHALF] per 16-bit lane; SWAP exchanges the halves of x (reversed rotation).  One
// v_pk_mad_u16 each, the selects are free (op_sel / op_sel_hi).
template <int WHALF, bool SWAP>
__device__ __forceinline__ void pk_mac(uint32_t &acc, uint32_t x, uint32_t wpk) {
    uint32_t r;   // three-address form: the register allocator decides whether the sum stays in place
    if constexpr (WHALF == 0 && !SWAP) asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "v"(wpk), "v"(acc));
    if constexpr (WHALF == 1 && !SWAP) asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "=v"(r) : "v"(x), "v"(wpk), "v"(acc));
    if constexpr (WHALF == 0 && SWAP) asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,0,1]" : "=v"(r) : "v"(x), "v"(wpk), "v"(acc));
    if constexpr (WHALF == 1 && SWAP) asm("v_pk_mad_u16 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(x), "v"(wpk), "v"(acc));
    acc = r;
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
constexpr int K1T_TW = 64, K1T_TH = 64, K1T_NT = 512;      // 512 threads take the tile's 64 x 64 sites in two halves of 32 rows
constexpr int K1T_PW = K1T_TW + 2 * kHalo, K1T_PH = K1T_TH + 2 * kHalo;
constexpr int kU1tTileBytes = 3 * K1T_PH * K1T_PW * 2;
constexpr int kU1tLdsBytes = 3 * kTube1BandBytes + kU1tTileBytes + 16;

// packed pair of window codes: low half = code at (R1, C1), high half = code at (R2, C2); window column c lives in
// dword c / 2, half c % 2.  One v_perm_b32 (selector bytes 0-3 pick from the second operand, 4-7 from the first).
template <int R1, int C1, int R2, int C2, int NW>
__device__ __forceinline__ uint32_t win_pair(const uint32_t (&w)[5][NW]) {
    static_assert(R1 >= 0 && R1 < 5 && R2 >= 0 && R2 < 5 && C1 >= 0 && C1 < 2 * NW && C2 >= 0 && C2 < 2 * NW, "window is 5 rows x 2 NW codes");
    constexpr uint32_t sel = ((C1 & 1) ? 0x0302u : 0x0100u) | (((C2 & 1) ? 0x0706u : 0x0504u) << 16);
    return __builtin_amdgcn_perm(w[R2][C2 / 2], w[R1][C1 / 2], sel);
}

// LDS read at an integer byte address (address space 3 pointers are 32-bit offsets into the workgroup's allocation)
__device__ __forceinline__ uint32_t lds_u32(uint32_t addr) {
    return *(const __attribute__((address_space(3))) uint32_t *)(uintptr_t)addr;
}
__device__ __forceinline__ uint32_t lds_addr_of(const void *p) {
    return (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void *)p;
}

// dst = a + (byte SEL of b): one full-rate SDWA add, the stride byte of a sort key needs no masking
template <int SEL>
__device__ __forceinline__ uint32_t add_byte(uint32_t a, uint32_t b) {
    uint32_t r;
    if constexpr (SEL == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_0" : "=v"(r) : "v"(a), "v"(b));
    else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// accumulators of one site: u == 1 one int32; u == 2 two rotation-pair sets of four 16-bit fields (value + 128 rows, as
// the u == 4 kernels: a02 holds rotations 0 and 2 -- the latter added in reversed element order -- a13 rotations 1 and 3)
template <int U> struct U1tAcc;
template <> struct U1tAcc<1> { int v; __device__ __forceinline__ void clear() { v = 0; } };
template <> struct U1tAcc<2> {
    uint32_t a02[2], a13[2];
    __device__ __forceinline__ void clear() { a02[0] = a02[1] = a13[0] = a13[1] = 0; }
};
__device__ __forceinline__ uint2 lds_u64(uint32_t addr) {
    typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
    const u32x2 v = *(const __attribute__((address_space(3))) u32x2 *)(uintptr_t)addr;
    return make_uint2(v.x, v.y);
}
template <int U> __host__ __device__ constexpr int u1t_band_bytes() { return U == 1 ? kTube1BandBytes : kTube2BandBytes; }

// rotations R and R + 2 of the pixel at window column I + 2, pattern PAT
template <int U, int PAT, int R, int I, int NW>
__device__ __forceinline__ void u1t_pair(const uint32_t (&win)[5][NW], uint32_t k0, uint32_t base_a, U1tAcc<U> &acc) {
    constexpr int SHIFT = U == 1 ? 2 : 3;
    constexpr int BAND = PAT * u1t_band_bytes<U>();      // LDS byte address of this pattern's band
    constexpr int yb = rot_dy(R, kPatDi[PAT][0], kPatDj[PAT][0]), xb = rot_dx(R, kPatDi[PAT][0], kPatDj[PAT][0]);
    constexpr int yc = rot_dy(R, kPatDi[PAT][1], kPatDj[PAT][1]), xc = rot_dx(R, kPatDi[PAT][1], kPatDj[PAT][1]);
    constexpr int yd = rot_dy(R, kPatDi[PAT][2], kPatDj[PAT][2]), xd = rot_dx(R, kPatDi[PAT][2], kPatDj[PAT][2]);
#if MULUT_ABLATE == 34   /* timing-only: no neighbour perms */
    const uint32_t pb = k0 + R, pc = k0 ^ (uint32_t)(PAT + 1), pd = k0 + 0x10u * I;
#else
    const uint32_t pb = win_pair<2 + yb, I + 2 + xb, 2 - yb, I + 2 - xb, NW>(win);
    const uint32_t pc = win_pair<2 + yc, I + 2 + xc, 2 - yc, I + 2 - xc, NW>(win);
    const uint32_t pd = win_pair<2 + yd, I + 2 + xd, 2 - yd, I + 2 - xd, NW>(win);
#endif
    TubePair1 bp;
    simplex4_tube_pair1<SHIFT>(k0, base_a, pb, pc, pd, bp);
    // byte offsets of rows 0..3 of both passes, unpacked: row j + 1 = row j + stride byte of sorted key j
    uint32_t aa[4], ab[4];
    aa[0] = bp.base & 0xFFFFu;
    ab[0] = bp.base >> 16;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        aa[j + 1] = add_byte<0>(aa[j], bp.ks[j]);
        ab[j + 1] = add_byte<2>(ab[j], bp.ks[j]);
    }
    constexpr int kRow4 = kTubeAll << SHIFT;
    if constexpr (U == 1) {
        uint32_t xa[5], xb2[5];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
#if MULUT_ABLATE == 31   /* timing-only: no band reads */
            xa[j] = aa[j & 3] + j; xb2[j] = ab[j & 3] ^ (uint32_t)j;
#else
            // LDS addresses as plain integers (dynamic LDS starts at 0 -- checked at kernel entry): going through the
            // `smem` symbol would cost one v_add of a link-time zero per read
            xa[j] = lds_u32(aa[j < 4 ? j : 0] + (uint32_t)(BAND + (j < 4 ? 0 : kRow4)));
            xb2[j] = lds_u32(ab[j < 4 ? j : 0] + (uint32_t)(BAND + (j < 4 ? 0 : kRow4)));
#endif
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            typedef short s16x2 __attribute__((ext_vector_type(2)));
            const uint32_t t = (xa[j] & 0x0000FFFFu) | (xb2[j] & 0xFFFF0000u);     // value of pass A | value of pass B
            acc.v = __builtin_amdgcn_sdot2(__builtin_bit_cast(s16x2, t), __builtin_bit_cast(s16x2, bp.w[j]), acc.v, false);
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
#if !defined(MULUT_VARIANT_k1ilp)
    if constexpr (U == 1) asm volatile("" : "+v"(acc.v), "+v"(k0));
    else asm volatile("" : "+v"(acc.a02[0]), "+v"(k0));
#endif
    u1t_pair<U, PAT, 1, I, 3>(win, k0, base_a, acc);
}

// one pixel: all modes, then the byte (u == 1) or the 2 x 2 block as four bytes, row-major (u == 2)
template <int U, int I>
__device__ __forceinline__ uint32_t u1t_pixel(const StageArgs &a, const uint32_t (&win)[5][3]) {
    U1tAcc<U> acc;
    acc.clear();
    // anchor terms, the same for every mode and rotation of the pixel
    const uint32_t ca_pk = win_pair<2, I + 2, 2, I + 2, 3>(win);
    uint32_t k0 = tube1_key(ca_pk, kTubeSA << (U == 1 ? 2 : 3));
    const uint32_t base_a = pk_mad(ca_pk, pk_dup(16 * kTubeSA), 0u);
    for (int mv = 0; mv < a.M; ++mv) {
        const int m = __builtin_amdgcn_readfirstlane(mv);
        const int pat = a.dj[m][0] == 2 ? 1 : a.di[m][0] == 1 ? 2 : 0;
        if (pat == 0) u1t_mode<U, 0, I>(win, k0, base_a, acc);
        else if (pat == 1) u1t_mode<U, 1, I>(win, k0, base_a, acc);
        else u1t_mode<U, 2, I>(win, k0, base_a, acc);
    }
    if constexpr (U == 1) {
        if (a.use_fma)       // wave-uniform: fused float epilogue proven exact; v_cvt_pk_u8_f32 rounds to nearest even and saturates
            return __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)acc.v, a.inv_d, a.epi_c), 0u, 0u);
        return rhe_clip_u8(acc.v + a.bias_num, a.div);
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

#if defined(MULUT_VARIANT_k1ilp) || defined(MULUT_VARIANT_k1w4)
#define K1T_WAVES 4
#else
#define K1T_WAVES 6      // 80 VGPRs: three 512-thread workgroups per CU (8 waves per SIMD would mean 64 VGPRs and spills in the pair loop)
#endif
template <int U>
__global__ void __launch_bounds__(K1T_NT, K1T_WAVES) stage_u1t_kernel(StageArgs a, BandArgs b, uint32_t detail_per_1024) {
    constexpr int TW = K1T_TW, TH = K1T_TH, NT = K1T_NT, PW = K1T_PW, PH = K1T_PH;
    constexpr int BB = u1t_band_bytes<U>();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *s_tile = smem + 3 * BB;
    uint32_t *s_cnt = (uint32_t *)(smem + 3 * BB + kU1tTileBytes);     // [0] detailed groups, [1] groups looked at
    if (lds_addr_of(smem) != 0u) __builtin_trap();      // the band reads assume the dynamic LDS block starts at address 0 (no static LDS here): fail loudly, never skip the work

    for (int m = 0; m < a.M; ++m) {
        const int pat = a.dj[m][0] == 2 ? 1 : a.di[m][0] == 1 ? 2 : 0;
        const uint32_t *src = (const uint32_t *)b.band[m];
        uint32_t *dst = (uint32_t *)(smem + pat * BB);
        for (int i = (int)threadIdx.x; i < BB / 4; i += NT) dst[i] = src[i];
    }
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

    for (int tile = first; tile < last; tile += step) {
        int n, y0, x0;
        decode_tile(a, tile, n, y0, x0, TW, TH);
        __syncthreads();      // everyone is done with the previous tile (and, first trip, the bands are staged)
        if (threadIdx.x == 0) {
            uint32_t z = 0;
            asm volatile("" : "+v"(z));      // made here: the compiler otherwise keeps a zero pair live across the whole kernel -- in scratch
            s_cnt[0] = z; s_cnt[1] = z;
        }
        constexpr int GR = (TW + 8) / 4;            // 18 four-pixel groups cover image columns x0-4 .. x0+67
        // a group of four pixels of one image row, as two packed byte pairs per channel (edge columns replicated)
        auto group_hwc = [&](int row, int g, uint32_t (&bp)[6]) {
            const int gy = imin(imax(y0 + row - kHalo, ylo), yhi);
            const int gx = x0 - 4 + 4 * g, cgx = imin(imax(gx, 0), a.W - 4);
            const uint32_t *src = (const uint32_t *)view_addr(a.in, n, 0, gy, cgx);
            const uint32_t d0 = src[0], d1 = src[1], d2 = src[2];      // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
            bp[0] = __builtin_amdgcn_perm(0u, d0, 0x0C030C00u); bp[1] = __builtin_amdgcn_perm(d2, d1, 0x0C050C02u);
            bp[2] = __builtin_amdgcn_perm(d1, d0, 0x0C040C01u); bp[3] = __builtin_amdgcn_perm(d2, d1, 0x0C060C03u);
            bp[4] = __builtin_amdgcn_perm(d1, d0, 0x0C050C02u); bp[5] = __builtin_amdgcn_perm(0u, d2, 0x0C030C00u);
            if (gx < 0) {                 // left of the image: every column replicates column 0
                bp[0] = bp[1] = pk_dup(bp[0] & 0xFFFFu); bp[2] = bp[3] = pk_dup(bp[2] & 0xFFFFu); bp[4] = bp[5] = pk_dup(bp[4] & 0xFFFFu);
            } else if (gx > a.W - 4) {    // right of it: column W-1
                bp[0] = bp[1] = pk_dup(bp[1] >> 16); bp[2] = bp[3] = pk_dup(bp[3] >> 16); bp[4] = bp[5] = pk_dup(bp[5] >> 16);
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) bp[k] = codes_of(bp[k]);
        };
        auto group_planar = [&](int c, int row, int g, uint32_t &p01, uint32_t &p23) {
            const int gy = imin(imax(y0 + row - kHalo, ylo), yhi);
            const int gx = x0 - 4 + 4 * g, cgx = imin(imax(gx, 0), a.W - 4);
            const uint32_t d = *(const uint32_t *)view_addr(a.in, n, c, gy, cgx);
            p01 = __builtin_amdgcn_perm(0u, d, 0x0C010C00u); p23 = __builtin_amdgcn_perm(0u, d, 0x0C030C02u);
            if (gx < 0) p01 = p23 = pk_dup(p01 & 0xFFFFu);
            else if (gx > a.W - 4) p01 = p23 = pk_dup(p23 >> 16);
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
            for (int o = 32; o > 0; o >>= 1) { far += __shfl_down(far, o); seen += __shfl_down(seen, o); }
            __syncthreads();      // counters zeroed before anyone adds
            if ((threadIdx.x & 63) == 0 && seen) { atomicAdd(&s_cnt[0], far); atomicAdd(&s_cnt[1], seen); }
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
#if MULUT_ABLATE == 33   /* timing-only: no neighbourhood test */
                dirty = win8[0][0] >> 31;
#else
                dirty = u1t_dirty(win8);
#endif
                asm volatile("" : "+v"(dirty));     // computed HERE: sunk below the pixel loop, its 20 window registers would be parked in scratch
            }
            uint32_t packed = 0;
            // two pixels per step of a real loop: their 5 x 6 window is re-read (dword-aligned), nothing of a later step
            // can be scheduled into an earlier one
#pragma clang loop unroll(disable)
            for (int it = 0; it < 2; ++it) {
                int ty, tx4;
                coords(ty, tx4);
                const int y = y0 + ty, x = x0 + tx4;
                if (U == 2 && x + 2 * it >= a.W) break;
                uint32_t win[5][3];
                const uint32_t *row = (const uint32_t *)(s_tile + 2 * ((c * PH + ty) * PW + tx4 + 2 * it));
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    win[q][0] = row[q * (PW / 2)]; win[q][1] = row[q * (PW / 2) + 1]; win[q][2] = row[q * (PW / 2) + 2];
                }
                uint32_t b0 = u1t_pixel<U, 0>(a, win);
#if !defined(MULUT_VARIANT_k1ilp)
                asm volatile("" : "+v"(b0), "+v"(win[2][1]));       // the second pixel starts after the first is done
#endif
                const uint32_t b1 = u1t_pixel<U, 1>(a, win);
                if constexpr (U == 1) {
                    packed |= (b0 | (b1 << 8)) << (16 * it);
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
            // sites that may have left the tube: onto the fix-up list, one atomic per wave and pixel slot (rare)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool d = ((dirty >> i) & 1u) != 0u && x + i < a.W;
                const unsigned long long dm = __ballot(d);
                if (dm != 0ull) {
                    const int lane = (int)(threadIdx.x & 63), lead = __ffsll((long long)dm) - 1;
                    uint32_t at = 0;
                    if (lane == lead) at = atomicAdd(a.fix_count, (uint32_t)__popcll(dm));
                    at = (uint32_t)__shfl((int)at, lead);
                    if (d) a.fix_list[at + (uint32_t)__popcll(dm & ((1ull << lane) - 1ull))] = (uint32_t)(((n * a.C + c) * a.H + y) * a.W + x + i);
                }
            }
        }
        }
    }
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
int g_u1t_persist = 0;      // experiment knob (mulut_set_tuning "u1t_persist")

template <int U>
static hipError_t launch_u1t_t(const StageArgs &a, const BandArgs &b, unsigned detail_per_1024, int num_cus, hipStream_t st) {
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)stage_u1t_kernel<U>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    const long long ntiles = (long long)a.N * a.tiles_x * a.tiles_y;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL) return hipErrorInvalidValue;
    // persist_per_cu > 0: that many persistent workgroups per CU walk XCD-contiguous tile ranges; 0: one workgroup per tile
    const long long want = g_u1t_persist > 0 ? (long long)g_u1t_persist * num_cus : ntiles;     // (three 512-thread workgroups fit a CU)
    const unsigned grid = (unsigned)(ntiles < want ? ntiles : want);
    const size_t lds = 3 * (size_t)u1t_band_bytes<U>() + kU1tTileBytes + 16;
    hipLaunchKernelGGL(stage_u1t_kernel<U>, dim3(grid), dim3(K1T_NT), lds, st, a, b, (uint32_t)detail_per_1024);
    return hipGetLastError();
}

hipError_t launch_stage_u1t(const StageArgs &a, const BandArgs &b, unsigned detail_per_1024, int num_cus, hipStream_t st) {
    if (a.C > 3 || a.M > 3 || !a.fix_list || !a.fix_count) return hipErrorInvalidValue;
    return launch_u1t_t<1>(a, b, detail_per_1024, num_cus, st);
}

hipError_t launch_stage_u1w_list(const StageArgs &a, int num_cus, hipStream_t st) {
    if (a.C > 3 || !a.tile_list || !a.tile_count) return hipErrorInvalidValue;
    auto kern = stage_u1w_kernel<K1_TW, K1_TH, K1_NT, true>;
    const size_t lds = (size_t)kU1TableBytes + (size_t)a.C * (K1_TH + 2 * kHalo) * (K1_TW + 2 * kHalo);
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
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

// ------------------------------------------------------------------------------------------
// Tile statistic for the final stage: how many (pixel, channel) sites of each 64x16 tile have a
// 5x5 neighbour whose MSB differs from the centre's by more than 1, i.e. at least one of the 12
// passes of that site leaves the LDS band.  Smooth tiles (few such sites) go to the band kernel,
// detailed ones to the full-table kernel; both read the per-tile verdict from device memory, so the
// choice costs no host synchronisation and the pipeline stays capturable into a hipGraph.
// ------------------------------------------------------------------------------------------
// anchor-MSB histogram of tile `tile` (of ntiles), bin b, as uint16 (a tile has 3072 samples): eight bins of one tile per
// 16 bytes, tile-major inside a half.  The positions detail_plan_kernel derives (uint32) use the same scheme with four bins
// per 16 bytes.
__device__ __forceinline__ size_t detail_hist_index(uint32_t tile, uint32_t ntiles, int b) {
    return ((size_t)(b >> 3) * ntiles + tile) * 8 + (size_t)(b & 7);
}
__device__ __forceinline__ size_t detail_pos_index(uint32_t tile, uint32_t ntiles, int b) {
    return ((size_t)(b >> 2) * ntiles + tile) * 4 + (size_t)(b & 3);
}

template <int TW, int TH>
__global__ void __launch_bounds__(256) tile_stat_kernel(StageArgs a, uint32_t *verdict, uint32_t max_oob_per_1024, uint16_t *thist, uint32_t *any) {
    constexpr int PW = TW + 2 * kHalo, PH = TH + 2 * kHalo;
    __shared__ uint8_t s_h[3 * PH * PW];
    __shared__ uint32_t s_cnt, s_valid;
    __shared__ uint32_t s_hist[16];
    int n, y0, x0;
    const int id = xcd_remap(blockIdx.x, gridDim.x);   // neighbouring tiles on one XCD: halo lines fetched once
    decode_tile(a, id, n, y0, x0, TW, TH);
    if (a.k1_hdr && a.k1_hdr[2] != 0u) {
        // the statistic folded into the first stage: its tube kernel routed its 64x64 tiles by the detail of ITS input, and what it
        // left unmarked is smooth enough here too (a wrong guess costs fix-up work, never exactness)
        const int k1 = (n * a.k1_tiles_y + (y0 - a.k1_oy0) / 64) * a.k1_tiles_x + x0 / 64;
        if (a.k1_hdr[16 + k1] == 0u) {          // workgroup-uniform
            if (threadIdx.x == 0) verdict[id] = 0u;
            return;
        }
    }
    if (threadIdx.x == 0) { s_cnt = 0; s_valid = 0; }
    const int total = a.C * PH * PW;
    const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
    // Planar input whose rows start on dword boundaries (the pipeline's intermediate images): aligned dwords covering
    // image columns x0-4 .. x0+67, all in flight before the first LDS store (the kernel is nothing but latency); columns left
    // of 0 / right of W-1 replicate the edge byte.  Any other input: byte loads.
    const bool planar = a.in.sX == 1 && ((a.W | a.in.sY | a.in.sC) & 3) == 0 && (a.in.sN & 3) == 0 && (((uintptr_t)a.in.p) & 3) == 0;
    if (planar) {
        constexpr int GR = (TW + 8) / 4, PER4 = (3 * PH * GR + 255) / 256;
        uint32_t v[PER4];
#pragma unroll
        for (int k = 0; k < PER4; ++k) {
            const int i = (int)threadIdx.x + k * 256;
            const int g = i % GR, row = (i / GR) % PH, c = imin(i / (GR * PH), a.C - 1);
            const int gy = imin(imax(y0 + row - kHalo, ylo), yhi);
            const int gx = x0 - 4 + 4 * g, cgx = imin(imax(gx, 0), a.W - 4);
            uint32_t d = *(const uint32_t *)view_addr(a.in, n, c, gy, cgx);
            if (gx < 0) d = (d & 0xFFu) * 0x01010101u;
            else if (gx > a.W - 4) d = (d >> 24) * 0x01010101u;
            v[k] = (d >> 4) & 0x0F0F0F0Fu;
        }
#pragma unroll
        for (int k = 0; k < PER4; ++k) {
            const int i = (int)threadIdx.x + k * 256;
            if (i < a.C * PH * GR) {
                const int g = i % GR, row = (i / GR) % PH, c = i / (GR * PH);
                uint8_t *dst = s_h + (c * PH + row) * PW + 4 * g - 2;       // tile columns 4g-2 .. 4g+1
                if (g > 0) { dst[0] = (uint8_t)v[k]; dst[1] = (uint8_t)(v[k] >> 8); }
                if (4 * g + 1 < PW) { dst[2] = (uint8_t)(v[k] >> 16); dst[3] = (uint8_t)(v[k] >> 24); }
            }
        }
    } else {
        constexpr int PER = (3 * PH * PW + 255) / 256;
        uint8_t v[PER];
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = (int)threadIdx.x + k * 256;
            const int px = i % PW, py = (i / PW) % PH, c = imin(i / (PW * PH), a.C - 1);
            const int gy = imin(imax(y0 + py - kHalo, ylo), yhi);
            const int gx = imin(imax(x0 + px - kHalo, 0), a.W - 1);
            v[k] = *view_addr(a.in, n, c, gy, gx);
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int i = (int)threadIdx.x + k * 256;
            if (i < total) s_h[i] = v[k] >> 4;
        }
    }
    __syncthreads();
    // a verdict is a heuristic: every second site in x and y is enough (4x less LDS work)
    uint32_t oob = 0, valid = 0;
    for (int s = threadIdx.x; s < a.C * (TH / 2) * (TW / 2); s += 256) {
        const int tx = 2 * (s % (TW / 2)), ty = 2 * ((s / (TW / 2)) % (TH / 2)), c = s / ((TW / 2) * (TH / 2));
        if (y0 + ty >= a.oy1 || x0 + tx >= a.W) continue;
        const uint8_t *ctr = s_h + c * (PH * PW) + (ty + kHalo) * PW + (tx + kHalo);
        const int hc = ctr[0];
        int lo = hc, hi = hc;
#pragma unroll
        for (int dy = -2; dy <= 2; ++dy)
#pragma unroll
            for (int dx = -2; dx <= 2; ++dx) {
                const int v = ctr[dy * PW + dx];
                lo = imin(lo, v);
                hi = imax(hi, v);
            }
        oob += (hi - hc > 1 || hc - lo > 1) ? 1u : 0u;
        valid += 1u;
    }
    for (int o = 32; o > 0; o >>= 1) {   // wave reduction, then one LDS atomic per wave
        oob += __shfl_down(oob, o);
        valid += __shfl_down(valid, o);
    }
    if ((threadIdx.x & 63) == 0) {
        atomicAdd(&s_cnt, oob);
        atomicAdd(&s_valid, valid);
    }
    __syncthreads();
    const bool detailed = s_cnt * 1024u > max_oob_per_1024 * s_valid;       // workgroup-uniform
    if (threadIdx.x == 0) {
        verdict[id] = detailed ? 1u : 0u;
        if (detailed && any) *any = 1u;      // (every writer stores the same value)
    }
    // a detailed tile also leaves the anchor-MSB histogram of its samples for the anchor-slab path (launch_detail_slab)
    if (detailed && thist) {
        if (threadIdx.x < 16) s_hist[threadIdx.x] = 0;
        __syncthreads();
        // a thread counts its (at most 12) samples in sixteen 4-bit fields first and then adds one number per anchor MSB it met:
        // on photographs that is two or three LDS atomics per thread instead of twelve
        unsigned long long mine = 0ull;
        for (int s = threadIdx.x; s < a.C * TH * TW; s += 256) {
            const int tx = s % TW, ty = (s / TW) % TH, c = s / (TW * TH);
            const int x = x0 + tx;
            if (y0 + ty < a.oy1 && x < a.W && x >= kSlabXLo && x < a.W - slab_x_hi(a))
                mine += 1ull << (4 * s_h[c * (PH * PW) + (ty + kHalo) * PW + (tx + kHalo)]);
        }
        static_assert(3 * TW * TH / 256 < 16, "a 4-bit field holds a thread's samples");
        while (mine != 0ull) {
            const int b = (__ffsll((long long)mine) - 1) >> 2;
            atomicAdd(&s_hist[b], (uint32_t)((mine >> (4 * b)) & 15ull));
            mine &= ~(15ull << (4 * b));
        }
        __syncthreads();
        if (threadIdx.x < 16) thist[detail_hist_index((uint32_t)id, gridDim.x, (int)threadIdx.x)] = (uint16_t)s_hist[threadIdx.x];
    }
}

// ------------------------------------------------------------------------------------------
// Site flags for the tube kernel (and the hybrid's tile verdicts, replacing tile_stat_kernel on that path):
// flags[(n H + y) W + x] bit c = the 5x5 neighbourhood of pixel (y, x) in channel c spans more than one MSB step,
// i.e. one of the site's 12 passes may leave the tube.  MSBs are held as ONE-HOT 16-bit masks (1 << h), two pixels
// per dword: the set of MSBs in a neighbourhood is then a plain OR -- separable, 5 columns then 5 rows -- and
// "spans at most two adjacent values" is  M & ~(L | L << 1) == 0  with L = M & -M, all on packed halves.
// One workgroup per 64x16 tile (the tube kernel's tile); verdict[tile] = 1 when more than max_per_1024 of its
// pixels are flagged.
// LDS: [ one-hot tile 3 x 20 x 68 u16 ][ horizontal ORs 3 x 20 x 64 u16 ]
// ------------------------------------------------------------------------------------------
template <int TW, int TH>
__global__ void __launch_bounds__(256) site_flag_kernel(StageArgs a, uint32_t *verdict, uint8_t *flags, uint32_t max_per_1024) {
    constexpr int PW = TW + 2 * kHalo, PH = TH + 2 * kHalo, NT = 256;
    __shared__ __attribute__((aligned(16))) uint16_t s_oh[3 * PH * PW];
    __shared__ __attribute__((aligned(16))) uint16_t s_hr[3 * PH * TW];
    __shared__ uint32_t s_cnt, s_valid;
    int n, y0, x0;
    const int id = xcd_remap(blockIdx.x, gridDim.x);
    decode_tile(a, id, n, y0, x0, TW, TH);
    if (threadIdx.x == 0) { s_cnt = 0; s_valid = 0; }
    const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
    const bool planar = a.in.sX == 1 && ((a.W | a.in.sY | a.in.sC) & 3) == 0 && (a.in.sN & 3) == 0 && (((uintptr_t)a.in.p) & 3) == 0;
    if (planar) {
        constexpr int GR = (TW + 8) / 4;            // aligned dwords cover image columns x0-4 .. x0+67
        for (int i = threadIdx.x; i < a.C * PH * GR; i += NT) {
            const int g = i % GR, row = (i / GR) % PH, c = i / (GR * PH);
            const int gy = imin(imax(y0 + row - kHalo, ylo), yhi);
            const int gx = x0 - 4 + 4 * g, cgx = imin(imax(gx, 0), a.W - 4);
            uint32_t d = *(const uint32_t *)view_addr(a.in, n, c, gy, cgx);
            if (gx < 0) d = (d & 0xFFu) * 0x01010101u;                  // left of the image: column 0
            else if (gx > a.W - 4) d = (d >> 24) * 0x01010101u;          // right of it: column W-1
            const uint32_t p01 = (1u << ((d >> 4) & 15u)) | (0x10000u << ((d >> 12) & 15u));
            const uint32_t p23 = (1u << ((d >> 20) & 15u)) | (0x10000u << (d >> 28));
            uint32_t *dst = (uint32_t *)(s_oh + (c * PH + row) * PW + 4 * g - 2);      // tile columns 4g-2 .. 4g+1
            if (g > 0) dst[0] = p01;
            if (4 * g + 1 < PW) dst[1] = p23;
        }
    } else {
        for (int i = threadIdx.x; i < a.C * PH * PW; i += NT) {
            const int px = i % PW, row = (i / PW) % PH, c = i / (PW * PH);
            const int gy = imin(imax(y0 + row - kHalo, ylo), yhi);
            const int gx = imin(imax(x0 + px - kHalo, 0), a.W - 1);
            s_oh[i] = (uint16_t)(1u << (*view_addr(a.in, n, c, gy, gx) >> 4));
        }
    }
    __syncthreads();
    // columns: pixel pair (x, x+1), x even, takes tile columns x .. x+5 = three dwords
    for (int i = threadIdx.x; i < a.C * PH * (TW / 2); i += NT) {
        const int xp = i % (TW / 2), row = (i / (TW / 2)) % PH, c = i / ((TW / 2) * PH);
        const uint32_t *src = (const uint32_t *)(s_oh + (c * PH + row) * PW) + xp;
        const uint32_t d0 = src[0], d1 = src[1], d2 = src[2];
        const uint32_t mid = d1 | __builtin_amdgcn_alignbit(d1, d1, 16);            // columns x+2, x+3 in both halves
        const uint32_t e = __builtin_amdgcn_perm(d2, d0, 0x05040302u);              // low half: column x+1, high half: column x+4
        const uint32_t common = mid | e | __builtin_amdgcn_alignbit(e, e, 16);      // columns x+1 .. x+4 in both halves
        ((uint32_t *)(s_hr + (c * PH + row) * TW))[xp] = common | (d0 & 0x0000FFFFu) | (d2 & 0xFFFF0000u);
    }
    __syncthreads();
    // rows, test, flag bytes of two pixels at once
    uint32_t bad = 0, valid = 0;
    for (int i = threadIdx.x; i < TH * (TW / 2); i += NT) {
        const int xp = i % (TW / 2), ty = i / (TW / 2);
        const int y = y0 + ty, x = x0 + 2 * xp;
        if (y >= a.oy1 || x >= a.W) continue;
        uint32_t fl = 0;
        for (int c = 0; c < a.C; ++c) {
            const uint32_t *col = (const uint32_t *)(s_hr + (c * PH + ty) * TW) + xp;
            const uint32_t m = col[0] | col[TW / 2] | col[2 * (TW / 2)] | col[3 * (TW / 2)] | col[4 * (TW / 2)];
            const uint32_t low = m & pk_sub(0u, m);                                  // lowest set bit per half
            const uint32_t two = low | ((low << 1) & 0xFFFEFFFEu);                   // it and its upper neighbour
            const uint32_t out = m & ~two;
            fl |= (((out & 0xFFFFu) ? 1u : 0u) | ((out >> 16) ? 0x100u : 0u)) << c;
        }
        uint8_t *dst = flags + ((size_t)n * a.H + y) * a.W + x;
        dst[0] = (uint8_t)fl;
        bad += (fl & 0xFFu) ? 1u : 0u;
        valid += 1;
        if (x + 1 < a.W) {
            dst[1] = (uint8_t)(fl >> 8);
            bad += (fl >> 8) ? 1u : 0u;
            valid += 1;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { bad += __shfl_down(bad, o); valid += __shfl_down(valid, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&s_cnt, bad); atomicAdd(&s_valid, valid); }
    __syncthreads();
    if (threadIdx.x == 0 && verdict) verdict[id] = (s_cnt * 1024u > max_per_1024 * s_valid) ? 1u : 0u;   // 1 = detailed
}

hipError_t launch_site_flags(const StageArgs &a, uint32_t *verdict, uint8_t *flags, uint32_t max_per_1024, hipStream_t st) {
    const long long nb = (long long)a.N * a.tiles_x * a.tiles_y;   // a.tiles_* must be the 64x16 tiling
    if (nb <= 0 || nb > 0x7fffffffLL || a.C > 3 || !flags) return hipErrorInvalidValue;
    hipLaunchKernelGGL((site_flag_kernel<64, 16>), dim3((unsigned)nb), dim3(256), 0, st, a, verdict, flags, max_per_1024);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// K2: final stage, u*u-byte rows.
// One thread = one LR pixel, channels in sequence.  Per rotation the 5*M weighted rows are
// accumulated as 16-bit fields, two per dword:  lo[k] holds row elements 4k and 4k+2, hi[k] holds
// 4k+1 and 4k+3 (each table byte is value+128, so every field stays non-negative:
// M * 16 * 255 < 65536 per rotation for M <= 16).
//   stage_up_kernel    generic (u in {2,3,4}, any M): every row is gathered from the full table in
//                      global memory (L1/L2); TA-bound at ~38 cycles per gather instruction per CU.
//   stage_band_kernel  u == 4, M <= 3: the diagonal band of each table (mulut_core.h) is resident in
//                      LDS for the lifetime of a persistent workgroup; in-band passes gather with
//                      ds_read_b128, the rest fall back to the full table.  Two passes (rotations r
//                      and r+2) run side by side in packed 16-bit halves.
// ------------------------------------------------------------------------------------------
template <int U>
__device__ __forceinline__ void load_row(const void *lut, int idx, uint32_t (&row)[row_dwords(U)]) {
    constexpr int RW = row_dwords(U);
    if constexpr (RW == 4) {
        const uint4 v = *(const uint4 *)((const char *)lut + ((uint32_t)idx << 4));
        row[0] = v.x; row[1] = v.y; row[2] = v.z; row[3] = v.w;
    } else {
        const uint32_t *p = (const uint32_t *)lut + (uint32_t)idx * RW;
#pragma unroll
        for (int k = 0; k < RW; ++k) row[k] = p[k];
    }
}

// Per-rotation SWAR accumulators with compile-time names.  u == 4 merges rotation pairs (r, r+2)
// into one accumulator each (mulut_core.h "merged rotation pairs"): 16 VGPRs instead of 32.  The merged form adds
// all four rotations inside 16-bit fields (4 M 16 255 < 65536 only for M <= 4); with more modes u == 4 takes the
// per-rotation form too (MERGED = false: a field holds one rotation, M 16 255 < 65536 for M <= 16, and sum() adds
// the extracted fields in 32 bits).
template <int U, bool MERGED = (U == 4)>
struct RotAcc {
    static constexpr int RW = row_dwords(U);
    uint32_t lo0[RW], hi0[RW], lo1[RW], hi1[RW], lo2[RW], hi2[RW], lo3[RW], hi3[RW];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < RW; ++k) lo0[k] = hi0[k] = lo1[k] = hi1[k] = lo2[k] = hi2[k] = lo3[k] = hi3[k] = 0;
    }
    template <int R>
    __device__ __forceinline__ void fma(const uint32_t (&row)[RW], uint32_t w) {
        if constexpr (R == 0) swar_fma<RW>(lo0, hi0, row, w);
        if constexpr (R == 1) swar_fma<RW>(lo1, hi1, row, w);
        if constexpr (R == 2) swar_fma<RW>(lo2, hi2, row, w);
        if constexpr (R == 3) swar_fma<RW>(lo3, hi3, row, w);
    }
    // field sum of block position (sy, sx) over the four rotations
    template <int SY, int SX>
    __device__ __forceinline__ uint32_t sum() const {
        return swar_field<row_elem(0, SY, SX, U), RW>(lo0, hi0) + swar_field<row_elem(1, SY, SX, U), RW>(lo1, hi1) +
               swar_field<row_elem(2, SY, SX, U), RW>(lo2, hi2) + swar_field<row_elem(3, SY, SX, U), RW>(lo3, hi3);
    }
    __device__ __forceinline__ void finalize() {}
};

template <>
struct RotAcc<4, true> {
    static constexpr int RW = 4;
    uint32_t lo02[4], hi02[4], lo13[4], hi13[4];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < 4; ++k) lo02[k] = hi02[k] = lo13[k] = hi13[k] = 0;
    }
    // one row given as ready-made 16-bit fields (rlo[k] = e(4k) | e(4k+2) << 16, rhi[k] = e(4k+1) | e(4k+3) << 16),
    // weight = 16-bit half HALF of wpk: eight v_pk_mad_u16, the reversed rotations swap halves with op_sel
    template <int R, int HALF>
    __device__ __forceinline__ void mac_x(const uint32_t (&rlo)[4], const uint32_t (&rhi)[4], uint32_t wpk) {
        static_for<0, 4>([&](auto K) {
            constexpr int k = K;
            if constexpr (R == 0) { pk_mac<HALF, false>(lo02[k], rlo[k], wpk); pk_mac<HALF, false>(hi02[k], rhi[k], wpk); }
            if constexpr (R == 1) { pk_mac<HALF, false>(lo13[k], rlo[k], wpk); pk_mac<HALF, false>(hi13[k], rhi[k], wpk); }
            if constexpr (R == 2) { pk_mac<HALF, true>(lo02[3 - k], rhi[k], wpk); pk_mac<HALF, true>(hi02[3 - k], rlo[k], wpk); }
            if constexpr (R == 3) { pk_mac<HALF, true>(lo13[3 - k], rhi[k], wpk); pk_mac<HALF, true>(hi13[3 - k], rlo[k], wpk); }
        });
    }
    // one compact (value + 128 bytes) row, weight w <= 16 in the low half: split into fields (3 full-rate ops per dword), then mac_x
    template <int R>
    __device__ __forceinline__ void fma(const uint32_t (&row)[4], uint32_t w) {
#if MULUT_ABLATE == 4
        if constexpr (R == 0) swar_fma<4>(lo02, hi02, row, w);
        if constexpr (R == 1) swar_fma<4>(lo13, hi13, row, w);
        if constexpr (R == 2) swar_fma_rev4(lo02, hi02, row, w);
        if constexpr (R == 3) swar_fma_rev4(lo13, hi13, row, w);
#else
        uint32_t rlo[4], rhi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { rlo[k] = row[k] & 0x00FF00FFu; rhi[k] = (row[k] >> 8) & 0x00FF00FFu; }
        mac_x<R, 0>(rlo, rhi, w);
#endif
    }
    // weight = 16-bit half HALF of a packed register (band kernel): one v_pk_mad_u16 per dword
    template <int R, int HALF>
    __device__ __forceinline__ void fma_pk(const uint32_t (&row)[4], uint32_t wpk) {
        if constexpr (R == 0) swar_fma4_pk<HALF>(lo02, hi02, row, wpk);
        if constexpr (R == 1) swar_fma4_pk<HALF>(lo13, hi13, row, wpk);
        if constexpr (R == 2) swar_fma_rev4_pk<HALF>(lo02, hi02, row, wpk);
        if constexpr (R == 3) swar_fma_rev4_pk<HALF>(lo13, hi13, row, wpk);
    }
    // expanded 32-B band rows (ready-made SWAR fields)
    template <int R, int HALF>
    __device__ __forceinline__ void fma_x(const uint32_t (&rlo)[4], const uint32_t (&rhi)[4], uint32_t wpk) {
        if constexpr (R == 0) swar_fma_x4<HALF>(lo02, hi02, rlo, rhi, wpk);
        if constexpr (R == 1) swar_fma_x4<HALF>(lo13, hi13, rlo, rhi, wpk);
        if constexpr (R == 2) swar_fma_x4_rev<HALF>(lo02, hi02, rlo, rhi, wpk);
        if constexpr (R == 3) swar_fma_x4_rev<HALF>(lo13, hi13, rlo, rhi, wpk);
    }
    // after finalize(): lo02/hi02 hold the sum of all four rotations in block order
    __device__ __forceinline__ void finalize() {
        uint32_t lo[4], hi[4];
        combine_pairs4(lo02, hi02, lo13, hi13, lo, hi);
#pragma unroll
        for (int k = 0; k < 4; ++k) { lo02[k] = lo[k]; hi02[k] = hi[k]; }
    }
    template <int SY, int SX>
    __device__ __forceinline__ uint32_t sum() const {
        const uint32_t word = (SX & 1) ? hi02[SY] : lo02[SY];
        return (SX & 2) ? (word >> 16) : (word & 0xFFFFu);
    }
};

// rotate back + sum the four rotations, remove the +128 bias, divide / round-half-even / clip, and
// either store (planar / generic) or hand the packed rows to the RGB interleave.
template <int U, int OUT, class Acc>
__device__ __forceinline__ void finish_channel(const StageArgs &a, Acc &acc, int n, int c, int y, int x,
                                               uint32_t (&o)[U]) {
    const int unbias = 128 * kQ * 4 * a.M - a.bias_num;
    acc.finalize();
    static_for<0, U>([&](auto SY) {
        constexpr int sy = SY;
        uint32_t packed = 0;
        if constexpr (U == 4 && OUT != kOutGeneric) {
            const int k0 = (int)acc.template sum<sy, 0>() - unbias, k1 = (int)acc.template sum<sy, 1>() - unbias;
            const int k2 = (int)acc.template sum<sy, 2>() - unbias, k3 = (int)acc.template sum<sy, 3>() - unbias;
#if MULUT_ABLATE == 5   /* timing-only: no divide / round / clip */
            packed = (uint32_t)(k0 ^ k1 ^ k2 ^ k3);
#else
            if (a.use_f32)   // wave-uniform
                packed = rhe_pack4_f32(k0, k1, k2, k3, a.inv_d);
            else
                packed = rhe_clip_u8(k0, a.div) | (rhe_clip_u8(k1, a.div) << 8) | (rhe_clip_u8(k2, a.div) << 16) |
                         (rhe_clip_u8(k3, a.div) << 24);
#endif
        } else {
            static_for<0, U>([&](auto SX) {
                constexpr int sx = SX;
                const uint32_t v = rhe_clip_u8((int)acc.template sum<sy, sx>() - unbias, a.div);
                if constexpr (OUT == kOutGeneric) {
                    *const_cast<uint8_t *>(view_addr(a.out, n, c, y * U + sy, x * U + sx)) = (uint8_t)v;
                } else {
                    packed |= v << (8 * sx);
                }
            });
        }
        o[sy] = packed;
        if constexpr (OUT == kOutPlanarU4) {
            *(uint32_t *)const_cast<uint8_t *>(view_addr(a.out, n, c, y * U + sy, x * U)) = packed;
        }
    });
}

// one packed output row (4 bytes) of a finalized u == 4 accumulator
template <int SY, class Acc>
__device__ __forceinline__ uint32_t finish_row4(const StageArgs &a, const Acc &acc) {
    const int unbias = 128 * kQ * 4 * a.M - a.bias_num;
    const int k0 = (int)acc.template sum<SY, 0>() - unbias, k1 = (int)acc.template sum<SY, 1>() - unbias;
    const int k2 = (int)acc.template sum<SY, 2>() - unbias, k3 = (int)acc.template sum<SY, 3>() - unbias;
    if (a.use_f32) return rhe_pack4_f32(k0, k1, k2, k3, a.inv_d);   // wave-uniform
    return rhe_clip_u8(k0, a.div) | (rhe_clip_u8(k1, a.div) << 8) | (rhe_clip_u8(k2, a.div) << 16) |
           (rhe_clip_u8(k3, a.div) << 24);
}

// RGB epilogue row by row: only three packed rows are live at a time (finishing whole channels first
// parked 4-12 dwords per pixel in scratch, i.e. extra HBM writes)
template <class Acc>
__device__ __forceinline__ void finish_store_rgb4(const StageArgs &a, Acc &accR, Acc &accG, Acc &accB,
                                                  int n, int y, int x) {
    accR.finalize();
    accG.finalize();
    accB.finalize();
    static_for<0, 4>([&](auto SY) {
        constexpr int sy = SY;
        uint32_t w0, w1, w2;
        interleave_rgb4(finish_row4<sy>(a, accR), finish_row4<sy>(a, accG), finish_row4<sy>(a, accB), w0, w1, w2);
        uint32_t *dst = (uint32_t *)const_cast<uint8_t *>(view_addr(a.out, n, 0, y * 4 + sy, x * 4));
        dst[0] = w0;
        dst[1] = w1;
        dst[2] = w2;
    });
}

template <int U>
__device__ __forceinline__ void store_rgb(const StageArgs &a, int n, int y, int x, const uint32_t (&oR)[U],
                                          const uint32_t (&oG)[U], const uint32_t (&oB)[U]) {
#pragma unroll
    for (int sy = 0; sy < U; ++sy) {
        uint32_t w0, w1, w2;
        interleave_rgb4(oR[sy], oG[sy], oB[sy], w0, w1, w2);
        uint32_t *dst = (uint32_t *)const_cast<uint8_t *>(view_addr(a.out, n, 0, y * U + sy, x * U));
        dst[0] = w0;
        dst[1] = w1;
        dst[2] = w2;
    }
}

// keeps the channel's packed rows in named registers (c is wave-uniform -> scalar branches)
template <int U>
__device__ __forceinline__ void keep_rgb(int c, const uint32_t (&o)[U], uint32_t (&oR)[U], uint32_t (&oG)[U],
                                         uint32_t (&oB)[U]) {
    if (c == 0) {
#pragma unroll
        for (int sy = 0; sy < U; ++sy) oR[sy] = o[sy];
    } else if (c == 1) {
#pragma unroll
        for (int sy = 0; sy < U; ++sy) oG[sy] = o[sy];
    } else {
#pragma unroll
        for (int sy = 0; sy < U; ++sy) oB[sy] = o[sy];
    }
}

// one pass against the full table in global memory
template <int U, int R, class Acc>
__device__ __forceinline__ void pass_global(const void *lut, int va, int vb, int vc, int vd, const StageArgs &a,
                                            Acc &acc) {
    constexpr int RW = row_dwords(U);
    int idx[5], w[5];
    simplex4(va, vb, vc, vd, idx, w);
    uint32_t row[5][RW];
#if MULUT_ABLATE == 1
#pragma unroll
    for (int j = 0; j < 5; ++j) idx[j] &= (a.N >> 30);
#endif
#if MULUT_ABLATE == 2
#pragma unroll
    for (int j = 0; j < 5; ++j)
#pragma unroll
        for (int k = 0; k < RW; ++k) row[j][k] = (uint32_t)idx[j] + k;
#else
#pragma unroll
    for (int j = 0; j < 5; ++j) load_row<U>(lut, idx[j], row[j]);
#endif
#pragma unroll
    for (int j = 0; j < 5; ++j) acc.template fma<R>(row[j], (uint32_t)w[j]);
}

// HY = false: one block per TW x TH tile.
// HY = true (hybrid launch): four blocks per 64x16 verdict tile (its 2x2 sub-tiles); a block exits at once
// unless the statistic marked the tile for this kernel (an empty block costs ~0.3 us of one CU).
template <int U, int OUT, int TW, int TH, bool HY, bool WIDE = false>
__global__ void __launch_bounds__(TW *TH, 4) stage_up_kernel(StageArgs a) {
    constexpr int PW = TW + 2 * kHalo, PH = TH + 2 * kHalo;
    constexpr int NT = TW * TH;
    static_assert(!HY || (TW == 32 && TH == 8), "hybrid sub-tiling assumes 2x2 sub-tiles of 32x8 in a 64x16 tile");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    int n, y0, x0;
    const int id = HY ? xcd_remap(blockIdx.x >> 2, gridDim.x >> 2) : xcd_remap(blockIdx.x, gridDim.x);
    const int lt = threadIdx.x;
    const int sub = HY ? (blockIdx.x & 3) : 0;
    uint8_t *s_img = smem;
    if constexpr (HY) {
        if ((int)a.verdict[id] != a.verdict_take) return;              // block-uniform
        int b = id;
        const int vx = b % a.vt_x;
        b /= a.vt_x;
        const int vy = b % a.vt_y;
        n = b / a.vt_y;
        y0 = a.oy0 + vy * 16 + (sub >> 1) * TH;
        x0 = vx * 64 + (sub & 1) * TW;
    } else {
        decode_tile(a, id, n, y0, x0, TW, TH);
    }
    {   // this group's tile (load_tile, strided by the group's NT threads)
        const int total = a.C * PH * PW;
        const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
        for (int i = lt; i < total; i += NT) {
            const int px = i % PW, py = (i / PW) % PH, c = i / (PW * PH);
            const int gy = imin(imax(y0 + py - kHalo, ylo), yhi);
            const int gx = imin(imax(x0 + px - kHalo, 0), a.W - 1);
            s_img[i] = *view_addr(a.in, n, c, gy, gx);
        }
    }
    __syncthreads();

    const int tx = lt % TW, ty = lt / TW;
    const int y = y0 + ty, x = x0 + tx;
    if (y >= a.oy1 || x >= a.W) return;

    uint32_t oR[U], oG[U], oB[U];
    for (int c = 0; c < a.C; ++c) {
        const uint8_t *ctr = s_img + c * (PH * PW) + (ty + kHalo) * PW + (tx + kHalo);
        const int va = ctr[0];
        RotAcc<U, (U == 4) && !WIDE> acc;
        acc.clear();
        for (int mv = 0; mv < a.M; ++mv) {
            // the mode index is wave-uniform: pin it to an SGPR so the per-mode kernel arguments
            // (table pointer, pattern offsets) are fetched with scalar loads, not per-lane VMEM
            const int m = __builtin_amdgcn_readfirstlane(mv);
            const void *lut = a.lut[m];
            const int di0 = a.di[m][0], di1 = a.di[m][1], di2 = a.di[m][2];
            const int dj0 = a.dj[m][0], dj1 = a.dj[m][1], dj2 = a.dj[m][2];
            static_for<0, 4>([&](auto R) {
                constexpr int r = R;
                int dy, dx, v0, v1, v2;
                sample_offset(r, di0, dj0, dy, dx); v0 = ctr[dy * PW + dx];
                sample_offset(r, di1, dj1, dy, dx); v1 = ctr[dy * PW + dx];
                sample_offset(r, di2, dj2, dy, dx); v2 = ctr[dy * PW + dx];
                pass_global<U, r>(lut, va, v0, v1, v2, a, acc);
            });
        }
        uint32_t o[U];
        finish_channel<U, OUT>(a, acc, n, c, y, x, o);
        if constexpr (OUT == kOutPackedRGBU4) keep_rgb<U>(c, o, oR, oG, oB);
    }
    if constexpr (OUT == kOutPackedRGBU4) store_rgb<U>(a, n, y, x, oR, oG, oB);
}

constexpr int K2_TW = 32, K2_TH = 8;
void stage_up_tile(int &tw, int &th) { tw = K2_TW; th = K2_TH; }

const char *stage_up_name(int u, int out_mode) {
    (void)u;
    switch (out_mode) {
        case kOutPlanarU4: return "stage_up_kernel<4,planar>";
        case kOutPackedRGBU4: return "stage_up_kernel<4,rgb>";
        default: return "stage_up_kernel<generic>";
    }
}

template <int U, int OUT>
static hipError_t launch_up(const StageArgs &a, hipStream_t st) {
    const size_t tile_bytes = ((3 * (size_t)(K2_TH + 2 * kHalo) * (K2_TW + 2 * kHalo) + 15) / 16) * 16;
    if (a.verdict_take >= 0) {
        if constexpr (U == 4) {
            const long long nb = 4LL * a.N * a.vt_x * a.vt_y;
            if (nb <= 0 || nb > 0x7fffffffLL) return hipErrorInvalidValue;
            hipLaunchKernelGGL((stage_up_kernel<U, OUT, K2_TW, K2_TH, true>), dim3((unsigned)nb), dim3(K2_TW * K2_TH),
                               tile_bytes, st, a);
            return hipGetLastError();
        } else {
            return hipErrorInvalidValue;
        }
    }
    const long long nb = (long long)a.N * a.tiles_x * a.tiles_y;
    if (nb <= 0 || nb > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((stage_up_kernel<U, OUT, K2_TW, K2_TH, false>), dim3((unsigned)nb), dim3(K2_TW * K2_TH), tile_bytes, st, a);
    return hipGetLastError();
}

hipError_t launch_stage_up(const StageArgs &a, int u, int out_mode, hipStream_t st) {
    if (a.C > 3) return hipErrorInvalidValue;
    if (u == 4 && out_mode == kOutPlanarU4) return launch_up<4, kOutPlanarU4>(a, st);
    if (u == 4 && out_mode == kOutPackedRGBU4 && a.C == 3) return launch_up<4, kOutPackedRGBU4>(a, st);
    switch (u) {
        case 2: return launch_up<2, kOutGeneric>(a, st);
        case 3: return launch_up<3, kOutGeneric>(a, st);
        case 4: return launch_up<4, kOutGeneric>(a, st);
        default: return hipErrorInvalidValue;
    }
}

// u == 4 with more than four modes: per-rotation accumulators (the merged ones would overflow their 16-bit fields)
hipError_t launch_stage_up_wide4(const StageArgs &a, hipStream_t st) {
    if (a.C > 3 || a.verdict_take >= 0) return hipErrorInvalidValue;
    const size_t tile_bytes = ((3 * (size_t)(K2_TH + 2 * kHalo) * (K2_TW + 2 * kHalo) + 15) / 16) * 16;
    const long long nb = (long long)a.N * a.tiles_x * a.tiles_y;
    if (nb <= 0 || nb > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((stage_up_kernel<4, kOutGeneric, K2_TW, K2_TH, false, true>), dim3((unsigned)nb), dim3(K2_TW * K2_TH), tile_bytes, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// K2-band: persistent workgroups, band tables resident in LDS
// LDS: [ band of mode 0 | band of mode 1 | band of mode 2 : 34000 B each ][ image tile ]
// ------------------------------------------------------------------------------------------
constexpr int kBandBytes = kBandRows * 16;
static_assert(kBandBytes % 16 == 0, "band image must keep 16-byte alignment");

// one pass of a pair: low (HALF == 0) or high (HALF == 1) 16-bit half of the packed results
template <int R, int HALF>
__device__ __forceinline__ void pass_band(const uint8_t *band, const void *lut, const BandPair &bp, int va, int vb,
                                          int vc, int vd, const StageArgs &a, RotAcc<4> &acc) {
    const uint32_t t = HALF ? (bp.t_band >> 16) : (bp.t_band & 0xFFFFu);
    if (t <= 32u) {
        uint32_t row[5][4];
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const uint32_t off = HALF ? (bp.addr[j] >> 16) : (bp.addr[j] & 0xFFFFu);
            const uint4 v = *(const uint4 *)(band + off);
            row[j][0] = v.x; row[j][1] = v.y; row[j][2] = v.z; row[j][3] = v.w;
        }
#pragma unroll
        for (int j = 0; j < 5; ++j) acc.template fma_pk<R, HALF>(row[j], bp.w[j]);
    } else {
        pass_global<4, R>(lut, va, vb, vc, vd, a, acc);
    }
}

template <int OUT, int TW, int TH>
__global__ void __launch_bounds__(TW *TH) stage_band_kernel(StageArgs a, BandArgs b) {
    constexpr int PW = TW + 2 * kHalo, PH = TH + 2 * kHalo;
    constexpr int NT = TW * TH;
    constexpr int U = 4;
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *s_band = smem;
    // per-thread staging of the finished channels' packed rows (RGB path): [c][tid] x 16 B.  Keeping
    // them in "registers" across the runtime channel loop made the compiler spill them to scratch,
    // which doubled the kernel's HBM write traffic (profiles/r01_v2_pmc_*).
    uint4 *s_out = (uint4 *)(smem + a.M * kBandBytes);
    uint8_t *s_img = smem + a.M * kBandBytes + (OUT == kOutPackedRGBU4 ? 3 * NT * 16 : 0);

    for (int m = 0; m < a.M; ++m) {
        const uint4 *src = (const uint4 *)b.band[m];
        uint4 *dst = (uint4 *)(s_band + m * kBandBytes);
        for (int i = threadIdx.x; i < kBandBytes / 16; i += NT) dst[i] = src[i];
    }
    const int tx = threadIdx.x % TW, ty = threadIdx.x / TW;
    const int ntiles = a.N * a.tiles_x * a.tiles_y;
    // persistent workgroups; XCD x (= blockIdx % 8) walks its own contiguous eighth of the tiles
    const int G = gridDim.x;
    const bool by_xcd = (G & 7) == 0;
    const int per = (ntiles + 7) >> 3;
    const int first = by_xcd ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int last = by_xcd ? imin(((int)(blockIdx.x & 7) + 1) * per, ntiles) : ntiles;
    const int step = by_xcd ? (G >> 3) : G;
    for (int tile = first; tile < last; tile += step) {
        int n, y0, x0;
        decode_tile(a, tile, n, y0, x0, TW, TH);
        __syncthreads();  // band staged (first trip) / everyone done reading the previous tile
        load_tile<TW, TH, NT>(a, n, y0, x0, s_img);
        __syncthreads();
        const int y = y0 + ty, x = x0 + tx;
        if (y >= a.oy1 || x >= a.W) continue;   // no barrier below this point inside the trip

        for (int c = 0; c < a.C; ++c) {
            const uint8_t *ctr = s_img + c * (PH * PW) + (ty + kHalo) * PW + (tx + kHalo);
            const int va = ctr[0];
            RotAcc<4> acc;
            acc.clear();
            for (int mv = 0; mv < a.M; ++mv) {
                const int m = __builtin_amdgcn_readfirstlane(mv);   // SGPR: scalar loads of the per-mode arguments
                const uint8_t *band = s_band + m * kBandBytes;
                const void *lut = a.lut[m];
                const int di0 = a.di[m][0], di1 = a.di[m][1], di2 = a.di[m][2];
                const int dj0 = a.dj[m][0], dj1 = a.dj[m][1], dj2 = a.dj[m][2];
                static_for<0, 2>([&](auto P) {
                    constexpr int r = P;          // pair (r, r + 2): opposite sampling offsets
                    int dy, dx;
                    sample_offset(r, di0, dj0, dy, dx); const int o0 = dy * PW + dx;
                    sample_offset(r, di1, dj1, dy, dx); const int o1 = dy * PW + dx;
                    sample_offset(r, di2, dj2, dy, dx); const int o2 = dy * PW + dx;
#if MULUT_ABLATE == 8   /* timing-only: no pixel reads, index math hoistable out of the mode loop */
                    const int b0 = va, b1 = va, c0 = va, c1 = va, d0 = va, d1 = va;
#else
                    const int b0 = ctr[o0], b1 = ctr[-o0], c0 = ctr[o1], c1 = ctr[-o1], d0 = ctr[o2], d1 = ctr[-o2];
#endif
                    BandPair bp;
                    simplex4_band_pair((uint32_t)va, (uint32_t)b0 | ((uint32_t)b1 << 16), (uint32_t)c0 | ((uint32_t)c1 << 16),
                                       (uint32_t)d0 | ((uint32_t)d1 << 16), bp);
                    // both halves <= 32 in every lane (wave-uniform): straight-line band path, the ten row
                    // reads of the two passes issue together and overlap the first pass's MACs
                    const bool in_both = ((bp.t_band & 0xFFFFu) <= 32u) & ((bp.t_band >> 16) <= 32u);
                    if (__all(in_both)) {
                        uint32_t rowA[5][4], rowB[5][4];
#if MULUT_ABLATE == 7   /* timing-only: no LDS row gathers */
#pragma unroll
                        for (int j = 0; j < 5; ++j)
#pragma unroll
                            for (int k = 0; k < 4; ++k) { rowA[j][k] = bp.addr[j] + k; rowB[j][k] = bp.addr[j] ^ k; }
#else
#pragma unroll
                        for (int j = 0; j < 5; ++j) {
                            const uint4 v = *(const uint4 *)(band + (bp.addr[j] & 0xFFFFu));
                            rowA[j][0] = v.x; rowA[j][1] = v.y; rowA[j][2] = v.z; rowA[j][3] = v.w;
                        }
#pragma unroll
                        for (int j = 0; j < 5; ++j) {
                            const uint4 v = *(const uint4 *)(band + (bp.addr[j] >> 16));
                            rowB[j][0] = v.x; rowB[j][1] = v.y; rowB[j][2] = v.z; rowB[j][3] = v.w;
                        }
#endif
#if MULUT_ABLATE == 6   /* timing-only: rows folded with one xor each instead of the SWAR MAC */
#pragma unroll
                        for (int j = 0; j < 5; ++j)
#pragma unroll
                            for (int k = 0; k < 4; ++k) { acc.lo02[k] ^= rowA[j][k] + bp.w[j]; acc.lo13[k] ^= rowB[j][k] + bp.w[j]; }
#else
#pragma unroll
                        for (int j = 0; j < 5; ++j) acc.template fma_pk<r, 0>(rowA[j], bp.w[j]);
#pragma unroll
                        for (int j = 0; j < 5; ++j) acc.template fma_pk<r + 2, 1>(rowB[j], bp.w[j]);
#endif
                    } else {
                        pass_band<r, 0>(band, lut, bp, va, b0, c0, d0, a, acc);
                        pass_band<r + 2, 1>(band, lut, bp, va, b1, c1, d1, a, acc);
                    }
                });
            }
            uint32_t o[U];
            finish_channel<U, OUT>(a, acc, n, c, y, x, o);
            if constexpr (OUT == kOutPackedRGBU4) s_out[c * NT + threadIdx.x] = make_uint4(o[0], o[1], o[2], o[3]);
        }
        if constexpr (OUT == kOutPackedRGBU4) {
            const uint4 R = s_out[threadIdx.x], Gc = s_out[NT + threadIdx.x], B = s_out[2 * NT + threadIdx.x];
            const uint32_t oR[4] = {R.x, R.y, R.z, R.w}, oG[4] = {Gc.x, Gc.y, Gc.z, Gc.w}, oB[4] = {B.x, B.y, B.z, B.w};
            store_rgb<U>(a, n, y, x, oR, oG, oB);
        }
    }
}

constexpr int KB_TW = 64, KB_TH = 16;
void stage_band_tile(int &tw, int &th) { tw = KB_TW; th = KB_TH; }

hipError_t launch_tile_stat(const StageArgs &a, uint32_t *verdict, uint32_t max_oob_per_1024, hipStream_t st, uint16_t *thist, uint32_t *any) {
    const long long nb = (long long)a.N * a.tiles_x * a.tiles_y;   // a.tiles_* must be the 64x16 tiling
    if (nb <= 0 || nb > 0x7fffffffLL || a.C > 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL((tile_stat_kernel<KB_TW, KB_TH>), dim3((unsigned)nb), dim3(256), 0, st, a, verdict, max_oob_per_1024, thist, any);
    return hipGetLastError();
}
const char *stage_band_name(int out_mode) {
    return out_mode == kOutPackedRGBU4 ? "stage_band_kernel<rgb>" : out_mode == kOutPlanarU4 ? "stage_band_kernel<planar>"
                                                                                               : "stage_band_kernel<generic>";
}

template <int OUT>
static hipError_t launch_band_t(const StageArgs &a, const BandArgs &b, int num_cus, hipStream_t st) {
    auto kern = stage_band_kernel<OUT, KB_TW, KB_TH>;
    const size_t lds = (size_t)a.M * kBandBytes + (OUT == kOutPackedRGBU4 ? 3 * KB_TW * KB_TH * 16 : 0) +
                       (size_t)a.C * (KB_TH + 2 * kHalo) * (KB_TW + 2 * kHalo);
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    const long long ntiles = (long long)a.N * a.tiles_x * a.tiles_y;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)(ntiles < num_cus ? ntiles : num_cus);   // one persistent workgroup per CU
    hipLaunchKernelGGL(kern, dim3(grid), dim3(KB_TW * KB_TH), lds, st, a, b);
    return hipGetLastError();
}

hipError_t launch_stage_band(const StageArgs &a, const BandArgs &b, int out_mode, int num_cus, hipStream_t st) {
    if (a.C > 3 || a.M > 3) return hipErrorInvalidValue;
    if (out_mode == kOutPlanarU4) return launch_band_t<kOutPlanarU4>(a, b, num_cus, st);
    if (out_mode == kOutPackedRGBU4 && a.C == 3) return launch_band_t<kOutPackedRGBU4>(a, b, num_cus, st);
    return launch_band_t<kOutGeneric>(a, b, num_cus, st);
}

// ------------------------------------------------------------------------------------------
// K2-band-x: as stage_band_kernel, but the band rows live in LDS EXPANDED to 16-bit fields, so the
// MAC is 8 v_pk_mad_u16 per row with no unpack.  A band is two planes of 16-byte rows -- LO
// (elements 4k | 4k+2 << 16) and HI (4k+1 | 4k+3 << 16) -- so both reads of a row use the compact
// band offset (the second with an immediate) and bank behaviour equals the compact band's.
// 68 KB per mode: only the active mode is resident, the mode loop is outermost inside a tile and
// the accumulators of all channels of a pixel (3 x 16 VGPRs) stay in registers across it.  The next
// mode's band is brought in by LDS-DMA (global_load_lds_dwordx4) into the other buffer while the
// current one is being used.
// LDS: [ band buffer 0 : 69632 B ][ band buffer 1 : 69632 B ][ image tile 0 ][ image tile 1 ]
// ------------------------------------------------------------------------------------------
constexpr int kPlaneBytes = ((kBandRows * 16 + 1023) / 1024) * 1024;   // 34816: one LDS-DMA piece is 1 KiB
constexpr int kBandXBytes = 2 * kPlaneBytes;                          // 69632

// asynchronous global -> LDS copy of one expanded band; every wave moves whole 1-KiB pieces
template <int NT>
__device__ __forceinline__ void band_dma(const uint8_t *gsrc, uint8_t *lds_dst) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int piece = wave; piece < kBandXBytes / 1024; piece += NT / 64) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc + piece * 1024 + lane * 16),
                                         (__attribute__((address_space(3))) void *)(lds_dst + piece * 1024), 16, 0, 0);
    }
}

template <int R, int HALF>
__device__ __forceinline__ void rows_x(const uint8_t *band, const BandPair &bp, RotAcc<4> &acc) {
    // depth-1 software pipeline over the five rows: row j+1 is in flight while row j is accumulated
    // (all five at once would need 40 VGPRs next to the 48 accumulators of the three channels)
    uint4 c0, c1, n0, n1;
    auto ld = [&](int j, uint4 &v0, uint4 &v1) {
        const uint32_t off = HALF ? (bp.addr[j] >> 16) : (bp.addr[j] & 0xFFFFu);
        v0 = *(const uint4 *)(band + off);
        v1 = *(const uint4 *)(band + off + kPlaneBytes);
    };
#if MULUT_ABLATE == 7   /* timing-only: no LDS row gathers */
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        const uint32_t rlo[4] = {bp.addr[j], bp.addr[j] + 1, bp.addr[j] + 2, bp.addr[j] + 3}, rhi[4] = {bp.addr[j] ^ 1, bp.addr[j] ^ 2, bp.addr[j] ^ 3, bp.addr[j] ^ 4};
        acc.template fma_x<R, HALF>(rlo, rhi, bp.w[j]);
    }
    (void)band; (void)c0; (void)c1; (void)n0; (void)n1; (void)ld;
#else
    ld(0, c0, c1);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        if (j < 4) ld(j + 1, n0, n1);
        const uint32_t rlo[4] = {c0.x, c0.y, c0.z, c0.w}, rhi[4] = {c1.x, c1.y, c1.z, c1.w};
#if MULUT_ABLATE == 6   /* timing-only: one xor per dword instead of the MAC */
#pragma unroll
        for (int k = 0; k < 4; ++k) { acc.lo02[k] ^= rlo[k] + bp.w[j]; acc.hi13[k] ^= rhi[k] + bp.w[j]; }
#else
        acc.template fma_x<R, HALF>(rlo, rhi, bp.w[j]);
#endif
        c0 = n0;
        c1 = n1;
    }
#endif
}

// rare out-of-band pass inside the expanded-band kernel: one row in flight at a time, so that this path
// does not set the kernel's register allocation (latency is irrelevant here)
template <int R>
__device__ __forceinline__ void pass_global_lean(const void *lut, int va, int vb, int vc, int vd, RotAcc<4> &acc) {
    int idx[5], w[5];
    simplex4(va, vb, vc, vd, idx, w);
#pragma unroll 1
    for (int j = 0; j < 5; ++j) {
        int ij = idx[0], wj = w[0];
        if (j == 1) { ij = idx[1]; wj = w[1]; }
        if (j == 2) { ij = idx[2]; wj = w[2]; }
        if (j == 3) { ij = idx[3]; wj = w[3]; }
        if (j == 4) { ij = idx[4]; wj = w[4]; }
        uint32_t row[4];
        load_row<4>(lut, ij, row);
        acc.template fma<R>(row, (uint32_t)wj);
    }
}

#if defined(MULUT_PROFILE)
struct WaveProf {
    unsigned long long t_slow, n_slow, t_fast, n_fast;
};
__device__ WaveProf *g_prof_dummy;
#define PROF_ARG , WaveProf &prof
#define PROF_PASS , prof
#else
#define PROF_ARG
#define PROF_PASS
#endif
template <int R>
__device__ __forceinline__ void pair_x(const uint8_t *band, const void *lut, const uint16_t *ctr, int o0, int o1, int o2,
                                       RotAcc<4> &acc PROF_ARG) {
    // the tile holds pixel codes: a key is one v_and_or of a packed pair, the 16*h term one v_and
    const uint32_t ca = ctr[0];
#if MULUT_ABLATE == 8   /* timing-only: no neighbour reads, index math hoistable */
    const uint32_t pb = ca * 0x10001u, pc = pb, pd = pb;
    (void)o0; (void)o1; (void)o2;
#else
    // (ds_read_u16_d16_hi cannot be used to fill the high half directly: with SRAM-ECC on, as on this part,
    // d16 loads zero the other half of the destination)
    const uint32_t pb = ctr[o0] | ((uint32_t)ctr[-o0] << 16);
    const uint32_t pc = ctr[o1] | ((uint32_t)ctr[-o1] << 16);
    const uint32_t pd = ctr[o2] | ((uint32_t)ctr[-o2] << 16);
#endif
    BandPair bp;
    simplex4_band_pair_code(ca, pb, pc, pd, bp);
#if defined(MULUT_PROFILE)
    const unsigned long long pq0 = __builtin_amdgcn_s_memtime();
    const bool pq_fast = __all(bp.t_band == 0u);
    struct PQ {
        WaveProf &p; unsigned long long t0; bool fast;
        __device__ ~PQ() {
            const unsigned long long d = __builtin_amdgcn_s_memtime() - t0;
            if (fast) { p.t_fast += d; p.n_fast += 1; } else { p.t_slow += d; p.n_slow += 1; }
        }
    } pq{prof, pq0, pq_fast};
#endif
    if (__all(bp.t_band == 0u)) {          // both passes in band in every lane
        rows_x<R, 0>(band, bp, acc);
        rows_x<R + 2, 1>(band, bp, acc);
    } else {
        const int va = pixel_value(ca);
        if ((bp.t_band & 0xFFFFu) == 0u) rows_x<R, 0>(band, bp, acc);
        else pass_global_lean<R>(lut, va, pixel_value(pb & 0xFFFFu), pixel_value(pc & 0xFFFFu), pixel_value(pd & 0xFFFFu), acc);
        if ((bp.t_band >> 16) == 0u) rows_x<R + 2, 1>(band, bp, acc);
        else pass_global_lean<R + 2>(lut, va, pixel_value(pb >> 16), pixel_value(pc >> 16), pixel_value(pd >> 16), acc);
    }
}

template <int OUT, int TW, int TH>
__global__ void __launch_bounds__(TW *TH) stage_bandx_kernel(StageArgs a, BandArgs b) {
    constexpr int PW = TW + 2 * kHalo, PH = TH + 2 * kHalo;
    constexpr int NT = TW * TH;
    constexpr int U = 4;
    constexpr int kTileBytes = ((2 * 3 * PH * PW + 15) / 16) * 16;   // 16-bit pixel codes
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *s_band = smem;                       // two buffers of kBandXBytes
    uint8_t *s_tile = smem + 2 * kBandXBytes;     // two buffers of kTileBytes

    const int tx = threadIdx.x % TW, ty = threadIdx.x / TW;
    const int ntiles = a.N * a.tiles_x * a.tiles_y;
    const int G = gridDim.x;
    const bool by_xcd = (G & 7) == 0;
    const int per = (ntiles + 7) >> 3;
    const int first = by_xcd ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int last = by_xcd ? imin(((int)(blockIdx.x & 7) + 1) * per, ntiles) : ntiles;
    const int step = by_xcd ? (G >> 3) : G;

#if defined(MULUT_PROFILE)
    WaveProf prof = {0, 0, 0, 0};
    unsigned long long pt_bar = 0, pt_epi = 0, pt_load = 0;
    const unsigned long long pt_begin = __builtin_amdgcn_s_memtime();
#endif
    int phase = 0;   // counts (tile, mode) steps of this workgroup: band buffer = phase & 1
    if (first < last) band_dma<NT>((const uint8_t *)b.band[0], s_band);
    for (int tile = first, it = 0; tile < last; tile += step) {
        if (a.verdict_take >= 0 && (int)a.verdict[tile] != a.verdict_take) continue;   // hybrid: not a smooth tile
        int n, y0, x0;
        decode_tile(a, tile, n, y0, x0, TW, TH);
        uint16_t *s_img = (uint16_t *)(s_tile + (it & 1) * kTileBytes);
        ++it;
#if defined(MULUT_PROFILE)
        const unsigned long long pt0 = __builtin_amdgcn_s_memtime();
#endif
        load_tile_code<TW, TH, NT>(a, n, y0, x0, s_img);   // the buffer last read two tiles ago
#if defined(MULUT_PROFILE)
        pt_load += __builtin_amdgcn_s_memtime() - pt0;
#endif
        const int y = y0 + ty, x = x0 + tx;
        const bool valid = y < a.oy1 && x < a.W;
        const uint16_t *ctr = s_img + (ty + kHalo) * PW + (tx + kHalo);
        RotAcc<4> acc0, acc1, acc2;
        acc0.clear(); acc1.clear(); acc2.clear();
        for (int mv = 0; mv < a.M; ++mv, ++phase) {
            const int m = __builtin_amdgcn_readfirstlane(mv);
            // my DMA pieces of this phase's band have landed; after the barrier everyone's have, the image
            // tile is visible, and every wave has finished the previous phase (its band buffer is free)
#if defined(MULUT_PROFILE)
            const unsigned long long pt1 = __builtin_amdgcn_s_memtime();
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#if defined(MULUT_PROFILE)
            pt_bar += __builtin_amdgcn_s_memtime() - pt1;
#endif
            const uint8_t *band = s_band + (phase & 1) * kBandXBytes;
            {   // prefetch the next phase's band (next mode, or mode 0 of the next tile) into the other buffer
                const int mnext = mv + 1 < a.M ? mv + 1 : 0;
#if MULUT_ABLATE != 9   /* 9 = timing-only: bands never restaged */
                if (mv + 1 < a.M || tile + step < last)
#else
                if (false)
#endif
                    band_dma<NT>((const uint8_t *)b.band[__builtin_amdgcn_readfirstlane(mnext)],
                                 s_band + ((phase + 1) & 1) * kBandXBytes);
            }
            if (valid) {
                const void *lut = a.lut[m];
                // wave-uniform LDS offsets of keys b, c, d for rotations 0 and 1 (2 and 3 are their negatives):
                // pinned to SGPRs, they must not compete with the 48 accumulator VGPRs
                int dy, dx;
                sample_offset(0, a.di[m][0], a.dj[m][0], dy, dx); const int p0 = __builtin_amdgcn_readfirstlane(dy * PW + dx);
                sample_offset(0, a.di[m][1], a.dj[m][1], dy, dx); const int p1 = __builtin_amdgcn_readfirstlane(dy * PW + dx);
                sample_offset(0, a.di[m][2], a.dj[m][2], dy, dx); const int p2 = __builtin_amdgcn_readfirstlane(dy * PW + dx);
                sample_offset(1, a.di[m][0], a.dj[m][0], dy, dx); const int q0 = __builtin_amdgcn_readfirstlane(dy * PW + dx);
                sample_offset(1, a.di[m][1], a.dj[m][1], dy, dx); const int q1 = __builtin_amdgcn_readfirstlane(dy * PW + dx);
                sample_offset(1, a.di[m][2], a.dj[m][2], dy, dx); const int q2 = __builtin_amdgcn_readfirstlane(dy * PW + dx);
                pair_x<0>(band, lut, ctr, p0, p1, p2, acc0 PROF_PASS);
                pair_x<1>(band, lut, ctr, q0, q1, q2, acc0 PROF_PASS);
                if (a.C > 1) {
                    pair_x<0>(band, lut, ctr + PH * PW, p0, p1, p2, acc1 PROF_PASS);
                    pair_x<1>(band, lut, ctr + PH * PW, q0, q1, q2, acc1 PROF_PASS);
                }
                if (a.C > 2) {
                    pair_x<0>(band, lut, ctr + 2 * PH * PW, p0, p1, p2, acc2 PROF_PASS);
                    pair_x<1>(band, lut, ctr + 2 * PH * PW, q0, q1, q2, acc2 PROF_PASS);
                }
            }
        }
#if defined(MULUT_PROFILE)
        const unsigned long long pt3 = __builtin_amdgcn_s_memtime();
#endif
        if (valid) {
            if constexpr (OUT == kOutPackedRGBU4) {
                finish_store_rgb4(a, acc0, acc1, acc2, n, y, x);
            } else {
                uint32_t o[U];
                finish_channel<U, OUT>(a, acc0, n, 0, y, x, o);
                if (a.C > 1) finish_channel<U, OUT>(a, acc1, n, 1, y, x, o);
                if (a.C > 2) finish_channel<U, OUT>(a, acc2, n, 2, y, x, o);
            }
        }
#if defined(MULUT_PROFILE)
        pt_epi += __builtin_amdgcn_s_memtime() - pt3;
#endif
    }
#if defined(MULUT_PROFILE)
    if ((threadIdx.x & 63) == 0) {
        unsigned long long *dst = (unsigned long long *)(a.out.p + (long long)a.N * a.out.sN) + ((size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6)) * 8;
        dst[0] = __builtin_amdgcn_s_memtime() - pt_begin;
        dst[1] = pt_load; dst[2] = pt_bar; dst[3] = prof.t_fast; dst[4] = pt_epi; dst[5] = prof.n_slow; dst[6] = prof.t_slow; dst[7] = prof.n_fast;
    }
#endif
}

const char *stage_bandx_name(int out_mode) {
    return out_mode == kOutPackedRGBU4 ? "stage_bandx_kernel<rgb>" : out_mode == kOutPlanarU4 ? "stage_bandx_kernel<planar>"
                                                                                               : "stage_bandx_kernel<generic>";
}

template <int OUT>
static hipError_t launch_bandx_t(const StageArgs &a, const BandArgs &b, int num_cus, hipStream_t st) {
    auto kern = stage_bandx_kernel<OUT, KB_TW, KB_TH>;
    const size_t lds = 2 * (size_t)kBandXBytes + 2 * (size_t)(((2 * 3 * (KB_TH + 2 * kHalo) * (KB_TW + 2 * kHalo) + 15) / 16) * 16);
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    const long long ntiles = (long long)a.N * a.tiles_x * a.tiles_y;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)(ntiles < num_cus ? ntiles : num_cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(KB_TW * KB_TH), lds, st, a, b);
    return hipGetLastError();
}

hipError_t launch_stage_bandx(const StageArgs &a, const BandArgs &b, int out_mode, int num_cus, hipStream_t st) {
    if (a.C > 3 || a.M > 3) return hipErrorInvalidValue;
    if (out_mode == kOutPlanarU4) return launch_bandx_t<kOutPlanarU4>(a, b, num_cus, st);
    if (out_mode == kOutPackedRGBU4 && a.C == 3) return launch_bandx_t<kOutPackedRGBU4>(a, b, num_cus, st);
    return launch_bandx_t<kOutGeneric>(a, b, num_cus, st);
}


// ------------------------------------------------------------------------------------------
// K2-tube: final stage (u == 4, M <= 3) with the bands of ALL modes resident in LDS.
// The band is the "tube" of mulut_core.h (rows whose keys span <= 2 MSB steps: 1041 slots), expanded to
// 16-bit fields in two planes: 33,312 B per mode, 99,936 B for s, d and y together, so nothing is swapped
// and a tile needs no barrier except the one that publishes the next image tile.  With every band at hand
// the loops nest channel -> mode -> rotation pair: 16 accumulator VGPRs are live instead of 48, which
// leaves room for all ten row reads of a pass in flight and for a general path that does not spill.
// The pattern of a mode is a template parameter behind a scalar switch: every neighbour read is a
// ds_read_u16 with an immediate offset from one per-channel window address.
// LDS: [ band s | band d | band y : 33,312 B each ][ image tile 0 ][ image tile 1 ]  (pixel codes, 20 x 72 x 3 x 2 B each)
// ------------------------------------------------------------------------------------------
constexpr int kTubeHaloX = 4;      // the tile image starts 4 columns left of the tile: whole aligned dwords of the input row
constexpr int kTubeTileBytes = ((2 * 3 * (16 + 2 * kHalo) * (64 + 2 * kTubeHaloX) + 15) / 16) * 16;
constexpr int kTubeLdsBytes = 3 * kTubeBandBytes + 2 * kTubeTileBytes;
// the packed row offsets carry this bias so that (plane address - bias) fits ds_read's 16-bit immediate
__host__ __device__ constexpr int tube_bias(int pat) { return pat == 2 ? 36000 : pat == 1 ? 2048 : 0; }

// one row (LO + HI plane dwords) into the accumulators of rotation R, weight half HALF
template <int R, int HALF>
__device__ __forceinline__ void tube_mac_row(RotAcc<4> &acc, const uint4 &lo, const uint4 &hi, uint32_t wpk) {
    const uint32_t rlo[4] = {lo.x, lo.y, lo.z, lo.w}, rhi[4] = {hi.x, hi.y, hi.z, hi.w};
    acc.template mac_x<R, HALF>(rlo, rhi, wpk);
}


// the five rows of one pass: all ten reads issued, then accumulated in order.  The byte offsets are built unpacked (one
// SDWA add per row extracts the pass's half of the packed stride and adds it); row 4 (vertex 1111) sits a fixed 65
// slots after row 0: it shares row 0's address register and differs in the immediate only.
template <int R, int HALF, int IMM>
__device__ __forceinline__ void tube_rows(const uint8_t *smem, const TubePair &bp, RotAcc<4> &acc) {
    uint32_t a[4];
    a[0] = HALF ? (bp.base >> 16) : (bp.base & 0xFFFFu);
#pragma unroll
    for (int j = 0; j < 3; ++j) a[j + 1] = add_word<HALF>(a[j], bp.step[j]);
    uint4 lo[5], hi[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        constexpr int kRow4 = kTubeAll * 16;
        lo[j] = *(const uint4 *)(smem + a[j < 4 ? j : 0] + (IMM + (j < 4 ? 0 : kRow4)));
        hi[j] = *(const uint4 *)(smem + a[j < 4 ? j : 0] + (IMM + kTubePlaneBytes + (j < 4 ? 0 : kRow4)));
    }
#pragma unroll
    for (int j = 0; j < 5; ++j) tube_mac_row<R, HALF>(acc, lo[j], hi[j], bp.w[j]);
}

// rotations R and R + 2 of one site and mode.  win = LDS byte address of the site's 5x5 window corner; k0 / ha16 / base_a:
// the anchor's key, MSB term and slot term (+ this pattern's bias), hoisted by the caller.
template <int PAT, int R, int PW, bool FLAGGED>
__device__ __forceinline__ void tube_pair(const uint8_t *smem, uint32_t win, uint32_t k0, uint32_t ha16, uint32_t base_a, RotAcc<4> &acc, uint32_t &dirty) {
    constexpr int IMM = PAT * kTubeBandBytes - tube_bias(PAT);
    static_assert(IMM >= 0 && IMM + kTubeAll * 16 + kTubePlaneBytes <= 65535 && tube_bias(PAT) + kTubePlaneBytes <= 65536 && tube_bias(PAT) % 16 == 0,
                  "ds_read immediate / packed offset range");
    const uint16_t *w = (const uint16_t *)(smem + win);
    constexpr int yb = rot_dy(R, kPatDi[PAT][0], kPatDj[PAT][0]), xb = rot_dx(R, kPatDi[PAT][0], kPatDj[PAT][0]);
    constexpr int yc = rot_dy(R, kPatDi[PAT][1], kPatDj[PAT][1]), xc = rot_dx(R, kPatDi[PAT][1], kPatDj[PAT][1]);
    constexpr int yd = rot_dy(R, kPatDi[PAT][2], kPatDj[PAT][2]), xd = rot_dx(R, kPatDi[PAT][2], kPatDj[PAT][2]);
    // rotation R + 2 samples the opposite offsets
    const uint32_t pb = w[(2 + yb) * PW + 2 + xb] | ((uint32_t)w[(2 - yb) * PW + 2 - xb] << 16);
    const uint32_t pc = w[(2 + yc) * PW + 2 + xc] | ((uint32_t)w[(2 - yc) * PW + 2 - xc] << 16);
    const uint32_t pd = w[(2 + yd) * PW + 2 + xd] | ((uint32_t)w[(2 - yd) * PW + 2 - xd] << 16);
    TubePair bp;
    simplex4_tube_pair<FLAGGED>(k0, ha16, base_a, pb, pc, pd, bp);
    // A pass outside the tube still walks the band (any key combination maps to a slot inside it, so the reads stay
    // in range) and adds garbage; the site is marked -- by site_flag_kernel ahead of this launch (FLAGGED), else by the
    // per-pass test here -- and recomputed from the full table by stage_up_fix_kernel.
    if constexpr (!FLAGGED) dirty |= bp.t_oob;
    tube_rows<R, 0, IMM>(smem, bp, acc);
    tube_rows<R + 2, 1, IMM>(smem, bp, acc);
}

template <int PAT, int PW, bool FLAGGED>
__device__ __forceinline__ void tube_mode(const uint8_t *smem, uint32_t win, uint32_t k0, uint32_t ha16, uint32_t ha27, RotAcc<4> &acc, uint32_t &dirty) {
    const uint32_t base_a = ha27 + pk_dup((uint32_t)tube_bias(PAT));
    tube_pair<PAT, 0, PW, FLAGGED>(smem, win, k0, ha16, base_a, acc, dirty);
    tube_pair<PAT, 1, PW, FLAGGED>(smem, win, k0, ha16, base_a, acc, dirty);
}

// Epilogue of one channel straight from the pair accumulators: the block value at (sy, sx) is the field of
// element 4 sy + sx in the (0,2) accumulators plus the field of element (3 - sx) 4 + sy in the (1,3) ones (one
// 16-bit-select add each), then the fused cvt / fma / rndne / cvt_pk_u8.  o[sy] = the four bytes of block row sy.
template <int E>
__device__ __forceinline__ uint32_t tube_field(const uint32_t (&lo)[4], const uint32_t (&hi)[4]) {
    const uint32_t word = (E & 1) ? hi[E >> 2] : lo[E >> 2];
    return (E & 2) ? (word >> 16) : (word & 0xFFFFu);
}
__device__ __forceinline__ void tube_finish_rows(const StageArgs &a, RotAcc<4> &acc, uint32_t (&o)[4]) {
    if (a.use_fma) {      // wave-uniform
        static_for<0, 4>([&](auto SY) {
            constexpr int sy = SY;
            const uint32_t s0 = tube_field<4 * sy + 0>(acc.lo02, acc.hi02) + tube_field<12 + sy>(acc.lo13, acc.hi13);
            const uint32_t s1 = tube_field<4 * sy + 1>(acc.lo02, acc.hi02) + tube_field<8 + sy>(acc.lo13, acc.hi13);
            const uint32_t s2 = tube_field<4 * sy + 2>(acc.lo02, acc.hi02) + tube_field<4 + sy>(acc.lo13, acc.hi13);
            const uint32_t s3 = tube_field<4 * sy + 3>(acc.lo02, acc.hi02) + tube_field<0 + sy>(acc.lo13, acc.hi13);
            o[sy] = rhe_pack4_fma(s0, s1, s2, s3, a.inv_d, a.epi_c);
        });
    } else {
        acc.finalize();
        o[0] = finish_row4<0>(a, acc); o[1] = finish_row4<1>(a, acc);
        o[2] = finish_row4<2>(a, acc); o[3] = finish_row4<3>(a, acc);
    }
}

template <int OUT, int TW, int TH, bool FLAGGED>
__global__ void __launch_bounds__(TW *TH) stage_tube_kernel(StageArgs a, BandArgs b) {
    constexpr int PW = TW + 2 * kTubeHaloX, PH = TH + 2 * kHalo;     // tile image: columns x0-4 .. x0+TW+3, rows y0-2 .. y0+TH+1
    constexpr int NT = TW * TH;
    constexpr int DW = PW / 4, PER4 = (3 * PH * DW + NT - 1) / NT;    // aligned dwords per tile row / per thread (dword path)
    static_assert(((2 * 3 * PH * PW + 15) / 16) * 16 == kTubeTileBytes, "tile buffer size");
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];

    const int tx = threadIdx.x % TW, ty = threadIdx.x / TW;
    const int ntiles = a.N * a.tiles_x * a.tiles_y;
    const int G = gridDim.x;
    const bool by_xcd = (G & 7) == 0;
    const int per = (ntiles + 7) >> 3;
    const int first = by_xcd ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int last = by_xcd ? imin(((int)(blockIdx.x & 7) + 1) * per, ntiles) : ntiles;
    const int step = by_xcd ? (G >> 3) : G;
    // next tile of this workgroup at or after t that the verdict (if any) assigns to this kernel
    auto next_tile = [&](int t) {
        if (a.verdict_take >= 0)
            while (t < last && (int)a.verdict[t] != a.verdict_take) t += step;
        return t;
    };
    const int total = a.C * PH * PW;
    const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
    // Planar input whose rows start on dword boundaries (the pipeline's intermediate images): a tile is fetched as
    // aligned dwords, in flight while the previous tile is computed.  Columns left of 0 / right of W-1 replicate the
    // edge byte of the nearest valid dword.  Any other input takes the byte path at stash time (not prefetched).
    const bool dw_ok = a.in.sX == 1 && ((a.W | a.in.sY | a.in.sC) & 3) == 0 && (a.in.sN & 3) == 0 && (((uintptr_t)a.in.p) & 3) == 0;
    auto fetch = [&](int tile, uint32_t (&v)[PER4]) {
        if (!dw_ok) return;
        int n, y0, x0;
        decode_tile(a, tile, n, y0, x0, TW, TH);
#pragma unroll
        for (int k = 0; k < PER4; ++k) {
            const int i = (int)threadIdx.x + k * NT;
            const int q = i % DW, py = (i / DW) % PH, c = imin(i / (DW * PH), a.C - 1);
            const int gy = imin(imax(y0 + py - kHalo, ylo), yhi);
            const int gx = imin(imax(x0 - kTubeHaloX + 4 * q, 0), a.W - 4);
            v[k] = *(const uint32_t *)view_addr(a.in, n, c, gy, gx);
        }
    };
    auto stash = [&](int tile, int buf, const uint32_t (&v)[PER4]) {
        int n, y0, x0;
        decode_tile(a, tile, n, y0, x0, TW, TH);
        uint8_t *dst = smem + 3 * kTubeBandBytes + buf * kTubeTileBytes;
        if (dw_ok) {
            // the index math is redone from an opaque copy of the thread id: nothing but the fetched dwords themselves
            // may stay live across the tile's computation (the compiler would otherwise park shared terms in scratch)
            int tid = (int)threadIdx.x;
            asm volatile("" : "+v"(tid));
#pragma unroll
            for (int k = 0; k < PER4; ++k) {
                const int i = tid + k * NT;
                if (i < a.C * PH * DW) {
                    const int gx = x0 - kTubeHaloX + 4 * (i % DW);
                    // bytes (b0,b1) / (b2,b3) into 16-bit lanes; a dword clamped at an image edge replicates the edge byte
                    const uint32_t sel_lo = gx < 0 ? 0x0C000C00u : gx > a.W - 4 ? 0x0C030C03u : 0x0C010C00u;
                    const uint32_t sel_hi = gx < 0 ? 0x0C000C00u : gx > a.W - 4 ? 0x0C030C03u : 0x0C030C02u;
                    const uint32_t lo = __builtin_amdgcn_perm(0u, v[k], sel_lo), hi = __builtin_amdgcn_perm(0u, v[k], sel_hi);
                    // (b, 0) * 0x1001 = f << 12 | b per 16-bit lane; keeping the two nibbles gives pixel_code(b)
                    uint2 c2;
                    c2.x = pk_mad(lo, pk_dup(0x1001u), 0u) & 0xF0F0F0F0u;
                    c2.y = pk_mad(hi, pk_dup(0x1001u), 0u) & 0xF0F0F0F0u;
                    *(uint2 *)(dst + 8 * i) = c2;
                }
            }
        } else {
            for (int i = threadIdx.x; i < total; i += NT) {
                const int px = i % PW, py = (i / PW) % PH, c = i / (PW * PH);
                const int gy = imin(imax(y0 + py - kHalo, ylo), yhi);
                const int gx = imin(imax(x0 + px - kTubeHaloX, 0), a.W - 1);
                ((uint16_t *)dst)[i] = (uint16_t)pixel_code(*view_addr(a.in, n, c, gy, gx));
            }
        }
    };

    int tile = next_tile(first);
    if (tile >= last) return;              // workgroup-uniform
#if defined(MULUT_VARIANT_clk)
    const unsigned long long clk0 = __builtin_amdgcn_s_memtime();
#endif
    uint32_t pix[PER4];
    fetch(tile, pix);
    // bands: slot = pattern id of the mode (s, d, y); absent patterns are never read
    for (int m = 0; m < a.M; ++m) {
        const int pat = a.dj[m][0] == 2 ? 1 : a.di[m][0] == 1 ? 2 : 0;
        const uint4 *src = (const uint4 *)b.band[m];
        uint4 *dst = (uint4 *)(smem + pat * kTubeBandBytes);
        for (int i = threadIdx.x; i < kTubeBandBytes / 16; i += NT) dst[i] = src[i];
    }
    stash(tile, 0, pix);
    __syncthreads();

    for (int it = 0; tile < last; ++it) {
        const int nxt = next_tile(tile + step);
        if (nxt < last) fetch(nxt, pix);
        int n, y0, x0;
        decode_tile(a, tile, n, y0, x0, TW, TH);
        const int y = y0 + ty, x = x0 + tx;
        if (y < a.oy1 && x < a.W) {
            // LDS byte address of the 5x5 window corner (y-2, x-2) of this site, channel 0
            uint32_t win = (uint32_t)(3 * kTubeBandBytes + (it & 1) * kTubeTileBytes + 2 * (ty * PW + tx + kTubeHaloX - kHalo));
            uint32_t o0[4], o1[4], o2[4];     // packed output rows of the finished channels (RGB path)
#pragma unroll
            for (int k = 0; k < 4; ++k) o0[k] = o1[k] = o2[k] = 0;
            // per channel: != 0 when some pass of the sample may have left the tube (FLAGGED: bit c of site_flag_kernel's byte)
            const uint32_t sflags = FLAGGED ? (uint32_t)a.site_flags[((size_t)n * a.H + y) * a.W + x] : 0u;
            uint32_t dmask = FLAGGED ? sflags & 7u : 0u;      // bit c: channel c is dirty
#pragma clang loop unroll(disable)
            for (int c = 0; c < a.C; ++c, win += 2 * PH * PW) {
                uint32_t dirty = 0u;
                const uint32_t ca = *(const uint16_t *)(smem + win + 2 * (2 * PW + 2));
                const uint32_t k0 = tube_anchor_key(ca), ha16 = tube_anchor_h16(ca), ha27 = pk_mad(ha16, pk_dup(kTubeSA), 0u);
                RotAcc<4> acc;
                acc.clear();
                for (int mv = 0; mv < a.M; ++mv) {
                    const int m = __builtin_amdgcn_readfirstlane(mv);
                    const int pat = a.dj[m][0] == 2 ? 1 : a.di[m][0] == 1 ? 2 : 0;     // scalar
                    if (pat == 0) tube_mode<0, PW, FLAGGED>(smem, win, k0, ha16, ha27, acc, dirty);
                    else if (pat == 1) tube_mode<1, PW, FLAGGED>(smem, win, k0, ha16, ha27, acc, dirty);
                    else tube_mode<2, PW, FLAGGED>(smem, win, k0, ha16, ha27, acc, dirty);
                }
                if constexpr (OUT == kOutPackedRGBU4) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { o0[k] = o1[k]; o1[k] = o2[k]; }
                    tube_finish_rows(a, acc, o2);
                } else {
                    uint32_t o[4];
                    finish_channel<4, OUT>(a, acc, n, c, y, x, o);
                }
                if constexpr (!FLAGGED) dmask |= (dirty != 0u ? 1u : 0u) << c;
            }
            if constexpr (OUT == kOutPackedRGBU4) store_rgb<4>(a, n, y, x, o0, o1, o2);
            // dirty samples (pixel, channel) go on the fix-up list: one atomic per wave and channel (rare), compacted by lane rank
            if (__ballot(dmask != 0u) != 0ull) {
                const uint32_t pixel_id = (uint32_t)((n * a.H + y) * a.W + x);
                for (int c = 0; c < a.C; ++c) {
                    const bool d = ((dmask >> c) & 1u) != 0u;
                    const unsigned long long dm = __ballot(d);
                    if (dm == 0ull) continue;
                    const int lane = (int)(threadIdx.x & 63);
                    uint32_t at = 0;
                    if (lane == __ffsll((long long)dm) - 1) at = atomicAdd(a.fix_count, (uint32_t)__popcll(dm));
                    at = (uint32_t)__shfl((int)at, __ffsll((long long)dm) - 1);
                    if (d) a.fix_list[at + (uint32_t)__popcll(dm & ((1ull << lane) - 1ull))] = pixel_id | ((uint32_t)c << 30);
                }
            }
        }
        if (nxt < last) stash(nxt, (it + 1) & 1, pix);
        __syncthreads();     // next tile published; everyone is done reading the current one
        tile = nxt;
    }
#if defined(MULUT_VARIANT_clk)   /* probe build: shader-clock ticks this workgroup lived, into the first bytes of the output */
    if (blockIdx.x == 0 && threadIdx.x == 0) *(unsigned long long *)a.out.p = __builtin_amdgcn_s_memtime() - clk0;
#endif
}

// Fix-up of the tube kernel: every listed pixel (id = (n H + y) W + x) is recomputed, all channels and passes,
// with its rows taken from the full tables in global memory -- the arithmetic of stage_up_kernel, with the neighbours
// read straight from the stage input.  A fixed grid walks the list; its length is read from device memory, so the
// launch is unconditional (hipGraph-capturable) and costs a few microseconds when the list is empty.
template <int OUT>
__global__ void __launch_bounds__(256) stage_up_fix_kernel(StageArgs a) {
    // a thread's 5x5 window as five 8-byte rows (from column x - 2) in its own LDS slot: the 36 neighbour reads of a sample are
    // LDS byte reads then, not global ones (the kernel is bound by the texture path).  Border columns (and inputs that are
    // not planar) read the image directly, with edge replication.
    __shared__ uint2 s_win[5][256];
    const uint32_t count = *a.fix_count;
    const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
        // entry: pixel id (n H + y) W + x in the low 30 bits, channel in the top two (3 = every channel)
        const uint32_t ent = a.fix_list[i], id = ent & 0x3FFFFFFFu, only = ent >> 30;
        const int x = (int)(id % (uint32_t)a.W), y = (int)((id / (uint32_t)a.W) % (uint32_t)a.H), n = (int)(id / ((uint32_t)a.W * (uint32_t)a.H));
        for (int c = 0; c < a.C; ++c) {
            if (only != 3u && (uint32_t)c != only) continue;
            const bool inner = a.in.sX == 1 && x >= kSlabXLo && x < a.W - slab_x_hi(a);      // 8 bytes from x - 2 stay inside the row / the padding
            if (inner) {
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    uint2 v;
                    __builtin_memcpy(&v, view_addr(a.in, n, c, imin(imax(y + q - 2, ylo), yhi), x - 2), 8);
                    s_win[q][threadIdx.x] = v;
                }
            }
            auto px = [&](int dy, int dx) {
                if (inner) return (int)((const uint8_t *)&s_win[dy + 2][threadIdx.x])[dx + 2];
                const int gy = imin(imax(y + dy, ylo), yhi), gx = imin(imax(x + dx, 0), a.W - 1);
                return (int)*view_addr(a.in, n, c, gy, gx);
            };
            const int va = px(0, 0);
            RotAcc<4> acc;
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
                    pass_global<4, r>(lut, va, v0, v1, v2, a, acc);
                });
            }
            uint32_t o[4];
            finish_channel<4, OUT>(a, acc, n, c, y, x, o);
            if constexpr (OUT == kOutPackedRGBU4) {      // one channel of the packed RGB block: bytes at stride 3
#pragma unroll
                for (int sy = 0; sy < 4; ++sy) {
                    uint8_t *dst = const_cast<uint8_t *>(view_addr(a.out, n, 0, y * 4 + sy, x * 4)) + c;
#pragma unroll
                    for (int sx = 0; sx < 4; ++sx) dst[3 * sx] = (uint8_t)(o[sy] >> (8 * sx));
                }
            }
        }
    }
}

// The same fix-up with one PASS per lane.  The list is short (~0.1 % of the samples on smooth content) and a sample's 12 passes
// are independent: the kernel above serialises them in one thread -- twelve dependent trips to L2 per entry, 110 us per launch for
// ~50 k entries with most of the chip idle.  Here a 16-lane group takes an entry; lane p computes passes p, p + 16, ... of the
// sample (mode p / 4, rotation p % 4): four neighbour bytes and five 16-byte rows straight from global memory -- two dependent
// trips for the whole sample -- multiplied out per row element and added into the group's 16 LDS sums at the block positions
// that rotation maps the elements to; lane e then finishes block position e (divide, round half to even, clip) and stores its byte.
__global__ void __launch_bounds__(256) stage_up_fix2_kernel(StageArgs a) {
    __shared__ int s_sum[16][16];
    const uint32_t count = *a.fix_count;
    const int grp = (int)(threadIdx.x >> 4), ln = (int)(threadIdx.x & 15);
    const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
    const int unbias = 128 * kQ * 4 * a.M - a.bias_num;
    // a group's 16 lanes sit in one wave and LDS serves a wave's operations in order: no workgroup barrier anywhere, the groups run
    // independently (the fences only keep the compiler from moving the LDS accesses across each other)
    for (uint32_t i = blockIdx.x * 16u + (uint32_t)grp; i < count; i += gridDim.x * 16u) {
        // entry: pixel id (n H + y) W + x in the low 30 bits, channel in the top two (3 = every channel)
        const uint32_t ent = a.fix_list[i], id = ent & 0x3FFFFFFFu, only = ent >> 30;
        const int x = (int)(id % (uint32_t)a.W), y = (int)((id / (uint32_t)a.W) % (uint32_t)a.H), n = (int)(id / ((uint32_t)a.W * (uint32_t)a.H));
        const int c_lo = only == 3u ? 0 : (int)only, c_hi = only == 3u ? a.C : (int)only + 1;
        for (int c = c_lo; c < c_hi; ++c) {
            s_sum[grp][ln] = 0;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            auto px = [&](int dy, int dx) {
                const int gy = imin(imax(y + dy, ylo), yhi), gx = imin(imax(x + dx, 0), a.W - 1);
                return (int)*view_addr(a.in, n, c, gy, gx);
            };
            const int va = px(0, 0);
            for (int p = ln; p < 4 * a.M; p += 16) {
                const int m = p >> 2, r = p & 3;
                int dy, dx, v[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    sample_offset(r, a.di[m][k], a.dj[m][k], dy, dx);
                    v[k] = px(dy, dx);
                }
                int idx[5], w[5];
                simplex4(va, v[0], v[1], v[2], idx, w);
                const uint4 *tab = (const uint4 *)a.lut[m];
                uint32_t row[5][4];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const uint4 t = tab[idx[j]];
                    row[j][0] = t.x; row[j][1] = t.y; row[j][2] = t.z; row[j][3] = t.w;
                }
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    int sum = 0;
#pragma unroll
                    for (int j = 0; j < 5; ++j) sum += w[j] * (int)((row[j][e >> 2] >> (8 * (e & 3))) & 0xFFu);
                    // block position (sy, sx) that rotation r gives row element e (the inverse of row_elem)
                    const int pos = r == 0 ? e : r == 1 ? (e & 3) * 4 + 3 - (e >> 2) : r == 2 ? 15 - e : (3 - (e & 3)) * 4 + (e >> 2);
                    atomicAdd(&s_sum[grp][pos], sum);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const uint32_t b = rhe_clip_u8(s_sum[grp][ln] - unbias, a.div);
            *const_cast<uint8_t *>(view_addr(a.out, n, c, y * 4 + (ln >> 2), x * 4 + (ln & 3))) = (uint8_t)b;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // sums read before the next round clears them
        }
    }
}

// variant (tuning "fix_kernel"): 0 = one pass per lane (stage_up_fix2_kernel), 1 = one entry per thread (stage_up_fix_kernel)
hipError_t launch_stage_up_fix(const StageArgs &a, int out_mode, int num_cus, hipStream_t st, int variant) {
    if (a.C > 3 || !a.fix_list || !a.fix_count) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(4 * num_cus)), block(256);
    if (variant == 0) hipLaunchKernelGGL(stage_up_fix2_kernel, dim3((unsigned)(8 * num_cus)), block, 0, st, a);
    else if (out_mode == kOutPlanarU4) hipLaunchKernelGGL((stage_up_fix_kernel<kOutPlanarU4>), grid, block, 0, st, a);
    else if (out_mode == kOutPackedRGBU4 && a.C == 3) hipLaunchKernelGGL((stage_up_fix_kernel<kOutPackedRGBU4>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((stage_up_fix_kernel<kOutGeneric>), grid, block, 0, st, a);
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
// tube band; no tile routing; flagged sites go to stage_up_fix_site_kernel through a.fix_list
hipError_t launch_stage_u2t(const StageArgs &a, const BandArgs &b, int num_cus, hipStream_t st) {
    if (a.C > 3 || a.M > 3 || !a.fix_list || !a.fix_count || a.verdict_take >= 0) return hipErrorInvalidValue;
    hipError_t e = launch_u1t_t<2>(a, b, 0u, num_cus, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(stage_up_fix_site_kernel<2>, dim3((unsigned)(4 * num_cus)), dim3(256), 0, st, a);
    return hipGetLastError();
}

const char *stage_tube_name(int out_mode) {
    return out_mode == kOutPackedRGBU4 ? "stage_tube_kernel<rgb>" : out_mode == kOutPlanarU4 ? "stage_tube_kernel<planar>"
                                                                                              : "stage_tube_kernel<generic>";
}

template <int OUT>
static hipError_t launch_tube_t(const StageArgs &a, const BandArgs &b, int num_cus, hipStream_t st) {
    auto kern_t = stage_tube_kernel<OUT, KB_TW, KB_TH, true>;
    auto kern_f = stage_tube_kernel<OUT, KB_TW, KB_TH, false>;
    auto kern = a.site_flags ? kern_t : kern_f;
    static bool attr_set[64][2] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev][a.site_flags ? 1 : 0]) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set[dev][a.site_flags ? 1 : 0] = true;
    }
    const long long ntiles = (long long)a.N * a.tiles_x * a.tiles_y;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)(ntiles < num_cus ? ntiles : num_cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(KB_TW * KB_TH), (size_t)kTubeLdsBytes, st, a, b);
    return hipGetLastError();
}

hipError_t launch_stage_tube(const StageArgs &a, const BandArgs &b, int out_mode, int num_cus, hipStream_t st) {
    if (a.C > 3 || a.M > 3) return hipErrorInvalidValue;
    if (out_mode == kOutPlanarU4) return launch_tube_t<kOutPlanarU4>(a, b, num_cus, st);
    if (out_mode == kOutPackedRGBU4 && a.C == 3) return launch_tube_t<kOutPackedRGBU4>(a, b, num_cus, st);
    return launch_tube_t<kOutGeneric>(a, b, num_cus, st);
}

// ------------------------------------------------------------------------------------------
// K2-tube2: the tube kernel with every LDS read hand-scheduled (tools/gen_tube2_asm.py -> mulut_tube2_asm.inc).
// Same LDS image, same arithmetic, same fix-up list as stage_tube_kernel; what changes is WHEN things are issued:
//   * the five rows of a pass live in 40 VGPRs above the compiler's register budget, so the ten ds_read_b128 of the
//     NEXT pass are in flight while the current pass's 40 v_pk_mad_u16 run (a row's registers are refilled right after
//     its eight MACs); the neighbour codes of the pair after next are fetched under the MACs of a pair's second pass;
//   * the index math of a pair (compiler-scheduled C++ between the blocks) therefore never waits for LDS;
//   * the pipeline runs on across the channels of a site; it drains once per tile (one site per thread and tile).
// Pixel codes are code1 (f << 12 | h): a sort key is code | stride (one OR), the slot sum a v_pk_mad_u16 chain on the
// raw codes (the LSB nibble multiplies out of the 16-bit half), as in the first-stage tube kernel.
// The mode list is a template parameter (PATS = M | p0 << 2 | p1 << 4 | p2 << 6): every neighbour offset and band offset is
// an immediate.  Instantiated for the mode strings launch_stage_tube2 lists; the others take stage_tube_kernel.
// ------------------------------------------------------------------------------------------
#if defined(MULUT_VARIANT_t2dbg1)
#include "mulut_tube2_asm_dbg1.inc"
#elif defined(MULUT_VARIANT_t2dbg2)
#include "mulut_tube2_asm_dbg2.inc"
#elif defined(MULUT_VARIANT_t2dbg3)
#include "mulut_tube2_asm_dbg3.inc"
#elif defined(MULUT_VARIANT_t2dbg4)
#include "mulut_tube2_asm_dbg4.inc"
#elif defined(MULUT_VARIANT_t2dbg5)
#include "mulut_tube2_asm_dbg5.inc"
#else
#include "mulut_tube2_asm.inc"
#endif

struct T2Pair {
    uint32_t base, s0, s1, s2;      // packed per pass: byte offset of row 0 (+ bias), byte strides of path steps 1..3
    uint32_t w0, w1, w2, w3, w4;    // packed weights
};
__host__ __device__ constexpr int t2_modes(int pats) { return pats & 3; }
__host__ __device__ constexpr int t2_pat(int pats, int m) { return (pats >> (2 + 2 * m)) & 3; }
constexpr int kT2W = 16, kT2H = 4;        // a wave's tile: 16 x 4 pixels, one site per lane (192 contiguous output bytes per HR row)
constexpr int kT2PW = kT2W + 2 * kTubeHaloX, kT2PH = kT2H + 2 * kHalo, kT2Chan = 2 * kT2PH * kT2PW;
constexpr int kT2WaveTileBytes = 3 * kT2Chan;
// LDS: [ band s | band d | band y ][ 16 wave images of pixel codes ][ parked output rows of channels 0 and 1 (RGB path) ][ work counter ]
constexpr int kTube2LdsBytes = 3 * kTubeBandBytes + 16 * kT2WaveTileBytes + 2 * 16 * KB_TW * KB_TH + 16;
static_assert(kTube2LdsBytes <= 160 * 1024, "LDS budget");
static_assert(KB_TW % kT2W == 0 && KB_TH % kT2H == 0 && (KB_TW / kT2W) * (KB_TH / kT2H) == 16 && KB_TW / kT2W == 4, "16 wave tiles per verdict tile, four across");
// byte offset (from the window corner) of neighbour K of pattern PAT under rotation R; SIGN -1: rotation R + 2
__host__ __device__ constexpr int t2_nb(int pat, int r, int k, int sign) {
    return 2 * ((2 + sign * rot_dy(r, kPatDi[pat][k], kPatDj[pat][k])) * kT2PW + 2 + sign * rot_dx(r, kPatDi[pat][k], kPatDj[pat][k]));
}
__host__ __device__ constexpr int t2_imm(int pat) { return pat * kTubeBandBytes - tube_bias(pat); }

// float of the signed 16-bit value in the low half of x: one SDWA convert with sign extension
__device__ __forceinline__ float t2_f32_of_i16(uint32_t x) {
    float f;
    asm("v_cvt_f32_i32_sdwa %0, sext(%1) dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "=v"(f) : "v"(x));
    return f;
}

template <int PAT>
__device__ __forceinline__ void t2_index(uint32_t k0, uint32_t ha, uint32_t base_a0, uint32_t pb, uint32_t pc, uint32_t pd, T2Pair &o, uint32_t &dirty) {
    constexpr uint32_t SB = kTubeSB * 16, SC = kTubeSC * 16, SD = kTubeSD * 16;
    uint32_t k1 = pb | pk_dup(SB), k2 = pc | pk_dup(SC), k3 = pd | pk_dup(SD);
    const uint32_t hb = pb & 0x000F000Fu, hc = pc & 0x000F000Fu, hd = pd & 0x000F000Fu;
    pk_cmpx_desc(k0, k1);
    pk_cmpx_desc(k2, k3);
    pk_cmpx_desc(k0, k2);
    pk_cmpx_desc(k1, k3);
    pk_cmpx_desc(k1, k2);
    const uint32_t f1 = pk_shr12(k0), f2 = pk_shr12(k1), f3 = pk_shr12(k2), f4 = pk_shr12(k3);
    // code * (16 * stride) = 16 * h * stride per half: the f nibble (bits 12..15) times a multiple of 16 leaves the half
    o.base = pk_mad(pb, pk_dup(SB), pk_mad(pc, pk_dup(SC), pk_mad(pd, pk_dup(SD), base_a0 + pk_dup((uint32_t)tube_bias(PAT)))));
    o.s0 = k0 & 0x0FF00FF0u;
    o.s1 = k1 & 0x0FF00FF0u;
    o.s2 = k2 & 0x0FF00FF0u;
    o.w0 = pk_dup(kQ) - f1;
    o.w1 = f1 - f2;
    o.w2 = f2 - f3;
    o.w3 = f3 - f4;
    o.w4 = f4;
    const uint32_t mx = pk_max(pk_max(hb, hc), pk_max(hd, ha));
    const uint32_t mn = pk_min(pk_min(hb, hc), pk_min(hd, ha));
    dirty |= (mx - mn) & 0xFFFEFFFEu;       // in the tube iff the MSBs span at most one step
}

#define T2_ACC_OPS(lo, hi) [l0] "+v"(lo[0]), [l1] "+v"(lo[1]), [l2] "+v"(lo[2]), [l3] "+v"(lo[3]), [h0] "+v"(hi[0]), [h1] "+v"(hi[1]), [h2] "+v"(hi[2]), [h3] "+v"(hi[3])
#define T2_TMP_OPS [a0] "=&v"(a0), [a1] "=&v"(a1), [a2] "=&v"(a2), [a3] "=&v"(a3)
#define T2_W_OPS(c) [w0] "v"(c.w0), [w1] "v"(c.w1), [w2] "v"(c.w2), [w3] "v"(c.w3), [w4] "v"(c.w4)
#define T2_ADDR_OPS(n) [base] "v"(n.base), [s0] "v"(n.s0), [s1] "v"(n.s1), [s2] "v"(n.s2)
#define T2_IMM_OPS(PAT) [ilo] "i"(t2_imm(PAT)), [ihi] "i"(t2_imm(PAT) + kTubePlaneBytes), [ilo4] "i"(t2_imm(PAT) + kTubeAll * 16), [ihi4] "i"(t2_imm(PAT) + kTubeAll * 16 + kTubePlaneBytes)
#define T2_NB_OPS(PAT, R, OFF) [n0] "i"(t2_nb(PAT, R, 0, 1) + (OFF)), [n1] "i"(t2_nb(PAT, R, 0, -1) + (OFF)), [n2] "i"(t2_nb(PAT, R, 1, 1) + (OFF)), \
    [n3] "i"(t2_nb(PAT, R, 1, -1) + (OFF)), [n4] "i"(t2_nb(PAT, R, 2, 1) + (OFF)), [n5] "i"(t2_nb(PAT, R, 2, -1) + (OFF)), [nan] "i"(2 * (2 * kT2PW + 2) + (OFF))

#define T2_NBOUT_OPS [pb] "=&v"(pb), [pc] "=&v"(pc), [pd] "=&v"(pd), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2)
// first pass of a pair (weights = low halves of cur): MACs, refill with the pair's second pass
template <int PAT>
__device__ __forceinline__ void t2_block_a(uint32_t (&lo)[4], uint32_t (&hi)[4], const T2Pair &cur) {
    uint32_t a0, a1, a2, a3;
    asm volatile(TUBE2_ASM_A : T2_ACC_OPS(lo, hi), T2_TMP_OPS : T2_W_OPS(cur), T2_ADDR_OPS(cur), T2_IMM_OPS(PAT) : TUBE2_CLOBBERS);
}
// second pass (weights = high halves of cur, reversed element order): MACs, refill with the first pass of the next pair (pattern
// NPAT, addresses from nxt).  NN: neighbour codes fetched under the MACs -- 0 none, 6 those of rotation pair TR of pattern TPAT in the
// window at win + TOFF (out: pb, pc, pd), 7 the same + that window's anchor code (out: ca, in both halves).
// LAST: no refill (the site's very last pass).
template <int NPAT, int NN, int TPAT, int TR, int TOFF, bool LAST>
__device__ __forceinline__ void t2_block_b(uint32_t (&lo)[4], uint32_t (&hi)[4], const T2Pair &cur, const T2Pair &nxt, uint32_t win,
                                           uint32_t &pb, uint32_t &pc, uint32_t &pd, uint32_t &ca) {
    uint32_t a0, a1, a2, a3, t0, t1, t2;
    if constexpr (LAST)
        asm volatile(TUBE2_ASM_B_LAST : T2_ACC_OPS(lo, hi) : T2_W_OPS(cur) : TUBE2_CLOBBERS);
    else if constexpr (NN == 0)
        asm volatile(TUBE2_ASM_B_N0 : T2_ACC_OPS(lo, hi), T2_TMP_OPS : T2_W_OPS(cur), T2_ADDR_OPS(nxt), T2_IMM_OPS(NPAT) : TUBE2_CLOBBERS);
    else if constexpr (NN == 6)
        asm volatile(TUBE2_ASM_B_N6 : T2_ACC_OPS(lo, hi), T2_TMP_OPS, T2_NBOUT_OPS : T2_W_OPS(cur), T2_ADDR_OPS(nxt), T2_IMM_OPS(NPAT), [win] "v"(win), T2_NB_OPS(TPAT, TR, TOFF) : TUBE2_CLOBBERS);
    else
        asm volatile(TUBE2_ASM_B_N7 : T2_ACC_OPS(lo, hi), T2_TMP_OPS, T2_NBOUT_OPS, [ca] "=&v"(ca) : T2_W_OPS(cur), T2_ADDR_OPS(nxt), T2_IMM_OPS(NPAT), [win] "v"(win), T2_NB_OPS(TPAT, TR, TOFF) : TUBE2_CLOBBERS);
}

// Work decomposition: NO workgroup barrier after the bands are staged.  The phase stamps of the first version (one 64 x 16 tile per
// workgroup and barrier) showed every wave parked at the tile barrier for 30 % of its life: the SIMD arbitrates oldest-first, so a
// tile's waves finish far apart, and while the early ones wait the SIMD runs at the issue rate of one or two waves.  Here a WAVE owns
// a 16 x 4 pixel tile (one site per lane) with a private 24 x 8 x C image of pixel codes in LDS; it draws its next tile from a
// workgroup counter in LDS (a workgroup still owns an XCD-contiguous run of 64 x 16 verdict tiles = 16 wave tiles each), fetches it
// while it computes the current one, and never waits for another wave.
template <int OUT, int PATS>
__global__ void __launch_bounds__(KB_TW *KB_TH) __attribute__((amdgpu_waves_per_eu(TUBE2_WAVES_PER_EU, TUBE2_WAVES_PER_EU))) stage_tube2_kernel(StageArgs a, BandArgs b) {
    constexpr int TW = KB_TW, TH = KB_TH, PW = kT2PW, PH = kT2PH, NT = TW * TH;
    constexpr int M = t2_modes(PATS), NP = 2 * M;
    constexpr int DW = PW / 4, PER4 = (3 * PH * DW + 63) / 64;         // aligned dwords per image row / per lane
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *s_next = (uint32_t *)(smem + kTube2LdsBytes - 16);       // the workgroup's next work item
    if (lds_addr_of(smem) != 0u) __builtin_trap();      // every LDS address below is absolute: the dynamic block must start at 0 (no static LDS here)

    const int ntiles = a.N * a.tiles_x * a.tiles_y;
    const int G = gridDim.x;
    const bool by_xcd = (G & 7) == 0;
    const int per = (ntiles + 7) >> 3;
    const int first = by_xcd ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int last = by_xcd ? imin(((int)(blockIdx.x & 7) + 1) * per, ntiles) : ntiles;
    const int step = by_xcd ? (G >> 3) : G;
    if (first >= last) return;              // workgroup-uniform
    const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
    const bool dw_ok = __builtin_amdgcn_readfirstlane((int)(a.in.sX == 1 && ((a.W | a.in.sY | a.in.sC) & 3) == 0 && (a.in.sN & 3) == 0 && (((uintptr_t)a.in.p) & 3) == 0)) != 0;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint8_t *img = smem + 3 * kTubeBandBytes + wave * kT2WaveTileBytes;      // this wave's image of pixel codes

    // work item j of the workgroup = wave tile j & 15 of its (j >> 4)-th verdict tile; -1 = none left.  Lane-uniform.
    auto grab = [&]() {
        for (;;) {
            uint32_t j = 0;
            if ((threadIdx.x & 63) == 0) j = atomicAdd(s_next, 1u);
            j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
            const long long tile = (long long)first + (long long)(j >> 4) * step;
            if (tile >= last) return -1;
            if (a.verdict_take >= 0 && (int)a.verdict[tile] != a.verdict_take) continue;
            int n, ty0, tx0;
            decode_tile(a, (int)tile, n, ty0, tx0, TW, TH);
            if (tx0 + kT2W * (int)(j & 3u) >= a.W || ty0 + kT2H * (int)((j >> 2) & 3u) >= a.oy1) continue;
            return (int)j;
        }
    };
    auto origin = [&](int j, int &n, int &y0, int &x0) {
        decode_tile(a, first + (j >> 4) * step, n, y0, x0, TW, TH);
        x0 += kT2W * (j & 3);
        y0 += kT2H * ((j >> 2) & 3);
    };
    // Planar input whose rows start on dword boundaries (the pipeline's intermediate images): a wave tile is fetched as aligned
    // dwords, in flight while the previous tile is computed.  Any other input takes the byte path at stash time.
    auto fetch = [&](int j, uint32_t (&v)[PER4]) {
        if (!dw_ok || j < 0) return;
        int n, y0, x0;
        origin(j, n, y0, x0);
        int lane = (int)(threadIdx.x & 63);       // opaque: no per-lane term of this may stay live across a tile's computation
        asm volatile("" : "+v"(lane));
#pragma unroll
        for (int k = 0; k < PER4; ++k) {
            const int i = lane + 64 * k;
            const int q = i % DW, py = (i / DW) % PH, c = imin(i / (DW * PH), a.C - 1);
            const int gy = imin(imax(y0 + py - kHalo, ylo), yhi);
            const int gx = imin(imax(x0 - kTubeHaloX + 4 * q, 0), a.W - 4);
            v[k] = *(const uint32_t *)view_addr(a.in, n, c, gy, gx);
        }
    };
    auto stash = [&](int j, const uint32_t (&v)[PER4]) {
        if (j < 0) return;
        int n, y0, x0;
        origin(j, n, y0, x0);
        int lane = (int)(threadIdx.x & 63);
        asm volatile("" : "+v"(lane));
        if (dw_ok) {
#pragma unroll
            for (int k = 0; k < PER4; ++k) {
                const int i = lane + 64 * k;
                if (i < a.C * PH * DW) {
                    const int gx = x0 - kTubeHaloX + 4 * (i % DW);
                    // bytes (b0,b1) / (b2,b3) into 16-bit lanes; a dword clamped at an image edge replicates the edge byte
                    const uint32_t sel_lo = gx < 0 ? 0x0C000C00u : gx > a.W - 4 ? 0x0C030C03u : 0x0C010C00u;
                    const uint32_t sel_hi = gx < 0 ? 0x0C000C00u : gx > a.W - 4 ? 0x0C030C03u : 0x0C030C02u;
                    const uint32_t lo = __builtin_amdgcn_perm(0u, v[k], sel_lo), hi = __builtin_amdgcn_perm(0u, v[k], sel_hi);
                    // code1 per 16-bit lane: (b << 12) keeps the LSB nibble in bits 12..15, b >> 4 is the MSB nibble
                    uint2 c2;
                    c2.x = pk_mad(lo, pk_dup(0x1000u), pk_shr4(lo));
                    c2.y = pk_mad(hi, pk_dup(0x1000u), pk_shr4(hi));
                    *(uint2 *)(img + 8 * i) = c2;
                }
            }
        } else {
            for (int i = lane; i < a.C * PH * PW; i += 64) {
                const int px = i % PW, py = (i / PW) % PH, c = i / (PW * PH);
                const int gy = imin(imax(y0 + py - kHalo, ylo), yhi);
                const int gx = imin(imax(x0 + px - kTubeHaloX, 0), a.W - 1);
                ((uint16_t *)img)[i] = (uint16_t)pixel_code1(*view_addr(a.in, n, c, gy, gx));
            }
        }
    };

#if defined(MULUT_VARIANT_t2prof)   /* probe build: shader-clock ticks per phase, summed over all waves into the context's probe buffer */
    const unsigned long long t_first = __builtin_amdgcn_s_memtime(), r_first = __builtin_amdgcn_s_memrealtime();
    uint32_t t_prev = (uint32_t)t_first, t_ph0 = 0, t_ph1 = 0, t_ph2 = 0, t_ph3 = 0, t_ph4 = 0, t_ph5 = 0;      // wave-uniform (scalar registers)
#define T2_STAMP(PH) do { const uint32_t t_now = (uint32_t)__builtin_amdgcn_s_memtime(); t_ph##PH += t_now - t_prev; t_prev = t_now; } while (0)
#else
#define T2_STAMP(PH) do { } while (0)
#endif
    // bands: slot = pattern id of the mode; patterns the mode list lacks are never read
    static_for<0, M>([&](auto MI) {
        constexpr int pat = t2_pat(PATS, MI);
        const uint4 *src = (const uint4 *)b.band[MI];
        uint4 *dst = (uint4 *)(smem + pat * kTubeBandBytes);
        for (int i = threadIdx.x; i < kTubeBandBytes / 16; i += NT) dst[i] = src[i];
    });
    if (threadIdx.x == 0) *s_next = 0u;
    __syncthreads();          // the only barrier of the kernel
    T2_STAMP(5);

    uint32_t pix[PER4];
    int item = grab();
    fetch(item, pix);
    stash(item, pix);
    while (item >= 0) {
        const int nxt_item = grab();
        fetch(nxt_item, pix);
        // the site of this lane; re-derived (from an opaque copy of the lane id) wherever it is needed: a value computed before the
        // pipelined loop and used after it would be parked in scratch (the loop needs every register)
        auto site = [&](int &n_, int &y_, int &x_, int &lx_, int &ly_) {
            int y0, x0;
            origin(item, n_, y0, x0);
            int l = (int)(threadIdx.x & 63);
            asm volatile("" : "+v"(l));
            lx_ = l % kT2W; ly_ = l / kT2W;
            y_ = y0 + ly_; x_ = x0 + lx_;
        };
        int n, y, x, lx, ly;
        site(n, y, x, lx, ly);
        T2_STAMP(0);       // work item drawn, next tile's fetch issued
        if (y < a.oy1 && x < a.W) {
            // LDS byte address of the 5x5 window corner (y-2, x-2) of this site, channel 0
            uint32_t win = (uint32_t)(3 * kTubeBandBytes + wave * kT2WaveTileBytes + 2 * (ly * PW + lx + kTubeHaloX - kHalo));
            // finished channels wait in LDS for the RGB interleave (a uint4 per thread and channel): registers are what the
            // pipelined loop below is short of
            auto park = [&]() {
                int t = (int)threadIdx.x;
                asm volatile("" : "+v"(t));
                return (uint4 *)(smem + 3 * kTubeBandBytes + 16 * kT2WaveTileBytes) + t;
            };
            uint32_t dmask = 0u, dirty = 0u, dirty_n = 0u;
            uint32_t o[4] = {0u, 0u, 0u, 0u};
            uint32_t pb, pc, pd, ca;
            uint32_t k0, ha, ba0;
            auto anchor = [&](uint32_t c2) {      // c2 = the anchor's code in both halves
                k0 = (c2 & 0xF000F000u) | pk_dup((uint32_t)kTubeSA * 16);
                ha = c2 & 0x000F000Fu;
                ba0 = pk_mad(c2, pk_dup((uint32_t)kTubeSA * 16), 0u);
            };
            T2Pair cur, nxt;
            {
                constexpr int p0 = t2_pat(PATS, 0);
                uint32_t t0, t1, t2;
                asm volatile(TUBE2_ASM_LOAD_NB_ANCHOR : T2_NBOUT_OPS, [ca] "=&v"(ca) : [win] "v"(win), T2_NB_OPS(p0, 0, 0) : TUBE2_CLOBBERS);
                anchor(ca);
                t2_index<p0>(k0, ha, ba0, pb, pc, pd, cur, dirty);
                asm volatile(TUBE2_ASM_LOAD_NB : T2_NBOUT_OPS : [win] "v"(win), T2_NB_OPS(p0, 1, 0) : TUBE2_CLOBBERS);
                uint32_t a0, a1, a2, a3;
                asm volatile(TUBE2_ASM_FIRST_ROWS : T2_TMP_OPS : T2_ADDR_OPS(cur), T2_IMM_OPS(p0) : TUBE2_CLOBBERS);
                t2_index<p0>(k0, ha, ba0, pb, pc, pd, nxt, dirty);
            }
            RotAcc<4> acc;
            // the (0,2) fields start at -unbias (the rows are value + 128: unbias = 128 * 16 * 4 M <= 24576) where the epilogue works on K
            auto acc_start = [&]() {
                acc.clear();
                if constexpr (OUT != kOutGeneric) {
                    const uint32_t nb = pk_dup((uint32_t)(65536 - 128 * kQ * 4 * M));
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc.lo02[k] = acc.hi02[k] = nb;
                }
            };
            acc_start();
            T2_STAMP(1);   // pipeline prologue: neighbours of pairs 0 and 1, index math, first rows requested
#pragma clang loop unroll(disable)
            for (int c = 0; c < a.C; ++c, win += kT2Chan) {
                const bool more = c + 1 < a.C;       // wave-uniform
                asm volatile("; MULUT_T2_STREAM_BEGIN (tools/ubench/gen_stream_ubench.py cuts the ISA here)");
                static_for<0, NP>([&](auto PI) {
                    constexpr int p = PI;
                    if constexpr (p == (NP >= 6 ? 4 : NP - 1)) asm volatile("; MULUT_T2_STREAM_END");      // pairs 0..3: straight-line, no `more` variants
                    constexpr int pat = t2_pat(PATS, p >> 1), R = p & 1;
                    // the pair whose neighbours are fetched now (two pairs ahead) and the pair whose first rows refill the registers
                    constexpr bool t_here = p + 2 < NP;
                    constexpr int tp = t_here ? p + 2 : p + 2 - NP, tpat = t2_pat(PATS, tp >> 1), tr = tp & 1, toff = t_here ? 0 : kT2Chan;
                    constexpr bool n_here = p + 1 < NP;
                    constexpr int npat = t2_pat(PATS, n_here ? (p + 1) >> 1 : 0);
                    T2Pair nn;
                    auto &lo = R == 0 ? acc.lo02 : acc.lo13;
                    auto &hi = R == 0 ? acc.hi02 : acc.hi13;
                    t2_block_a<pat>(lo, hi, cur);
                    if (t_here || more) {      // (then the next pair exists too: t_here implies n_here)
                        t2_block_b<npat, (tp == 0 ? 7 : 6), tpat, tr, toff, false>(lo, hi, cur, nxt, win, pb, pc, pd, ca);
                        if constexpr (tp == 0) anchor(ca);
                        if constexpr (t_here) t2_index<tpat>(k0, ha, ba0, pb, pc, pd, nn, dirty);
                        else t2_index<tpat>(k0, ha, ba0, pb, pc, pd, nn, dirty_n);
                    } else {
                        if constexpr (n_here) t2_block_b<npat, 0, 0, 0, 0, false>(lo, hi, cur, nxt, win, pb, pc, pd, ca);
                        else t2_block_b<npat, 0, 0, 0, 0, true>(lo, hi, cur, nxt, win, pb, pc, pd, ca);
                        nn = nxt;
                    }
                    cur = nxt;
                    nxt = nn;
                });
                if constexpr (OUT != kOutGeneric) {
                    // Epilogue on the numerators themselves: the (0,2) accumulators started at -unbias (mod 2^16), so the sum of the
                    // two fields of a block position IS K = 16 M pred (mod 2^16, |K| < 2^15).  Per byte: one 16-bit-select add, one
                    // sign-extending convert, one multiply by fl(1/d), one v_cvt_pk_u8_f32 (it rounds to nearest even and saturates:
                    // tools/probe_cvt.hip).  Exact for this divisor: StageArgs::use_f32, proven by brute force at configure time.
                    static_for<0, 4>([&](auto SY) {
                        constexpr int sy = SY;
                        const uint32_t s0 = tube_field<4 * sy + 0>(acc.lo02, acc.hi02) + tube_field<12 + sy>(acc.lo13, acc.hi13);
                        const uint32_t s1 = tube_field<4 * sy + 1>(acc.lo02, acc.hi02) + tube_field<8 + sy>(acc.lo13, acc.hi13);
                        const uint32_t s2 = tube_field<4 * sy + 2>(acc.lo02, acc.hi02) + tube_field<4 + sy>(acc.lo13, acc.hi13);
                        const uint32_t s3 = tube_field<4 * sy + 3>(acc.lo02, acc.hi02) + tube_field<0 + sy>(acc.lo13, acc.hi13);
                        uint32_t r = __builtin_amdgcn_cvt_pk_u8_f32(t2_f32_of_i16(s0) * a.inv_d, 0u, 0u);
                        r = __builtin_amdgcn_cvt_pk_u8_f32(t2_f32_of_i16(s1) * a.inv_d, 1u, r);
                        r = __builtin_amdgcn_cvt_pk_u8_f32(t2_f32_of_i16(s2) * a.inv_d, 2u, r);
                        r = __builtin_amdgcn_cvt_pk_u8_f32(t2_f32_of_i16(s3) * a.inv_d, 3u, r);
                        o[sy] = r;
                    });
                    if constexpr (OUT == kOutPackedRGBU4) {
                        if (c < 2) park()[c * NT] = make_uint4(o[0], o[1], o[2], o[3]);
                    } else {
                        int n2, y2, x2, lx2, ly2;
                        site(n2, y2, x2, lx2, ly2);
#pragma unroll
                        for (int sy = 0; sy < 4; ++sy) *(uint32_t *)const_cast<uint8_t *>(view_addr(a.out, n2, c, y2 * 4 + sy, x2 * 4)) = o[sy];
                    }
                } else {
                    int n2, y2, x2, lx2, ly2;
                    site(n2, y2, x2, lx2, ly2);
                    finish_channel<4, OUT>(a, acc, n2, c, y2, x2, o);
                }
                acc_start();
                dmask |= (dirty != 0u ? 1u : 0u) << c;
                dirty = dirty_n;
                dirty_n = 0u;
            }
            T2_STAMP(2);   // the channels: 12 passes + epilogue each
            site(n, y, x, lx, ly);
            if constexpr (OUT == kOutPackedRGBU4) {
                const uint4 r = park()[0], g = park()[NT];
                const uint32_t oR[4] = {r.x, r.y, r.z, r.w}, oG[4] = {g.x, g.y, g.z, g.w};
                store_rgb<4>(a, n, y, x, oR, oG, o);
            }
            // dirty samples (pixel, channel) go on the fix-up list: one atomic per wave and channel (rare), compacted by lane rank
            if (__ballot(dmask != 0u) != 0ull) {
                const uint32_t pixel_id = (uint32_t)((n * a.H + y) * a.W + x);
                for (int c = 0; c < a.C; ++c) {
                    const bool d = ((dmask >> c) & 1u) != 0u;
                    const unsigned long long dm = __ballot(d);
                    if (dm == 0ull) continue;
                    const int lane = (int)(threadIdx.x & 63);
                    uint32_t at = 0;
                    if (lane == __ffsll((long long)dm) - 1) at = atomicAdd(a.fix_count, (uint32_t)__popcll(dm));
                    at = (uint32_t)__shfl((int)at, __ffsll((long long)dm) - 1);
                    if (d) a.fix_list[at + (uint32_t)__popcll(dm & ((1ull << lane) - 1ull))] = pixel_id | ((uint32_t)c << 30);
                }
            }
        }
        T2_STAMP(3);       // output stores, fix-up list
        stash(nxt_item, pix);      // the wave's image is its own: every read of the current tile has returned (the pipeline drained)
        T2_STAMP(4);       // next tile's pixel codes into LDS (waits for its fetch)
        item = nxt_item;
    }
#if defined(MULUT_VARIANT_t2prof)
    if (a.dbg && (threadIdx.x & 63) == 0) {
        atomicAdd(a.dbg + 0, (unsigned long long)t_ph0); atomicAdd(a.dbg + 1, (unsigned long long)t_ph1); atomicAdd(a.dbg + 2, (unsigned long long)t_ph2);
        atomicAdd(a.dbg + 3, (unsigned long long)t_ph3); atomicAdd(a.dbg + 4, (unsigned long long)t_ph4); atomicAdd(a.dbg + 5, (unsigned long long)t_ph5);
        atomicAdd(a.dbg + 6, __builtin_amdgcn_s_memtime() - t_first);          // wave lifetime in shader-clock ticks ...
        atomicAdd(a.dbg + 7, __builtin_amdgcn_s_memrealtime() - r_first);      // ... and in 100 MHz ticks: their ratio is the in-kernel clock
        atomicAdd(a.dbg + 8, 1ull);
    }
#endif
#undef T2_STAMP
}

constexpr int kT2PatsSDY = 3 | (0 << 2) | (1 << 4) | (2 << 6);

// pattern list of the launch as PATS (0 if the list is not one stage_tube2_kernel is built for)
static int tube2_pats(const StageArgs &a) {
    int pats = a.M;
    for (int m = 0; m < a.M; ++m) pats |= (a.dj[m][0] == 2 ? 1 : a.di[m][0] == 1 ? 2 : 0) << (2 + 2 * m);
    return pats == kT2PatsSDY ? pats : 0;
}
bool stage_tube2_supported(const StageArgs &a) {
    // the float epilogue must be exact for the divisor (StageArgs::use_f32, proven at configure time), the bias the numerator bias of a final stage
    return a.C <= 3 && a.M <= 3 && a.site_flags == nullptr && tube2_pats(a) != 0 && a.use_f32 && a.bias_num == 0;
}

template <int OUT>
static hipError_t launch_tube2_t(const StageArgs &a, const BandArgs &b, int num_cus, hipStream_t st) {
    auto kern = stage_tube2_kernel<OUT, kT2PatsSDY>;
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    const long long ntiles = (long long)a.N * a.tiles_x * a.tiles_y;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)(ntiles < num_cus ? ntiles : num_cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(KB_TW * KB_TH), (size_t)kTube2LdsBytes, st, a, b);
    return hipGetLastError();
}

hipError_t launch_stage_tube2(const StageArgs &a, const BandArgs &b, int out_mode, int num_cus, hipStream_t st) {
    if (!stage_tube2_supported(a)) return hipErrorInvalidValue;
    if (out_mode == kOutPlanarU4) return launch_tube2_t<kOutPlanarU4>(a, b, num_cus, st);
    if (out_mode == kOutPackedRGBU4 && a.C == 3) return launch_tube2_t<kOutPackedRGBU4>(a, b, num_cus, st);
    return launch_tube2_t<kOutGeneric>(a, b, num_cus, st);
}

// ------------------------------------------------------------------------------------------
// Detailed tiles of the final stage (u == 4): anchor slabs in LDS instead of row gathers from L2.
//
// On detailed content the full-table kernel is bound by its gathers: 60 rows of 16 bytes per sample, nearly every one
// of them a separate 128-byte line from L2 (17-145 cycles per gather instruction per CU).  The anchor of a sample (the
// first key) is the pixel itself in all 12 passes, so the passes of a sample touch only the slab pair of its anchor MSB
// (mulut_core.h "slab pairs": 157,216 bytes, LDS-sized).  The samples (pixel, channel) of the tiles the statistic marked
// detailed are therefore grouped by anchor MSB, on the device and without host synchronisation:
//   detail_bucket_kernel<false>  counts the samples of every anchor MSB (LDS histogram per tile, 16 atomics per tile)
//   detail_plan_kernel           turns the 16 counts into list starts and work items of <= 4096 samples of one anchor
//   detail_bucket_kernel<true>   writes the sample ids (tile << 12 | c << 10 | ty << 6 | tx) into the 16 lists
//   stage_slab_kernel            one persistent workgroup per CU walks the items: per mode it copies the item's slab pair
//                                into LDS (a straight 157 KB copy, L2-resident) and runs the mode's four passes of its
//                                4 samples per thread -- rotation pairs in packed 16-bit halves as in the tube kernel,
//                                rows by ds_read_b128, accumulated from the raw bytes (three operations per dword) -- keeping
//                                the accumulators in registers across the modes; the finished 4x4 block of a sample is
//                                one 16-byte store at blocks[id]
//   detail_retile_kernel         writes the blocks of the detailed tiles to the output image in its layout
// A sample's 5x5 window is read straight from the stage input (L2-resident), 8 unaligned bytes per row from column
// x - 2; pixels in the first 2 / last 6 columns of the image would need edge replication inside those 8 bytes and go to
// the pixel fix-up list (stage_up_fix_kernel) instead.
// ------------------------------------------------------------------------------------------
#if defined(MULUT_VARIANT_slabs3) || defined(MULUT_VARIANT_slabs3nopf)
constexpr int kSlabNT = 1024, kSlabS = 3, kSlabItem = kSlabNT * kSlabS;
#else
constexpr int kSlabNT = 1024, kSlabS = 4, kSlabItem = kSlabNT * kSlabS;
#endif
#if defined(MULUT_VARIANT_slabs3nopf)
#define MULUT_VARIANT_slabnopf 1
#endif
constexpr int kSlabLdsBytes = ((kSlabPairBytes + 1023) / 1024) * 1024;      // whole 1-KiB LDS-DMA pieces: 157,696

__device__ __forceinline__ uint4 lds_u128(uint32_t addr) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = *(const __attribute__((address_space(3))) u32x4 *)(uintptr_t)addr;
    return make_uint4(v.x, v.y, v.z, v.w);
}

// One workgroup: the per-tile histograms of the detailed tiles (thist[tile][16], written by tile_stat_kernel) become
// absolute positions in the id lists (exclusive scan over tiles per anchor MSB, in place), the 16 totals become list
// starts and work items, and the detailed tiles are listed (dlist) -- no atomics, so the lists are deterministic.
__global__ void __launch_bounds__(1024) detail_plan_kernel(DetailArgs d, const uint32_t *verdict, uint32_t ntiles, uint32_t want_items) {
    if (!d.dirty_list && d.ctl[kDetAny] == 0u) return;      // no tile was marked detailed (workgroup-uniform): nothing to plan, fill, compute or retile
    __shared__ uint32_t s_wave[16][17];
    __shared__ uint32_t s_start[17], s_item0[17], s_isz;
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    // thread t takes tiles t, t + 1024, ...: the histogram halves (detail_hist_index) are then read as whole coalesced KiB per
    // wave; the lists follow this order
    const uint4 *half[2] = {(const uint4 *)d.thist, (const uint4 *)d.thist + ntiles};
    uint32_t excl[17];       // this thread's exclusive prefix per anchor MSB (16: detailed tiles)
    {
        uint32_t local[17];
#pragma unroll
        for (int b = 0; b < 17; ++b) local[b] = 0;
        // chunks of 8 tiles per thread, fully unrolled: all 24 loads of a chunk are in flight together (a rolled loop makes a
        // round trip to L2 per iteration, and this is one workgroup)
        for (uint32_t base = 0; base < ntiles; base += 8 * 1024) {
            uint32_t det[8], any = 0;
            uint4 r[8][2];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t t = base + (uint32_t)k * 1024u + threadIdx.x, tc = t < ntiles ? t : ntiles - 1u;
                det[k] = (t < ntiles && verdict[tc] == 1u) ? 1u : 0u;
                any |= det[k];
            }
            if (!__any((int)any)) continue;           // smooth content: no histogram is read at all
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t t = base + (uint32_t)k * 1024u + threadIdx.x, tc = t < ntiles ? t : ntiles - 1u;
                r[k][0] = half[0][tc];
                r[k][1] = half[1][tc];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t w[8] = {r[k][0].x, r[k][0].y, r[k][0].z, r[k][0].w, r[k][1].x, r[k][1].y, r[k][1].z, r[k][1].w};
#pragma unroll
                for (int b = 0; b < 16; ++b) local[b] += det[k] ? ((w[b >> 1] >> (16 * (b & 1))) & 0xFFFFu) : 0u;
                local[16] += det[k];
            }
        }
#pragma unroll
        for (int b = 0; b < 17; ++b) {
            uint32_t inc = local[b];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)inc, o);
                if (lane >= o) inc += up;
            }
            excl[b] = inc - local[b];
            if (lane == 63) s_wave[wave][b] = inc;
        }
    }
    __syncthreads();
    if (threadIdx.x < 17) {          // wave totals -> exclusive wave bases, column totals into s_start
        uint32_t run = 0;
        for (int w = 0; w < 16; ++w) {
            const uint32_t v = s_wave[w][threadIdx.x];
            s_wave[w][threadIdx.x] = run;
            run += v;
        }
        s_start[threadIdx.x] = run;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // item size: whole items (4 samples per thread) when there is work for every workgroup, else fewer samples per thread
        // (a multiple of the workgroup size) so that the few samples still spread over the workgroups
        uint32_t total = 0;
        for (int b = 0; b < 16; ++b) total += s_start[b] + d.ctl[kDetDirty + b];
        uint32_t isz = ((total / (want_items ? want_items : 1u) + kSlabNT - 1) / kSlabNT) * kSlabNT;
        isz = isz < (uint32_t)kSlabNT ? (uint32_t)kSlabNT : isz > (uint32_t)kSlabItem ? (uint32_t)kSlabItem : isz;
        s_isz = isz;
        uint32_t start = 0, item0 = 0;
        for (int b = 0; b < 16; ++b) {
            // the list of anchor MSB b: the samples of the detailed tiles, then the tube kernel's dirty samples
            const uint32_t cnt = s_start[b] + d.ctl[kDetDirty + b];
            d.ctl[kDetCount + b] = cnt;
            d.ctl[kDetStart + b] = start;
            d.ctl[kDetDirtyBase + b] = start + s_start[b];
            s_start[b] = start;
            s_item0[b] = item0;
            start += cnt;
            item0 += (cnt + isz - 1) / isz;
        }
        d.ctl[kDetTiles] = s_start[16];
        d.ctl[kDetItems] = item0;
    }
    __syncthreads();
    for (int b = 0; b < 16; ++b) {
        const uint32_t isz = s_isz, cnt = d.ctl[kDetCount + b], ni = (cnt + isz - 1) / isz;
        for (uint32_t i = threadIdx.x; i < ni; i += 1024) {
            const uint32_t left = cnt - i * isz;
            d.items[2 * (s_item0[b] + i)] = ((uint32_t)b << 28) | (left < isz ? left : isz);
            d.items[2 * (s_item0[b] + i) + 1] = s_start[b] + i * isz;
        }
    }
    {
        uint32_t run[16], slot = s_wave[wave][16] + excl[16];
#pragma unroll
        for (int b = 0; b < 16; ++b) run[b] = s_start[b] + s_wave[wave][b] + excl[b];
        uint4 *quarter[4] = {(uint4 *)d.tpos, (uint4 *)d.tpos + ntiles, (uint4 *)d.tpos + 2 * (size_t)ntiles, (uint4 *)d.tpos + 3 * (size_t)ntiles};
        for (uint32_t base = 0; base < ntiles; base += 8 * 1024) {
            uint32_t det[8], any = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t t = base + (uint32_t)k * 1024u + threadIdx.x, tc = t < ntiles ? t : ntiles - 1u;
                det[k] = (t < ntiles && verdict[tc] == 1u) ? 1u : 0u;
                any |= det[k];
            }
            if (!__any((int)any)) continue;           // no detailed tile in this chunk of the wave: nothing to read or write
            uint4 r[8][2];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t t = base + (uint32_t)k * 1024u + threadIdx.x, tc = t < ntiles ? t : ntiles - 1u;
                r[k][0] = half[0][tc];
                r[k][1] = half[1][tc];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (det[k]) {
                    const uint32_t t = base + (uint32_t)k * 1024u + threadIdx.x;
                    const uint32_t w[8] = {r[k][0].x, r[k][0].y, r[k][0].z, r[k][0].w, r[k][1].x, r[k][1].y, r[k][1].z, r[k][1].w};
                    uint32_t o[16];
#pragma unroll
                    for (int b = 0; b < 16; ++b) { o[b] = run[b]; run[b] += (w[b >> 1] >> (16 * (b & 1))) & 0xFFFFu; }
                    quarter[0][t] = make_uint4(o[0], o[1], o[2], o[3]);
                    quarter[1][t] = make_uint4(o[4], o[5], o[6], o[7]);
                    quarter[2][t] = make_uint4(o[8], o[9], o[10], o[11]);
                    quarter[3][t] = make_uint4(o[12], o[13], o[14], o[15]);
                    d.dlist[slot++] = t;
                }
        }
    }
}

// ids and descriptors of the samples of the listed tiles, written at the positions detail_plan_kernel assigned (the rank of
// a sample inside its tile's share of a list comes from an LDS counter); pixels in the image's border columns go to the
// pixel fix-up list instead
__global__ void __launch_bounds__(256) detail_fill_kernel(StageArgs a, DetailArgs d) {
    if (!d.dirty_list && d.ctl[kDetAny] == 0u) return;      // no tile was marked detailed (workgroup-uniform): nothing to plan, fill, compute or retile
    constexpr int TW = KB_TW, TH = KB_TH, PER = 3 * TW * TH / 256;
    static_assert(TW == 64 && TH == 16, "sample ids assume the 64x16 verdict tile");
    __shared__ uint32_t s_rank[16], s_base[16], s_fix[2];
    const uint32_t ndet = d.ctl[kDetTiles];
    const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
    for (uint32_t li = blockIdx.x; li < ndet; li += gridDim.x) {
        const int tile = (int)d.dlist[li];
        int n, y0, x0;
        decode_tile(a, tile, n, y0, x0, TW, TH);
        __syncthreads();              // the previous tile's counters are no longer read
        if (threadIdx.x < 16) {
            s_rank[threadIdx.x] = 0;
            s_base[threadIdx.x] = d.tpos[detail_pos_index((uint32_t)tile, (uint32_t)(a.N * a.tiles_x * a.tiles_y), (int)threadIdx.x)];
        }
        if (threadIdx.x == 16) s_fix[0] = 0;
        __syncthreads();
        uint32_t pos[PER], fixr[4];
        uint8_t val[PER];
        // every byte of the thread in flight before the first LDS atomic (behind the per-sample condition each load would be
        // a round trip of its own: twelve in a row)
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int s = (int)threadIdx.x + k * 256;
            const int c = imin(s >> 10, a.C - 1), y = imin(y0 + ((s >> 6) & 15), a.oy1 - 1), x = imin(x0 + (s & 63), a.W - 1);
            val[k] = *view_addr(a.in, n, c, y, x);
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int s = (int)threadIdx.x + k * 256;
            const int c = s >> 10, ty = (s >> 6) & 15, tx = s & 63;
            const int y = y0 + ty, x = x0 + tx;
            const bool inside = c < a.C && y < a.oy1 && x < a.W;
            const bool slab = inside && x >= kSlabXLo && x < a.W - slab_x_hi(a);
            pos[k] = 0xFFFFFFFFu;
            if (slab) {
                const uint32_t h = (uint32_t)(val[k] >> 4);
                pos[k] = s_base[h] + atomicAdd(&s_rank[h], 1u);
            }
            if (k < 4) fixr[k] = (inside && !slab) ? atomicAdd(&s_fix[0], 1u) : 0xFFFFFFFFu;      // k < 4 <=> channel 0: each pixel once
        }
        __syncthreads();
        if (threadIdx.x == 0 && s_fix[0]) s_fix[1] = atomicAdd(a.fix_count, s_fix[0]);      // one atomic per tile that has border pixels
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (fixr[k] != 0xFFFFFFFFu) {
                const int s = (int)threadIdx.x + k * 256;
                a.fix_list[s_fix[1] + fixr[k]] = (uint32_t)((n * a.H + y0 + ((s >> 6) & 15)) * a.W + x0 + (s & 63)) | (3u << 30);
            }
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (pos[k] != 0xFFFFFFFFu) {
                const int s = (int)threadIdx.x + k * 256;
                const int c = s >> 10, y = y0 + ((s >> 6) & 15), x = x0 + (s & 63);
                d.desc[pos[k]] = (uint32_t)(view_addr(a.in, n, c, y, x - 2) - a.in.p) | ((uint32_t)imin(y - ylo, 2) << 28) | ((uint32_t)imin(yhi - y, 2) << 30);
            }
    }
}

// The tube kernel's dirty samples (d.dirty_list: pixel id | channel << 30; their passes left the tube) join the lists of the
// anchor-slab kernel instead of being recomputed by gathers from the full tables:
//   dirty_count_kernel    samples per anchor MSB (LDS histogram per workgroup, 16 atomics per workgroup) -> ctl[kDetDirty..]
//   detail_plan_kernel    reserves their places behind the detailed tiles' samples of the same anchor   -> ctl[kDetDirtyBase..]
//   dirty_scatter_kernel  writes their descriptors there (rank from an LDS counter + one atomic per workgroup, round and anchor)
//   dirty_retile_kernel   stores each finished block's channel into the output image
// Samples in the image's border columns go to the pixel fix-up list (a.fix_list), as those of the detailed tiles do.
__device__ __forceinline__ void dirty_decode(const StageArgs &a, uint32_t ent, int &n, int &c, int &y, int &x) {
    const uint32_t id = ent & 0x3FFFFFFFu;
    c = (int)(ent >> 30);
    x = (int)(id % (uint32_t)a.W);
    y = (int)((id / (uint32_t)a.W) % (uint32_t)a.H);
    n = (int)(id / ((uint32_t)a.W * (uint32_t)a.H));
}
__global__ void __launch_bounds__(256) dirty_count_kernel(StageArgs a, DetailArgs d) {
    __shared__ uint32_t s_hist[16];
    if (threadIdx.x < 16) s_hist[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t count = *d.dirty_count;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
        int n, c, y, x;
        dirty_decode(a, d.dirty_list[i], n, c, y, x);
        if (x >= kSlabXLo && x < a.W - slab_x_hi(a)) atomicAdd(&s_hist[*view_addr(a.in, n, c, y, x) >> 4], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 16 && s_hist[threadIdx.x]) atomicAdd(&d.ctl[kDetDirty + threadIdx.x], s_hist[threadIdx.x]);
}
__global__ void __launch_bounds__(256) dirty_scatter_kernel(StageArgs a, DetailArgs d) {
    __shared__ uint32_t s_cnt[16], s_base[16];
    const uint32_t count = *d.dirty_count;
    const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
    for (uint32_t i0 = blockIdx.x * 256u; i0 < count; i0 += gridDim.x * 256u) {      // workgroup-uniform rounds
        __syncthreads();
        if (threadIdx.x < 16) s_cnt[threadIdx.x] = 0;
        __syncthreads();
        const uint32_t i = i0 + threadIdx.x;
        uint32_t h = 16, rank = 0, desc = 0;
        if (i < count) {
            const uint32_t ent = d.dirty_list[i];
            int n, c, y, x;
            dirty_decode(a, ent, n, c, y, x);
            if (x >= kSlabXLo && x < a.W - slab_x_hi(a)) {
                h = (uint32_t)(*view_addr(a.in, n, c, y, x) >> 4);
                rank = atomicAdd(&s_cnt[h], 1u);
                desc = (uint32_t)(view_addr(a.in, n, c, y, x - 2) - a.in.p) | ((uint32_t)imin(y - ylo, 2) << 28) | ((uint32_t)imin(yhi - y, 2) << 30);
            } else {
                a.fix_list[atomicAdd(a.fix_count, 1u)] = ent;          // border column (rare)
            }
        }
        __syncthreads();
        if (threadIdx.x < 16) s_base[threadIdx.x] = d.ctl[kDetDirtyBase + threadIdx.x] + (s_cnt[threadIdx.x] ? atomicAdd(&d.ctl[kDetDirtyCursor + threadIdx.x], s_cnt[threadIdx.x]) : 0u);
        __syncthreads();
        if (h < 16) d.desc[s_base[h] + rank] = desc;
    }
}
template <int OUT>
__global__ void __launch_bounds__(256) dirty_retile_kernel(StageArgs a, DetailArgs d) {
    const uint32_t count = *d.dirty_count;
    for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < count; i += gridDim.x * 256u) {
        int n, c, y, x;
        dirty_decode(a, d.dirty_list[i], n, c, y, x);
        if (x < kSlabXLo || x >= a.W - slab_x_hi(a)) continue;
        const uint4 v = d.blocks[(size_t)(view_addr(a.in, n, c, y, x) - a.in.p)];
        const uint32_t o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int sy = 0; sy < 4; ++sy) {
            if constexpr (OUT == kOutPlanarU4) {
                *(uint32_t *)const_cast<uint8_t *>(view_addr(a.out, n, c, y * 4 + sy, x * 4)) = o[sy];
            } else {
#pragma unroll
                for (int sx = 0; sx < 4; ++sx) *const_cast<uint8_t *>(view_addr(a.out, n, c, y * 4 + sy, x * 4 + sx)) = (uint8_t)(o[sy] >> (8 * sx));
            }
        }
    }
}

// accumulators of one sample: raw (F) and odd-byte (H) sums of the rotation pairs (0, 2) and (1, 3)
struct SlabAcc {
    uint32_t F02[4], H02[4], F13[4], H13[4];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < 4; ++k) F02[k] = H02[k] = F13[k] = H13[k] = 0;
    }
    template <int R, int HALF>
    __device__ __forceinline__ void mac_row(const uint4 &row, uint32_t wpk) {
        const uint32_t rd[4] = {row.x, row.y, row.z, row.w};
        static_for<0, 4>([&](auto K) {
            constexpr int k = K;
            if constexpr (R == 0) { pk_mac<HALF, false>(F02[k], rd[k], wpk); pk_mac<HALF, false>(H02[k], slab_odd_bytes(rd[k]), wpk); }
            if constexpr (R == 1) { pk_mac<HALF, false>(F13[k], rd[k], wpk); pk_mac<HALF, false>(H13[k], slab_odd_bytes(rd[k]), wpk); }
            if constexpr (R == 2) { pk_mac<HALF, false>(F02[3 - k], slab_rev_bytes(rd[k]), wpk); pk_mac<HALF, false>(H02[3 - k], slab_rev_odd_bytes(rd[k]), wpk); }
            if constexpr (R == 3) { pk_mac<HALF, false>(F13[3 - k], slab_rev_bytes(rd[k]), wpk); pk_mac<HALF, false>(H13[3 - k], slab_rev_odd_bytes(rd[k]), wpk); }
        });
    }
    __device__ __forceinline__ void to_fields(RotAcc<4> &r) const {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            r.lo02[k] = slab_even_sums(F02[k], H02[k]); r.hi02[k] = H02[k];
            r.lo13[k] = slab_even_sums(F13[k], H13[k]); r.hi13[k] = H13[k];
        }
    }
};

// dst = (16-bit half HALF of x) * 16 + acc: one v_mad_u32_u16, the half picked by op_sel
template <int HALF>
__device__ __forceinline__ uint32_t mad16_half(uint32_t x, uint32_t acc) {
    uint32_t r;
    if constexpr (HALF == 0) asm("v_mad_u32_u16 %0, %1, 16, %2 op_sel:[0,0,0,0]" : "=v"(r) : "v"(x), "v"(acc));
    else asm("v_mad_u32_u16 %0, %1, 16, %2 op_sel:[1,0,0,0]" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

// LDS byte addresses of rows 0..3 of the pass in half HALF: 16 * (running sum of the path's unit steps); row 4 = row 0 + kSlabAll * 16
template <int HALF>
__device__ __forceinline__ void slab_row_addrs(const SlabPair &sp, uint32_t (&ad)[4]) {
#if !defined(MULUT_VARIANT_slabmad16)
    ad[0] = HALF ? (sp.base >> 16) : (sp.base & 0xFFFFu);
#pragma unroll
    for (int j = 0; j < 3; ++j) ad[j + 1] = add_word<HALF>(ad[j], sp.step[j]);
#pragma unroll
    for (int j = 0; j < 4; ++j) ad[j] <<= 4;
#else
    ad[0] = mad16_half<HALF>(sp.base, 0u);
#pragma unroll
    for (int j = 0; j < 3; ++j) ad[j + 1] = mad16_half<HALF>(sp.step[j], ad[j]);
#endif
}
template <int J>
__device__ __forceinline__ uint4 slab_row(const uint32_t (&ad)[4]) {
#if defined(MULUT_ABLATE) && MULUT_ABLATE == 51   /* timing-only: every lane of a wave reads row J's address of lane 0 (no bank conflicts) */
    return lds_u128((uint32_t)__builtin_amdgcn_readfirstlane((int)ad[J < 4 ? J : 0]) + (uint32_t)(J < 4 ? 0 : kSlabAll * 16) + ((threadIdx.x & 15u) << 4));
#else
    return lds_u128(ad[J < 4 ? J : 0] + (uint32_t)(J < 4 ? 0 : kSlabAll * 16));
#endif
}

// Both passes of a rotation pair (R in the low halves of sp, R + 2 in the high halves) from the slab pair at LDS address 0.
// The second pass's rows are requested one by one as the first pass's rows are consumed -- into the registers those free --
// so the LDS latency of every second pass is covered by accumulation instead of being waited for.
template <int R>
__device__ __forceinline__ void slab_pair_rows(const SlabPair &sp, SlabAcc &acc) {
    uint32_t a0[4], a1[4];
    slab_row_addrs<0>(sp, a0);
    slab_row_addrs<1>(sp, a1);
#if defined(MULUT_VARIANT_slabseq)
    uint4 r0[5], r1[5];
    static_for<0, 5>([&](auto J) { r0[J] = slab_row<J>(a0); });
    static_for<0, 5>([&](auto J) { acc.template mac_row<R, 0>(r0[J], sp.w[J]); });
    static_for<0, 5>([&](auto J) { r1[J] = slab_row<J>(a1); });
    static_for<0, 5>([&](auto J) { acc.template mac_row<R + 2, 1>(r1[J], sp.w[J]); });
#else
    uint4 r0[5], r1[5];
    static_for<0, 5>([&](auto J) { r0[J] = slab_row<J>(a0); });
    static_for<0, 5>([&](auto J) {
        acc.template mac_row<R, 0>(r0[J], sp.w[J]);
        r1[J] = slab_row<J>(a1);
        __builtin_amdgcn_sched_barrier(0);      // keep the order: row J of the first pass consumed, row J of the second requested
    });
    static_for<0, 5>([&](auto J) { acc.template mac_row<R + 2, 1>(r1[J], sp.w[J]); });
#endif
}

// The part of a sample's 5x5 window a pattern touches.  Per window row: nothing, one dword from column x - 1
// (columns -1..2), or 8 bytes from column x - 2 (columns -2..5).  Row 0 is always 8 bytes (the anchor is byte 2 of it).
//   s: rows -1..1, columns -1..1;  d: rows -2/0/2, columns -2/0/2;  y: rows +-2 columns +-1, rows +-1 columns +-1, +-2
template <int PAT>
__host__ __device__ constexpr int slab_row_kind(int dy) {
    return dy == 0 ? 2 : PAT == 0 ? ((dy == 1 || dy == -1) ? 1 : 0) : PAT == 1 ? ((dy & 1) == 0 ? 2 : 0) : ((dy == 1 || dy == -1) ? 2 : 1);
}
struct SlabWin {
    uint32_t lo[5], hi[5];
    // the byte at window offset (dy, dx) of pattern PAT's loads: register and byte index
    template <int PAT, int DY, int DX>
    __device__ __forceinline__ uint32_t reg() const {
        constexpr int kind = slab_row_kind<PAT>(DY);
        static_assert(kind != 0 && (kind == 2 || (DX >= -1 && DX <= 2)), "window byte not loaded for this pattern");
        constexpr int idx = kind == 1 ? DX + 1 : DX + 2;
        return idx < 4 ? lo[DY + 2] : hi[DY + 2];
    }
    template <int PAT, int DY, int DX>
    static constexpr int byte_idx() { return (slab_row_kind<PAT>(DY) == 1 ? DX + 1 : DX + 2) & 3; }
};

// sample descriptor: byte offset of (n, c, y, x - 2) in the stage input (< 2^28) | min(y - ylo, 2) << 28 | min(yhi - y, 2) << 30
template <int PAT>
__device__ __forceinline__ void slab_load_window_t(const StageArgs &a, uint32_t desc, SlabWin &w) {
    const int top = (int)((desc >> 28) & 3u), bot = (int)(desc >> 30);
    const uint8_t *p0 = a.in.p + (desc & 0x0FFFFFFFu);
    static_for<0, 5>([&](auto RW) {
        constexpr int r = RW, dy = r - 2, kind = slab_row_kind<PAT>(dy);
        // every register of the window is assigned on every pattern's path (rows the pattern does not touch: zero): stores
        // to different elements in the three branches would be merged into one indexed store, i.e. the window put in scratch
        w.lo[r] = 0;
        w.hi[r] = 0;
        if constexpr (kind != 0) {
            const int dyc = dy < 0 ? -imin(-dy, top) : imin(dy, bot);       // edge replication at the true image borders
            const uint8_t *p = p0 + dyc * a.in.sY;
#if defined(MULUT_ABLATE) && MULUT_ABLATE == 52   /* timing-only: no window loads */
            w.lo[r] = (uint32_t)(uintptr_t)p * 0x9E3779B1u;
            w.hi[r] = w.lo[r] >> 7;
#else
            if constexpr (kind == 1) {
                uint32_t v;
                __builtin_memcpy(&v, p + 1, 4);
                w.lo[r] = v;
            } else {
                uint2 v;
                __builtin_memcpy(&v, p, 8);
                w.lo[r] = v.x;
                w.hi[r] = v.y;
            }
#endif
        }
    });
}
__device__ __forceinline__ void slab_load_window(const StageArgs &a, int pat, uint32_t desc, SlabWin &w) {
    asm volatile("" : "+v"(desc));      // opaque: the row addresses are rebuilt here every time (hoisted out of the mode loop they would be parked in scratch)
    if (pat == 0) slab_load_window_t<0>(a, desc, w);          // scalar branches
    else if (pat == 1) slab_load_window_t<1>(a, desc, w);
    else slab_load_window_t<2>(a, desc, w);
}

// neighbour K of pattern PAT: rotation R's byte in the low half, rotation R + 2's (the opposite offset) in the high half
template <int PAT, int R, int K>
__device__ __forceinline__ uint32_t slab_nb(const SlabWin &w) {
    constexpr int dy = rot_dy(R, kPatDi[PAT][K], kPatDj[PAT][K]), dx = rot_dx(R, kPatDi[PAT][K], kPatDj[PAT][K]);
    constexpr uint32_t sel = 0x0C000C00u | ((uint32_t)(4 + SlabWin::byte_idx<PAT, -dy, -dx>()) << 16) | (uint32_t)SlabWin::byte_idx<PAT, dy, dx>();
    return __builtin_amdgcn_perm(w.template reg<PAT, -dy, -dx>(), w.template reg<PAT, dy, dx>(), sel);
}
template <int PAT, int R>
__device__ __forceinline__ void slab_pair_index(const SlabWin &w, uint32_t k0, SlabPair &sp) {
    simplex4_slab_pair(k0, slab_nb<PAT, R, 0>(w), slab_nb<PAT, R, 1>(w), slab_nb<PAT, R, 2>(w), sp);
}

// All four passes of one sample and mode.  Only the index math is specific to the pattern (a scalar switch per rotation
// pair, ~45 instructions each); the row reads and the accumulation -- most of the code -- are shared by the patterns, which
// keeps the loop over an item (4 samples x 3 modes) inside the instruction cache: with the pattern as a template
// parameter of the whole body the item loop was 77 KB of code and ran 1.5x slower.
__device__ __forceinline__ void slab_sample(int pat, const SlabWin &w, SlabAcc &acc) {
    uint32_t k0 = slab_anchor_key((w.lo[2] >> 16) & 0xFFu);
    static_for<0, 2>([&](auto RR) {
        constexpr int R = RR;
        SlabPair sp;
        if (pat == 0) slab_pair_index<0, R>(w, k0, sp);
        else if (pat == 1) slab_pair_index<1, R>(w, k0, sp);
        else slab_pair_index<2, R>(w, k0, sp);
        slab_pair_rows<R>(sp, acc);
        // one pair at a time: the next pair's index math must not be scheduled into this one (VGPR budget)
        asm volatile("" : "+v"(acc.F02[0]), "+v"(acc.F13[0]), "+v"(k0));
    });
}

#if defined(MULUT_VARIANT_slabclk)
#define SLAB_CLK(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); clk[i] += t_ - clk_last; clk_last = t_; } while (0)
#define SLAB_CLK_ARGS , unsigned long long (&clk)[4], unsigned long long &clk_last
#define SLAB_CLK_PASS , clk, clk_last
#else
#define SLAB_CLK(i) do { } while (0)
#define SLAB_CLK_ARGS
#define SLAB_CLK_PASS
#endif
// one mode of one item: the slab pair into LDS (LDS-DMA: 154 pieces of 1 KiB, wave w takes pieces w, w + 16, ...), then the
// mode's four passes of the thread's samples; the window of the next sample is in flight while the current one is computed,
// the first one while the slab pair is copied
__device__ __forceinline__ void slab_mode(const StageArgs &a, int pat, uint32_t cnt, const uint8_t *pair, bool in_lds, uint8_t *smem, const uint32_t (&desc)[kSlabS], SlabAcc (&acc)[kSlabS] SLAB_CLK_ARGS) {
    SlabWin wa, wb;
    slab_load_window(a, pat, desc[0], wa);
    if (!in_lds) {                     // workgroup-uniform
    __syncthreads();                   // everyone is done with the previous slab pair
    SLAB_CLK(0);                       // waiting for the slowest wave of the previous mode
    {
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
        constexpr int kPieces = kSlabLdsBytes / 1024;
#pragma unroll
        for (int k = 0; k < (kPieces + 15) / 16; ++k) {
            const int piece = wave + 16 * k;
            if (piece < kPieces)       // wave-uniform
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pair + piece * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(smem + piece * 1024), 16, 0, 0);
        }
    }
    __syncthreads();
    }
    SLAB_CLK(1);                       // slab pair copy
#if defined(MULUT_VARIANT_slabnopf)
    static_for<0, kSlabS>([&](auto S) {
        constexpr int s = S;
        if constexpr (s > 0) slab_load_window(a, pat, desc[s], wa);
        slab_sample(pat, wa, acc[s]);
    });
#else
    static_for<0, kSlabS>([&](auto S) {
        constexpr int s = S;
        if ((uint32_t)s * kSlabNT < cnt) {          // workgroup-uniform: a short item leaves sample slots empty
            if constexpr (s + 1 < kSlabS) slab_load_window(a, pat, desc[s + 1], (s & 1) ? wa : wb);
            slab_sample(pat, (s & 1) ? wb : wa, acc[s]);
        }
    });
#endif
    SLAB_CLK(2);                       // the mode's passes
}

__global__ void __launch_bounds__(kSlabNT) stage_slab_kernel(StageArgs a, DetailArgs d) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    if (!d.dirty_list && d.ctl[kDetAny] == 0u) return;      // no tile was marked detailed (workgroup-uniform)
    if (lds_addr_of(smem) != 0u) __builtin_trap();      // the row reads assume the slab pair starts at LDS address 0 (no static LDS here): fail loudly, never leave blocks unwritten
    const uint32_t nitems = d.ctl[kDetItems];
#if defined(MULUT_VARIANT_slabclk)
    unsigned long long clk[4] = {0, 0, 0, 0}, clk_last = __builtin_amdgcn_s_memtime();
#endif
    const uint8_t *resident = nullptr;      // the slab pair in LDS
    bool snake = false;
    for (uint32_t it = blockIdx.x; it < nitems; it += gridDim.x) {
        const uint32_t hdr = d.items[2 * it], first = d.items[2 * it + 1];
        const uint32_t h = hdr >> 28, cnt = hdr & 0x0FFFFFFFu;
        uint32_t desc[kSlabS];
#pragma unroll
        for (int s = 0; s < kSlabS; ++s) {
            const uint32_t i = (uint32_t)s * kSlabNT + threadIdx.x;
            desc[s] = d.desc[first + (i < cnt ? i : cnt - 1u)];          // surplus lanes recompute the last sample (never stored)
        }
        SlabAcc acc[kSlabS];
#pragma unroll
        for (int s = 0; s < kSlabS; ++s) acc[s].clear();
        // the modes in alternating order from item to item: consecutive items of a workgroup mostly share their anchor, and
        // the pair the last mode left in LDS then serves the next item's first mode (the sums do not depend on the order)
        for (int mv = 0; mv < a.M; ++mv) {
            const int m = __builtin_amdgcn_readfirstlane(snake ? a.M - 1 - mv : mv);
            const int pat = a.dj[m][0] == 2 ? 1 : a.di[m][0] == 1 ? 2 : 0;     // scalar
            const uint8_t *pair = d.slab[m] + (size_t)h * kSlabPairBytes;
            SLAB_CLK(3);                   // item set-up / epilogue / stores
#if defined(MULUT_ABLATE) && MULUT_ABLATE == 54   /* timing-only: the slab pair is copied once per workgroup */
            slab_mode(a, pat, cnt, pair, resident != nullptr, smem, desc, acc SLAB_CLK_PASS);
#else
            slab_mode(a, pat, cnt, pair, pair == resident, smem, desc, acc SLAB_CLK_PASS);
#endif
            resident = pair;
        }
        snake = !snake;
#pragma unroll
        for (int s = 0; s < kSlabS; ++s) {
            RotAcc<4> r;
            acc[s].to_fields(r);
            uint32_t o[4];
#if defined(MULUT_ABLATE) && MULUT_ABLATE == 53   /* timing-only: no divide / round / clip */
            o[0] = r.lo02[0] ^ r.lo13[0]; o[1] = r.lo02[1] ^ r.hi13[1]; o[2] = r.hi02[2] ^ r.lo13[2]; o[3] = r.hi02[3] ^ r.hi13[3];
#else
            tube_finish_rows(a, r, o);
#endif
            const uint32_t i = (uint32_t)s * kSlabNT + threadIdx.x;
            if (i < cnt) d.blocks[(desc[s] & 0x0FFFFFFFu) + 2u] = make_uint4(o[0], o[1], o[2], o[3]);      // indexed by the sample's byte offset in the stage input
        }
    }
#if defined(MULUT_VARIANT_slabclk)   /* probe build: shader-clock ticks / 1024 of wave 0 of every workgroup per phase, summed into ctl[48..51] */
    SLAB_CLK(3);
    if (threadIdx.x == 0)
        for (int k = 0; k < 4; ++k) atomicAdd(&d.ctl[48 + k], (uint32_t)(clk[k] >> 10));
#endif
}

template <int OUT>
__global__ void __launch_bounds__(KB_TW *KB_TH) detail_retile_kernel(StageArgs a, DetailArgs d) {
    if (!d.dirty_list && d.ctl[kDetAny] == 0u) return;      // no tile was marked detailed (workgroup-uniform)
    const uint32_t ndet = d.ctl[kDetTiles];
    const int tx = threadIdx.x % KB_TW, ty = threadIdx.x / KB_TW;
    for (uint32_t li = blockIdx.x; li < ndet; li += gridDim.x) {
    const int tile = (int)d.dlist[li];
    int n, y0, x0;
    decode_tile(a, tile, n, y0, x0, KB_TW, KB_TH);
    const int y = y0 + ty, x = x0 + tx;
    if (y >= a.oy1 || x >= a.W || x < kSlabXLo || x >= a.W - slab_x_hi(a)) continue;      // border columns: the fix-up kernel's
    const size_t id = (size_t)(view_addr(a.in, n, 0, y, x) - a.in.p), cs = (size_t)a.in.sC;      // block index = byte offset in the stage input
    if constexpr (OUT == kOutPackedRGBU4) {
        const uint4 r = d.blocks[id], g = d.blocks[id + cs], b = d.blocks[id + 2 * cs];
        const uint32_t oR[4] = {r.x, r.y, r.z, r.w}, oG[4] = {g.x, g.y, g.z, g.w}, oB[4] = {b.x, b.y, b.z, b.w};
        store_rgb<4>(a, n, y, x, oR, oG, oB);
    } else {
        for (int c = 0; c < a.C; ++c) {
            const uint4 v = d.blocks[id + (size_t)c * cs];
            const uint32_t o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int sy = 0; sy < 4; ++sy) {
                if constexpr (OUT == kOutPlanarU4) {
                    *(uint32_t *)const_cast<uint8_t *>(view_addr(a.out, n, c, y * 4 + sy, x * 4)) = o[sy];
                } else {
#pragma unroll
                    for (int sx = 0; sx < 4; ++sx)
                        *const_cast<uint8_t *>(view_addr(a.out, n, c, y * 4 + sy, x * 4 + sx)) = (uint8_t)(o[sy] >> (8 * sx));
                }
            }
        }
    }
    }
}

bool detail_slab_supported(const StageArgs &a) {
    const long long tiles = (long long)a.N * a.tiles_x * a.tiles_y;      // 64x16 verdict tiling
    const unsigned long long bytes = (unsigned long long)a.N * (unsigned long long)(a.in.sN < 0 ? -a.in.sN : a.in.sN);
    return a.C <= 3 && a.M <= 3 && a.in.sX == 1 && tiles > 0 && tiles < (1ll << 20) &&
           bytes < (1ull << 28) && (unsigned long long)a.N * a.H * a.W < (1ull << 32);
}
size_t detail_ids_count(const StageArgs &a) { return (size_t)a.N * a.tiles_x * a.tiles_y * 3 * KB_TW * KB_TH; }
size_t detail_items_max(const StageArgs &a) { return detail_ids_count(a) / kSlabItem + 16 + 8192; }      // + the small-item case of detail_plan_kernel (<= 2 x CUs x 4 items)
size_t detail_blocks_count(const StageArgs &a) { return (size_t)a.N * (size_t)(a.in.sN < 0 ? -a.in.sN : a.in.sN); }      // one per byte of the stage input

// the detailed tiles (a.verdict[tile] == 1, histograms in d.thist from launch_tile_stat) of a u == 4 final stage
hipError_t launch_detail_slab(const StageArgs &a, const DetailArgs &d, int out_mode, int num_cus, hipStream_t st) {
    if (!detail_slab_supported(a) || !a.verdict || !a.fix_list || !a.fix_count) return hipErrorInvalidValue;
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)stage_slab_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    const unsigned tiles = (unsigned)((long long)a.N * a.tiles_x * a.tiles_y);
    const unsigned walk = tiles < (unsigned)(8 * num_cus) ? tiles : (unsigned)(8 * num_cus);      // workgroups walking the list of detailed tiles
    const bool dirty = d.dirty_list != nullptr;          // the tube kernel's dirty samples join the lists (d.ctl zeroed by the caller)
    if (dirty) hipLaunchKernelGGL(dirty_count_kernel, dim3((unsigned)num_cus), dim3(256), 0, st, a, d);
    hipLaunchKernelGGL(detail_plan_kernel, dim3(1), dim3(1024), 0, st, d, (const uint32_t *)a.verdict, tiles, (uint32_t)num_cus);      // few samples: about one item per workgroup (an item's three slab copies make a second round dearer than longer items)
    hipLaunchKernelGGL(detail_fill_kernel, dim3(tiles < 2 * walk ? tiles : 2 * walk), dim3(256), 0, st, a, d);
    if (dirty) hipLaunchKernelGGL(dirty_scatter_kernel, dim3((unsigned)num_cus), dim3(256), 0, st, a, d);
    #if defined(MULUT_VARIANT_slablds64)
    hipLaunchKernelGGL(stage_slab_kernel, dim3((unsigned)num_cus), dim3(kSlabNT), (size_t)65536, st, a, d);      // experiment: empty-launch cost against the LDS size (wrong results when there are items)
#else
    hipLaunchKernelGGL(stage_slab_kernel, dim3((unsigned)num_cus), dim3(kSlabNT), (size_t)kSlabLdsBytes, st, a, d);
#endif
    if (out_mode == kOutPlanarU4) hipLaunchKernelGGL(detail_retile_kernel<kOutPlanarU4>, dim3(walk), dim3(KB_TW * KB_TH), 0, st, a, d);
    else if (out_mode == kOutPackedRGBU4 && a.C == 3) hipLaunchKernelGGL(detail_retile_kernel<kOutPackedRGBU4>, dim3(walk), dim3(KB_TW * KB_TH), 0, st, a, d);
    else hipLaunchKernelGGL(detail_retile_kernel<kOutGeneric>, dim3(walk), dim3(KB_TW * KB_TH), 0, st, a, d);
    if (dirty) {
        if (out_mode == kOutPlanarU4) hipLaunchKernelGGL(dirty_retile_kernel<kOutPlanarU4>, dim3((unsigned)(2 * num_cus)), dim3(256), 0, st, a, d);
        else hipLaunchKernelGGL(dirty_retile_kernel<kOutGeneric>, dim3((unsigned)(2 * num_cus)), dim3(256), 0, st, a, d);
    }
    return hipGetLastError();
}

}  // namespace mulut
