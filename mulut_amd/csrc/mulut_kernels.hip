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

__device__ __forceinline__ void decode_tile(const StageArgs &a, int &n, int &y0, int &x0, int TW, int TH) {
    int b = blockIdx.x;
    const int tx = b % a.tiles_x;
    b /= a.tiles_x;
    const int ty = b % a.tiles_y;
    n = b / a.tiles_y;
    y0 = a.oy0 + ty * TH;
    x0 = tx * TW;
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
    decode_tile(a, n, y0, x0, TW, TH);
    load_tile<TW, TH, NT>(a, n, y0, x0, s_img);

    const int nsamp = a.C * TH * TW;
    int acc[SPT];
#pragma unroll
    for (int k = 0; k < SPT; ++k) acc[k] = 0;

    for (int m = 0; m < a.M; ++m) {
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
            const int s = threadIdx.x + k * NT;
            if (s < nsamp) {
                const int tx = s % TW;
                const int ty = (s / TW) % TH;
                const int c = s / (TW * TH);
                const uint8_t *ctr = s_img + c * (PH * PW) + (ty + kHalo) * PW + (tx + kHalo);
                const int va = ctr[0];
                int sum = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    int idx[5], w[5];
                    simplex4(va, ctr[off[r][0]], ctr[off[r][1]], ctr[off[r][2]], idx, w);
#if MULUT_ABLATE == 11
#pragma unroll
                    for (int j = 0; j < 5; ++j) idx[j] &= (a.N >> 30);
#endif
#if MULUT_ABLATE == 12
#pragma unroll
                    for (int j = 0; j < 5; ++j) sum += w[j] * idx[j];
#else
#pragma unroll
                    for (int j = 0; j < 5; ++j) sum += w[j] * (int)s_lut[idx[j]];
#endif
                }
                acc[k] += sum;
            }
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

constexpr int K1_TW = 64, K1_TH = 32, K1_NT = 1024, K1_SPT = 6;  // 3 ch * 64*32 / 1024 = 6
static_assert(K1_SPT * K1_NT >= 3 * K1_TW * K1_TH, "SPT too small for 3 channels");

void stage_u1_tile(int &tw, int &th) { tw = K1_TW; th = K1_TH; }
const char *stage_u1_name() { return "stage_u1_kernel"; }

hipError_t launch_stage_u1(const StageArgs &a, hipStream_t st) {
    if (a.C > 3) return hipErrorInvalidValue;
    auto kern = stage_u1_kernel<K1_TW, K1_TH, K1_NT, K1_SPT>;
    const size_t lds = (size_t)kU1TableBytes + (size_t)a.C * (K1_TH + 2 * kHalo) * (K1_TW + 2 * kHalo);
    static bool attr_set[64] = {};  // per device: >64 KB of dynamic LDS has to be opted into
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 120 * 1024);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    const long long nb = (long long)a.N * a.tiles_x * a.tiles_y;
    if (nb <= 0 || nb > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(kern, dim3((unsigned)nb), dim3(K1_NT), lds, st, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// K2: final stage, u*u-byte rows gathered from the L2-resident tables.
// One thread = one LR pixel, channels in sequence.  Per rotation the 15 (= M*5) weighted rows are
// accumulated as 16-bit fields, two per dword:  lo[k] holds row elements 4k and 4k+2, hi[k] holds
// 4k+1 and 4k+3 (each table byte is value+128, so every field stays non-negative:
// 4 rot * M * 16 * 255 < 65536 for M <= 4).
// ------------------------------------------------------------------------------------------
template <int U>
__device__ __forceinline__ void load_row(const void *lut, int idx, uint32_t (&row)[row_dwords(U)]) {
    constexpr int RW = row_dwords(U);
    if constexpr (RW == 4) {
        const uint4 v = ((const uint4 *)lut)[idx];
        row[0] = v.x; row[1] = v.y; row[2] = v.z; row[3] = v.w;
    } else {
        const uint32_t *p = (const uint32_t *)lut + (long long)idx * RW;
#pragma unroll
        for (int k = 0; k < RW; ++k) row[k] = p[k];
    }
}

template <int U, int OUT, int TW, int TH>
__global__ void __launch_bounds__(TW *TH) stage_up_kernel(StageArgs a) {
    constexpr int PW = TW + 2 * kHalo, PH = TH + 2 * kHalo;
    constexpr int NT = TW * TH;
    constexpr int RW = row_dwords(U);
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint8_t *s_img = smem;

    int n, y0, x0;
    decode_tile(a, n, y0, x0, TW, TH);
    load_tile<TW, TH, NT>(a, n, y0, x0, s_img);
    __syncthreads();

    const int tx = threadIdx.x % TW, ty = threadIdx.x / TW;
    const int y = y0 + ty, x = x0 + tx;
    if (y >= a.oy1 || x >= a.W) return;
    // packed-RGB path keeps the three channels' rows in named registers until the interleave
    uint32_t oR[U], oG[U], oB[U];
    for (int c = 0; c < a.C; ++c) {
        const uint8_t *ctr = s_img + c * (PH * PW) + (ty + kHalo) * PW + (tx + kHalo);
        const int va = ctr[0];
        uint32_t lo0[RW], hi0[RW], lo1[RW], hi1[RW], lo2[RW], hi2[RW], lo3[RW], hi3[RW];
#pragma unroll
        for (int k = 0; k < RW; ++k) lo0[k] = hi0[k] = lo1[k] = hi1[k] = lo2[k] = hi2[k] = lo3[k] = hi3[k] = 0;
        for (int m = 0; m < a.M; ++m) {
            const void *lut = a.lut[m];
            const int di0 = a.di[m][0], di1 = a.di[m][1], di2 = a.di[m][2];
            const int dj0 = a.dj[m][0], dj1 = a.dj[m][1], dj2 = a.dj[m][2];
            static_for<0, 4>([&](auto R) {
                constexpr int r = R;
                int dy, dx, v0, v1, v2;
                sample_offset(r, di0, dj0, dy, dx); v0 = ctr[dy * PW + dx];
                sample_offset(r, di1, dj1, dy, dx); v1 = ctr[dy * PW + dx];
                sample_offset(r, di2, dj2, dy, dx); v2 = ctr[dy * PW + dx];
                int idx[5], w[5];
                simplex4(va, v0, v1, v2, idx, w);
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
#if MULUT_ABLATE == 3
#pragma unroll
                for (int j = 0; j < 5; ++j)
#pragma unroll
                    for (int k = 0; k < RW; ++k) {
                        if constexpr (r == 0) lo0[k] ^= row[j][k] + w[j];
                        if constexpr (r == 1) lo1[k] ^= row[j][k] + w[j];
                        if constexpr (r == 2) lo2[k] ^= row[j][k] + w[j];
                        if constexpr (r == 3) lo3[k] ^= row[j][k] + w[j];
                    }
#else
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    if constexpr (r == 0) swar_fma<RW>(lo0, hi0, row[j], (uint32_t)w[j]);
                    if constexpr (r == 1) swar_fma<RW>(lo1, hi1, row[j], (uint32_t)w[j]);
                    if constexpr (r == 2) swar_fma<RW>(lo2, hi2, row[j], (uint32_t)w[j]);
                    if constexpr (r == 3) swar_fma<RW>(lo3, hi3, row[j], (uint32_t)w[j]);
                }
#endif
            });
        }
        // rotate back + sum the four rotations, remove the +128 bias, divide/round/clip
        const int unbias = 128 * kQ * 4 * a.M;
        uint32_t o[U];
        static_for<0, U>([&](auto SY) {
            constexpr int sy = SY;
            uint32_t packed = 0;
            static_for<0, U>([&](auto SX) {
                constexpr int sx = SX;
                const uint32_t sum = swar_field<row_elem(0, sy, sx, U), RW>(lo0, hi0) +
                                     swar_field<row_elem(1, sy, sx, U), RW>(lo1, hi1) +
                                     swar_field<row_elem(2, sy, sx, U), RW>(lo2, hi2) +
                                     swar_field<row_elem(3, sy, sx, U), RW>(lo3, hi3);
                const uint32_t v = rhe_clip_u8((int)sum - unbias + a.bias_num, a.div);
                if constexpr (OUT == kOutGeneric) {
                    *const_cast<uint8_t *>(view_addr(a.out, n, c, y * U + sy, x * U + sx)) = (uint8_t)v;
                } else {
                    packed |= v << (8 * sx);
                }
            });
            o[sy] = packed;
            if constexpr (OUT == kOutPlanarU4) {
                *(uint32_t *)const_cast<uint8_t *>(view_addr(a.out, n, c, y * U + sy, x * U)) = packed;
            }
        });
        if constexpr (OUT == kOutPackedRGBU4) {  // c is wave-uniform: scalar branches, static register names
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
    }
    if constexpr (OUT == kOutPackedRGBU4) {
        // [r0 r1 r2 r3],[g0..g3],[b0..b3] -> r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3 (12 bytes per HR row)
#pragma unroll
        for (int sy = 0; sy < U; ++sy) {
            const uint32_t R = oR[sy], G = oG[sy], B = oB[sy];
            uint32_t w0, w1, w2;
            interleave_rgb4(R, G, B, w0, w1, w2);
            uint32_t *dst = (uint32_t *)const_cast<uint8_t *>(view_addr(a.out, n, 0, y * U + sy, x * U));
            dst[0] = w0;
            dst[1] = w1;
            dst[2] = w2;
        }
    }
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
    const size_t lds = (size_t)a.C * (K2_TH + 2 * kHalo) * (K2_TW + 2 * kHalo);
    const long long nb = (long long)a.N * a.tiles_x * a.tiles_y;
    if (nb <= 0 || nb > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((stage_up_kernel<U, OUT, K2_TW, K2_TH>), dim3((unsigned)nb), dim3(K2_TW * K2_TH), lds, st, a);
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

}  // namespace mulut
