// mulut_dev.h -- device helpers shared by the kernel translation units (mulut_kernels.hip, mulut_k1.hip, mulut_detail.hip):
// image views and tile staging, pattern / rotation constants, packed 16-bit MAC and SDWA helpers, the per-rotation SWAR
// accumulators of the final stage with their epilogues.  Device code only; see mulut_core.h for the per-site arithmetic.
#ifndef MULUT_DEV_H_
#define MULUT_DEV_H_

#include <hip/hip_runtime.h>

#include "mulut_kernels.h"

namespace mulut {

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

constexpr int kPatDi[3][3] = {{0, 1, 1}, {0, 2, 2}, {1, 1, 2}};   // s, d, y: row offsets of keys b, c, d (pattern_offsets)
constexpr int kPatDj[3][3] = {{1, 0, 1}, {2, 0, 2}, {1, 2, 1}};
constexpr int rot_dy(int r, int di, int dj) { return r == 0 ? di : r == 1 ? dj : r == 2 ? -di : -dj; }   // sample_offset
constexpr int rot_dx(int r, int di, int dj) { return r == 0 ? dj : r == 1 ? -di : r == 2 ? -dj : di; }

// dst = a + (16-bit half SEL of b): one SDWA add extracts and adds
template <int SEL>
__device__ __forceinline__ uint32_t add_word(uint32_t a, uint32_t b) {
    uint32_t r;
    if constexpr (SEL == 0) asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_0" : "=v"(r) : "v"(a), "v"(b));
    else asm("v_add_u32_sdwa %0, %1, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:WORD_1" : "=v"(r) : "v"(a), "v"(b));
    return r;
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

// packed pair of window codes: low half = code at (R1, C1), high half = code at (R2, C2); window column c lives in
// dword c / 2, half c % 2.  One v_perm_b32 (selector bytes 0-3 pick from the second operand, 4-7 from the first).
template <int R1, int C1, int R2, int C2, int NW>
__device__ __forceinline__ uint32_t win_pair(const uint32_t (&w)[5][NW]) {
    static_assert(R1 >= 0 && R1 < 5 && R2 >= 0 && R2 < 5 && C1 >= 0 && C1 < 2 * NW && C2 >= 0 && C2 < 2 * NW, "window is 5 rows x 2 NW codes");
    constexpr uint32_t sel = ((C1 & 1) ? 0x0302u : 0x0100u) | (((C2 & 1) ? 0x0706u : 0x0504u) << 16);
    return __builtin_amdgcn_perm(w[R2][C2 / 2], w[R1][C1 / 2], sel);
}

// inclusive prefix sum over the 64 lanes of a wave: v_add_u32_dpp row_shr:1, 2, 4, 8 (zero fill) inside each row of 16, then the last
// lane of row 0 / 2 into rows 1 / 3 (row_bcast:15, rows 1 and 3) and the last lane of row 1 into rows 2 and 3 (row_bcast:31)
__device__ __forceinline__ uint32_t wave_scan_add(uint32_t x) {
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xF, 0xF, true);      // row_shr:1
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xF, 0xF, true);      // row_shr:2
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x114, 0xF, 0xF, true);      // row_shr:4
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x118, 0xF, 0xF, true);      // row_shr:8
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x142, 0xA, 0xF, false);     // row_bcast:15 into rows 1 and 3
    x += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x143, 0xC, 0xF, false);     // row_bcast:31 into rows 2 and 3
    return x;
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

constexpr int KB_TW = 64, KB_TH = 16;      // tile of the final-stage LDS kernels and of the per-tile verdicts

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
        uint32_t rlo[4], rhi[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { rlo[k] = row[k] & 0x00FF00FFu; rhi[k] = (row[k] >> 8) & 0x00FF00FFu; }
        mac_x<R, 0>(rlo, rhi, w);
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
            if (a.use_f32)   // wave-uniform
                packed = rhe_pack4_f32(k0, k1, k2, k3, a.inv_d);
            else
                packed = rhe_clip_u8(k0, a.div) | (rhe_clip_u8(k1, a.div) << 8) | (rhe_clip_u8(k2, a.div) << 16) |
                         (rhe_clip_u8(k3, a.div) << 24);
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
#pragma unroll
    for (int j = 0; j < 5; ++j) load_row<U>(lut, idx[j], row[j]);
#pragma unroll
    for (int j = 0; j < 5; ++j) acc.template fma<R>(row[j], (uint32_t)w[j]);
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


// The same for mode lists of more than four modes: a block value is the sum of two fields of up to 8160 x modes each (17 bits for 8 modes:
// added in 32 bits, never merged in a field), the numerator K = sum - unbias goes through the float epilogue where it is proven exact for the
// divisor (use_f32: 1..6 and 8 modes) and the integer one otherwise.
__device__ __forceinline__ void tube_finish_rows_wide(const StageArgs &a, const RotAcc<4> &acc, uint32_t (&o)[4]) {
    const int unbias = 128 * kQ * 4 * a.M - a.bias_num;
    static_for<0, 4>([&](auto SY) {
        constexpr int sy = SY;
        const int k0 = (int)(tube_field<4 * sy + 0>(acc.lo02, acc.hi02) + tube_field<12 + sy>(acc.lo13, acc.hi13)) - unbias;
        const int k1 = (int)(tube_field<4 * sy + 1>(acc.lo02, acc.hi02) + tube_field<8 + sy>(acc.lo13, acc.hi13)) - unbias;
        const int k2 = (int)(tube_field<4 * sy + 2>(acc.lo02, acc.hi02) + tube_field<4 + sy>(acc.lo13, acc.hi13)) - unbias;
        const int k3 = (int)(tube_field<4 * sy + 3>(acc.lo02, acc.hi02) + tube_field<0 + sy>(acc.lo13, acc.hi13)) - unbias;
        if (a.use_f32) o[sy] = rhe_pack4_f32(k0, k1, k2, k3, a.inv_d);      // wave-uniform
        else o[sy] = rhe_clip_u8(k0, a.div) | (rhe_clip_u8(k1, a.div) << 8) | (rhe_clip_u8(k2, a.div) << 16) | (rhe_clip_u8(k3, a.div) << 24);
    });
}

// where a tile's anchor-MSB histogram / list positions live (tile_stat_kernel writes, the detailed-tile path reads)
__device__ __forceinline__ size_t detail_hist_index(uint32_t tile, uint32_t ntiles, int b) {
    return ((size_t)(b >> 3) * ntiles + tile) * 8 + (size_t)(b & 7);
}
__device__ __forceinline__ size_t detail_pos_index(uint32_t tile, uint32_t ntiles, int b) {
    return ((size_t)(b >> 2) * ntiles + tile) * 4 + (size_t)(b & 3);
}


}  // namespace mulut
#endif  // MULUT_DEV_H_
