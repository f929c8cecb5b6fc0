// mulut_core.h -- per-site arithmetic shared by every kernel in mulut_kernels.hip.
//
// Pure integer math with no device intrinsics, so the very same functions are compiled by hipcc
// into the gfx950 kernels and by g++ into tests/host_emul (a CPU unit test of this header; it is
// NOT a product path -- mulut_amd never runs it).
//
// Reference behaviour restated here (paths relative to the reference repo):
//   simplex4()      : MSB/LSB split, corner indices, 24-case simplex weights   sr/4_test_lut.py:15-51,56-109,140-230
//   sample_offset() : np.rot90 + bottom/right edge pad as seen from the un-rotated image  :294-296
//   row_elem()      : block->image reshuffle + rotate back                      :232-235
//   rhe_clip_u8()   : pred/avg + bias, np.round (half to even), clip            :300-306
#ifndef MULUT_CORE_H_
#define MULUT_CORE_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define MULUT_HD __host__ __device__ __forceinline__
#else
#define MULUT_HD static inline
#endif

namespace mulut {

constexpr int kInterval = 4;            // --interval 4 (common/option.py:23)
constexpr int kQ = 1 << kInterval;      // 16
constexpr int kL = (1 << (8 - kInterval)) + 1;  // 17
constexpr int kStrideA = kL * kL * kL;  // 4913  key a (anchor pixel) is most significant (:61)
constexpr int kStrideB = kL * kL;       // 289
constexpr int kStrideC = kL;            // 17
constexpr int kStrideD = 1;
constexpr int kRows = kL * kL * kL * kL;            // 83521
constexpr int kAllStrides = kStrideA + kStrideB + kStrideC + kStrideD;  // 5220: p1111 - p0000

MULUT_HD int imin(int a, int b) { return a < b ? a : b; }
MULUT_HD int imax(int a, int b) { return a > b ? a : b; }

// Pattern offsets (row, col) of keys b, c, d relative to the anchor a (= (0,0)).
// 's' :20-23, 'd' :32-35, 'y' :43-46.  Returns false for an unknown mode (:54 raises).
MULUT_HD bool pattern_offsets(char mode, int (&di)[3], int (&dj)[3]) {
    switch (mode) {
        case 's': di[0] = 0; dj[0] = 1; di[1] = 1; dj[1] = 0; di[2] = 1; dj[2] = 1; return true;
        case 'd': di[0] = 0; dj[0] = 2; di[1] = 2; dj[1] = 0; di[2] = 2; dj[2] = 2; return true;
        case 'y': di[0] = 1; dj[0] = 1; di[1] = 1; dj[1] = 2; di[2] = 2; dj[2] = 1; return true;
        default: return false;
    }
}

// Rotation r turns the pattern offset (di,dj) into this displacement in the un-rotated image;
// clamping the displaced coordinate to the image is what the bottom/right edge pad amounts to.
MULUT_HD void sample_offset(int r, int di, int dj, int &dy, int &dx) {
    switch (r & 3) {
        case 0: dy = di; dx = dj; break;
        case 1: dy = dj; dx = -di; break;
        case 2: dy = -di; dx = -dj; break;
        default: dy = -dj; dx = di; break;
    }
}

// Element of the u*u table row that lands on HR sub-pixel (sy,sx) of the site's block.
MULUT_HD constexpr int row_elem(int r, int sy, int sx, int u) {
    return (r & 3) == 0   ? sy * u + sx
           : (r & 3) == 1 ? (u - 1 - sx) * u + sy
           : (r & 3) == 2 ? (u - 1 - sy) * u + (u - 1 - sx)
                          : sx * u + (u - 1 - sy);
}

MULUT_HD void cmpx_desc(uint32_t &a, uint32_t &b) {
    const uint32_t hi = a > b ? a : b, lo = a > b ? b : a;
    a = hi;
    b = lo;
}

// One site: four key values (0..255) -> five table row indices along the monotone vertex path
// 0000 -> ... -> 1111 and their integer weights (sum 16).  The fractional parts are sorted
// descending by a 5-comparator network on (f << 16 | stride) keys; ties only ever reorder
// zero-weight vertices, so any tie order reproduces the reference's 24-case cascade.
MULUT_HD void simplex4(int va, int vb, int vc, int vd, int (&idx)[5], int (&w)[5]) {
    const int base = (va >> 4) * kStrideA + (vb >> 4) * kStrideB + (vc >> 4) * kStrideC + (vd >> 4);
    uint32_t k0 = ((uint32_t)(va & 15) << 16) | kStrideA;
    uint32_t k1 = ((uint32_t)(vb & 15) << 16) | kStrideB;
    uint32_t k2 = ((uint32_t)(vc & 15) << 16) | kStrideC;
    uint32_t k3 = ((uint32_t)(vd & 15) << 16) | kStrideD;
    cmpx_desc(k0, k1);
    cmpx_desc(k2, k3);
    cmpx_desc(k0, k2);
    cmpx_desc(k1, k3);
    cmpx_desc(k1, k2);
    const int f1 = (int)(k0 >> 16), f2 = (int)(k1 >> 16), f3 = (int)(k2 >> 16), f4 = (int)(k3 >> 16);
    idx[0] = base;
    idx[1] = idx[0] + (int)(k0 & 0xFFFFu);
    idx[2] = idx[1] + (int)(k1 & 0xFFFFu);
    idx[3] = idx[2] + (int)(k2 & 0xFFFFu);
    idx[4] = base + kAllStrides;
    w[0] = kQ - f1;
    w[1] = f1 - f2;
    w[2] = f2 - f3;
    w[3] = f3 - f4;
    w[4] = f4;
}

// Division magic for round-half-even by a small runtime-uniform divisor d (d = 16*M or 64*M):
// floor(n/d) == (n * magic) >> 32 for 0 <= n < 2^17 when magic = ceil(2^32/d), d <= 1024.
struct DivMagic {
    uint32_t d;
    uint32_t magic;
};
MULUT_HD DivMagic make_div_magic(uint32_t d) {
    DivMagic m;
    m.d = d;
    m.magic = (uint32_t)(((1ull << 32) + d - 1) / d);
    return m;
}

// clip(round_half_even(n / d), 0, 255) for an integer numerator n (may be negative).
MULUT_HD uint32_t rhe_clip_u8(int n, DivMagic m) {
    const uint32_t nn = (uint32_t)imax(n, 0);  // negative quotients round to <= 0 and clip to 0
    const uint32_t q = (uint32_t)(((uint64_t)nn * m.magic) >> 32);
    const uint32_t r = nn - q * m.d;
    const uint32_t up = (2u * r + (q & 1u)) > m.d ? 1u : 0u;
    const uint32_t v = q + up;
    return v > 255u ? 255u : v;
}

// Stage epilogue numerators: K = q * pred summed over modes x 4 rotations.
//   non-final stage: out = clip(rhe((K + 127*64M) / 64M))   (avg = 4M, bias = 127, :286)
//   final stage    : out = clip(rhe( K           / 16M))    (avg = M,  bias = 0,   :283)
MULUT_HD int stage_divisor(int n_modes, bool is_last) { return is_last ? kQ * n_modes : kQ * 4 * n_modes; }
MULUT_HD int stage_bias_num(int n_modes, bool is_last) { return is_last ? 0 : 127 * kQ * 4 * n_modes; }

// ---- 16-bit SWAR accumulation of u*u-byte table rows (final-stage kernel) -------------------------
// Device tables with v_num > 1 store value+128 as uint8, so every partial sum is non-negative.  A
// row dword holds elements 4k..4k+3; lo[k] accumulates elements 4k (bits 0-15) and 4k+2 (bits
// 16-31), hi[k] elements 4k+1 and 4k+3.  One accumulator pair serves one rotation:
// M modes * 16 * 255 < 65536 for M <= 16.
template <int RW>
MULUT_HD void swar_fma(uint32_t (&lo)[RW], uint32_t (&hi)[RW], const uint32_t (&row)[RW], uint32_t w) {
    for (int k = 0; k < RW; ++k) {
#if defined(MULUT_ABLATE) && MULUT_ABLATE == 4   /* timing-only: no unpack (as if rows were u16 pairs) */
        lo[k] += (row[k] & 0x00FFFFFFu) * w;
        hi[k] += (row[k] >> 8) * w;
#else
        lo[k] += (row[k] & 0x00FF00FFu) * w;
        hi[k] += ((row[k] >> 8) & 0x00FF00FFu) * w;
#endif
    }
}

template <int E, int RW>
MULUT_HD uint32_t swar_field(const uint32_t (&lo)[RW], const uint32_t (&hi)[RW]) {
    const uint32_t word = (E & 1) ? hi[E >> 2] : lo[E >> 2];
    return (E & 2) ? (word >> 16) : (word & 0xFFFFu);
}

// [r0 r1 r2 r3],[g0..g3],[b0..b3] -> r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3 (12 bytes of one HR row)
MULUT_HD void interleave_rgb4(uint32_t R, uint32_t G, uint32_t B, uint32_t &w0, uint32_t &w1, uint32_t &w2) {
#if defined(__HIP_DEVICE_COMPILE__)
    // two v_perm_b32 per output dword (selector bytes 0-3 pick from the second operand, 4-7 from the first)
    const uint32_t rg0 = __builtin_amdgcn_perm(G, R, 0x01000400u);   // r0 g0 -- r1
    const uint32_t gr1 = __builtin_amdgcn_perm(R, G, 0x02060001u);   // g1 -- r2 g2   (G = second operand)
    const uint32_t rg2 = __builtin_amdgcn_perm(G, R, 0x00070300u);   // -- r3 g3 --
    w0 = __builtin_amdgcn_perm(B, rg0, 0x03040100u);                 // r0 g0 b0 r1
    w1 = __builtin_amdgcn_perm(B, gr1, 0x03020500u);                 // g1 b1 r2 g2
    w2 = __builtin_amdgcn_perm(B, rg2, 0x07020106u);                 // b2 r3 g3 b3
#else
    w0 = (R & 0xFFu) | ((G & 0xFFu) << 8) | ((B & 0xFFu) << 16) | ((R & 0xFF00u) << 16);
    w1 = ((G >> 8) & 0xFFu) | (B & 0xFF00u) | (R & 0xFF0000u) | ((G & 0xFF0000u) << 8);
    w2 = ((B >> 16) & 0xFFu) | ((R >> 16) & 0xFF00u) | ((G >> 8) & 0xFF0000u) | (B & 0xFF000000u);
#endif
}

// compile-time loop: the body receives an IC<I>, so every register-array index is a constant
// expression (runtime-indexed arrays would be demoted to scratch memory on the GPU)
template <int I>
struct IC {
    static constexpr int value = I;
    constexpr operator int() const { return I; }
};
template <int B, int E, class F>
MULUT_HD void static_for(F &&f) {
    if constexpr (B < E) {
        f(IC<B>{});
        static_for<B + 1, E>(f);
    }
}

// ---- packed 2 x u16 helpers ------------------------------------------------------------------------
// Two passes of one site are processed side by side in the low / high half of a dword.  On the
// GPU these compile to v_pk_*_u16 (one instruction for both halves); the host versions exist for
// tests/host_emul only.
#if defined(__clang__)
typedef unsigned short mulut_u16x2 __attribute__((ext_vector_type(2)));
#define MULUT_PK(x) __builtin_bit_cast(mulut_u16x2, (uint32_t)(x))
#define MULUT_UNPK(v) __builtin_bit_cast(uint32_t, (v))
#endif
#if defined(__HIP_DEVICE_COMPILE__)
MULUT_HD uint32_t pk_add(uint32_t a, uint32_t b) { return MULUT_UNPK(MULUT_PK(a) + MULUT_PK(b)); }
MULUT_HD uint32_t pk_sub(uint32_t a, uint32_t b) { return MULUT_UNPK(MULUT_PK(a) - MULUT_PK(b)); }
MULUT_HD uint32_t pk_min(uint32_t a, uint32_t b) { return MULUT_UNPK(__builtin_elementwise_min(MULUT_PK(a), MULUT_PK(b))); }
MULUT_HD uint32_t pk_max(uint32_t a, uint32_t b) { return MULUT_UNPK(__builtin_elementwise_max(MULUT_PK(a), MULUT_PK(b))); }
MULUT_HD uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) { return MULUT_UNPK(MULUT_PK(a) * MULUT_PK(b) + MULUT_PK(c)); }
MULUT_HD uint32_t pk_shr12(uint32_t a) { return MULUT_UNPK(MULUT_PK(a) >> (unsigned short)12); }
MULUT_HD uint32_t pk_shr4(uint32_t a) { return MULUT_UNPK(MULUT_PK(a) >> (unsigned short)4); }
MULUT_HD uint32_t pk_shr11(uint32_t a) { return MULUT_UNPK(MULUT_PK(a) >> (unsigned short)11); }
MULUT_HD uint32_t pk_sub_sat(uint32_t a, uint32_t b) { return MULUT_UNPK(__builtin_elementwise_sub_sat(MULUT_PK(a), MULUT_PK(b))); }
#else
MULUT_HD uint32_t pk_add(uint32_t a, uint32_t b) { return ((a + b) & 0xFFFFu) | (((a >> 16) + (b >> 16)) << 16); }
MULUT_HD uint32_t pk_sub(uint32_t a, uint32_t b) { return ((a - b) & 0xFFFFu) | (((a >> 16) - (b >> 16)) << 16); }
MULUT_HD uint32_t pk_min(uint32_t a, uint32_t b) {
    const uint32_t l = (a & 0xFFFFu) < (b & 0xFFFFu) ? (a & 0xFFFFu) : (b & 0xFFFFu);
    const uint32_t h = (a >> 16) < (b >> 16) ? (a >> 16) : (b >> 16);
    return l | (h << 16);
}
MULUT_HD uint32_t pk_max(uint32_t a, uint32_t b) {
    const uint32_t l = (a & 0xFFFFu) > (b & 0xFFFFu) ? (a & 0xFFFFu) : (b & 0xFFFFu);
    const uint32_t h = (a >> 16) > (b >> 16) ? (a >> 16) : (b >> 16);
    return l | (h << 16);
}
MULUT_HD uint32_t pk_mad(uint32_t a, uint32_t b, uint32_t c) {
    return (((a & 0xFFFFu) * (b & 0xFFFFu) + (c & 0xFFFFu)) & 0xFFFFu) | ((((a >> 16) * (b >> 16) + (c >> 16)) & 0xFFFFu) << 16);
}
MULUT_HD uint32_t pk_shr12(uint32_t a) { return ((a & 0xFFFFu) >> 12) | (((a >> 16) >> 12) << 16); }
MULUT_HD uint32_t pk_shr4(uint32_t a) { return ((a & 0xFFFFu) >> 4) | (((a >> 16) >> 4) << 16); }
MULUT_HD uint32_t pk_shr11(uint32_t a) { return ((a & 0xFFFFu) >> 11) | (((a >> 16) >> 11) << 16); }
MULUT_HD uint32_t pk_sub_sat(uint32_t a, uint32_t b) {
    const uint32_t l = (a & 0xFFFFu) > (b & 0xFFFFu) ? (a & 0xFFFFu) - (b & 0xFFFFu) : 0u;
    const uint32_t h = (a >> 16) > (b >> 16) ? (a >> 16) - (b >> 16) : 0u;
    return l | (h << 16);
}
#endif
MULUT_HD uint32_t pk_dup(uint32_t lo16) { return lo16 | (lo16 << 16); }
MULUT_HD void pk_cmpx_desc(uint32_t &a, uint32_t &b) {
    const uint32_t hi = pk_max(a, b), lo = pk_min(a, b);
    a = hi;
    b = lo;
}

// SWAR accumulate with the weight taken from one 16-bit half of a packed register (HALF = 0 low,
// 1 high).  On the GPU this is one v_pk_mad_u16 per accumulator dword with op_sel picking the half for
// both lanes -- no weight extraction and no separate add.  Fields never carry: value <= 255, w <= 16.
template <int HALF>
MULUT_HD uint32_t pk_mad_w(uint32_t x, uint32_t wpk, uint32_t acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    const mulut_u16x2 wv = MULUT_PK(wpk);
    const mulut_u16x2 ws = __builtin_shufflevector(wv, wv, HALF, HALF);
    return MULUT_UNPK(MULUT_PK(x) * ws + MULUT_PK(acc));
#else
    const uint32_t w = HALF ? (wpk >> 16) : (wpk & 0xFFFFu);
    return pk_mad(x, pk_dup(w), acc);
#endif
}
template <int HALF>
MULUT_HD void swar_fma4_pk(uint32_t (&lo)[4], uint32_t (&hi)[4], const uint32_t (&row)[4], uint32_t wpk) {
    for (int k = 0; k < 4; ++k) {
        lo[k] = pk_mad_w<HALF>(row[k] & 0x00FF00FFu, wpk, lo[k]);
        hi[k] = pk_mad_w<HALF>((row[k] >> 8) & 0x00FF00FFu, wpk, hi[k]);
    }
}

// ---- diagonal band of a final-stage table (LDS-resident part) ---------------------------------------
// Natural images keep the four keys of a patch within one MSB step of each other, so almost every
// gather lands in the "band"  { (A,B,C,D) : B-A, C-A, D-A in [-2,2] }  of the 17^4 table: 17*125
// rows (34 KB at 16 B/row) instead of 83521.  Band slot of a row:
//      A*125 + (B-A+2)*25 + (C-A+2)*5 + (D-A+2)  =  A*94 + B*25 + C*5 + D + 62
// i.e. fixed per-key strides (94, 25, 5, 1), so the simplex vertex walk works unchanged.  A pass is
// "in band" iff |hb-ha|, |hc-ha|, |hd-ha| <= 1 (then all five vertices are); other passes take the
// full table in global memory.
constexpr int kBandSpan = 5;
constexpr int kBandRowsPerA = kBandSpan * kBandSpan * kBandSpan;   // 125
constexpr int kBandRows = kL * kBandRowsPerA;                      // 2125
constexpr int kBandStrideA = kBandRowsPerA - 31, kBandStrideB = 25, kBandStrideC = 5, kBandStrideD = 1;
constexpr int kBandBase = 62;
MULUT_HD constexpr bool band_contains(int A, int B, int C, int D) {
    return B - A >= -2 && B - A <= 2 && C - A >= -2 && C - A <= 2 && D - A >= -2 && D - A <= 2;
}
MULUT_HD constexpr int band_slot(int A, int B, int C, int D) {
    return A * kBandStrideA + B * kBandStrideB + C * kBandStrideC + D * kBandStrideD + kBandBase;
}

// Two passes (low half = pass A, high half = pass B) of one site against a band table with 16-byte
// rows.  va: anchor value (shared); pb/pc/pd: key values of both passes packed (vA | vB << 16).
// Outputs, packed per pass: byte offsets addr[5] of the five rows inside the band image, weights
// w[5], and t_band (<= 32 in a half  <=>  that pass is in band).
struct BandPair {
    uint32_t addr[5];
    uint32_t w[5];
    uint32_t t_band;
};
MULUT_HD void simplex4_band_pair(uint32_t va, uint32_t pb, uint32_t pc, uint32_t pd, BandPair &o) {
    constexpr uint32_t RB = 16;  // row bytes
    constexpr uint32_t SA = kBandStrideA * RB, SB = kBandStrideB * RB, SC = kBandStrideC * RB, SD = kBandStrideD * RB;
    const uint32_t hb16 = pb & 0x00F000F0u, hc16 = pc & 0x00F000F0u, hd16 = pd & 0x00F000F0u;  // 16*h per half
    const uint32_t ha16 = va & 0xF0u;
    // keys: f << 12 | byte stride (< 4096), sorted descending per half by a 5-comparator network
    uint32_t k0 = pk_dup(((va & 15u) << 12) | SA);
    uint32_t k1 = ((pb << 12) & 0xF000F000u) | pk_dup(SB);
    uint32_t k2 = ((pc << 12) & 0xF000F000u) | pk_dup(SC);
    uint32_t k3 = ((pd << 12) & 0xF000F000u) | pk_dup(SD);
    pk_cmpx_desc(k0, k1);
    pk_cmpx_desc(k2, k3);
    pk_cmpx_desc(k0, k2);
    pk_cmpx_desc(k1, k3);
    pk_cmpx_desc(k1, k2);
    const uint32_t f1 = pk_shr12(k0), f2 = pk_shr12(k1), f3 = pk_shr12(k2), f4 = pk_shr12(k3);
    // base byte offset: (ha*94 + hb*25 + hc*5 + hd + 62) * 16, built from the 16*h nibbles
    const uint32_t base_a = pk_dup((ha16 * kBandStrideA) + kBandBase * RB);
    const uint32_t base = pk_mad(hb16, pk_dup(kBandStrideB), pk_mad(hc16, pk_dup(kBandStrideC), pk_add(hd16, base_a)));
    o.addr[0] = base;
    o.addr[1] = pk_add(base, k0 & 0x0FFF0FFFu);
    o.addr[2] = pk_add(o.addr[1], k1 & 0x0FFF0FFFu);
    o.addr[3] = pk_add(o.addr[2], k2 & 0x0FFF0FFFu);
    o.addr[4] = pk_add(base, pk_dup(kBandRowsPerA * RB));
    o.w[0] = pk_sub(pk_dup(kQ), f1);
    o.w[1] = pk_sub(f1, f2);
    o.w[2] = pk_sub(f2, f3);
    o.w[3] = pk_sub(f3, f4);
    o.w[4] = f4;
    // in band <=> 16*(h - ha + 1) in {0,16,32} for b, c, d (out-of-range differences wrap to >= 0xFFF0 or reach 48+)
    const uint32_t off = pk_dup((16u - ha16) & 0xFFFFu);
    o.t_band = pk_max(pk_max(pk_add(hb16, off), pk_add(hc16, off)), pk_add(hd16, off));
}

// Pre-split pixel code used by the expanded-band kernel's image tile: the LSB nibble already sits where
// the sort key wants it and the MSB nibble where the band offset math wants it,
//      code(v) = (v & 15) << 12 | (v & 0xF0)            value(code) = (code >> 12) | (code & 0xF0)
// so building a key is one v_and_or and the 16*h term one v_and on a packed pair of codes.
MULUT_HD uint32_t pixel_code(uint32_t v) { return ((v & 15u) << 12) | (v & 0xF0u); }
MULUT_HD int pixel_value(uint32_t code) { return (int)(((code >> 12) & 15u) | (code & 0xF0u)); }

// simplex4_band_pair on pixel codes: ca = code of the anchor (low half only), pb/pc/pd = packed code pairs.
// Outputs as simplex4_band_pair, except t_band holds saturate(max - 32) per half: a half is in band iff 0.
MULUT_HD void simplex4_band_pair_code(uint32_t ca, uint32_t pb, uint32_t pc, uint32_t pd, BandPair &o) {
    constexpr uint32_t RB = 16;
    constexpr uint32_t SA = kBandStrideA * RB, SB = kBandStrideB * RB, SC = kBandStrideC * RB, SD = kBandStrideD * RB;
    const uint32_t hb16 = pb & 0x00F000F0u, hc16 = pc & 0x00F000F0u, hd16 = pd & 0x00F000F0u;
    const uint32_t ha16 = ca & 0xF0u;
    uint32_t k0 = pk_dup((ca & 0xF000u) | SA);
    uint32_t k1 = (pb & 0xF000F000u) | pk_dup(SB);
    uint32_t k2 = (pc & 0xF000F000u) | pk_dup(SC);
    uint32_t k3 = (pd & 0xF000F000u) | pk_dup(SD);
    pk_cmpx_desc(k0, k1);
    pk_cmpx_desc(k2, k3);
    pk_cmpx_desc(k0, k2);
    pk_cmpx_desc(k1, k3);
    pk_cmpx_desc(k1, k2);
    const uint32_t f1 = pk_shr12(k0), f2 = pk_shr12(k1), f3 = pk_shr12(k2), f4 = pk_shr12(k3);
    const uint32_t base_a = pk_dup((ha16 * kBandStrideA) + kBandBase * RB);
    const uint32_t base = pk_mad(hb16, pk_dup(kBandStrideB), pk_mad(hc16, pk_dup(kBandStrideC), pk_add(hd16, base_a)));
    o.addr[0] = base;
    o.addr[1] = pk_add(base, k0 & 0x0FFF0FFFu);
    o.addr[2] = pk_add(o.addr[1], k1 & 0x0FFF0FFFu);
    o.addr[3] = pk_add(o.addr[2], k2 & 0x0FFF0FFFu);
    o.addr[4] = pk_add(base, pk_dup(kBandRowsPerA * RB));
    o.w[0] = pk_sub(pk_dup(kQ), f1);
    o.w[1] = pk_sub(f1, f2);
    o.w[2] = pk_sub(f2, f3);
    o.w[3] = pk_sub(f3, f4);
    o.w[4] = f4;
    const uint32_t off = pk_dup((16u - ha16) & 0xFFFFu);
    const uint32_t t = pk_max(pk_max(pk_add(hb16, off), pk_add(hc16, off)), pk_add(hd16, off));
    o.t_band = pk_sub_sat(t, pk_dup(32u));
}

// ---- "tube" band: rows whose four keys span at most two MSB steps -----------------------------------------
// A pass whose four MSBs satisfy max - min <= 1 touches only rows with max - min <= 2.  There are 991 such
// rows in the 17^4 table, and the affine map
//      slot = 27 A + 18 B + 12 C + 8 D                      (27 + 18 + 12 + 8 = 65)
// is injective on them with range 0..1040 (found by exhaustive search; checked in tests/host_emul): 1041
// 16-byte slots per plane, so the bands of all three modes, expanded to 16-bit fields (two planes each),
// fit LDS together (3 x 33,312 B) and no band is ever swapped.  The strides are 11, 2, 12, 8 mod 16: any two
// rows whose keys differ by at most one step per key land in different 16-byte bank groups of a
// ds_read_b128 lane group (|11a + 2b + 12c + 8d| mod 16 != 0 for a,b,c,d in {-1,0,1} not all zero).
constexpr int kTubeSA = 27, kTubeSB = 18, kTubeSC = 12, kTubeSD = 8;
constexpr int kTubeAll = kTubeSA + kTubeSB + kTubeSC + kTubeSD;       // 65: slot(p1111) - slot(p0000)
constexpr int kTubeSlots = (kL - 1) * kTubeAll + 1;                    // 1041
constexpr int kTubePlaneBytes = kTubeSlots * 16;                       // 16656
constexpr int kTubeBandBytes = 2 * kTubePlaneBytes;                    // 33312: LO plane, HI plane
MULUT_HD constexpr bool tube_contains(int A, int B, int C, int D) {
    const int mx = A > B ? (A > C ? (A > D ? A : D) : (C > D ? C : D)) : (B > C ? (B > D ? B : D) : (C > D ? C : D));
    const int mn = A < B ? (A < C ? (A < D ? A : D) : (C < D ? C : D)) : (B < C ? (B < D ? B : D) : (C < D ? C : D));
    return mx - mn <= 2;
}
MULUT_HD constexpr int tube_slot(int A, int B, int C, int D) { return A * kTubeSA + B * kTubeSB + C * kTubeSC + D * kTubeSD; }

// Two passes (low half = pass A, high half = pass B) of one site against a tube band, on pixel codes
// (pixel_code below): ca = anchor code in the low half, pb/pc/pd = packed code pairs.  `bias` (a multiple of
// 16, duplicated into both halves by the caller's constant) is added to every row offset so that the caller
// can keep the plane's LDS address within the 16-bit immediate of ds_read.  t_oob is zero in a half iff that
// pass is in the tube (max - min of its four MSBs <= 1).
struct TubePair {
    uint32_t base;      // byte offset of row 0 inside a plane (+ bias), packed per pass (row 4 = base + 16 * 65)
    uint32_t step[3];   // byte strides of path steps 1..3, packed per pass (each < 4096)
    uint32_t w[5];      // weights, packed per pass
    uint32_t t_oob;
};
// anchor terms of a site, the same for every mode and rotation: the anchor's sort key and MSB term (16 h) in both halves
MULUT_HD uint32_t tube_anchor_key(uint32_t ca) { return pk_dup((ca & 0xF000u) | (uint32_t)(kTubeSA * 16)); }
MULUT_HD uint32_t tube_anchor_h16(uint32_t ca) { return pk_dup(ca & 0xF0u); }
// base_a = ha16 * 27 + bias per half (bias: a multiple of 16 the caller keeps the plane's LDS address within reach with)
template <bool SKIP_TEST = false>
MULUT_HD void simplex4_tube_pair(uint32_t k0, uint32_t ha16, uint32_t base_a, uint32_t pb, uint32_t pc, uint32_t pd, TubePair &o) {
    constexpr uint32_t SB = kTubeSB * 16, SC = kTubeSC * 16, SD = kTubeSD * 16;   // byte strides < 4096
    const uint32_t hb16 = pb & 0x00F000F0u, hc16 = pc & 0x00F000F0u, hd16 = pd & 0x00F000F0u;     // 16*h per half
    uint32_t k1 = (pb & 0xF000F000u) | pk_dup(SB);
    uint32_t k2 = (pc & 0xF000F000u) | pk_dup(SC);
    uint32_t k3 = (pd & 0xF000F000u) | pk_dup(SD);
    pk_cmpx_desc(k0, k1);
    pk_cmpx_desc(k2, k3);
    pk_cmpx_desc(k0, k2);
    pk_cmpx_desc(k1, k3);
    pk_cmpx_desc(k1, k2);
    const uint32_t f1 = pk_shr12(k0), f2 = pk_shr12(k1), f3 = pk_shr12(k2), f4 = pk_shr12(k3);
    // row offset in bytes = 16 * slot = (16 h) * slot stride, summed over the four keys
    o.base = pk_mad(hb16, pk_dup(kTubeSB), pk_mad(hc16, pk_dup(kTubeSC), pk_mad(hd16, pk_dup(kTubeSD), base_a)));
    o.step[0] = k0 & 0x0FFF0FFFu;
    o.step[1] = k1 & 0x0FFF0FFFu;
    o.step[2] = k2 & 0x0FFF0FFFu;
    // plain 32-bit subtracts from here on (full-rate ops; the packed forms are half rate): no half ever borrows -- the
    // sorted f's are descending, max >= min
    o.w[0] = pk_dup(kQ) - f1;
    o.w[1] = f1 - f2;
    o.w[2] = f2 - f3;
    o.w[3] = f3 - f4;
    o.w[4] = f4;
    if constexpr (SKIP_TEST) {
        o.t_oob = 0;            // the caller has the site's flag from site_flag_kernel
        (void)ha16;
    } else {
        const uint32_t mx = pk_max(pk_max(hb16, hc16), pk_max(hd16, ha16));
        const uint32_t mn = pk_min(pk_min(hb16, hc16), pk_min(hd16, ha16));
        o.t_oob = (mx - mn) & 0xFFE0FFE0u;     // differences are multiples of 16: in the tube iff 0 or 16
    }
}
// byte offsets of rows 0..3 of one pass (HALF 0 / 1) from the packed results
MULUT_HD void tube_pair_rows(const TubePair &p, int half, uint32_t (&a)[4]) {
    a[0] = half ? (p.base >> 16) : (p.base & 0xFFFFu);
    for (int j = 0; j < 3; ++j) a[j + 1] = a[j] + (half ? (p.step[j] >> 16) : (p.step[j] & 0xFFFFu));
}

// The same pair math for tables with 1-byte rows (non-final stages): the tube band holds one dword per slot, so row
// offsets are 4 * slot, and a stride (<= 4 * 27) fits the low byte of a sort key -- the kernel adds it to the running
// offset with a byte-select (SDWA) add, no masking.  Pixel codes here are  code1(v) = (v & 15) << 12 | v >> 4:  with the
// MSB in the low nibble, code1 * (16 * stride) = 16 * h * stride (mod 2^16) -- the LSB nibble multiplies out of the
// half -- so the slot sum needs no MSB extraction either.
//   in : k0 = anchor key pair (f << 12 | 4 * 27 per half), base_a = code1(anchor) * (16 * 27) per half, pb/pc/pd = code1 pairs
//   out: base = byte offset of row 0 per half (row 4 = base + 4 * 65), ks[0..2] = sorted keys whose low bytes are the
//        strides of path steps 1..3, w[5] = weights.  The in-tube test is the caller's (a 5 x 5 neighbourhood at once).
struct TubePair1 {
    uint32_t base;
    uint32_t ks[3];
    uint32_t w[5];
};
constexpr int kTube1BandBytes = ((kTubeSlots * 4 + 15) / 16) * 16;      // 4176
constexpr int kTube2BandBytes = ((kTubeSlots * 8 + 15) / 16) * 16;      // 8336: u == 2 rows as four 16-bit fields (8 bytes per slot)
// u == 3 rows: the nine values as ten 16-bit fields e0 e1 e2 e3 e4 e4 e5 e6 e7 e8 -- the centre twice, so that the 180-degree
// rotation of the 3 x 3 block (element q -> 8 - q) is the reversal of the ten fields (whole dwords reversed, halves swapped) and the
// passes of rotations r and r + 2 share one accumulator set, as for u == 2 and u == 4.  24 bytes per slot (20 used; 8-byte aligned reads)
constexpr int kTube3SlotBytes = 24;
constexpr int kTube3BandBytes = ((kTubeSlots * kTube3SlotBytes + 15) / 16) * 16;      // 24992
MULUT_HD constexpr int tube3_field(int q) { return q <= 4 ? q : q + 1; }               // field of block element q (the centre's second copy is field 5)
MULUT_HD uint32_t pixel_code1(uint32_t v) { return ((v & 15u) << 12) | (v >> 4); }
MULUT_HD uint32_t tube1_key(uint32_t code_pk, uint32_t stride4) { return (code_pk & 0xF000F000u) | pk_dup(stride4); }
// SLOT = bytes per slot: 4 for 1-byte rows (one dword per slot), 8 for u == 2 rows (four 16-bit fields per slot) -- a stride then fits
// the low BYTE of a key -- and 24 for u == 3 rows, whose strides (<= 24 * 27) take the twelve bits under f
template <int SLOT>
MULUT_HD void simplex4_tube_pair1_slot(uint32_t k0, uint32_t base_a, uint32_t pb, uint32_t pc, uint32_t pd, TubePair1 &o) {
    static_assert(SLOT == 4 || SLOT == 8 || SLOT == 24, "slot size");
    uint32_t k1 = tube1_key(pb, kTubeSB * SLOT), k2 = tube1_key(pc, kTubeSC * SLOT), k3 = tube1_key(pd, kTubeSD * SLOT);
#if !(defined(MULUT_ABLATE) && MULUT_ABLATE == 32)   /* 32 = timing-only: no sort */
    pk_cmpx_desc(k0, k1);
    pk_cmpx_desc(k2, k3);
    pk_cmpx_desc(k0, k2);
    pk_cmpx_desc(k1, k3);
    pk_cmpx_desc(k1, k2);
#endif
    const uint32_t f1 = pk_shr12(k0), f2 = pk_shr12(k1), f3 = pk_shr12(k2), f4 = pk_shr12(k3);
    // 16 * slot summed over the keys (< 16 * 1041); a quarter of it is the byte offset.  One 32-bit shift: both halves
    // are multiples of 16, so no set bit crosses over.
    const uint32_t base16 = pk_mad(pb, pk_dup(16 * kTubeSB), pk_mad(pc, pk_dup(16 * kTubeSC), pk_mad(pd, pk_dup(16 * kTubeSD), base_a)));
    o.base = SLOT == 24 ? base16 + (base16 >> 1) : base16 >> (SLOT == 4 ? 2 : 1);      // (24 = 16 * 1.5; the halves are multiples of 16 and stay below 2^16)
    o.ks[0] = k0; o.ks[1] = k1; o.ks[2] = k2;
    o.w[0] = pk_dup(kQ) - f1;
    o.w[1] = f1 - f2;
    o.w[2] = f2 - f3;
    o.w[3] = f3 - f4;
    o.w[4] = f4;
}
template <int SHIFT = 2>
MULUT_HD void simplex4_tube_pair1(uint32_t k0, uint32_t base_a, uint32_t pb, uint32_t pc, uint32_t pd, TubePair1 &o) {
    simplex4_tube_pair1_slot<(1 << SHIFT)>(k0, base_a, pb, pc, pd, o);
}

// ---- slab pairs: the full table, two anchor slabs at a time (detailed content) ----------------------------
// The anchor (first key) of a sample is the pixel itself in every mode and rotation, so all 12 passes of a sample use
// rows (A, *, *, *) before the anchor's step of the path and (A + 1, *, *, *) after it.  The device image of a final
// stage table therefore also exists as 16 "slab pairs":  pair A = rows (A, b, c, d) and (A + 1, b, c, d) interleaved,
// 32 bytes per (b, c, d), value + 128 as uint8 -- 2 * 4913 * 16 = 157,216 bytes, which fits the LDS of one CU.  In units
// of 16 bytes the strides are  anchor step 1, D 2, C 34, B 578: each fits the 12 bits under f in a packed sort key, and a
// pass's row offsets are plain sums of the sorted keys' low bits, as in the tube pair math.
//   in : k0 = anchor key pair (f << 12 | 1 per half), pb/pc/pd = neighbour BYTES of rotations r (low half) and r + 2
//        (high half), 0x00vv00vv
//   out: base = row 0 offset per half (16-byte units; row 4 = base + kSlabAll), step[0..2] = unit steps of path steps
//        1..3 per half, w[5] = weights per half
constexpr int kSlabUA = 1, kSlabUD = 2, kSlabUC = 34, kSlabUB = 578, kSlabAll = kSlabUA + kSlabUD + kSlabUC + kSlabUB;
constexpr int kSlabPairBytes = 2 * 17 * 17 * 17 * 16;      // 157,216
constexpr int kSlabTableBytes = 16 * kSlabPairBytes;       // one mode: 2,515,456
struct SlabPair {
    uint32_t base;
    uint32_t step[3];
    uint32_t w[5];
};
MULUT_HD uint32_t slab_anchor_key(uint32_t va) { return pk_dup(((va & 15u) << 12) | (uint32_t)kSlabUA); }
MULUT_HD void simplex4_slab_pair(uint32_t k0, uint32_t pb, uint32_t pc, uint32_t pd, SlabPair &o) {
    // byte * 4096 keeps the LSB nibble at bits 12..15 of its half (the MSB nibble leaves the half)
    uint32_t k1 = pk_mad(pb, pk_dup(4096u), pk_dup(kSlabUB));
    uint32_t k2 = pk_mad(pc, pk_dup(4096u), pk_dup(kSlabUC));
    uint32_t k3 = pk_mad(pd, pk_dup(4096u), pk_dup(kSlabUD));
    const uint32_t hb = pk_shr4(pb), hc = pk_shr4(pc), hd = pk_shr4(pd);
    pk_cmpx_desc(k0, k1);
    pk_cmpx_desc(k2, k3);
    pk_cmpx_desc(k0, k2);
    pk_cmpx_desc(k1, k3);
    pk_cmpx_desc(k1, k2);
    const uint32_t f1 = pk_shr12(k0), f2 = pk_shr12(k1), f3 = pk_shr12(k2), f4 = pk_shr12(k3);
    o.base = pk_mad(hb, pk_dup(kSlabUB), pk_mad(hc, pk_dup(kSlabUC), hd + hd));     // <= 16 * 614: no carry between halves
    o.step[0] = k0 & 0x0FFF0FFFu;
    o.step[1] = k1 & 0x0FFF0FFFu;
    o.step[2] = k2 & 0x0FFF0FFFu;
    o.w[0] = pk_dup(kQ) - f1;
    o.w[1] = f1 - f2;
    o.w[2] = f2 - f3;
    o.w[3] = f3 - f4;
    o.w[4] = f4;
}
// Rows stay bytes (value + 128, four per dword) and are accumulated WITHOUT being split into 16-bit fields first:
//   F += dword * w   (v_pk_mad_u16 on the raw dword: a field holds 256 * odd byte + even byte, sums wrap mod 2^16)
//   H += (odd bytes as fields) * w
// and at the end  even sums = F - 256 H (mod 2^16: exact, an even sum is < 2^16 for <= 4 modes),  odd sums = H:
// three operations per dword where split + accumulate takes four.  Rotations r + 2 add the byte-reversed dword into
// accumulator dword 3 - k (merged rotation pairs, below).
#if defined(__HIP_DEVICE_COMPILE__)
MULUT_HD uint32_t slab_odd_bytes(uint32_t x) { return __builtin_amdgcn_perm(0u, x, 0x0C030C01u); }       // (b1, b3) as fields
MULUT_HD uint32_t slab_rev_bytes(uint32_t x) { return __builtin_amdgcn_perm(0u, x, 0x00010203u); }
MULUT_HD uint32_t slab_rev_odd_bytes(uint32_t x) { return __builtin_amdgcn_perm(0u, x, 0x0C000C02u); }   // odd bytes of the reversed dword: (b2, b0)
#else
MULUT_HD uint32_t slab_odd_bytes(uint32_t x) { return (x >> 8) & 0x00FF00FFu; }
MULUT_HD uint32_t slab_rev_bytes(uint32_t x) { return (x >> 24) | ((x >> 8) & 0xFF00u) | ((x << 8) & 0xFF0000u) | (x << 24); }
MULUT_HD uint32_t slab_rev_odd_bytes(uint32_t x) { return ((x >> 16) & 0xFFu) | ((x & 0xFFu) << 16); }
#endif
MULUT_HD uint32_t slab_even_sums(uint32_t F, uint32_t H) { return pk_mad(H, pk_dup(0xFF00u), F); }
// Round 4: the second accumulator takes the dword shifted right by 8 bits -- one full-rate 32-bit shift instead of a half-rate byte
// permute per dword and row:
//   F += x * w          fields (b0 + 256 b1, b2 + 256 b3) * w
//   G += (x >> 8) * w   fields (b1 + 256 b2, b3) * w
// With E_i / O_i the sums of the even / odd bytes:  F = (E0 + 256 O1, E2 + 256 O3),  G = (O1 + 256 E2, O3)  (mod 2^16 per field), so
//   E = F - 256 G = (E0, E2)                  (one packed multiply-add: 256 * 256 E2 leaves the field)
//   O = G - ((256 E2 mod 2^16), 0) = (O1, O3)
// exact while every true sum is below 2^16 (as for slab_even_sums).
MULUT_HD void slab_split_sums(uint32_t F, uint32_t G, uint32_t &even, uint32_t &odd) {
    even = pk_mad(G, pk_dup(0xFF00u), F);
    odd = pk_sub(G, (even >> 8) & 0x0000FF00u);
}

// ---- full-table pairs for 1-byte rows (first-stage kernel on detailed tiles) ------------------------------
// Rotations r / r + 2 of one site in packed halves against the WHOLE table (row index A 4913 + B 289 + C 17 + D).  The
// strides of b, c, d fit the 12 bits under f in a sort key; the anchor's 4913 does not, so the anchor's key carries a marker
// (bit 11, above any sum of the other strides: 289 + 17 + 1 = 307) and a row's offset is
//     base + (cum & 0x7FF) + (cum >> 11 & 1) * 4913        cum = running sum of the sorted keys' low 12 bits.
//   in : k0 = anchor key pair (f << 12 | 0x800 per half), pb/pc/pd = neighbour BYTES of rotation r (low half) and r + 2 (high)
//   out: base = B 289 + C 17 + D per half (the caller adds A 4913), cum[0..2] = offsets of rows 1..3 from row 0 per half
//        (the marker already turned into the anchor's stride), w[5] = weights per half; row 4 = base + kAllStrides
struct FullPair1 {
    uint32_t base;
    uint32_t cum[3];
    uint32_t w[5];
};
constexpr uint32_t kFullMark = 0x800u;
MULUT_HD uint32_t full1_anchor_key(uint32_t va) { return pk_dup(((va & 15u) << 12) | kFullMark); }
MULUT_HD void simplex4_full_pair1(uint32_t k0, uint32_t pb, uint32_t pc, uint32_t pd, FullPair1 &o) {
    uint32_t k1 = pk_mad(pb, pk_dup(4096u), pk_dup(kStrideB));
    uint32_t k2 = pk_mad(pc, pk_dup(4096u), pk_dup(kStrideC));
    uint32_t k3 = pk_mad(pd, pk_dup(4096u), pk_dup(kStrideD));
    const uint32_t hb = pk_shr4(pb), hc = pk_shr4(pc), hd = pk_shr4(pd);
    pk_cmpx_desc(k0, k1);
    pk_cmpx_desc(k2, k3);
    pk_cmpx_desc(k0, k2);
    pk_cmpx_desc(k1, k3);
    pk_cmpx_desc(k1, k2);
    const uint32_t f1 = pk_shr12(k0), f2 = pk_shr12(k1), f3 = pk_shr12(k2), f4 = pk_shr12(k3);
    o.base = pk_mad(hb, pk_dup(kStrideB), pk_mad(hc, pk_dup(kStrideC), hd));      // <= 15 * 307: no carry between halves
    const uint32_t c0 = k0 & 0x0FFF0FFFu;
    const uint32_t c1 = c0 + (k1 & 0x0FFF0FFFu);                                   // <= 0x800 + 307 per half
    const uint32_t c2 = c1 + (k2 & 0x0FFF0FFFu);
    // marker (2048) -> the anchor's stride: + (cum >> 11) * (4913 - 2048), both halves at once
    o.cum[0] = pk_mad(pk_shr11(c0), pk_dup((uint32_t)kStrideA - kFullMark), c0);
    o.cum[1] = pk_mad(pk_shr11(c1), pk_dup((uint32_t)kStrideA - kFullMark), c1);
    o.cum[2] = pk_mad(pk_shr11(c2), pk_dup((uint32_t)kStrideA - kFullMark), c2);
    o.w[0] = pk_dup(kQ) - f1;
    o.w[1] = f1 - f2;
    o.w[2] = f2 - f3;
    o.w[3] = f3 - f4;
    o.w[4] = f4;
}
// row offsets (from table + A 4913) of one pass
MULUT_HD void full_pair1_rows(const FullPair1 &p, int half, uint32_t (&r)[5]) {
    const uint32_t b = half ? (p.base >> 16) : (p.base & 0xFFFFu);
    r[0] = b;
    for (int j = 0; j < 3; ++j) {
        r[j + 1] = b + (half ? (p.cum[j] >> 16) : (p.cum[j] & 0xFFFFu));
    }
    r[4] = b + (uint32_t)kAllStrides;
}

// ---- merged rotation pairs ----------------------------------------------------------------------------
// Rotation r+2 maps row element e to the block position that rotation r gives element 15-e
// (row_elem(r+2,sy,sx,4) == 15 - row_elem(r,sy,sx,4)), so the rows of rotation r+2 can be added
// into rotation r's accumulator in reversed element order: 8 accumulator dwords per pair instead
// of 16.  In the SWAR layout reversal sends dword k' -> 3-k', swaps lo <-> hi and swaps the two
// 16-bit fields; the byte shuffles below (one v_perm_b32 each on the GPU) produce the swapped
// fields directly.
MULUT_HD uint32_t bytes_3_1(uint32_t x) {  // (b3, 0, b1, 0): fields (elem 3, elem 1) of the dword
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(0u, x, 0x0C010C03u);
#else
    return (x >> 24) | (((x >> 8) & 0xFFu) << 16);
#endif
}
MULUT_HD uint32_t bytes_2_0(uint32_t x) {  // (b2, 0, b0, 0)
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_perm(0u, x, 0x0C000C02u);
#else
    return ((x >> 16) & 0xFFu) | ((x & 0xFFu) << 16);
#endif
}
MULUT_HD void swar_fma_rev4(uint32_t (&lo)[4], uint32_t (&hi)[4], const uint32_t (&row)[4], uint32_t w) {
    for (int k = 0; k < 4; ++k) {
        lo[3 - k] += bytes_3_1(row[k]) * w;   // elems (4k+3, 4k+1) -> reversed (4(3-k)+0, 4(3-k)+2)
        hi[3 - k] += bytes_2_0(row[k]) * w;   // elems (4k+2, 4k+0) -> reversed (4(3-k)+1, 4(3-k)+3)
    }
}

template <int HALF>
MULUT_HD void swar_fma_rev4_pk(uint32_t (&lo)[4], uint32_t (&hi)[4], const uint32_t (&row)[4], uint32_t wpk) {
    for (int k = 0; k < 4; ++k) {
        lo[3 - k] = pk_mad_w<HALF>(bytes_3_1(row[k]), wpk, lo[3 - k]);
        hi[3 - k] = pk_mad_w<HALF>(bytes_2_0(row[k]), wpk, hi[3 - k]);
    }
}

// ---- expanded band rows (32 B): [lo0 lo1 lo2 lo3 | hi0 hi1 hi2 hi3], lo_k = e(4k) | e(4k+2) << 16,
// hi_k = e(4k+1) | e(4k+3) << 16 -- the SWAR fields ready-made, so a row costs 8 v_pk_mad_u16 and
// no unpack.  The reversed (rotation r+2) form swaps the halves of src0 with op_sel for free.
template <int HALF>
MULUT_HD uint32_t pk_mad_w_swap(uint32_t x, uint32_t wpk, uint32_t acc) {
#if defined(__HIP_DEVICE_COMPILE__)
    const mulut_u16x2 wv = MULUT_PK(wpk);
    const mulut_u16x2 ws = __builtin_shufflevector(wv, wv, HALF, HALF);
    const mulut_u16x2 xv = MULUT_PK(x);
    return MULUT_UNPK(__builtin_shufflevector(xv, xv, 1, 0) * ws + MULUT_PK(acc));
#else
    return pk_mad_w<HALF>((x >> 16) | (x << 16), wpk, acc);
#endif
}
template <int HALF>
MULUT_HD void swar_fma_x4(uint32_t (&lo)[4], uint32_t (&hi)[4], const uint32_t (&rlo)[4], const uint32_t (&rhi)[4], uint32_t wpk) {
    for (int k = 0; k < 4; ++k) {
        lo[k] = pk_mad_w<HALF>(rlo[k], wpk, lo[k]);
        hi[k] = pk_mad_w<HALF>(rhi[k], wpk, hi[k]);
    }
}
template <int HALF>
MULUT_HD void swar_fma_x4_rev(uint32_t (&lo)[4], uint32_t (&hi)[4], const uint32_t (&rlo)[4], const uint32_t (&rhi)[4], uint32_t wpk) {
    for (int k = 0; k < 4; ++k) {
        lo[3 - k] = pk_mad_w_swap<HALF>(rhi[k], wpk, lo[3 - k]);
        hi[3 - k] = pk_mad_w_swap<HALF>(rlo[k], wpk, hi[3 - k]);
    }
}

// Sum of both pair accumulators in output order (u = 4): S[e0] = acc02[e0] + acc13[e1] with
// e0 = 4*sy+sx (rotation-0 layout == block layout) and e1 = (3-sx)*4 + sy (rotation-1 layout).
MULUT_HD void combine_pairs4(const uint32_t (&lo02)[4], const uint32_t (&hi02)[4], const uint32_t (&lo13)[4],
                             const uint32_t (&hi13)[4], uint32_t (&lo)[4], uint32_t (&hi)[4]) {
    // out dword sy, position sx  <-  acc13 dword 3-sx, position sy
    //   positions 0,2 live in lo[], 1,3 in hi[]; positions 0,1 are the low 16 bits, 2,3 the high 16
    lo[0] = lo02[0] + ((lo13[3] & 0xFFFFu) | (lo13[1] << 16));
    hi[0] = hi02[0] + ((lo13[2] & 0xFFFFu) | (lo13[0] << 16));
    lo[1] = lo02[1] + ((hi13[3] & 0xFFFFu) | (hi13[1] << 16));
    hi[1] = hi02[1] + ((hi13[2] & 0xFFFFu) | (hi13[0] << 16));
    lo[2] = lo02[2] + ((lo13[3] >> 16) | (lo13[1] & 0xFFFF0000u));
    hi[2] = hi02[2] + ((lo13[2] >> 16) | (lo13[0] & 0xFFFF0000u));
    lo[3] = lo02[3] + ((hi13[3] >> 16) | (hi13[1] & 0xFFFF0000u));
    hi[3] = hi02[3] + ((hi13[2] >> 16) | (hi13[0] & 0xFFFF0000u));
}

// ---- float form of the epilogue -------------------------------------------------------------------------
// clip(rhe(K/d)) as  cvt -> mul by fl(1/d) -> round-to-nearest-even -> saturating u8 convert.  It is
// exact iff K*fl(1/d) rounds to the exact tie value whenever K/d is a tie; rhe_f32_valid() checks that
// by brute force over the whole numerator range at configure time (it holds for d = 16*M, 64*M with
// M in 1..6 and 8), otherwise the integer form is used.
MULUT_HD uint32_t rhe_clip_u8_f32(int K, float inv_d) {
#if defined(__HIP_DEVICE_COMPILE__)
    const float q = __builtin_rintf((float)K * inv_d);
    return __builtin_amdgcn_cvt_pk_u8_f32(q, 0u, 0u);
#else
    const float q = __builtin_rintf((float)K * inv_d);
    return q < 0.0f ? 0u : (q > 255.0f ? 255u : (uint32_t)q);
#endif
}

// four results packed into one dword (byte i = clip(rhe(K_i / d))): one v_cvt_pk_u8_f32 per byte on the GPU
MULUT_HD uint32_t rhe_pack4_f32(int K0, int K1, int K2, int K3, float inv_d) {
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf((float)K0 * inv_d), 0u, 0u);
    r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf((float)K1 * inv_d), 1u, r);
    r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf((float)K2 * inv_d), 2u, r);
    r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_rintf((float)K3 * inv_d), 3u, r);
    return r;
#else
    return rhe_clip_u8_f32(K0, inv_d) | (rhe_clip_u8_f32(K1, inv_d) << 8) | (rhe_clip_u8_f32(K2, inv_d) << 16) |
           (rhe_clip_u8_f32(K3, inv_d) << 24);
#endif
}

// Fused form for biased sums: clip(rhe((S - unbias) / d)) as  cvt -> fma(S, fl(1/d), c) -> rndne -> saturating u8
// convert, with c = fl(-unbias * fl(1/d)): the subtraction rides in the fma.  Exactness is again proven by brute
// force over every reachable sum at configure time (rhe_fma_valid).
MULUT_HD uint32_t rhe_clip_u8_fma(uint32_t S, float inv_d, float c) {
    const float q = __builtin_rintf(__builtin_fmaf((float)S, inv_d, c));
#if defined(__HIP_DEVICE_COMPILE__)
    return __builtin_amdgcn_cvt_pk_u8_f32(q, 0u, 0u);
#else
    return q < 0.0f ? 0u : (q > 255.0f ? 255u : (uint32_t)q);
#endif
}
MULUT_HD uint32_t rhe_pack4_fma(uint32_t S0, uint32_t S1, uint32_t S2, uint32_t S3, float inv_d, float c) {
#if defined(__HIP_DEVICE_COMPILE__)
    // v_cvt_pk_u8_f32 itself rounds to nearest even and saturates to 0..255 (measured on gfx950 over -8..504 in steps
    // of 1/128, tools/probe_cvt.hip: 0 differences from rintf + clamp), so no v_rndne_f32 is needed in front of it
    uint32_t r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)S0, inv_d, c), 0u, 0u);
    r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)S1, inv_d, c), 1u, r);
    r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)S2, inv_d, c), 2u, r);
    r = __builtin_amdgcn_cvt_pk_u8_f32(__builtin_fmaf((float)S3, inv_d, c), 3u, r);
    return r;
#else
    return rhe_clip_u8_fma(S0, inv_d, c) | (rhe_clip_u8_fma(S1, inv_d, c) << 8) | (rhe_clip_u8_fma(S2, inv_d, c) << 16) |
           (rhe_clip_u8_fma(S3, inv_d, c) << 24);
#endif
}
MULUT_HD bool rhe_fma_valid(uint32_t smax, int unbias, DivMagic m, float inv_d, float c) {
    for (uint32_t s = 0; s <= smax; ++s)
        if (rhe_clip_u8_fma(s, inv_d, c) != rhe_clip_u8((int)s - unbias, m)) return false;
    return true;
}

// the same fused form on a signed numerator K with the bias folded into the addend: clip(rhe((K + bias) / d)) as
// cvt -> fma(K, fl(1/d), c) -> (rounding) saturating u8 convert
MULUT_HD uint32_t rhe_clip_u8_fma_i(int K, float inv_d, float c) {
    const float q = __builtin_rintf(__builtin_fmaf((float)K, inv_d, c));
    return q < 0.0f ? 0u : (q > 255.0f ? 255u : (uint32_t)q);
}
MULUT_HD bool rhe_fma_valid_i(int kmin, int kmax, int bias, DivMagic m, float inv_d, float c) {
    for (int k = kmin; k <= kmax; ++k)
        if (rhe_clip_u8_fma_i(k, inv_d, c) != rhe_clip_u8(k + bias, m)) return false;
    return true;
}

// brute-force proof that the float epilogue equals the integer one for every numerator in [kmin, kmax]
MULUT_HD bool rhe_f32_valid(int kmin, int kmax, DivMagic m, float inv_d) {
    for (int k = kmin; k <= kmax; ++k)
        if (rhe_clip_u8_f32(k, inv_d) != rhe_clip_u8(k, m)) return false;
    return true;
}

}  // namespace mulut
#endif  // MULUT_CORE_H_
