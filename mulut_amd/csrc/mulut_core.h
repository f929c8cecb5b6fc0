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
        lo[k] += (row[k] & 0x00FF00FFu) * w;
        hi[k] += ((row[k] >> 8) & 0x00FF00FFu) * w;
    }
}

template <int E, int RW>
MULUT_HD uint32_t swar_field(const uint32_t (&lo)[RW], const uint32_t (&hi)[RW]) {
    const uint32_t word = (E & 1) ? hi[E >> 2] : lo[E >> 2];
    return (E & 2) ? (word >> 16) : (word & 0xFFFFu);
}

// [r0 r1 r2 r3],[g0..g3],[b0..b3] -> r0 g0 b0 r1 | g1 b1 r2 g2 | b2 r3 g3 b3 (12 bytes of one HR row)
MULUT_HD void interleave_rgb4(uint32_t R, uint32_t G, uint32_t B, uint32_t &w0, uint32_t &w1, uint32_t &w2) {
    w0 = (R & 0xFFu) | ((G & 0xFFu) << 8) | ((B & 0xFFu) << 16) | ((R & 0xFF00u) << 16);
    w1 = ((G >> 8) & 0xFFu) | (B & 0xFF00u) | (R & 0xFF0000u) | ((G & 0xFF0000u) << 8);
    w2 = ((B >> 16) & 0xFFu) | ((R >> 16) & 0xFF00u) | ((G >> 8) & 0xFF0000u) | (B & 0xFF000000u);
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

}  // namespace mulut
#endif  // MULUT_CORE_H_
