// CPU unit-test harness for mulut_amd/csrc/mulut_core.h -- TEST ONLY, never a product path.
// It drives the exact per-site functions the gfx950 kernels are built from (simplex4, swar_fma /
// swar_field with row_elem, rhe_clip_u8, interleave_rgb4) over a whole image with plain loops, so
// arithmetic bugs show up on the CPU before a GPU box is spent on them.  Launch geometry, LDS
// tiling and the C ABI are covered by the -m gpu tests.
#include <cstddef>
#include <cstdint>
#include <vector>

#include "../../mulut_amd/csrc/mulut_core.h"

using namespace mulut;

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

template <int U>
static void stage_up(const int8_t *const *luts, const char *modes, int M, const uint8_t *in, int H, int W, int C,
                     uint8_t *out_hwc, bool is_last) {
    constexpr int RW = (U * U + 3) / 4;
    // device table image: value+128, row stride RW*4 bytes
    std::vector<std::vector<uint32_t>> tabs(M);
    for (int m = 0; m < M; ++m) {
        tabs[m].assign((size_t)kRows * RW, 0x80808080u);
        uint8_t *b = (uint8_t *)tabs[m].data();
        for (int i = 0; i < kRows; ++i)
            for (int e = 0; e < U * U; ++e) b[(size_t)i * RW * 4 + e] = (uint8_t)((int)luts[m][(size_t)i * U * U + e] + 128);
    }
    const DivMagic dv = make_div_magic((uint32_t)stage_divisor(M, is_last));
    const int bias = stage_bias_num(M, is_last);
    const int Wo = W * U;
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            uint32_t o[3][U];
            for (int c = 0; c < C; ++c) {
                const uint8_t *pl = in + (size_t)c * H * W;  // planar input
                const int va = pl[(size_t)y * W + x];
                uint32_t lo[4][RW] = {}, hi[4][RW] = {};
                uint32_t lo02[4] = {}, hi02[4] = {}, lo13[4] = {}, hi13[4] = {};   // u == 4: merged rotation pairs
                for (int m = 0; m < M; ++m) {
                    int di[3], dj[3];
                    pattern_offsets(modes[m], di, dj);
                    for (int r = 0; r < 4; ++r) {
                        int v[3];
                        for (int k = 0; k < 3; ++k) {
                            int dy, dx;
                            sample_offset(r, di[k], dj[k], dy, dx);
                            v[k] = pl[(size_t)clampi(y + dy, 0, H - 1) * W + clampi(x + dx, 0, W - 1)];
                        }
                        int idx[5], w[5];
                        simplex4(va, v[0], v[1], v[2], idx, w);
                        for (int j = 0; j < 5; ++j) {
                            uint32_t row[RW];
                            for (int k = 0; k < RW; ++k) row[k] = tabs[m][(size_t)idx[j] * RW + k];
                            if (U == 4 && M <= 4) {      // the kernels' dispatch: merged rotation pairs hold four modes at most
                                uint32_t(&row4)[4] = reinterpret_cast<uint32_t(&)[4]>(row);
                                if (r == 0) swar_fma<4>(lo02, hi02, row4, (uint32_t)w[j]);
                                if (r == 1) swar_fma<4>(lo13, hi13, row4, (uint32_t)w[j]);
                                if (r == 2) swar_fma_rev4(lo02, hi02, row4, (uint32_t)w[j]);
                                if (r == 3) swar_fma_rev4(lo13, hi13, row4, (uint32_t)w[j]);
                            } else {
                                swar_fma<RW>(lo[r], hi[r], row, (uint32_t)w[j]);
                            }
                        }
                    }
                }
                const int unbias = 128 * kQ * 4 * M - bias;
                if (U == 4 && M <= 4) {
                    uint32_t tl[4], th[4];
                    combine_pairs4(lo02, hi02, lo13, hi13, tl, th);
                    const float inv_d = 1.0f / (float)dv.d;
                    const bool f32ok = rhe_f32_valid(-128 * kQ * 4 * M + bias, 128 * kQ * 4 * M + bias, dv, inv_d);
                    for (int sy = 0; sy < 4; ++sy) {
                        const int k0 = (int)(tl[sy] & 0xFFFFu) - unbias, k1 = (int)(th[sy] & 0xFFFFu) - unbias;
                        const int k2 = (int)(tl[sy] >> 16) - unbias, k3 = (int)(th[sy] >> 16) - unbias;
                        o[c][sy] = f32ok ? rhe_pack4_f32(k0, k1, k2, k3, inv_d)
                                         : (rhe_clip_u8(k0, dv) | (rhe_clip_u8(k1, dv) << 8) | (rhe_clip_u8(k2, dv) << 16) |
                                            (rhe_clip_u8(k3, dv) << 24));
                    }
                } else {
                    static_for<0, U>([&](auto SY) {
                        constexpr int sy = SY;
                        uint32_t packed = 0;
                        static_for<0, U>([&](auto SX) {
                            constexpr int sx = SX;
                            const uint32_t sum = swar_field<row_elem(0, sy, sx, U), RW>(lo[0], hi[0]) +
                                                 swar_field<row_elem(1, sy, sx, U), RW>(lo[1], hi[1]) +
                                                 swar_field<row_elem(2, sy, sx, U), RW>(lo[2], hi[2]) +
                                                 swar_field<row_elem(3, sy, sx, U), RW>(lo[3], hi[3]);
                            packed |= rhe_clip_u8((int)sum - unbias, dv) << (8 * sx);
                        });
                        o[c][sy] = packed;
                    });
                }
            }
            for (int sy = 0; sy < U; ++sy) {
                uint8_t *dst = out_hwc + ((size_t)(y * U + sy) * Wo + (size_t)x * U) * C;
                if (U == 4 && C == 3) {
                    uint32_t w0, w1, w2;
                    interleave_rgb4(o[0][sy], o[1][sy], o[2][sy], w0, w1, w2);
                    const uint32_t ws[3] = {w0, w1, w2};
                    for (int b = 0; b < 12; ++b) dst[b] = (uint8_t)(ws[b / 4] >> (8 * (b % 4)));
                } else {
                    for (int sx = 0; sx < U; ++sx)
                        for (int c = 0; c < C; ++c) dst[sx * C + c] = (uint8_t)(o[c][sy] >> (8 * sx));
                }
            }
        }
}

// packed band-pair index math vs the scalar simplex: returns the number of mismatches over all
// (va, vb, vc, vd) sampled on a grid; in-band passes must give band_slot() of the scalar indices.
extern "C" long emul_check_band_pair(int step) {
    long bad = 0;
    for (int va = 0; va < 256; va += step)
        for (int vb = 0; vb < 256; vb += step)
            for (int vc = 0; vc < 256; vc += step)
                for (int vd = 0; vd < 256; vd += 1) {
                    // pass B uses a permuted / different key set so both halves are exercised
                    const int vb2 = (vb * 7 + 3) & 255, vc2 = (vc * 5 + 11) & 255, vd2 = 255 - vd;
                    BandPair bp;
                    simplex4_band_pair((uint32_t)va, (uint32_t)vb | ((uint32_t)vb2 << 16), (uint32_t)vc | ((uint32_t)vc2 << 16),
                                       (uint32_t)vd | ((uint32_t)vd2 << 16), bp);
                    {   // the pixel-code form must agree with the value form in every output
                        BandPair bc;
                        simplex4_band_pair_code(pixel_code(va), pixel_code(vb) | (pixel_code(vb2) << 16),
                                                pixel_code(vc) | (pixel_code(vc2) << 16), pixel_code(vd) | (pixel_code(vd2) << 16), bc);
                        for (int j = 0; j < 5; ++j)
                            if (bc.addr[j] != bp.addr[j] || bc.w[j] != bp.w[j]) ++bad;
                        if (((bc.t_band & 0xFFFFu) == 0) != ((bp.t_band & 0xFFFFu) <= 32u)) ++bad;
                        if (((bc.t_band >> 16) == 0) != ((bp.t_band >> 16) <= 32u)) ++bad;
                        if (pixel_value(pixel_code(va)) != va) ++bad;
                    }
                    for (int half = 0; half < 2; ++half) {
                        const int b = half ? vb2 : vb, c = half ? vc2 : vc, d = half ? vd2 : vd;
                        int idx[5], w[5];
                        simplex4(va, b, c, d, idx, w);
                        const int ha = va >> 4, hb = b >> 4, hc = c >> 4, hd = d >> 4;
                        const bool in = (hb - ha >= -1 && hb - ha <= 1) && (hc - ha >= -1 && hc - ha <= 1) && (hd - ha >= -1 && hd - ha <= 1);
                        const uint32_t t = half ? (bp.t_band >> 16) : (bp.t_band & 0xFFFFu);
                        if ((t <= 32u) != in) { ++bad; continue; }
                        int wsum = 0;
                        // tie order may differ between the two sorts, so compare weight per row, not per slot
                        uint32_t rows_s[5], rows_p[5];
                        int wt_s[5], wt_p[5];
                        for (int j = 0; j < 5; ++j) {
                            const int wj = (int)(half ? (bp.w[j] >> 16) : (bp.w[j] & 0xFFFFu));
                            wsum += wj;
                            const int A = idx[j] / kStrideA, B = (idx[j] / kStrideB) % kL, C = (idx[j] / kStrideC) % kL, D = idx[j] % kL;
                            if (in && !band_contains(A, B, C, D)) ++bad;
                            rows_s[j] = in ? (uint32_t)band_slot(A, B, C, D) * 16u : 0u;
                            wt_s[j] = w[j];
                            rows_p[j] = half ? (bp.addr[j] >> 16) : (bp.addr[j] & 0xFFFFu);
                            wt_p[j] = wj;
                        }
                        if (in)
                            for (int j = 0; j < 5; ++j) {
                                int ws = 0, wp = 0;
                                for (int i = 0; i < 5; ++i) {
                                    if (rows_s[i] == rows_s[j]) ws += wt_s[i];
                                    if (rows_p[i] == rows_s[j]) wp += wt_p[i];
                                }
                                if (ws != wp) ++bad;
                            }
                        if (wsum != kQ) ++bad;
                    }
                }
    return bad;
}

// Full-table pairs for 1-byte rows: rows and weights of the packed pair math == the scalar simplex, for every key combination
// (sampled in three of the keys).
extern "C" long emul_check_full_pair1(int step) {
    long bad = 0;
    for (int va = 0; va < 256; va += step)
        for (int vb = 0; vb < 256; vb += step)
            for (int vc = 0; vc < 256; vc += step)
                for (int vd = 0; vd < 256; vd += 1) {
                    const int vb2 = (vb * 7 + 3) & 255, vc2 = (vc * 5 + 11) & 255, vd2 = 255 - vd;
                    FullPair1 fp;
                    simplex4_full_pair1(full1_anchor_key((uint32_t)va), (uint32_t)vb | ((uint32_t)vb2 << 16), (uint32_t)vc | ((uint32_t)vc2 << 16),
                                        (uint32_t)vd | ((uint32_t)vd2 << 16), fp);
                    for (int half = 0; half < 2; ++half) {
                        const int b = half ? vb2 : vb, c = half ? vc2 : vc, d = half ? vd2 : vd;
                        int idx[5], w[5];
                        simplex4(va, b, c, d, idx, w);
                        uint32_t r[5];
                        full_pair1_rows(fp, half, r);
                        int wp[5];
                        for (int j = 0; j < 5; ++j) {
                            wp[j] = (int)(half ? (fp.w[j] >> 16) : (fp.w[j] & 0xFFFFu));
                            r[j] += (uint32_t)((va >> 4) * kStrideA);
                            if (r[j] >= (uint32_t)kRows) ++bad;
                        }
                        for (int j = 0; j < 5; ++j) {
                            int ws = 0, wq = 0;
                            for (int i = 0; i < 5; ++i) {
                                if (idx[i] == idx[j]) ws += w[i];
                                if ((int)r[i] == idx[j]) wq += wp[i];
                            }
                            if (ws != wq) ++bad;
                        }
                    }
                }
    return bad;
}

// Slab pairs (mulut_core.h): for every key combination the packed pair math must give the rows of the scalar simplex
// as (anchor slab flag, (b, c, d) offset) with the same weight per row, and the raw-byte accumulation over a whole
// sample (3 modes x 4 rotations, rotations r + 2 reversed into the pair's accumulator) must reproduce the field sums.
extern "C" long emul_check_slab_pair(int step) {
    long bad = 0;
    for (int va = 0; va < 256; va += step)
        for (int vb = 0; vb < 256; vb += step)
            for (int vc = 0; vc < 256; vc += step)
                for (int vd = 0; vd < 256; vd += 1) {
                    const int vb2 = (vb * 7 + 3) & 255, vc2 = (vc * 5 + 11) & 255, vd2 = 255 - vd;
                    SlabPair sp;
                    simplex4_slab_pair(slab_anchor_key((uint32_t)va), (uint32_t)vb | ((uint32_t)vb2 << 16), (uint32_t)vc | ((uint32_t)vc2 << 16),
                                       (uint32_t)vd | ((uint32_t)vd2 << 16), sp);
                    for (int half = 0; half < 2; ++half) {
                        const int b = half ? vb2 : vb, c = half ? vc2 : vc, d = half ? vd2 : vd;
                        int idx[5], w[5];
                        simplex4(va, b, c, d, idx, w);
                        uint32_t cu[5];
                        cu[0] = half ? (sp.base >> 16) : (sp.base & 0xFFFFu);
                        for (int j = 0; j < 3; ++j) cu[j + 1] = cu[j] + (half ? (sp.step[j] >> 16) : (sp.step[j] & 0xFFFFu));
                        cu[4] = cu[0] + (uint32_t)kSlabAll;
                        uint32_t rows_s[5], rows_p[5];
                        int wt_p[5];
                        for (int j = 0; j < 5; ++j) {
                            // scalar row -> unit offset inside the anchor's slab pair
                            const int A = idx[j] / kStrideA, bcd = idx[j] % kStrideA;
                            const int flag = A - (va >> 4);
                            if (flag < 0 || flag > 1) { ++bad; continue; }
                            rows_s[j] = (uint32_t)(2 * bcd + flag);
                            rows_p[j] = cu[j];
                            wt_p[j] = (int)(half ? (sp.w[j] >> 16) : (sp.w[j] & 0xFFFFu));
                            if (cu[j] * 16u + 16u > (uint32_t)kSlabPairBytes) ++bad;
                        }
                        for (int j = 0; j < 5; ++j) {
                            int ws = 0, wp = 0;
                            for (int i = 0; i < 5; ++i) {
                                if (rows_s[i] == rows_s[j]) ws += w[i];
                                if (rows_p[i] == rows_s[j]) wp += wt_p[i];
                            }
                            if (ws != wp) ++bad;
                        }
                    }
                }
    // raw-byte accumulation against the field form (RotAcc<4>'s layout: lo = elements 4k, 4k+2; hi = 4k+1, 4k+3)
    uint32_t rng = 12345u;
    auto next = [&]() { rng = rng * 1664525u + 1013904223u; return rng >> 8; };
    for (int trial = 0; trial < 2000; ++trial) {
        uint32_t F[4] = {0, 0, 0, 0}, H[4] = {0, 0, 0, 0}, G[4] = {0, 0, 0, 0}, lo[4] = {0, 0, 0, 0}, hi[4] = {0, 0, 0, 0};
        const bool extreme = trial < 4;
        for (int pass = 0; pass < 16; ++pass) {      // four modes x (rotation r, rotation r + 2) x two visits: the merged field bound
            const bool rev = (pass & 1) != 0;
            int wl = 16;
            for (int j = 0; j < 5; ++j) {
                const int wj = j == 4 ? wl : (extreme ? (j == 0 ? 16 : 0) : (int)(next() % (uint32_t)(wl + 1)));
                wl -= wj;
                for (int k = 0; k < 4; ++k) {
                    const uint32_t row = extreme ? 0xFFFFFFFFu : (next() << 8) ^ next();
                    const uint32_t wpk = pk_dup((uint32_t)wj);
                    if (!rev) {
                        F[k] = pk_mad(row, wpk, F[k]);
                        H[k] = pk_mad(slab_odd_bytes(row), wpk, H[k]);
                        G[k] = pk_mad(row >> 8, wpk, G[k]);
                        lo[k] += (row & 0x00FF00FFu) * (uint32_t)wj;
                        hi[k] += ((row >> 8) & 0x00FF00FFu) * (uint32_t)wj;
                    } else {
                        F[3 - k] = pk_mad(slab_rev_bytes(row), wpk, F[3 - k]);
                        H[3 - k] = pk_mad(slab_rev_odd_bytes(row), wpk, H[3 - k]);
                        G[3 - k] = pk_mad(slab_rev_bytes(row) >> 8, wpk, G[3 - k]);
                        lo[3 - k] += bytes_3_1(row) * (uint32_t)wj;
                        hi[3 - k] += bytes_2_0(row) * (uint32_t)wj;
                    }
                }
            }
        }
        for (int k = 0; k < 4; ++k)
            if (slab_even_sums(F[k], H[k]) != lo[k] || H[k] != hi[k]) ++bad;
        for (int k = 0; k < 4; ++k) {      // the shifted-dword form of round 4
            uint32_t ev, od;
            slab_split_sums(F[k], G[k], ev, od);
            if (ev != lo[k] || od != hi[k]) ++bad;
        }
    }
    return bad;
}

// Tube pairs of the 1-byte-row kernel family (simplex4_tube_pair1_slot: slots of 4, 8 and 24 bytes = u 1, 2, 3): for every in-tube key
// combination the packed pair math must give byte offsets SLOT * (tube slot of the scalar simplex row) with the same weight per row --
// strides from the low byte of the sorted keys (4, 8) or their low twelve bits (24), as stage_u1t_kernel unpacks them.
template <int SLOT>
static long check_tube_pair1_slot(int step) {
    long bad = 0;
    for (int va = 0; va < 256; va += step)
        for (int vb = 0; vb < 256; vb += step)
            for (int vc = 0; vc < 256; vc += step)
                for (int vd = 0; vd < 256; vd += 1) {
                    const int vb2 = (vb * 7 + 3) & 255, vc2 = (vc * 5 + 11) & 255, vd2 = 255 - vd;
                    const uint32_t ca = pk_dup(pixel_code1((uint32_t)va));
                    TubePair1 tp;
                    simplex4_tube_pair1_slot<SLOT>(tube1_key(ca, (uint32_t)(kTubeSA * SLOT)), pk_mad(ca, pk_dup(16 * kTubeSA), 0u),
                                                   pixel_code1((uint32_t)vb) | (pixel_code1((uint32_t)vb2) << 16),
                                                   pixel_code1((uint32_t)vc) | (pixel_code1((uint32_t)vc2) << 16),
                                                   pixel_code1((uint32_t)vd) | (pixel_code1((uint32_t)vd2) << 16), tp);
                    for (int half = 0; half < 2; ++half) {
                        const int b = half ? vb2 : vb, c = half ? vc2 : vc, d = half ? vd2 : vd;
                        const int ha = va >> 4, hb = b >> 4, hc = c >> 4, hd = d >> 4;
                        if (imax(imax(ha, hb), imax(hc, hd)) - imin(imin(ha, hb), imin(hc, hd)) > 1) continue;      // (the in-tube test is the caller's)
                        int idx[5], w[5];
                        simplex4(va, b, c, d, idx, w);
                        uint32_t off[5];
                        off[0] = half ? (tp.base >> 16) : (tp.base & 0xFFFFu);
                        for (int j = 0; j < 3; ++j) {
                            const uint32_t k = half ? (tp.ks[j] >> 16) : (tp.ks[j] & 0xFFFFu);
                            off[j + 1] = off[j] + (SLOT == 24 ? (k & 0xFFFu) : (k & 0xFFu));
                        }
                        off[4] = off[0] + (uint32_t)(kTubeAll * SLOT);
                        for (int j = 0; j < 5; ++j) {
                            int ws = 0, wq = 0;
                            const int A = idx[j] / kStrideA, B = (idx[j] / kStrideB) % kL, C = (idx[j] / kStrideC) % kL, D = idx[j] % kL;
                            const uint32_t want = (uint32_t)(tube_slot(A, B, C, D) * SLOT);
                            for (int i = 0; i < 5; ++i) {
                                if (idx[i] == idx[j]) ws += w[i];
                                if (off[i] == want) wq += (int)(half ? (tp.w[i] >> 16) : (tp.w[i] & 0xFFFFu));
                            }
                            if (ws != wq || want + (uint32_t)SLOT > (uint32_t)(kTubeSlots * SLOT)) ++bad;
                        }
                    }
                }
    return bad;
}
extern "C" long emul_check_tube_pair1(int step) {
    long bad = check_tube_pair1_slot<4>(step) + check_tube_pair1_slot<8>(step) + check_tube_pair1_slot<24>(step);
    // u == 3 band fields: element q and element 8 - q sit at mirrored fields of the ten (the centre at 4 and 5)
    for (int q = 0; q < 9; ++q)
        if (q != 4 && tube3_field(q) + tube3_field(8 - q) != 9) ++bad;
    if (tube3_field(4) != 4) ++bad;
    return bad;
}

// tube band (mulut_core.h): the slot map must be injective on the 991 tube rows and stay inside
// [0, kTubeSlots) for EVERY key combination; the packed pair math must give, for in-tube passes, the
// tube slots of the scalar simplex rows with the same weight per row, for every bias the kernels use.
extern "C" long emul_check_tube_pair(int step) {
    long bad = 0;
    {
        static int owner[kTubeSlots];
        for (int i = 0; i < kTubeSlots; ++i) owner[i] = -1;
        int rows = 0;
        for (int A = 0; A < kL; ++A)
            for (int B = 0; B < kL; ++B)
                for (int C = 0; C < kL; ++C)
                    for (int D = 0; D < kL; ++D) {
                        const int s = tube_slot(A, B, C, D);
                        if (s < 0 || s >= kTubeSlots) { ++bad; continue; }
                        if (!tube_contains(A, B, C, D)) continue;
                        ++rows;
                        if (owner[s] != -1) ++bad;
                        owner[s] = ((A * kL + B) * kL + C) * kL + D;
                    }
        if (rows != 991) ++bad;
        // bank property: rows whose keys differ by at most one step each never share a 16-byte bank group
        for (int a = -1; a <= 1; ++a)
            for (int b = -1; b <= 1; ++b)
                for (int c = -1; c <= 1; ++c)
                    for (int d = -1; d <= 1; ++d)
                        if ((a || b || c || d) && ((a * kTubeSA + b * kTubeSB + c * kTubeSC + d * kTubeSD) & 15) == 0) ++bad;
    }
    for (uint32_t bias = 0; bias <= 32768u; bias += 32768u)
        for (int va = 0; va < 256; va += step)
            for (int vb = 0; vb < 256; vb += step)
                for (int vc = 0; vc < 256; vc += step)
                    for (int vd = 0; vd < 256; vd += 1) {
                        const int vb2 = (vb * 7 + 3) & 255, vc2 = (vc * 5 + 11) & 255, vd2 = 255 - vd;
                        TubePair tp;
                        const uint32_t ca = pixel_code(va);
                        simplex4_tube_pair(tube_anchor_key(ca), tube_anchor_h16(ca), pk_mad(tube_anchor_h16(ca), pk_dup(kTubeSA), pk_dup(bias)),
                                           pixel_code(vb) | (pixel_code(vb2) << 16), pixel_code(vc) | (pixel_code(vc2) << 16),
                                           pixel_code(vd) | (pixel_code(vd2) << 16), tp);
                        for (int half = 0; half < 2; ++half) {
                            const int b = half ? vb2 : vb, c = half ? vc2 : vc, d = half ? vd2 : vd;
                            int idx[5], w[5];
                            simplex4(va, b, c, d, idx, w);
                            const int ha = va >> 4, hb = b >> 4, hc = c >> 4, hd = d >> 4;
                            const int mx = imax(imax(ha, hb), imax(hc, hd)), mn = imin(imin(ha, hb), imin(hc, hd));
                            const bool in = mx - mn <= 1;
                            const uint32_t t = half ? (tp.t_oob >> 16) : (tp.t_oob & 0xFFFFu);
                            if ((t == 0u) != in) { ++bad; continue; }
                            int wsum = 0;
                            uint32_t rows_s[5], rows_p[5];
                            int wt_s[5], wt_p[5];
                            for (int j = 0; j < 5; ++j) {
                                const int wj = (int)(half ? (tp.w[j] >> 16) : (tp.w[j] & 0xFFFFu));
                                wsum += wj;
                                const int A = idx[j] / kStrideA, B = (idx[j] / kStrideB) % kL, C = (idx[j] / kStrideC) % kL, D = idx[j] % kL;
                                if (in && !tube_contains(A, B, C, D)) ++bad;
                                rows_s[j] = (uint32_t)tube_slot(A, B, C, D) * 16u + bias;
                                wt_s[j] = w[j];
                                {
                                    uint32_t ar[4];
                                    tube_pair_rows(tp, half, ar);
                                    rows_p[j] = j < 4 ? ar[j] : ar[0] + (uint32_t)(kTubeAll * 16);
                                }
                                wt_p[j] = wj;
                                // in the tube or not, every offset the kernel would read lies inside the plane
                                if (rows_p[j] < bias || rows_p[j] - bias > (uint32_t)(kTubePlaneBytes - 16) || (rows_p[j] & 15u)) ++bad;
                            }
                            for (int j = 0; j < 5; ++j) {
                                int ws = 0, wp = 0;
                                for (int i = 0; i < 5; ++i) {
                                    if (rows_s[i] == rows_s[j]) ws += wt_s[i];
                                    if (rows_p[i] == rows_s[j]) wp += wt_p[i];
                                }
                                if (ws != wp) ++bad;
                            }
                            if (wsum != kQ) ++bad;
                        }
                    }
    return bad;
}

extern "C" int emul_stage(const int8_t *const *luts, const char *modes, int M, int is_last, const uint8_t *in_chw,
                          int H, int W, int C, int u, uint8_t *out_hwc) {
    if (C > 3) return -1;
    switch (u) {
        case 2: stage_up<2>(luts, modes, M, in_chw, H, W, C, out_hwc, is_last); return 0;
        case 3: stage_up<3>(luts, modes, M, in_chw, H, W, C, out_hwc, is_last); return 0;
        case 4: stage_up<4>(luts, modes, M, in_chw, H, W, C, out_hwc, is_last); return 0;
        case 1: break;
        default: return -1;
    }
    // u == 1: the K1 arithmetic (int8 table, int32 accumulate)
    const DivMagic dv = make_div_magic((uint32_t)stage_divisor(M, is_last));
    const int bias = stage_bias_num(M, is_last);
    for (int c = 0; c < C; ++c)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                const uint8_t *pl = in_chw + (size_t)c * H * W;
                int acc = 0;
                for (int m = 0; m < M; ++m) {
                    int di[3], dj[3];
                    if (!pattern_offsets(modes[m], di, dj)) return -2;
                    for (int r = 0; r < 4; ++r) {
                        int v[3];
                        for (int k = 0; k < 3; ++k) {
                            int dy, dx;
                            sample_offset(r, di[k], dj[k], dy, dx);
                            v[k] = pl[(size_t)clampi(y + dy, 0, H - 1) * W + clampi(x + dx, 0, W - 1)];
                        }
                        int idx[5], w[5];
                        simplex4(pl[(size_t)y * W + x], v[0], v[1], v[2], idx, w);
                        for (int j = 0; j < 5; ++j) acc += w[j] * (int)luts[m][idx[j]];
                    }
                }
                out_hwc[((size_t)y * W + x) * C + c] = (uint8_t)rhe_clip_u8(acc + bias, dv);
            }
    return 0;
}
