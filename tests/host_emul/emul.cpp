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
                            swar_fma<RW>(lo[r], hi[r], row, (uint32_t)w[j]);
                        }
                    }
                }
                const int unbias = 128 * kQ * 4 * M;
                static_for<0, U>([&](auto SY) {
                    constexpr int sy = SY;
                    uint32_t packed = 0;
                    static_for<0, U>([&](auto SX) {
                        constexpr int sx = SX;
                        const uint32_t sum = swar_field<row_elem(0, sy, sx, U), RW>(lo[0], hi[0]) +
                                             swar_field<row_elem(1, sy, sx, U), RW>(lo[1], hi[1]) +
                                             swar_field<row_elem(2, sy, sx, U), RW>(lo[2], hi[2]) +
                                             swar_field<row_elem(3, sy, sx, U), RW>(lo[3], hi[3]);
                        packed |= rhe_clip_u8((int)sum - unbias + bias, dv) << (8 * sx);
                    });
                    o[c][sy] = packed;
                });
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
