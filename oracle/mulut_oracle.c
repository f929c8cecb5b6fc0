/*
 * mulut_oracle.c -- CPU ORACLE for the MuLUT LUT-inference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and
 * only as the checker.  The product path (mulut_amd/) never links, imports or calls it.
 *
 * It is an exact-integer restatement, written from the algorithm description in SURVEY.md 8(a),
 * of what the reference computes in floating point:
 *   - pass   : FourSimplexInterpFaster            /root/reference/sr/4_test_lut.py:14-237
 *   - stage  : mode x rotation loop + combine      /root/reference/sr/4_test_lut.py:279-306
 *   - cascade: the `for s in range(stages)` loop    /root/reference/sr/4_test_lut.py:279
 * Parity is PINNED: tests/test_oracle.py checks every function below against
 * tests/golden/{pass,pipeline}_fixtures.npz, which tests/golden/gen_golden.py produced by running
 * the reference itself in the authoring container, and against the reference's five committed
 * Set5 output PNGs (results/sr_x2sdy/Set5/X4/).
 *
 * Why integers are exact: LUT entries are int8, the five simplex weights are integers summing to
 * q = 2^interval, so q*out is an integer (|.| <= q*128); the reference's float32 products and
 * float64 sums hold these exactly, and its `pred/avg + bias` followed by np.round
 * (round-half-to-even) equals rhe((K + bias*q*avg) / (q*avg)) -- ties are exactly representable.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define MULUT_ORACLE_OK 0
#define MULUT_ORACLE_EBADMODE -1
#define MULUT_ORACLE_EBADARG -2

/* sampling patterns: (row, col) offsets of the four LUT keys a,b,c,d.
 * 's' sr/4_test_lut.py:20-23, 'd' :32-35, 'y' :43-46 */
static int pattern(char mode, int off[4][2]) {
    static const int S[4][2] = {{0, 0}, {0, 1}, {1, 0}, {1, 1}};
    static const int D[4][2] = {{0, 0}, {0, 2}, {2, 0}, {2, 2}};
    static const int Y[4][2] = {{0, 0}, {1, 1}, {1, 2}, {2, 1}};
    const int(*p)[2];
    switch (mode) {
        case 's': p = S; break;
        case 'd': p = D; break;
        case 'y': p = Y; break;
        default: return MULUT_ORACLE_EBADMODE; /* reference raises ValueError, :54 */
    }
    memcpy(off, p, sizeof(S));
    return MULUT_ORACLE_OK;
}

static inline int imin(int a, int b) { return a < b ? a : b; }
static inline int imax(int a, int b) { return a > b ? a : b; }

/* Where rotation r (the driver's np.rot90(img, r) + bottom/right edge pad, :294-296) makes the key
 * with pattern offset (di,dj) of output site (y,x) come from, in the UN-rotated image. */
static inline void sample_coord(int r, int y, int x, int di, int dj, int H, int W, int *sy, int *sx) {
    switch (r & 3) {
        case 0: *sy = imin(y + di, H - 1); *sx = imin(x + dj, W - 1); break;
        case 1: *sy = imin(y + dj, H - 1); *sx = imax(x - di, 0); break;
        case 2: *sy = imax(y - di, 0);     *sx = imax(x - dj, 0); break;
        default: *sy = imax(y - dj, 0);    *sx = imin(x + di, W - 1); break;
    }
}

/* Which element of the u*u LUT row lands on HR sub-pixel (sy,sx) after the rotate-back
 * np.rot90(out, 4-r) (:232-235). */
static inline int row_elem(int r, int sy, int sx, int u) {
    switch (r & 3) {
        case 0: return sy * u + sx;
        case 1: return (u - 1 - sx) * u + sy;
        case 2: return (u - 1 - sy) * u + (u - 1 - sx);
        default: return sx * u + (u - 1 - sy);
    }
}

/* 4-simplex vertex walk for one site: given the four key values, produce the five LUT row
 * indices and the five integer weights (sum q). :56-109 (corner indices), :140-230 (24 cases). */
static inline void simplex(const int v[4], int interval, int64_t idx[5], int wgt[5]) {
    const int q = 1 << interval;
    const int L = (1 << (8 - interval)) + 1;
    const int64_t stride[4] = {(int64_t)L * L * L, (int64_t)L * L, L, 1};
    int f[4], ord[4] = {0, 1, 2, 3};
    int64_t base = 0;
    for (int k = 0; k < 4; ++k) {
        base += (int64_t)(v[k] >> interval) * stride[k];
        f[k] = v[k] & (q - 1);
    }
    /* insertion sort of dimension ids by fractional part, descending (ties: zero-weight vertex) */
    for (int i = 1; i < 4; ++i) {
        int o = ord[i], j = i - 1;
        while (j >= 0 && f[ord[j]] < f[o]) { ord[j + 1] = ord[j]; --j; }
        ord[j + 1] = o;
    }
    idx[0] = base;
    wgt[0] = q - f[ord[0]];
    for (int j = 1; j < 5; ++j) {
        idx[j] = idx[j - 1] + stride[ord[j - 1]];
        wgt[j] = f[ord[j - 1]] - (j < 4 ? f[ord[j]] : 0);
    }
}

/* ---- pass: mirrors FourSimplexInterpFaster but returns q*out as int32 --------------------------
 * lut : int8 [L^4][u*u]   (np.load(...).reshape(-1, v_num), :333)
 * in  : uint8 planar [C][H][W]  (UN-padded, UN-rotated; the rotation/pad is folded into `r`)
 * out : int32 planar [C][H*u][W*u] in the un-rotated frame
 * r   : the driver's rotation counter (0..3); the reference function itself receives rot=4-r. */
int mulut_oracle_pass(const int8_t *lut, const uint8_t *in, int H, int W, int C, int interval, int u,
                      char mode, int r, int32_t *out, int accumulate) {
    int off[4][2];
    if (pattern(mode, off)) return MULUT_ORACLE_EBADMODE;
    if (H <= 0 || W <= 0 || C <= 0 || u <= 0 || interval < 1 || interval > 7) return MULUT_ORACLE_EBADARG;
    const int uu = u * u;
    const int Ho = H * u, Wo = W * u;
    for (int c = 0; c < C; ++c) {
        const uint8_t *pl = in + (size_t)c * H * W;
        int32_t *po = out + (size_t)c * Ho * Wo;
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                int v[4], wgt[5];
                int64_t idx[5];
                for (int k = 0; k < 4; ++k) {
                    int yy, xx;
                    sample_coord(r, y, x, off[k][0], off[k][1], H, W, &yy, &xx);
                    v[k] = pl[(size_t)yy * W + xx];
                }
                simplex(v, interval, idx, wgt);
                for (int sy = 0; sy < u; ++sy)
                    for (int sx = 0; sx < u; ++sx) {
                        const int e = row_elem(r, sy, sx, u);
                        int32_t acc = 0;
                        for (int j = 0; j < 5; ++j) acc += wgt[j] * (int32_t)lut[idx[j] * uu + e];
                        int32_t *dst = &po[(size_t)(y * u + sy) * Wo + (x * u + sx)];
                        *dst = accumulate ? *dst + acc : acc;
                    }
            }
    }
    return MULUT_ORACLE_OK;
}

/* round-half-to-even of n/d, d > 0 (np.round, :302) */
static inline int32_t rhe_div(int64_t n, int64_t d) {
    int64_t qf = n / d, rm = n % d;
    if (rm < 0) { rm += d; qf -= 1; } /* floor division */
    if (2 * rm > d || (2 * rm == d && (qf & 1))) qf += 1;
    return (int32_t)qf;
}

/* ---- stage: sum over modes x 4 rotations, then combine/round/clip (:279-306) -------------------
 * luts[m] : table of modes[m];  in/out planar uint8;  is_last selects (avg,bias) = (M,0) vs (4M,127)
 * scratch : int32 [C*H*u*W*u] or NULL (allocated here). */
int mulut_oracle_stage(const int8_t *const *luts, const char *modes, int M, int is_last, const uint8_t *in, int H,
                       int W, int C, int interval, int u, uint8_t *out) {
    const size_t n = (size_t)C * H * u * W * u;
    int32_t *K = (int32_t *)calloc(n, sizeof(int32_t));
    if (!K) return MULUT_ORACLE_EBADARG;
    for (int m = 0; m < M; ++m)
        for (int r = 0; r < 4; ++r) {
            int rc = mulut_oracle_pass(luts[m], in, H, W, C, interval, u, modes[m], r, K, 1);
            if (rc) { free(K); return rc; }
        }
    const int64_t q = 1 << interval;
    const int64_t d = is_last ? q * M : q * 4 * M;
    const int64_t b = is_last ? 0 : 127 * d;
    for (size_t i = 0; i < n; ++i) {
        int32_t vv = rhe_div((int64_t)K[i] + b, d);
        out[i] = (uint8_t)(vv < 0 ? 0 : (vv > 255 ? 255 : vv));
    }
    free(K);
    return MULUT_ORACLE_OK;
}

/* ---- cascade: S stages, the last one with upscale `scale` (:279-306) ---------------------------
 * luts : S*M tables ordered [stage][mode]; planar uint8 in [C][H][W], out [C][H*scale][W*scale] */
int mulut_oracle_pipeline(const int8_t *const *luts, int S, const char *modes, int M, int scale, int interval,
                          const uint8_t *in, int H, int W, int C, uint8_t *out) {
    if (S < 1) return MULUT_ORACLE_EBADARG;
    const size_t n = (size_t)C * H * W;
    uint8_t *cur = (uint8_t *)malloc(n), *nxt = (uint8_t *)malloc(n);
    if (!cur || !nxt) { free(cur); free(nxt); return MULUT_ORACLE_EBADARG; }
    memcpy(cur, in, n);
    int rc = MULUT_ORACLE_OK;
    for (int s = 0; s < S && !rc; ++s) {
        if (s + 1 == S) {
            rc = mulut_oracle_stage(luts + (size_t)s * M, modes, M, 1, cur, H, W, C, interval, scale, out);
        } else {
            rc = mulut_oracle_stage(luts + (size_t)s * M, modes, M, 0, cur, H, W, C, interval, 1, nxt);
            uint8_t *t = cur; cur = nxt; nxt = t;
        }
    }
    free(cur);
    free(nxt);
    return rc;
}
