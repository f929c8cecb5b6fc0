// old_generations.hip -- kernel generations superseded in rounds 1-3, moved out of the product translation units in round 3.
// NOT part of libmulut_hip.so.  Kept as the record of what was measured against (profiles/r01_*, r02_*):
//   stage_u1_kernel      first K1: one LDS read per neighbour, full table in LDS
//   u1w_mode             the unpacked window kernel body (49 VALU instructions per pass)
//   site_flag_kernel     per-pixel tube flags ahead of the tube kernel (conservative 5x5 test)
//   stage_band_kernel    compact diagonal band in LDS (8-bit rows)
//   stage_bandx_kernel   expanded band, one mode resident, mode-outer loops (+ the MULUT_PROFILE phase stamps)
//   dirty_*_kernel       the tube kernel's dirty samples through the anchor-slab path
// with the MULUT_ABLATE / MULUT_VARIANT timing-only switches they were studied with.  To revive one: include mulut_dev.h,
// paste the kernel back next to its launcher and add the launcher to mulut_kernels.h.

#include "../../mulut_amd/csrc/mulut_dev.h"
namespace mulut {

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
}  // namespace mulut
