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
#include <string.h>

#include <mutex>
#include <set>
#include <utility>

#include "mulut_dev.h"

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

hipError_t raise_lds_limit(const void *kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<int, const void *>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({dev, kernel})) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) done.insert({dev, kernel});
    return e;
}

hipError_t launch_pass(const PassArgs &a, hipStream_t st) {
    const long long nsite = (long long)a.C * a.H * a.W;
    const int nb = (int)((nsite + 255) / 256);
    hipLaunchKernelGGL(pass_kernel, dim3(nb), dim3(256), 0, st, a);
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
        const int k1 = ((n + a.k1_n0) * a.k1_tiles_y + (y0 - a.k1_oy0) / 64) * a.k1_tiles_x + x0 / 64;
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
    // routed launch of a u == 2 / u == 3 final stage (a.tile_list set, not HY): one workgroup per 64 x 64 tile of the tube-band kernel
    // (stage_u1t_kernel: its tile grid starts at (oy0, 0) like this one); it leaves at once unless that kernel marked the tile as
    // detailed and left it out, else walks the tile's TW x TH sub-tiles
    const bool routed = !HY && a.tile_list != nullptr;
    int n_sub = 1, ky0 = 0, kx0 = 0;
    if constexpr (HY) {
        if ((int)a.verdict[id] != a.verdict_take) return;              // block-uniform
        int b = id;
        const int vx = b % a.vt_x;
        b /= a.vt_x;
        const int vy = b % a.vt_y;
        n = b / a.vt_y;
        y0 = a.oy0 + vy * 16 + (sub >> 1) * TH;
        x0 = vx * 64 + (sub & 1) * TW;
    } else if (routed) {
        if (a.tile_list[id] == 0u) return;                              // block-uniform
        const int kx = (a.W + 63) >> 6, ky = (a.oy1 - a.oy0 + 63) >> 6;
        int b = id;
        kx0 = (b % kx) * 64;
        b /= kx;
        ky0 = a.oy0 + (b % ky) * 64;
        n = b / ky;
        n_sub = (64 / TW) * (64 / TH);
        y0 = ky0; x0 = kx0;
    } else {
        decode_tile(a, id, n, y0, x0, TW, TH);
    }
    for (int q = 0; q < n_sub; ++q) {
    if (routed) {
        y0 = ky0 + (q / (64 / TW)) * TH;
        x0 = kx0 + (q % (64 / TW)) * TW;
        if (y0 >= a.oy1 || x0 >= a.W) continue;                         // block-uniform
        if (q) __syncthreads();                                         // everyone is done with the previous sub-tile's pixels
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
    if (y >= a.oy1 || x >= a.W) continue;      // (no barrier below in this trip; the next trip's barrier is reached by every thread)

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
    // a.tile_list set: routed launch, one workgroup per 64 x 64 tile of the tube-band kernel that wrote the marks
    const long long nb = a.tile_list ? (long long)a.N * ((a.W + 63) >> 6) * ((a.oy1 - a.oy0 + 63) >> 6) : (long long)a.N * a.tiles_x * a.tiles_y;
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
    if (a.C > 3) return hipErrorInvalidValue;
    const size_t tile_bytes = ((3 * (size_t)(K2_TH + 2 * kHalo) * (K2_TW + 2 * kHalo) + 15) / 16) * 16;
    if (a.verdict_take >= 0) {      // hybrid launch: the tiles the statistic marked for this kernel (four blocks per 64x16 verdict tile)
        const long long nbv = 4LL * a.N * a.vt_x * a.vt_y;
        if (nbv <= 0 || nbv > 0x7fffffffLL) return hipErrorInvalidValue;
        hipLaunchKernelGGL((stage_up_kernel<4, kOutGeneric, K2_TW, K2_TH, true, true>), dim3((unsigned)nbv), dim3(K2_TW * K2_TH), tile_bytes, st, a);
        return hipGetLastError();
    }
    const long long nb = (long long)a.N * a.tiles_x * a.tiles_y;
    if (nb <= 0 || nb > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL((stage_up_kernel<4, kOutGeneric, K2_TW, K2_TH, false, true>), dim3((unsigned)nb), dim3(K2_TW * K2_TH), tile_bytes, st, a);
    return hipGetLastError();
}

void stage_band_tile(int &tw, int &th) { tw = KB_TW; th = KB_TH; }

hipError_t launch_tile_stat(const StageArgs &a, uint32_t *verdict, uint32_t max_oob_per_1024, hipStream_t st, uint16_t *thist, uint32_t *any) {
    const long long nb = (long long)a.N * a.tiles_x * a.tiles_y;   // a.tiles_* must be the 64x16 tiling
    if (nb <= 0 || nb > 0x7fffffffLL || a.C > 3) return hipErrorInvalidValue;
    hipLaunchKernelGGL((tile_stat_kernel<KB_TW, KB_TH>), dim3((unsigned)nb), dim3(256), 0, st, a, verdict, max_oob_per_1024, thist, any);
    return hipGetLastError();
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
template <int PAT, int R, int PW>
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
    simplex4_tube_pair<false>(k0, ha16, base_a, pb, pc, pd, bp);
    // A pass outside the tube still walks the band (any key combination maps to a slot inside it, so the reads stay
    // in range) and adds garbage; the site is marked by the per-pass test here and recomputed from the full table by the fix-up kernel.
    dirty |= bp.t_oob;
    tube_rows<R, 0, IMM>(smem, bp, acc);
    tube_rows<R + 2, 1, IMM>(smem, bp, acc);
}

template <int PAT, int PW>
__device__ __forceinline__ void tube_mode(const uint8_t *smem, uint32_t win, uint32_t k0, uint32_t ha16, uint32_t ha27, RotAcc<4> &acc, uint32_t &dirty) {
    const uint32_t base_a = ha27 + pk_dup((uint32_t)tube_bias(PAT));
    tube_pair<PAT, 0, PW>(smem, win, k0, ha16, base_a, acc, dirty);
    tube_pair<PAT, 1, PW>(smem, win, k0, ha16, base_a, acc, dirty);
}

template <int OUT, int TW, int TH>
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
            uint32_t dmask = 0u;      // bit c: channel c is dirty
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
                    if (pat == 0) tube_mode<0, PW>(smem, win, k0, ha16, ha27, acc, dirty);
                    else if (pat == 1) tube_mode<1, PW>(smem, win, k0, ha16, ha27, acc, dirty);
                    else tube_mode<2, PW>(smem, win, k0, ha16, ha27, acc, dirty);
                }
                if constexpr (OUT == kOutPackedRGBU4) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { o0[k] = o1[k]; o1[k] = o2[k]; }
                    tube_finish_rows(a, acc, o2);
                } else {
                    uint32_t o[4];
                    finish_channel<4, OUT>(a, acc, n, c, y, x, o);
                }
                dmask |= (dirty != 0u ? 1u : 0u) << c;
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
        if (n >= a.N || y < a.oy0 || y >= a.oy1) continue;      // never follow an entry outside the launch (a list bug must show as a wrong pixel, not as a memory fault)
        const int c_lo = only == 3u ? 0 : (int)only, c_hi = only == 3u ? imin(a.C, 3) : imin((int)only + 1, a.C);
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

// stage_up_fix2_kernel with the list walk software-pipelined: an entry costs three dependent trips to memory (the entry, its
// pixels, the table rows); here a group reads the entry two iterations ahead and the pixels one iteration ahead, so that what is
// left per iteration is the trip for the rows.  (An experiment: no faster than stage_up_fix2_kernel, see launch_stage_up_fix.)  Same arithmetic, same order of the
// integer sums.  An entry that names every channel (border columns of the detailed-tile path) takes its first channel through
// the pipeline and the others in a plain loop.
__global__ void __launch_bounds__(256) stage_up_fix3_kernel(StageArgs a) {
    __shared__ int s_sum[16][16];
    const uint32_t count = *a.fix_count;
    const int grp = (int)(threadIdx.x >> 4), ln = (int)(threadIdx.x & 15);
    const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
    const int unbias = 128 * kQ * 4 * a.M - a.bias_num;
    const uint32_t stride = gridDim.x * 16u;
    constexpr uint32_t kNone = 0xFFFFFFFFu;      // (a pixel id never has all of its low 30 bits set: N H W < 2^30)
    struct Pix { int x, y, n, va, v[3]; };
    auto decode = [&](uint32_t ent, int &x, int &y, int &n) {
        const uint32_t id = ent & 0x3FFFFFFFu;
        x = (int)(id % (uint32_t)a.W); y = (int)((id / (uint32_t)a.W) % (uint32_t)a.H); n = (int)(id / ((uint32_t)a.W * (uint32_t)a.H));
    };
    // the anchor and this lane's three neighbours (pass ln: mode ln / 4, rotation ln % 4) of channel c of an entry
    auto fetch = [&](uint32_t ent, int c, Pix &p) {
        if (ent == kNone) return;
        decode(ent, p.x, p.y, p.n);
        auto px = [&](int dy, int dx) {
            const int gy = imin(imax(p.y + dy, ylo), yhi), gx = imin(imax(p.x + dx, 0), a.W - 1);
            return (int)*view_addr(a.in, p.n, c, gy, gx);
        };
        p.va = px(0, 0);
        if (ln < 4 * a.M) {
            const int m = ln >> 2, r = ln & 3;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                int dy, dx;
                sample_offset(r, a.di[m][k], a.dj[m][k], dy, dx);
                p.v[k] = px(dy, dx);
            }
        }
    };
    // passes pass0, pass0 + 16, ... of the sample whose pixels for pass pass0 are in p (later passes fetch their own), then the byte
    auto finish = [&](const Pix &p, int c) {
        s_sum[grp][ln] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        for (int q = ln; q < 4 * a.M; q += 16) {
            const int m = q >> 2, r = q & 3;
            int v[3] = {p.v[0], p.v[1], p.v[2]};
            if (q != ln) {      // more than four modes: the passes beyond the sixteenth
                auto px = [&](int dy, int dx) {
                    const int gy = imin(imax(p.y + dy, ylo), yhi), gx = imin(imax(p.x + dx, 0), a.W - 1);
                    return (int)*view_addr(a.in, p.n, c, gy, gx);
                };
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    int dy, dx;
                    sample_offset(r, a.di[m][k], a.dj[m][k], dy, dx);
                    v[k] = px(dy, dx);
                }
            }
            int idx[5], w[5];
            simplex4(p.va, v[0], v[1], v[2], idx, w);
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
                const int pos = r == 0 ? e : r == 1 ? (e & 3) * 4 + 3 - (e >> 2) : r == 2 ? 15 - e : (3 - (e & 3)) * 4 + (e >> 2);
                atomicAdd(&s_sum[grp][pos], sum);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const uint32_t b = rhe_clip_u8(s_sum[grp][ln] - unbias, a.div);
        *const_cast<uint8_t *>(view_addr(a.out, p.n, c, p.y * 4 + (ln >> 2), p.x * 4 + (ln & 3))) = (uint8_t)b;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // sums read before the next round clears them
    };
    auto first_channel = [&](uint32_t ent) { return (ent >> 30) == 3u ? 0 : (int)(ent >> 30); };

    uint32_t i = blockIdx.x * 16u + (uint32_t)grp;
    uint32_t ent_cur = i < count ? a.fix_list[i] : kNone;
    uint32_t ent_nxt = i + stride < count && i + stride >= i ? a.fix_list[i + stride] : kNone;
    Pix cur, nxt;
    cur.x = cur.y = cur.n = cur.va = 0; cur.v[0] = cur.v[1] = cur.v[2] = 0;
    nxt = cur;
    if (ent_cur != kNone) fetch(ent_cur, first_channel(ent_cur), cur);
    while (ent_cur != kNone) {      // uniform in the group
        const uint32_t i2 = i + 2u * stride;
        const uint32_t ent_nn = (i2 < count && i2 >= i) ? a.fix_list[i2] : kNone;
        if (ent_nxt != kNone) fetch(ent_nxt, first_channel(ent_nxt), nxt);
        const int c0 = first_channel(ent_cur);
        finish(cur, c0);
        if ((ent_cur >> 30) == 3u)
            for (int c = 1; c < a.C; ++c) {
                Pix p;
                p.v[0] = p.v[1] = p.v[2] = 0;
                fetch(ent_cur, c, p);
                finish(p, c);
            }
        ent_cur = ent_nxt; cur = nxt; ent_nxt = ent_nn; i += stride;
    }
}

// variant (tuning "fix_kernel"): 0 = one pass per lane (stage_up_fix2_kernel), 1 = one entry per thread (stage_up_fix_kernel),
// 2 = one pass per lane with the list walk pipelined (stage_up_fix3_kernel: measured equal to 0 within noise on every content --
// the walk is not what the kernel waits for -- kept as a variant)
hipError_t launch_stage_up_fix(const StageArgs &a, int out_mode, int num_cus, hipStream_t st, int variant) {
    if (a.C > 3 || !a.fix_list || !a.fix_count) return hipErrorInvalidValue;
    const dim3 grid((unsigned)(4 * num_cus)), block(256);
    if (a.M > 4 && variant == 1) variant = 0;      // the one-entry-per-thread kernel merges rotation pairs in 16-bit fields: four modes at most
    if (variant == 0) hipLaunchKernelGGL(stage_up_fix2_kernel, dim3((unsigned)(8 * num_cus)), block, 0, st, a);
    else if (variant == 2) hipLaunchKernelGGL(stage_up_fix3_kernel, dim3((unsigned)(8 * num_cus)), block, 0, st, a);
    else if (out_mode == kOutPlanarU4) hipLaunchKernelGGL((stage_up_fix_kernel<kOutPlanarU4>), grid, block, 0, st, a);
    else if (out_mode == kOutPackedRGBU4 && a.C == 3) hipLaunchKernelGGL((stage_up_fix_kernel<kOutPackedRGBU4>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((stage_up_fix_kernel<kOutGeneric>), grid, block, 0, st, a);
    return hipGetLastError();
}

const char *stage_tube_name(int out_mode) {
    return out_mode == kOutPackedRGBU4 ? "stage_tube_kernel<rgb>" : out_mode == kOutPlanarU4 ? "stage_tube_kernel<planar>"
                                                                                              : "stage_tube_kernel<generic>";
}

template <int OUT>
static hipError_t launch_tube_t(const StageArgs &a, const BandArgs &b, int num_cus, hipStream_t st) {
    auto kern = stage_tube_kernel<OUT, KB_TW, KB_TH>;
    {
        const hipError_t e = raise_lds_limit((const void *)kern, 160 * 1024);
        if (e != hipSuccess) return e;
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
#if defined(MULUT_TUBE2_ASM_INC)      /* timing-only generations of the blocks (tools/experiments/tube2_timing, TUBE2_DEBUG=1..5 of the generator) */
#include MULUT_TUBE2_ASM_INC
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
// LDS: [ band s | band d | band y ][ 16 wave images of pixel codes ][ parked output rows of channels 0 and 1 (RGB path) ][ 16 fix-up buffers ][ work counter ]
constexpr int kT2FixCap = 3 * kT2W * kT2H;      // a wave's fix-up buffer holds the flagged samples of (at least) one whole wave tile
constexpr int kT2FixOff = 3 * kTubeBandBytes + 16 * kT2WaveTileBytes + 2 * 16 * KB_TW * KB_TH;      // 16 waves x kT2FixCap entries
constexpr int kTube2LdsBytes = kT2FixOff + 16 * kT2FixCap * 4 + 16;
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
            // does the image (columns x0 - 4 .. x0 + 19) reach beyond the frame?  Wave-uniform: the selectors of an interior tile are constants
            // (chosen per lane they are two v_cndmask_b32 each -- 12.6 cycles per instruction and SIMD, profiles/r01_ubench_valu_issue_cost.txt)
            const bool edge = __builtin_amdgcn_readfirstlane((int)(x0 - kTubeHaloX < 0 || x0 - kTubeHaloX + 4 * (DW - 1) > a.W - 4)) != 0;
#pragma unroll
            for (int k = 0; k < PER4; ++k) {
                const int i = lane + 64 * k;
                if (i < a.C * PH * DW) {
                    // bytes (b0,b1) / (b2,b3) into 16-bit lanes; a dword clamped at an image edge replicates the edge byte
                    uint32_t lo, hi;
                    if (!edge) {
                        lo = __builtin_amdgcn_perm(0u, v[k], 0x0C010C00u); hi = __builtin_amdgcn_perm(0u, v[k], 0x0C030C02u);
                    } else {
                        const int gx = x0 - kTubeHaloX + 4 * (i % DW);
                        const uint32_t sel_lo = gx < 0 ? 0x0C000C00u : gx > a.W - 4 ? 0x0C030C03u : 0x0C010C00u;
                        const uint32_t sel_hi = gx < 0 ? 0x0C000C00u : gx > a.W - 4 ? 0x0C030C03u : 0x0C030C02u;
                        lo = __builtin_amdgcn_perm(0u, v[k], sel_lo); hi = __builtin_amdgcn_perm(0u, v[k], sel_hi);
                    }
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
#elif defined(MULUT_T2_MARKERS)     /* ISA analysis only (tools/asm_stats.sh ... -DMULUT_T2_MARKERS): phase boundaries as comments in the assembly */
#define T2_STAMP(PH) asm volatile("; MULUT_T2_PHASE_END " #PH)
#else
#define T2_STAMP(PH) do { } while (0)
#endif
    // bands: slot = pattern id.  A mode list is a multiset of patterns and the numerator a plain sum over its modes, so a pattern that
    // occurs k times is computed once on a band of k-fold values (fields <= 255 k; a merged rotation pair sums to <= 8160 x modes)
    static_for<0, M>([&](auto MI) {
        constexpr int pat = t2_pat(PATS, MI);
        const uint4 *src = (const uint4 *)b.band[MI];
        uint4 *dst = (uint4 *)(smem + pat * kTubeBandBytes);
        const uint32_t k = b.scale[MI];
        if (k == 0x00010001u) {
            for (int i = threadIdx.x; i < kTubeBandBytes / 16; i += NT) dst[i] = src[i];
        } else {
            for (int i = threadIdx.x; i < kTubeBandBytes / 16; i += NT) {
                const uint4 v = src[i];
                dst[i] = make_uint4(pk_mad(v.x, k, 0u), pk_mad(v.y, k, 0u), pk_mad(v.z, k, 0u), pk_mad(v.w, k, 0u));
            }
        }
    });
    if (threadIdx.x == 0) *s_next = 0u;
    __syncthreads();          // the only barrier of the kernel
    T2_STAMP(5);

    // the wave's share of the fix-up list in the making (entries in LDS, their number in a scalar register)
    uint32_t *fix_buf = (uint32_t *)(smem + kT2FixOff) + wave * kT2FixCap;
    uint32_t fix_have = 0u;
    auto fix_flush = [&]() {
        if (fix_have == 0u) return;      // wave-uniform
        // called where lanes outside the image are masked off: the copy is shared out among the ACTIVE lanes
        const unsigned long long act = __ballot(true);
        const uint32_t rank = (uint32_t)__popcll(act & ((1ull << (threadIdx.x & 63)) - 1ull)), nact = (uint32_t)__popcll(act);
        uint32_t at = 0u;
        if (rank == 0u) at = atomicAdd(a.fix_count, fix_have);
        at = (uint32_t)__builtin_amdgcn_readfirstlane((int)at);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the entries other lanes of this wave stored (LDS serves a wave's operations in order)
        for (uint32_t i = rank; i < fix_have; i += nact) a.fix_list[at + i] = fix_buf[i];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // read before the buffer is filled again
        fix_have = 0u;
    };
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
            // more than four modes: the numerator does not fit a signed 16-bit field any more (|K| <= 8192 x modes); the pair sums still fit
            // their unsigned fields (8160 x modes <= 65280 for 8 modes), so such lists start from zero and take the biased-sum epilogue
            const bool wide = a.M > 4;      // wave-uniform
            auto acc_start = [&]() {
                acc.clear();
                if (OUT != kOutGeneric && !wide) {
                    const uint32_t nb = pk_dup((uint32_t)(65536 - 128 * kQ * 4 * a.M));      // (a.M: the modes of the list, not the patterns)
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
                if (wide) {
                    tube_finish_rows_wide(a, acc, o);
                    if constexpr (OUT == kOutPackedRGBU4) {
                        if (c < 2) park()[c * NT] = make_uint4(o[0], o[1], o[2], o[3]);
                    } else {
                        int n2, y2, x2, lx2, ly2;
                        site(n2, y2, x2, lx2, ly2);
#pragma unroll
                        for (int sy = 0; sy < 4; ++sy) {
                            if constexpr (OUT == kOutPlanarU4) {
                                *(uint32_t *)const_cast<uint8_t *>(view_addr(a.out, n2, c, y2 * 4 + sy, x2 * 4)) = o[sy];
                            } else {       // any layout: byte by byte (finish_channel's generic form merges the pairs in 16-bit fields: not for these lists)
#pragma unroll
                                for (int sx = 0; sx < 4; ++sx) *const_cast<uint8_t *>(view_addr(a.out, n2, c, y2 * 4 + sy, x2 * 4 + sx)) = (uint8_t)(o[sy] >> (8 * sx));
                            }
                        }
                    }
                } else if constexpr (OUT != kOutGeneric) {
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
            // dirty samples (pixel, channel) are collected in the wave's LDS buffer and appended to the fix-up list when it is full (and
            // when the wave runs out of work): one memory-side atomic per ~190 flagged samples instead of one per wave tile and channel
#if defined(MULUT_VARIANT_nofixlist)    /* timing-only: nothing is listed (flagged samples stay wrong) */
            if (dmask == 0xFFFFFFFFu) a.fix_list[0] = 0u;
#else
            if (__ballot(dmask != 0u) != 0ull) {
                const uint32_t pixel_id = (uint32_t)((n * a.H + y) * a.W + x);
                const unsigned long long m0 = __ballot((dmask & 1u) != 0u), m1 = __ballot((dmask & 2u) != 0u), m2 = __ballot((dmask & 4u) != 0u);
                const uint32_t n0 = (uint32_t)__popcll(m0), n1 = (uint32_t)__popcll(m1), n2 = (uint32_t)__popcll(m2);
                if (fix_have + n0 + n1 + n2 > (uint32_t)kT2FixCap) fix_flush();
                const unsigned long long below = (1ull << (threadIdx.x & 63)) - 1ull;
                if (dmask & 1u) fix_buf[fix_have + (uint32_t)__popcll(m0 & below)] = pixel_id;
                if (dmask & 2u) fix_buf[fix_have + n0 + (uint32_t)__popcll(m1 & below)] = pixel_id | (1u << 30);
                if (dmask & 4u) fix_buf[fix_have + n0 + n1 + (uint32_t)__popcll(m2 & below)] = pixel_id | (2u << 30);
                fix_have += n0 + n1 + n2;
            }
#endif
        }
        T2_STAMP(3);       // output stores, fix-up list
        stash(nxt_item, pix);      // the wave's image is its own: every read of the current tile has returned (the pipeline drained)
        T2_STAMP(4);       // next tile's pixel codes into LDS (waits for its fetch)
        item = nxt_item;
    }
    fix_flush();
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

// how many modes of the launch have pattern s, d, y; false if a mode has none of them
static bool tube2_counts(const StageArgs &a, int (&cnt)[3]) {
    cnt[0] = cnt[1] = cnt[2] = 0;
    for (int m = 0; m < a.M; ++m) {
        int di[3], dj[3], p = -1;
        for (int q = 0; q < 3 && p < 0; ++q) {
            pattern_offsets("sdy"[q], di, dj);
            if (memcmp(di, a.di[m], sizeof(di)) == 0 && memcmp(dj, a.dj[m], sizeof(dj)) == 0) p = q;
        }
        if (p < 0) return false;
        ++cnt[p];
    }
    return true;
}
// The kernel is built for the pattern set {s, d, y}; any mode list that uses all three runs on it, in any order and with repeats
// (kMaxTube2Modes = 8 in all: a merged rotation pair sums to <= 8160 x modes in its unsigned 16-bit fields).  Lists that lack a pattern would
// pay for its passes: they stay with stage_tube_kernel.
bool stage_tube2_supported(const StageArgs &a) {
    int cnt[3];
    // the float epilogue must be exact for the divisor (StageArgs::use_f32, proven at configure time), the bias the numerator bias of a final stage
    // (up to four modes: the float epilogue on the signed 16-bit numerators must be exact for the divisor; five to eight: sums in 32 bits, float
    // or integer epilogue)
    return a.C <= 3 && a.M <= kMaxTube2Modes && tube2_counts(a, cnt) && cnt[0] && cnt[1] && cnt[2] && (a.M > 4 || a.use_f32) &&
           a.bias_num == 0;
}

template <int OUT>
static hipError_t launch_tube2_t(const StageArgs &a, const BandArgs &b, int num_cus, hipStream_t st) {
    auto kern = stage_tube2_kernel<OUT, kT2PatsSDY>;
    {
        const hipError_t e = raise_lds_limit((const void *)kern, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    const long long ntiles = (long long)a.N * a.tiles_x * a.tiles_y;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)(ntiles < num_cus ? ntiles : num_cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(KB_TW * KB_TH), (size_t)kTube2LdsBytes, st, a, b);
    return hipGetLastError();
}

// bm.band[m] = band of MODE m of the list (as for launch_stage_tube); handed to the kernel per pattern with its multiplicity
hipError_t launch_stage_tube2(const StageArgs &a, const BandArgs &bm, int out_mode, int num_cus, hipStream_t st) {
    int cnt[3];
    if (!stage_tube2_supported(a) || !tube2_counts(a, cnt)) return hipErrorInvalidValue;
    BandArgs b;
    memset(&b, 0, sizeof(b));
    for (int m = 0; m < a.M; ++m) {
        const int p = a.dj[m][0] == 2 ? 1 : a.di[m][0] == 1 ? 2 : 0;
        b.band[p] = bm.band[m];
        b.scale[p] = (uint32_t)cnt[p] * 0x00010001u;
    }
    if (out_mode == kOutPlanarU4) return launch_tube2_t<kOutPlanarU4>(a, b, num_cus, st);
    if (out_mode == kOutPackedRGBU4 && a.C == 3) return launch_tube2_t<kOutPackedRGBU4>(a, b, num_cus, st);
    return launch_tube2_t<kOutGeneric>(a, b, num_cus, st);
}

}  // namespace mulut
