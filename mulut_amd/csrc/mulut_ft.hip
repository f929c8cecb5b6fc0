// mulut_ft.hip -- LUT-aware fine-tuning kernels: the differentiable twin of the inference path.
//
// Reference: MuLUT.InterpTorchBatch + MuLUT.forward, sr/model.py:69-312 (driver sr/3_finetune_lut.py).
// One kernel pair per STAGE:
//   ft_stage_fwd  per LR site and channel: all modes x 4 rotations; each pass is the reference's float32
//                 expression  ((((q-f1)*p0 + (f1-f2)*p1) + (f2-f3)*p2) + (f3-f4)*p3) + f4*p4, /q  (no FMA
//                 contraction), accumulated as  pred = round(pred + pass)  after EVERY pass (:308), then
//                 x = round(clamp(pred/avg + bias, 0, 255)) (:309).
//   ft_stage_bwd  recomputes the stage forward (for the clamp mask), then scatters
//                   d/d(table row j)  = w_j/q * g            (atomic add into the quantised-table gradient)
//                   d/d(f of rank j)  = sum_e g*(p_j - p_{j-1})/q   (atomic add into the input gradient at
//                                       the clamped source pixel: replicate-pad and rot90 backward)
//                 rounding is BPDA identity (:59-67), clamp passes gradient inside [0,255] inclusive.
// The simplex case at TIES is the reference's 24-branch cascade with strict '>' (:191-282): forward
// values do not depend on it, gradients do.
// Tables are passed already quantised (clamp(round(w*127),-127,127), :74-76) by the Python module, which
// also applies that step's backward (x127, clamp mask).
#include <hip/hip_runtime.h>

#include <cstring>
#include <type_traits>

#include "../../include/mulut.h"
#include "mulut_core.h"
#include "mulut_kernels.h"

#pragma clang fp contract(off)

#ifndef MULUT_FT_ABL
#define MULUT_FT_ABL 0      // timing-only ablations (tools/ab_bench.py variants ftabl*): never in the product build
#endif

namespace mulut {

constexpr int kMaxFtModes = MULUT_MAX_MODES;

struct FtArgs {
    const float *w[kMaxFtModes];
    float *gw[kMaxFtModes];
    const float *x;     // [B][C][H][W], values 0..255
    const float *gout;  // [B][C][H*u][W*u]
    float *out;         // [B][C][H*u][W*u]
    float *gx;          // [B][C][H][W]
    uint16_t *inside;   // optional [B][C][H][W]: bit eo of a site = the stage's clamp lets gradient through at block position eo
                        // (0 <= pred / avg + bias <= 255); the forward writes it, a backward that gets it skips the forward recomputation
    int B, C, H, W, u, M, is_last;
    int di[kMaxFtModes][3], dj[kMaxFtModes][3];
};

// table row of every tube-band slot (-1: the slot holds no row), built at compile time: the flush of a workgroup's band sums walks the
// slots instead of re-deriving (A, B, C, D) -> slot for 16 x 125 candidates per mode
struct FtTubeRows { int row[kTubeSlots]; };
constexpr FtTubeRows ft_make_tube_rows() {
    FtTubeRows t{};
    for (int i = 0; i < kTubeSlots; ++i) t.row[i] = -1;
    for (int A = 0; A < kL; ++A)
        for (int B = (A < 2 ? 0 : A - 2); B < kL && B <= A + 2; ++B)
            for (int C = (A < 2 ? 0 : A - 2); C < kL && C <= A + 2; ++C)
                for (int D = (A < 2 ? 0 : A - 2); D < kL && D <= A + 2; ++D)
                    if (tube_contains(A, B, C, D)) t.row[tube_slot(A, B, C, D)] = A * kStrideA + B * kStrideB + C * kStrideC + D;
    return t;
}
__device__ const FtTubeRows kFtTubeRows = ft_make_tube_rows();

// The reference orders the four keys by a 24-branch cascade of strict '>' comparisons (sr/model.py:191-282).  For EVERY tie pattern that
// cascade equals the stable order "f descending, on equal f the key with the higher index first" (exhaustive check over all orderings
// and ties: tests/test_ft_order_cpu.py), so the rank of key i is the number of keys that come before it:
//   rank_i = #{ j > i : f_j >= f_i } + #{ j < i : f_j > f_i }.
// f lies in [0, 16): the int32 patterns of non-negative floats order like the floats, and [f_j < f_i] is the sign bit of their
// difference -- the ranks are adds and shifts, no compare + select (a v_cndmask_b32 costs six full-rate instructions on this chip and
// the cascade was ~20 of them per pass, the selects by rank below another ~36).
// Returns the keys by rank, two bits each: key of rank j in bits 2j, 2j + 1.
__device__ __forceinline__ int ft_order_code(float fa, float fb, float fc, float fd) {
    const int a = __float_as_int(fa), b = __float_as_int(fb), c = __float_as_int(fc), d = __float_as_int(fd);
    auto lt = [](int x, int y) { return (int)((unsigned)(x - y) >> 31); };      // [x < y] for 0 <= x, y < 2^31
    const int s10 = lt(b, a), s20 = lt(c, a), s30 = lt(d, a), s21 = lt(c, b), s31 = lt(d, b), s32 = lt(d, c);
    const int r1 = s10 + 2 - s21 - s31, r2 = s20 + s21 + 1 - s32, r3 = s30 + s31 + s32;      // (rank of key a: 3 - s10 - s20 - s30, contributes 0)
    return (1 << (2 * r1)) | (2 << (2 * r2)) | (3 << (2 * r3));
}

struct FtPass {
    int idx[5];      // table rows of the five vertices
    float wt[5];     // q-f1, f1-f2, f2-f3, f3-f4, f4
    int tslot[5];    // tube-band slots of the five vertices (mulut_core.h), valid when in_tube
    bool in_tube;    // the four MSBs span at most one step: every vertex lies in the 1041-slot tube band
    int ord;         // key (0 = a ... 3 = d) of rank j in bits 2j, 2j + 1
};

__device__ __forceinline__ void ft_pass_setup(const float *plane, int H, int W, int y, int x, int r, const int (&di)[3],
                                              const int (&dj)[3], FtPass &p) {
    int pix[4];
    float v[4];
    pix[0] = y * W + x;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        int dy, dx;
        sample_offset(r, di[k], dj[k], dy, dx);
        pix[k + 1] = imin(imax(y + dy, 0), H - 1) * W + imin(imax(x + dx, 0), W - 1);
    }
    int h[4];
    float f[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        v[k] = plane[pix[k]];
        const float hf = floorf(v[k] / (float)kQ);       // torch.floor_divide(img, q)
        h[k] = (int)hf;
        f[k] = v[k] - (float)kQ * hf;                    // img % q
    }
    const int ord = ft_order_code(f[0], f[1], f[2], f[3]);
    p.ord = ord;
    // the LSBs by rank: only the VALUES are needed, and a min / max network sorts values whatever the ties
    // (on the int32 patterns, as the ranks: integer min / max need no NaN canonicalisation of their operands)
    float fs[4];
    {
        const int i0 = __float_as_int(f[0]), i1 = __float_as_int(f[1]), i2 = __float_as_int(f[2]), i3 = __float_as_int(f[3]);
        const int a = imax(i0, i1), b = imin(i0, i1), c = imax(i2, i3), d = imin(i2, i3);
        const int t1 = imin(a, c), t2 = imax(b, d);
        fs[0] = __int_as_float(imax(a, c)); fs[1] = __int_as_float(imax(t1, t2)); fs[2] = __int_as_float(imin(t1, t2)); fs[3] = __int_as_float(imin(b, d));
    }
    // strides of the key of rank j: a shift of a packed constant by that key's id
    constexpr unsigned long long kRowStrides = (unsigned long long)kStrideA | ((unsigned long long)kStrideB << 16) | ((unsigned long long)kStrideC << 32) |
                                               ((unsigned long long)kStrideD << 48);
    constexpr uint32_t kTubeStrides = (uint32_t)kTubeSA | ((uint32_t)kTubeSB << 8) | ((uint32_t)kTubeSC << 16) | ((uint32_t)kTubeSD << 24);
    static_assert(kStrideA < 65536 && kTubeSA < 256 && kStrideD == 1, "packed stride constants");
    int ss[4], ts[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int d = (ord >> (2 * j)) & 3;
        ss[j] = (int)((uint32_t)(kRowStrides >> (16 * d)) & 0xFFFFu);
        ts[j] = (int)((kTubeStrides >> (8 * d)) & 0xFFu);
    }
    {
        p.tslot[0] = tube_slot(h[0], h[1], h[2], h[3]);
        p.tslot[1] = p.tslot[0] + ts[0];
        p.tslot[2] = p.tslot[1] + ts[1];
        p.tslot[3] = p.tslot[2] + ts[2];
        p.tslot[4] = p.tslot[3] + ts[3];
        const int hx = imax(imax(h[0], h[1]), imax(h[2], h[3])), hn = imin(imin(h[0], h[1]), imin(h[2], h[3]));
        p.in_tube = hx - hn <= 1 && hn >= 0 && hx <= 15;
    }
    p.idx[0] = h[0] * kStrideA + h[1] * kStrideB + h[2] * kStrideC + h[3];
    p.idx[1] = p.idx[0] + ss[0];
    p.idx[2] = p.idx[1] + ss[1];
    p.idx[3] = p.idx[2] + ss[2];
    p.idx[4] = p.idx[3] + ss[3];
    p.wt[0] = (float)kQ - fs[0];
    p.wt[1] = fs[0] - fs[1];
    p.wt[2] = fs[1] - fs[2];
    p.wt[3] = fs[2] - fs[3];
    p.wt[4] = fs[3];
}

// pred after all passes of the stage for one site (the order of additions and roundings is the reference's)
template <int U>
__device__ __forceinline__ void ft_site_forward(const FtArgs &a, const float *plane, int y, int x, float (&pred)[U * U]) {
#pragma unroll
    for (int e = 0; e < U * U; ++e) pred[e] = 0.0f;
    for (int m = 0; m < a.M; ++m) {
        const float *tab = a.w[m];
        const int di[3] = {a.di[m][0], a.di[m][1], a.di[m][2]}, dj[3] = {a.dj[m][0], a.dj[m][1], a.dj[m][2]};
        static_for<0, 4>([&](auto R) {
            constexpr int r = R;
            FtPass p;
            ft_pass_setup(plane, a.H, a.W, y, x, r, di, dj, p);
            const float *r0 = tab + (long long)p.idx[0] * (U * U), *r1 = tab + (long long)p.idx[1] * (U * U);
            const float *r2 = tab + (long long)p.idx[2] * (U * U), *r3 = tab + (long long)p.idx[3] * (U * U);
            const float *r4 = tab + (long long)p.idx[4] * (U * U);
            static_for<0, U * U>([&](auto E) {
                constexpr int eo = E;                                   // block position sy*U+sx
                constexpr int e = row_elem(r, eo / U, eo % U, U);       // table element landing there
                const float val = ((((p.wt[0] * r0[e] + p.wt[1] * r1[e]) + p.wt[2] * r2[e]) + p.wt[3] * r3[e]) + p.wt[4] * r4[e]) /
                                  (float)kQ;
                pred[eo] = rintf(pred[eo] + val);                       // pred += ...; pred = round_func(pred)
            });
        });
    }
}

template <int U>
__global__ void __launch_bounds__(256) ft_stage_fwd(FtArgs a) {
    const long long nsite = (long long)a.B * a.C * a.H * a.W;
    const long long s = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= nsite) return;
    const int x = (int)(s % a.W), y = (int)((s / a.W) % a.H);
    const long long bc = s / ((long long)a.W * a.H);
    const float *plane = a.x + bc * a.H * a.W;
    float pred[U * U];
    ft_site_forward<U>(a, plane, y, x, pred);
    const float avg = a.is_last ? (float)a.M : (float)(4 * a.M), bias = a.is_last ? 0.0f : 127.0f;
    float *po = a.out + bc * (long long)(a.H * U) * (a.W * U);
    uint32_t inside = 0;
    static_for<0, U * U>([&](auto E) {
        constexpr int eo = E;
        const float t0 = pred[eo] / avg + bias;
        inside |= (t0 >= 0.0f && t0 <= 255.0f) ? 1u << eo : 0u;
        const float t = fminf(fmaxf(t0, 0.0f), 255.0f);
        po[(long long)(y * U + eo / U) * (a.W * U) + (x * U + eo % U)] = rintf(t);
    });
    if (a.inside) a.inside[s] = (uint16_t)inside;
}

// g = dL/d pred of one site: the clamp's mask (saved by the forward, or from the stage forward recomputed here), then d(pred / avg)
template <int U, class F>
__device__ __forceinline__ void ft_site_g(const FtArgs &a, const float *plane, long long bc, int y, int x, bool valid, F &&put) {
    constexpr int EL = U * U;
    const float avg = a.is_last ? (float)a.M : (float)(4 * a.M), bias = a.is_last ? 0.0f : 127.0f;
    const float *pg = a.gout + bc * (long long)(a.H * U) * (a.W * U);
    if (a.inside) {
        const uint32_t inside = valid ? a.inside[(bc * a.H + y) * a.W + x] : 0u;
        static_for<0, EL>([&](auto E) {
            constexpr int eo = E;
            const float go = pg[(long long)(y * U + eo / U) * (a.W * U) + (x * U + eo % U)];
            put(eo, ((inside >> eo) & 1u) ? go / avg : 0.0f);
        });
    } else {
        float pred[EL];
#if MULUT_FT_ABL == 2
        for (int q = 0; q < EL; ++q) pred[q] = 100.0f;
#else
        ft_site_forward<U>(a, plane, y, x, pred);
#endif
        static_for<0, EL>([&](auto E) {
            constexpr int eo = E;
            const float t = pred[eo] / avg + bias;
            const float go = pg[(long long)(y * U + eo / U) * (a.W * U) + (x * U + eo % U)];
            put(eo, (valid && t >= 0.0f && t <= 255.0f) ? go / avg : 0.0f);
        });
    }
}

// float add into LDS as ds_add_f32: an atomicAdd on a pointer the compiler cannot prove to be LDS (here: one of two targets chosen
// at run time) becomes flat_atomic_add_f32, which reaches the LDS through the texture path
__device__ __forceinline__ void lds_add_f32(float *p, float v) {
    // (as an instruction of its own: written as an atomic on an address_space(3) pointer it is still merged with the global
    // atomic of the other branch into one flat atomic on a selected pointer.)  The compiler does not count this LDS operation:
    // LDS returns in order, so its own waits can only get stricter, and lds_adds_done() drains before a barrier publishes the sums.
    asm volatile("ds_add_f32 %0, %1" : : "v"((uint32_t)(uintptr_t)p), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_adds_done() { asm volatile("s_waitcnt lgkmcnt(0)" : : : "memory"); }

__device__ __forceinline__ float ft_sum16(float v) {      // sum over the 16 lanes of a DPP row; every lane gets it
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x128, 0xF, 0xF, true));      // row_ror:8
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x124, 0xF, 0xF, true));      // row_ror:4
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x122, 0xF, 0xF, true));      // row_ror:2
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xF, 0xF, true));      // row_ror:1
    return v;
}
// lane K of every 16-lane row to all lanes of its row (v_mov_b32_dpp row_newbcast:K)
template <int K> __device__ __forceinline__ int ft_bcast(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + K, 0xF, 0xF, true); }
template <int K> __device__ __forceinline__ float ft_bcast(float v) { return __int_as_float(ft_bcast<K>(__float_as_int(v))); }

// Backward of one stage.  A workgroup owns 256 consecutive sites.  Per site (one thread): recompute the stage forward for
// the clamp mask, g = dL/d pred; then per pass the five rows' dot products with g give the input gradient (one atomic
// per rank into the source pixel: adjacent sites hit adjacent floats, i.e. well-shaped 256-byte atomic instructions).
// The TABLE gradient is not scattered from the site threads (64 lanes -> 64 different rows = 64 memory-side atomic
// requests per instruction): every pass writes its (row, weight / q) items to LDS and the workgroup redistributes them
//   u > 1 : EPL lanes per item, one per row element: a wave instruction adds whole 16-float rows (64-byte segments);
//   u == 1: one float per item: items inside the tube (991 rows, the ones smooth content uses) are summed into a
//           per-workgroup LDS copy of the tube band with ds_add_f32 and flushed once per workgroup as contiguous
//           atomics; items outside it go to global memory directly.
constexpr int kFtGxTile = 1024;        // floats of a wave's input-gradient tile (ft_stage_bwd)
template <int U>
__device__ __forceinline__ int eo_of_elem(int r, int e) {      // block position whose table element is e under rotation r (inverse of row_elem)
    return r == 0 ? e : r == 1 ? U * (e % U) + (U - 1 - e / U) : r == 2 ? U * U - 1 - e : U * (U - 1 - e % U) + e / U;
}

template <int U>
__global__ void __launch_bounds__(256) ft_stage_bwd(FtArgs a) {
    constexpr int EL = U * U, EPL = EL <= 1 ? 1 : EL <= 4 ? 4 : 16, NT = 256;
    __shared__ float s_g[NT][EL + 1];
    __shared__ int s_idx[5][NT];
    __shared__ float s_wq[5][NT];
    __shared__ float s_band[U == 1 ? kMaxFtModes * kTubeSlots : 1];
    __shared__ float s_gxt[NT / 64][kFtGxTile];
    const long long nsite = (long long)a.B * a.C * a.H * a.W;
    const long long s = (long long)blockIdx.x * NT + threadIdx.x;
    const bool valid = s < nsite;
    const long long sc = valid ? s : nsite - 1;                 // surplus threads shadow the last site and contribute nothing
    const int x = (int)(sc % a.W), y = (int)((sc / a.W) % a.H);
    const long long bc = sc / ((long long)a.W * a.H);
    const float *plane = a.x + bc * a.H * a.W;
    float *gplane = a.gx + bc * a.H * a.W;
    // The input gradient: a pass gives d/d f of each key to that key's pixel.  The site's own pixel (key a) is summed in a register; for
    // key b, c or d the target is the site's pixel plus an offset that is the same for every site of the pass, so the 64 consecutive
    // sites of a WAVE hit 64 different positions of the wave's private tile -- the rows its sites lie in plus two on either side, each
    // plane with its own halo rows and columns (unclamped coordinates: replicate padding is applied when the tile is folded onto the
    // image at the end): a plain LDS read + add + write per key, no atomic (36 LDS float adds per site before, 83 % of this kernel's
    // launch with the LDS pipeline busy).  When the tile does not fit (crops wider than ~140) the adds go to memory.
    const int lane = (int)threadIdx.x & 63;
    float *tile = s_gxt[threadIdx.x >> 6];
    const int PWd = a.W + 4, PHt = a.H + 4;
    auto padded_row = [&](long long R) { return R + 4 * (R / a.H) + 2; };      // stacked image row (plane * H + y) -> row of the padded stack
    const long long w0 = (long long)blockIdx.x * NT + (threadIdx.x & ~63u);
    const long long R0 = (w0 < nsite ? w0 : nsite - 1) / a.W, R1 = (w0 + 63 < nsite ? w0 + 63 : nsite - 1) / a.W;
    const long long pr0 = padded_row(R0) - 2;
    const long long t_rows = padded_row(R1) + 2 - pr0 + 1;
    const int t_n = t_rows * PWd <= kFtGxTile ? (int)(t_rows * PWd) : 0;      // 0: does not fit
    for (int i = lane; i < t_n; i += 64) tile[i] = 0.0f;                      // (wave-private, and LDS serves a wave in order: no barrier)
    const int t_own = (int)(padded_row(bc * a.H + y) - pr0) * PWd + x + 2;
    if constexpr (U == 1)
        for (int i = threadIdx.x; i < a.M * kTubeSlots; i += NT) s_band[i] = 0.0f;
    float g[EL];
    ft_site_g<U>(a, plane, bc, y, x, valid, [&](int eo, float v) {
        g[eo] = v;
        s_g[threadIdx.x][eo] = v;
    });
    __syncthreads();      // the band and the gradient rows must be zero before any wave adds into them
    float own = 0.0f;
    for (int m = 0; m < a.M; ++m) {
        const float *tab = a.w[m];
        float *gtab = a.gw[m];
        const int di[3] = {a.di[m][0], a.di[m][1], a.di[m][2]}, dj[3] = {a.dj[m][0], a.dj[m][1], a.dj[m][2]};
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
            FtPass p;
            ft_pass_setup(plane, a.H, a.W, y, x, r, di, dj, p);
            float dsum[5];   // sum_e g * p_j[e]
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const float *row = tab + (long long)p.idx[j] * EL;
                float acc = 0.0f;
#pragma unroll
                for (int eo = 0; eo < EL; ++eo) {
                    const int e = r == 0 ? eo : r == 1 ? (U - 1 - eo % U) * U + eo / U : r == 2 ? EL - 1 - eo : (eo % U) * U + (U - 1 - eo / U);   // row_elem
                    acc += g[eo] * row[e];
                }
                dsum[j] = acc;
                const float wq = p.wt[j] / (float)kQ;
                if constexpr (U == 1) {
                    // one float per item: LDS tube band when the pass is inside the tube, else global memory.  Neighbouring sites of smooth
                    // content hit the SAME slot, and an LDS float add serialises over its lanes and again over equal addresses (this
                    // kernel's LDS pipeline was busy 83 % of the launch on 96 such adds per site): the lanes of a 16-lane row that share
                    // the row leader's slot are summed by DPP first and added once, by the leader; the others add on their own.
                    // (All 256 threads are here: the broadcast and the row sum read every lane.)
                    const float v = wq * g[0];
                    const int slot = p.in_tube ? m * kTubeSlots + p.tslot[j] : -1;
                    const int lead = ft_bcast<0>(slot);
                    const bool with_lead = slot == lead && slot >= 0;
                    const float sum = ft_sum16(with_lead ? v : 0.0f);
                    if ((threadIdx.x & 15) == 0 && lead >= 0 && sum != 0.0f) lds_add_f32(&s_band[lead], sum);
                    if (!with_lead && v != 0.0f) {
                        if (slot >= 0) lds_add_f32(&s_band[slot], v);
                        else atomicAdd(&gtab[p.idx[j]], v);
                    }
                } else {
                    s_idx[j][threadIdx.x] = valid ? p.idx[j] : -1;
                    s_wq[j][threadIdx.x] = wq;
                }
            }
            // d/d f of rank j = (p_{j+1} - p_j) . g / q belongs to the key of that rank
            float df[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) df[j] = (dsum[j + 1] - dsum[j]) / (float)kQ;
            const int o0 = p.ord & 3, o1 = (p.ord >> 2) & 3, o2 = (p.ord >> 4) & 3;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dk = o0 == k ? df[0] : o1 == k ? df[1] : o2 == k ? df[2] : df[3];
                if (k == 0) own += dk;
                else {
                    int dy, dx;
                    sample_offset(r, di[k - 1], dj[k - 1], dy, dx);
                    if (t_n) {
                        if (valid) {      // (a surplus thread shadows the last site: its read + add + write would race with that site's)
                            float *t = tile + t_own + dy * PWd + dx;
                            *t = *t + dk;
                        }
                    } else if (dk != 0.0f) atomicAdd(&gplane[imin(imax(y + dy, 0), a.H - 1) * a.W + imin(imax(x + dx, 0), a.W - 1)], dk);
                }
            }
            if constexpr (U > 1) {
                __syncthreads();
                const int e = (int)threadIdx.x % EPL;
                const int eo = eo_of_elem<U>(r, e < EL ? e : 0);
                for (int it = (int)threadIdx.x / EPL; it < 5 * NT; it += NT / EPL) {
                    const int j = it / NT, si = it % NT;
                    const int idx = s_idx[j][si];
                    if (e < EL && idx >= 0) {
                        const float v = s_wq[j][si] * s_g[si][eo];
                        if (v != 0.0f) atomicAdd(&gtab[(long long)idx * EL + e], v);
                    }
                }
                __syncthreads();
            }
        }
    }
    // the wave's tile onto the image: padded row / column -> plane and pixel, clamped into the plane (replicate padding)
    if (t_n) {
        if (valid) tile[t_own] += own;
        for (int i = lane; i < t_n; i += 64) {
            const float v = tile[i];
            if (v == 0.0f) continue;
            const long long pr = pr0 + i / PWd;
            const int cx = i % PWd - 2, yy = (int)(pr % PHt) - 2;
            const long long pl = pr / PHt;
            if (pl < (long long)a.B * a.C) atomicAdd(&a.gx[(pl * a.H + imin(imax(yy, 0), a.H - 1)) * a.W + imin(imax(cx, 0), a.W - 1)], v);
        }
    } else if (valid && own != 0.0f) atomicAdd(&gplane[y * a.W + x], own);
    lds_adds_done();
    __syncthreads();
    if constexpr (U == 1) {
        // flush the workgroup's tube-band sums: contiguous floats, a wave adds 256 bytes at a time
        for (int m = 0; m < a.M; ++m) {
            float *gtab = a.gw[m];
            for (int sl = threadIdx.x; sl < kTubeSlots; sl += NT) {
                const int row = kFtTubeRows.row[sl];
                const float v = s_band[m * kTubeSlots + sl];
                if (row >= 0 && v != 0.0f) atomicAdd(&gtab[row], v);
            }
        }
    }
}

// Backward of a stage with 16-float rows (u == 4), kFtB4Sites sites per workgroup.  Photograph-like batches send thousands of
// sites to the same few hundred table rows, and float adds are dear wherever they meet: memory-side atomics serialise per
// address, and an LDS float add (ds_add_f32) occupies the CU's LDS pipeline for 48 cycles per wave instruction whatever its
// addresses (tools/ubench/ubench_lds_atomic.hip; an integer add or a plain write takes 3.8, read + add + write of a private
// address 7).  So the table gradient is summed on three levels:
//   1. a 16-lane group (lane = row element) owns 16 sites (a 4x4 block) and a PRIVATE 16-entry cache of band rows in LDS, direct
//      mapped by slot mod 16 -- the 16 corners of one MSB cell have 16 different residues (the tube strides are 11, 2, 12, 8
//      mod 16), so a group whose sites stay inside a cell never evicts.  A hit is read + add + write, no atomic;
//   2. an evicted entry (the sites moved on to another cell) and, at the end of a mode, every entry goes into the workgroup's LDS
//      copy of the tube band (1041 x 16 floats) with ds_add_f32;
//   3. the band is flushed once per workgroup and mode as contiguous memory-side atomics.
// Passes outside the tube add to global memory directly (16 lanes per row: 64-byte segments).
// The sites of a group ARE its 16 lanes, so nothing crosses a wave and nothing goes through LDS items: a site lane keeps the five
// (table row | tube slot, weight / q) of its pass in registers and the group reads them with DPP row broadcasts; the 80 table rows of
// a pass (one coalesced 64-byte read per site and vertex) are requested before the first is used.  The group then walks its sites
// ONE SITE AT A TIME, all five vertices together: a site's vertices are corners of one cell, i.e. five DIFFERENT cache entries, so
// their five tag + entry reads, adds and writes are independent and in flight together (round 3 walked vertex by vertex, one
// dependent LDS round trip per site and vertex, at 2 waves per SIMD: the kernel waited 60 % of its wave cycles).  The dot products
// g . row -- the input gradient's ingredient -- are 16-lane DPP sums and stay in the site's lane.  The cache tags sit in LDS next to
// the entries (one broadcast read each).  No barrier inside a mode.
// The INPUT gradient: a pass gives d/d f of each key to that key's pixel.  The site's own pixel (key a) is summed in a register; for
// key b, c or d the target is the site's pixel plus an offset that is the same for every site of the pass, so the 16 sites of a
// group hit 16 DIFFERENT positions of the group's 8 x 8 tile (its 4 x 4 block and a halo of two, unclamped): a plain LDS read + add
// + write per key, no atomic.  The tile goes to memory once, at the end, folded onto the image (replicate padding: a position
// outside the plane belongs to the border pixel) -- 4 memory-side atomics per site instead of 37.
// LDS: [ band gradient 1041 x 16 f32 ][ g of the sites, 17 floats each ][ caches: 16 x 16 f32 per group ][ tags: 16 per group ]
//      [ input-gradient tiles: 8 x 8 f32 per group ]
constexpr int kFtB4Sites = 512, kFtB4Groups = kFtB4Sites / 16;
constexpr int kFtB4Lds = kTubeSlots * 16 * 4 + kFtB4Sites * 17 * 4 + kFtB4Groups * 16 * 16 * 4 + kFtB4Groups * 16 * 4 + kFtB4Groups * 64 * 4;
static_assert(kFtB4Lds <= 160 * 1024, "ft_stage_bwd4: LDS");

__global__ void __launch_bounds__(kFtB4Sites) ft_stage_bwd4(FtArgs a) {
    constexpr int U = 4, EL = 16, NT = kFtB4Sites, NG = kFtB4Groups;
    extern __shared__ __attribute__((aligned(16))) uint8_t ft_smem[];
    float *s_band = (float *)ft_smem;
    float (*s_g)[17] = (float (*)[17])(ft_smem + kTubeSlots * 16 * 4);
    float *s_cache = (float *)(ft_smem + kTubeSlots * 16 * 4 + NT * 17 * 4);
    int *s_tag = (int *)(s_cache + NG * 256);             // [NG][16]
    float *s_gxt = (float *)(s_tag + NG * 16);            // [NG][8][8]
    // a group's 16 sites are a 4x4 block of one plane (lane = 4 * row + column): neighbours in both directions share MSB cells, so
    // the group's cache sees fewer cell changes than with 16 sites along a row; lanes beyond the plane shadow its last site
    const int e = (int)threadIdx.x & 15, grp = (int)threadIdx.x >> 4, first = grp * 16;
    const int bw = (a.W + 3) / 4, bh = (a.H + 3) / 4;
    const long long nblock = (long long)a.B * a.C * bh * bw, block = (long long)blockIdx.x * NG + grp;
    const long long blk = block < nblock ? block : nblock - 1;
    const long long bc = blk / ((long long)bh * bw);
    const int brem = (int)(blk % ((long long)bh * bw));
    const int y0 = (brem / bw) * 4 + (e >> 2), x0 = (brem % bw) * 4 + (e & 3);
    const bool valid = block < nblock && y0 < a.H && x0 < a.W;
    const int y = imin(y0, a.H - 1), x = imin(x0, a.W - 1);
    const float *plane = a.x + bc * a.H * a.W;
    float *gplane = a.gx + bc * a.H * a.W;
    float *gxt = s_gxt + grp * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) gxt[e + 16 * i] = 0.0f;      // (only this group touches its tile, and LDS serves a wave in order)
    float *gxt_own = gxt + (2 + (e >> 2)) * 8 + 2 + (e & 3);
    ft_site_g<U>(a, plane, bc, y, x, valid, [&](int eo, float v) { s_g[threadIdx.x][eo] = v; });
    float *cache = s_cache + grp * 256;      // [16 entries][16 elements]
    int *tags = s_tag + grp * 16;            // slot held by entry c, -1: none
    float own = 0.0f;
    for (int m = 0; m < a.M; ++m) {
        const float *tab = a.w[m];
        float *gtab = a.gw[m];
        const int di[3] = {a.di[m][0], a.di[m][1], a.di[m][2]}, dj[3] = {a.dj[m][0], a.dj[m][1], a.dj[m][2]};
        for (int i = threadIdx.x; i < kTubeSlots * 16; i += NT) s_band[i] = 0.0f;
        tags[e] = -1;
        __syncthreads();      // band zeroed (and, first trip, s_g written) before any group adds into it
#pragma unroll 1
        for (int r = 0; r < 4; ++r) {
            // this lane's site, per vertex: byte offset of the table row, tube slot (-1: the pass is outside the tube -- all five or
            // none) and weight / q; a lane without a site has row 0, weight 0
            int roff[5], slotv[5], ord;
            float wq[5];
            {
                FtPass p;
                ft_pass_setup(plane, a.H, a.W, y, x, r, di, dj, p);
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    roff[j] = valid ? p.idx[j] * (EL * 4) : 0;
                    slotv[j] = valid && p.in_tube ? p.tslot[j] : -1;
                    wq[j] = valid ? p.wt[j] / (float)kQ : 0.0f;
                }
                ord = p.ord;
            }
            const int eo = eo_of_elem<U>(r, e);
            const char *tabe = (const char *)tab;
            float rowv[16][5];
            static_for<0, 16>([&](auto K) {
#pragma unroll
                for (int j = 0; j < 5; ++j) rowv[K][j] = *(const float *)(tabe + (uint32_t)(ft_bcast<K>(roff[j]) + e * 4));
            });
            float dm[5] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f};      // g . row of this lane's site, per vertex
            static_for<0, 16>([&](auto K) {
                asm volatile("" : : : "memory");      // one site at a time: keeps the scheduler from opening every site's reads and sums at once (registers)
                const float gv = s_g[first + K][eo];
                int slot[5];
                float v[5], d[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    slot[j] = ft_bcast<K>(slotv[j]);
                    v[j] = ft_bcast<K>(wq[j]) * gv;
                    d[j] = ft_sum16(gv * rowv[K][j]);
                }
                if (e == K) {
#pragma unroll
                    for (int j = 0; j < 5; ++j) dm[j] = d[j];
                }
#if MULUT_FT_ABL != 4
                if (slot[0] >= 0) {      // inside the tube (uniform in the group)
                    int tg[5];
                    float old[5];
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        tg[j] = tags[slot[j] & 15];
                        old[j] = cache[(slot[j] & 15) * 16 + e];
                    }
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        // hit: old + v; miss: v, and the entry's sum goes into the band
                        const bool hit = tg[j] == slot[j];
                        cache[(slot[j] & 15) * 16 + e] = hit ? old[j] + v[j] : v[j];
                        if (!hit) {
                            if (tg[j] >= 0) lds_add_f32(&s_band[tg[j] * 16 + e], old[j]);
                            tags[slot[j] & 15] = slot[j];
                        }
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        // (the broadcast outside the lane-divergent test below: a DPP read of a lane that is switched off returns 0)
                        const uint32_t off = (uint32_t)(ft_bcast<K>(roff[j]) + e * 4);
                        if (v[j] != 0.0f) atomicAdd((float *)((char *)gtab + off), v[j]);
                    }
                }
#endif
            });
            // d/d f of rank j = (g . p_{j+1} - g . p_j) / q belongs to the key of that rank
            float df[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) df[j] = (dm[j + 1] - dm[j]) / (float)kQ;
            const int o0 = ord & 3, o1 = (ord >> 2) & 3, o2 = (ord >> 4) & 3;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float dk = o0 == k ? df[0] : o1 == k ? df[1] : o2 == k ? df[2] : df[3];
                if (k == 0) own += dk;
#if MULUT_FT_ABL != 3
                else {
                    int dy, dx;
                    sample_offset(r, di[k - 1], dj[k - 1], dy, dx);
                    float *t = gxt_own + dy * 8 + dx;
                    *t = *t + dk;
                }
#endif
            }
        }
        // the group's cache into the band, then the band's rows (tube rows of anchor MSB A, 16 lanes per row) into the table gradient
#pragma unroll 1
        for (int c = 0; c < 16; ++c) {
            const int t = tags[c];
            if (t >= 0) lds_add_f32(&s_band[t * 16 + e], cache[c * 16 + e]);
        }
        lds_adds_done();
        __syncthreads();
        for (int sl = grp; sl < kTubeSlots; sl += NG) {
            const int row = kFtTubeRows.row[sl];
            const float v = s_band[sl * 16 + e];
            if (row >= 0 && v != 0.0f) atomicAdd(&gtab[(long long)row * EL + e], v);
        }
        __syncthreads();
    }
    // the tile onto the image: position (ty, tx) is pixel (4 by - 2 + ty, 4 bx - 2 + tx) clamped into the plane
    *gxt_own += valid ? own : 0.0f;
    if (block < nblock) {
        const int ty0 = (brem / bw) * 4 - 2, tx0 = (brem % bw) * 4 - 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int q = e + 16 * i;
            const float v = gxt[q];
            if (v != 0.0f) atomicAdd(&gplane[imin(imax(ty0 + (q >> 3), 0), a.H - 1) * a.W + imin(imax(tx0 + (q & 7), 0), a.W - 1)], v);
        }
    }
}

template <int U>
static hipError_t launch_ft(const FtArgs &a, bool backward, hipStream_t st) {
    const long long nsite = (long long)a.B * a.C * a.H * a.W;
    const long long nb = (nsite + 255) / 256;
    if (nb <= 0 || nb > 0x7fffffffLL) return hipErrorInvalidValue;
    if (backward && U == 4) {
        {
            const hipError_t e = mulut::raise_lds_limit((const void *)ft_stage_bwd4, 160 * 1024);
            if (e != hipSuccess) return e;
        }
        const long long nblock4 = (long long)a.B * a.C * ((a.H + 3) / 4) * ((a.W + 3) / 4);      // 4x4 site blocks, one per 16-lane group
        const long long nb4 = (nblock4 + kFtB4Groups - 1) / kFtB4Groups;
        hipLaunchKernelGGL(ft_stage_bwd4, dim3((unsigned)nb4), dim3(kFtB4Sites), (size_t)kFtB4Lds, st, a);
    } else if (backward) hipLaunchKernelGGL(ft_stage_bwd<U>, dim3((unsigned)nb), dim3(256), 0, st, a);
    else hipLaunchKernelGGL(ft_stage_fwd<U>, dim3((unsigned)nb), dim3(256), 0, st, a);
    return hipGetLastError();
}

// The module's quantisation step and its backward (sr/model.py:74-76: weight = clamp(round_func(weight * 127), -127, 127), round_func a
// BPDA identity), for all tables of a stage in one launch: as torch operations they are six launches per table and direction.
struct FtQuantArgs {
    const float *w[kMaxFtModes];
    float *o[kMaxFtModes];
    long long n;
};
template <bool BACKWARD>
__global__ void __launch_bounds__(256) ft_quantize_kernel(FtQuantArgs a) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.n) return;
    const int m = (int)blockIdx.y;
    const float r = rintf(a.w[m][i] * 127.0f);          // torch.round: half to even
    if (BACKWARD) a.o[m][i] = a.o[m][i] * ((r >= -127.0f && r <= 127.0f) ? 1.0f : 0.0f) * 127.0f;      // the clamp passes gradient inside [-127, 127] inclusive
    else a.o[m][i] = fminf(fmaxf(r, -127.0f), 127.0f);
}

}  // namespace mulut

using namespace mulut;

static int ft_quantize(int device, const float *const *w, float *const *o, int M, long long n, bool backward, void *stream) {
    if (!w || !o || n <= 0) return MULUT_EINVAL;
    if (M < 1 || M > kMaxFtModes) return MULUT_EUNSUPPORTED;
    FtQuantArgs a;
    memset(&a, 0, sizeof(a));
    for (int m = 0; m < M; ++m) {
        if (!w[m] || !o[m]) return MULUT_EINVAL;
        a.w[m] = w[m];
        a.o[m] = o[m];
    }
    a.n = n;
    const long long nb = (n + 255) / 256;
    if (nb > 0x7fffffffLL) return MULUT_EINVAL;
    if (hipSetDevice(device) != hipSuccess) return MULUT_ENODEVICE;
    if (backward) hipLaunchKernelGGL(ft_quantize_kernel<true>, dim3((unsigned)nb, (unsigned)M), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(ft_quantize_kernel<false>, dim3((unsigned)nb, (unsigned)M), dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? MULUT_OK : MULUT_EHIP;
}

static int ft_fill(FtArgs &a, const float *const *weights, float *const *grad_wq, const char *modes, int is_last, int u,
                   const float *x, int B, int C, int H, int W) {
    if (!weights || !modes || !x || B <= 0 || C <= 0 || H <= 0 || W <= 0) return MULUT_EINVAL;
    const size_t M = strlen(modes);
    if (M < 1 || M > (size_t)kMaxFtModes || u < 1 || u > 4) return MULUT_EUNSUPPORTED;
    memset(&a, 0, sizeof(a));
    for (size_t m = 0; m < M; ++m) {
        int di[3], dj[3];
        if (!pattern_offsets(modes[m], di, dj)) return MULUT_EMODE;
        if (!weights[m] || (grad_wq && !grad_wq[m])) return MULUT_EINVAL;
        a.w[m] = weights[m];
        a.gw[m] = grad_wq ? grad_wq[m] : nullptr;
        for (int k = 0; k < 3; ++k) {
            a.di[m][k] = di[k];
            a.dj[m][k] = dj[k];
        }
    }
    a.x = x;
    a.B = B; a.C = C; a.H = H; a.W = W; a.u = u; a.M = (int)M; a.is_last = is_last ? 1 : 0;
    return MULUT_OK;
}

extern "C" {

int mulut_ft_quantize(int device, const float *const *weights, float *const *weights_q, int M, long long n, void *stream) {
    return ft_quantize(device, weights, weights_q, M, n, false, stream);
}

int mulut_ft_quantize_backward(int device, const float *const *weights, float *const *grad, int M, long long n, void *stream) {
    return ft_quantize(device, weights, grad, M, n, true, stream);
}

static int ft_forward(int device, const float *const *weights_q, const char *modes, int is_last, int u, const float *x,
                      int B, int C, int H, int W, float *out, uint16_t *inside, void *stream) {
    FtArgs a;
    int rc = ft_fill(a, weights_q, nullptr, modes, is_last, u, x, B, C, H, W);
    if (rc) return rc;
    if (!out) return MULUT_EINVAL;
    a.out = out;
    a.inside = inside;
    if (hipSetDevice(device) != hipSuccess) return MULUT_ENODEVICE;
    hipError_t e = u == 1 ? launch_ft<1>(a, false, (hipStream_t)stream) : u == 2 ? launch_ft<2>(a, false, (hipStream_t)stream)
                 : u == 3 ? launch_ft<3>(a, false, (hipStream_t)stream) : launch_ft<4>(a, false, (hipStream_t)stream);
    return e == hipSuccess ? MULUT_OK : MULUT_EHIP;
}

int mulut_ft_stage_forward(int device, const float *const *weights_q, const char *modes, int is_last, int u, const float *x,
                           int B, int C, int H, int W, float *out, void *stream) {
    return ft_forward(device, weights_q, modes, is_last, u, x, B, C, H, W, out, nullptr, stream);
}

int mulut_ft_stage_forward_mask(int device, const float *const *weights_q, const char *modes, int is_last, int u, const float *x,
                                int B, int C, int H, int W, float *out, unsigned short *inside, void *stream) {
    if (!inside) return MULUT_EINVAL;
    return ft_forward(device, weights_q, modes, is_last, u, x, B, C, H, W, out, inside, stream);
}

static int ft_backward(int device, const float *const *weights_q, const char *modes, int is_last, int u, const float *x,
                       const float *grad_out, const uint16_t *inside, int B, int C, int H, int W, float *const *grad_wq, float *grad_x,
                       void *stream) {
    FtArgs a;
    int rc = ft_fill(a, weights_q, grad_wq, modes, is_last, u, x, B, C, H, W);
    if (rc) return rc;
    if (!grad_out || !grad_wq || !grad_x) return MULUT_EINVAL;
    a.gout = grad_out;
    a.gx = grad_x;
    a.inside = const_cast<uint16_t *>(inside);
    if (hipSetDevice(device) != hipSuccess) return MULUT_ENODEVICE;
    hipError_t e = u == 1 ? launch_ft<1>(a, true, (hipStream_t)stream) : u == 2 ? launch_ft<2>(a, true, (hipStream_t)stream)
                 : u == 3 ? launch_ft<3>(a, true, (hipStream_t)stream) : launch_ft<4>(a, true, (hipStream_t)stream);
    return e == hipSuccess ? MULUT_OK : MULUT_EHIP;
}

int mulut_ft_stage_backward(int device, const float *const *weights_q, const char *modes, int is_last, int u, const float *x,
                            const float *grad_out, int B, int C, int H, int W, float *const *grad_wq, float *grad_x,
                            void *stream) {
    return ft_backward(device, weights_q, modes, is_last, u, x, grad_out, nullptr, B, C, H, W, grad_wq, grad_x, stream);
}

int mulut_ft_stage_backward_mask(int device, const float *const *weights_q, const char *modes, int is_last, int u, const float *x,
                                 const float *grad_out, const unsigned short *inside, int B, int C, int H, int W,
                                 float *const *grad_wq, float *grad_x, void *stream) {
    if (!inside) return MULUT_EINVAL;
    return ft_backward(device, weights_q, modes, is_last, u, x, grad_out, inside, B, C, H, W, grad_wq, grad_x, stream);
}

}  // extern "C"
