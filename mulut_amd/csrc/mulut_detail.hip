// mulut_detail.hip -- detailed tiles of the final stage (u == 4): anchor slabs in LDS instead of row gathers from L2
#include <hip/hip_runtime.h>

#include "mulut_dev.h"

namespace mulut {


// ------------------------------------------------------------------------------------------
// Detailed tiles of the final stage (u == 4): anchor slabs in LDS instead of row gathers from L2.
//
// On detailed content the full-table kernel is bound by its gathers: 60 rows of 16 bytes per sample, nearly every one
// of them a separate 128-byte line from L2 (17-145 cycles per gather instruction per CU).  The anchor of a sample (the
// first key) is the pixel itself in all 12 passes, so the passes of a sample touch only the slab pair of its anchor MSB
// (mulut_core.h "slab pairs": 157,216 bytes, LDS-sized).  The samples (pixel, channel) of the tiles the statistic marked
// detailed are therefore grouped by anchor MSB, on the device and without host synchronisation:
//   detail_bucket_kernel<false>  counts the samples of every anchor MSB (LDS histogram per tile, 16 atomics per tile)
//   detail_plan_kernel           turns the 16 counts into list starts and work items of <= 4096 samples of one anchor
//   detail_bucket_kernel<true>   writes the sample ids (tile << 12 | c << 10 | ty << 6 | tx) into the 16 lists
//   stage_slab_kernel            one persistent workgroup per CU walks the items: per mode it copies the item's slab pair
//                                into LDS (a straight 157 KB copy, L2-resident) and runs the mode's four passes of its
//                                4 samples per thread -- rotation pairs in packed 16-bit halves as in the tube kernel,
//                                rows by ds_read_b128, accumulated from the raw bytes (three operations per dword) -- keeping
//                                the accumulators in registers across the modes; the finished 4x4 block of a sample is
//                                one 16-byte store at blocks[id]
//   detail_retile_kernel         writes the blocks of the detailed tiles to the output image in its layout
// A sample's 5x5 window is read straight from the stage input (L2-resident), 8 unaligned bytes per row from column
// x - 2; pixels in the first 2 / last 6 columns of the image would need edge replication inside those 8 bytes and go to
// the pixel fix-up list (stage_up_fix_kernel) instead.
// ------------------------------------------------------------------------------------------
constexpr int kSlabNT = 1024, kSlabS = 4, kSlabItem = kSlabNT * kSlabS;
constexpr int kSlabLdsBytes = ((kSlabPairBytes + 1023) / 1024) * 1024;      // whole 1-KiB LDS-DMA pieces: 157,696

__device__ __forceinline__ uint4 lds_u128(uint32_t addr) {
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = *(const __attribute__((address_space(3))) u32x4 *)(uintptr_t)addr;
    return make_uint4(v.x, v.y, v.z, v.w);
}

// One workgroup: the per-tile histograms of the detailed tiles (thist[tile][16], written by tile_stat_kernel) become
// absolute positions in the id lists (exclusive scan over tiles per anchor MSB, in place), the 16 totals become list
// starts and work items, and the detailed tiles are listed (dlist) -- no atomics, so the lists are deterministic.
__global__ void __launch_bounds__(1024) detail_plan_kernel(DetailArgs d, const uint32_t *verdict, uint32_t ntiles, uint32_t want_items) {
    if (!d.dirty_list && d.ctl[kDetAny] == 0u) return;      // no tile was marked detailed (workgroup-uniform): nothing to plan, fill, compute or retile
    __shared__ uint32_t s_wave[16][17];
    __shared__ uint32_t s_start[17], s_item0[17], s_isz;
    const int lane = (int)(threadIdx.x & 63), wave = (int)(threadIdx.x >> 6);
    // thread t takes tiles t, t + 1024, ...: the histogram halves (detail_hist_index) are then read as whole coalesced KiB per
    // wave; the lists follow this order
    const uint4 *half[2] = {(const uint4 *)d.thist, (const uint4 *)d.thist + ntiles};
    uint32_t excl[17];       // this thread's exclusive prefix per anchor MSB (16: detailed tiles)
    {
        uint32_t local[17];
#pragma unroll
        for (int b = 0; b < 17; ++b) local[b] = 0;
        // chunks of 8 tiles per thread, fully unrolled: all 24 loads of a chunk are in flight together (a rolled loop makes a
        // round trip to L2 per iteration, and this is one workgroup)
        for (uint32_t base = 0; base < ntiles; base += 8 * 1024) {
            uint32_t det[8], any = 0;
            uint4 r[8][2];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t t = base + (uint32_t)k * 1024u + threadIdx.x, tc = t < ntiles ? t : ntiles - 1u;
                det[k] = (t < ntiles && verdict[tc] == 1u) ? 1u : 0u;
                any |= det[k];
            }
            if (!__any((int)any)) continue;           // smooth content: no histogram is read at all
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t t = base + (uint32_t)k * 1024u + threadIdx.x, tc = t < ntiles ? t : ntiles - 1u;
                r[k][0] = half[0][tc];
                r[k][1] = half[1][tc];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t w[8] = {r[k][0].x, r[k][0].y, r[k][0].z, r[k][0].w, r[k][1].x, r[k][1].y, r[k][1].z, r[k][1].w};
#pragma unroll
                for (int b = 0; b < 16; ++b) local[b] += det[k] ? ((w[b >> 1] >> (16 * (b & 1))) & 0xFFFFu) : 0u;
                local[16] += det[k];
            }
        }
#pragma unroll
        for (int b = 0; b < 17; ++b) {
            uint32_t inc = local[b];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t up = (uint32_t)__shfl_up((int)inc, o);
                if (lane >= o) inc += up;
            }
            excl[b] = inc - local[b];
            if (lane == 63) s_wave[wave][b] = inc;
        }
    }
    __syncthreads();
    if (threadIdx.x < 17) {          // wave totals -> exclusive wave bases, column totals into s_start
        uint32_t run = 0;
        for (int w = 0; w < 16; ++w) {
            const uint32_t v = s_wave[w][threadIdx.x];
            s_wave[w][threadIdx.x] = run;
            run += v;
        }
        s_start[threadIdx.x] = run;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        // item size: whole items (4 samples per thread) when there is work for every workgroup, else fewer samples per thread
        // (a multiple of the workgroup size) so that the few samples still spread over the workgroups
        uint32_t total = 0;
        for (int b = 0; b < 16; ++b) total += s_start[b] + d.ctl[kDetDirty + b];
        uint32_t isz = ((total / (want_items ? want_items : 1u) + kSlabNT - 1) / kSlabNT) * kSlabNT;
        isz = isz < (uint32_t)kSlabNT ? (uint32_t)kSlabNT : isz > (uint32_t)kSlabItem ? (uint32_t)kSlabItem : isz;
        s_isz = isz;
        uint32_t start = 0, item0 = 0;
        for (int b = 0; b < 16; ++b) {
            // the list of anchor MSB b: the samples of the detailed tiles, then the tube kernel's dirty samples
            const uint32_t cnt = s_start[b] + d.ctl[kDetDirty + b];
            d.ctl[kDetCount + b] = cnt;
            d.ctl[kDetStart + b] = start;
            d.ctl[kDetDirtyBase + b] = start + s_start[b];
            s_start[b] = start;
            s_item0[b] = item0;
            start += cnt;
            item0 += (cnt + isz - 1) / isz;
        }
        d.ctl[kDetTiles] = s_start[16];
        d.ctl[kDetItems] = item0;
    }
    __syncthreads();
    for (int b = 0; b < 16; ++b) {
        const uint32_t isz = s_isz, cnt = d.ctl[kDetCount + b], ni = (cnt + isz - 1) / isz;
        for (uint32_t i = threadIdx.x; i < ni; i += 1024) {
            const uint32_t left = cnt - i * isz;
            d.items[2 * (s_item0[b] + i)] = ((uint32_t)b << 28) | (left < isz ? left : isz);
            d.items[2 * (s_item0[b] + i) + 1] = s_start[b] + i * isz;
        }
    }
    {
        uint32_t run[16], slot = s_wave[wave][16] + excl[16];
#pragma unroll
        for (int b = 0; b < 16; ++b) run[b] = s_start[b] + s_wave[wave][b] + excl[b];
        uint4 *quarter[4] = {(uint4 *)d.tpos, (uint4 *)d.tpos + ntiles, (uint4 *)d.tpos + 2 * (size_t)ntiles, (uint4 *)d.tpos + 3 * (size_t)ntiles};
        for (uint32_t base = 0; base < ntiles; base += 8 * 1024) {
            uint32_t det[8], any = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t t = base + (uint32_t)k * 1024u + threadIdx.x, tc = t < ntiles ? t : ntiles - 1u;
                det[k] = (t < ntiles && verdict[tc] == 1u) ? 1u : 0u;
                any |= det[k];
            }
            if (!__any((int)any)) continue;           // no detailed tile in this chunk of the wave: nothing to read or write
            uint4 r[8][2];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const uint32_t t = base + (uint32_t)k * 1024u + threadIdx.x, tc = t < ntiles ? t : ntiles - 1u;
                r[k][0] = half[0][tc];
                r[k][1] = half[1][tc];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (det[k]) {
                    const uint32_t t = base + (uint32_t)k * 1024u + threadIdx.x;
                    const uint32_t w[8] = {r[k][0].x, r[k][0].y, r[k][0].z, r[k][0].w, r[k][1].x, r[k][1].y, r[k][1].z, r[k][1].w};
                    uint32_t o[16];
#pragma unroll
                    for (int b = 0; b < 16; ++b) { o[b] = run[b]; run[b] += (w[b >> 1] >> (16 * (b & 1))) & 0xFFFFu; }
                    quarter[0][t] = make_uint4(o[0], o[1], o[2], o[3]);
                    quarter[1][t] = make_uint4(o[4], o[5], o[6], o[7]);
                    quarter[2][t] = make_uint4(o[8], o[9], o[10], o[11]);
                    quarter[3][t] = make_uint4(o[12], o[13], o[14], o[15]);
                    d.dlist[slot++] = t;
                }
        }
    }
}

// ids and descriptors of the samples of the listed tiles, written at the positions detail_plan_kernel assigned (the rank of
// a sample inside its tile's share of a list comes from an LDS counter); pixels in the image's border columns go to the
// pixel fix-up list instead
__global__ void __launch_bounds__(256) detail_fill_kernel(StageArgs a, DetailArgs d) {
    if (!d.dirty_list && d.ctl[kDetAny] == 0u) return;      // no tile was marked detailed (workgroup-uniform): nothing to plan, fill, compute or retile
    constexpr int TW = KB_TW, TH = KB_TH, PER = 3 * TW * TH / 256;
    static_assert(TW == 64 && TH == 16, "sample ids assume the 64x16 verdict tile");
    __shared__ uint32_t s_rank[16], s_base[16], s_fix[2];
    const uint32_t ndet = d.ctl[kDetTiles];
    const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
    for (uint32_t li = blockIdx.x; li < ndet; li += gridDim.x) {
        const int tile = (int)d.dlist[li];
        int n, y0, x0;
        decode_tile(a, tile, n, y0, x0, TW, TH);
        __syncthreads();              // the previous tile's counters are no longer read
        if (threadIdx.x < 16) {
            s_rank[threadIdx.x] = 0;
            s_base[threadIdx.x] = d.tpos[detail_pos_index((uint32_t)tile, (uint32_t)(a.N * a.tiles_x * a.tiles_y), (int)threadIdx.x)];
        }
        if (threadIdx.x == 16) s_fix[0] = 0;
        __syncthreads();
        uint32_t pos[PER], fixr[4];
        uint8_t val[PER];
        // every byte of the thread in flight before the first LDS atomic (behind the per-sample condition each load would be
        // a round trip of its own: twelve in a row)
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int s = (int)threadIdx.x + k * 256;
            const int c = imin(s >> 10, a.C - 1), y = imin(y0 + ((s >> 6) & 15), a.oy1 - 1), x = imin(x0 + (s & 63), a.W - 1);
            val[k] = *view_addr(a.in, n, c, y, x);
        }
#pragma unroll
        for (int k = 0; k < PER; ++k) {
            const int s = (int)threadIdx.x + k * 256;
            const int c = s >> 10, ty = (s >> 6) & 15, tx = s & 63;
            const int y = y0 + ty, x = x0 + tx;
            const bool inside = c < a.C && y < a.oy1 && x < a.W;
            const bool slab = inside && x >= kSlabXLo && x < a.W - slab_x_hi(a);
            pos[k] = 0xFFFFFFFFu;
            if (slab) {
                const uint32_t h = (uint32_t)(val[k] >> 4);
                pos[k] = s_base[h] + atomicAdd(&s_rank[h], 1u);
            }
            if (k < 4) fixr[k] = (inside && !slab) ? atomicAdd(&s_fix[0], 1u) : 0xFFFFFFFFu;      // k < 4 <=> channel 0: each pixel once
        }
        __syncthreads();
        if (threadIdx.x == 0 && s_fix[0]) s_fix[1] = atomicAdd(a.fix_count, s_fix[0]);      // one atomic per tile that has border pixels
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (fixr[k] != 0xFFFFFFFFu) {
                const int s = (int)threadIdx.x + k * 256;
                a.fix_list[s_fix[1] + fixr[k]] = (uint32_t)((n * a.H + y0 + ((s >> 6) & 15)) * a.W + x0 + (s & 63)) | (3u << 30);
            }
#pragma unroll
        for (int k = 0; k < PER; ++k)
            if (pos[k] != 0xFFFFFFFFu) {
                const int s = (int)threadIdx.x + k * 256;
                const int c = s >> 10, y = y0 + ((s >> 6) & 15), x = x0 + (s & 63);
                d.desc[pos[k]] = (uint32_t)(view_addr(a.in, n, c, y, x - 2) - a.in.p) | ((uint32_t)imin(y - ylo, 2) << 28) | ((uint32_t)imin(yhi - y, 2) << 30);
            }
    }
}


// accumulators of one sample: sums of the raw dwords (F) and of the dwords shifted right by one byte (G) of the rotation pairs (0, 2)
// and (1, 3) -- mulut_core.h slab_split_sums
struct SlabAcc {
    uint32_t F02[4], H02[4], F13[4], H13[4];
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int k = 0; k < 4; ++k) F02[k] = H02[k] = F13[k] = H13[k] = 0;
    }
    template <int R, int HALF>
    __device__ __forceinline__ void mac_row(const uint4 &row, uint32_t wpk) {
        const uint32_t rd[4] = {row.x, row.y, row.z, row.w};
        static_for<0, 4>([&](auto K) {
            constexpr int k = K;
            if constexpr (R == 0) { pk_mac<HALF, false>(F02[k], rd[k], wpk); pk_mac<HALF, false>(H02[k], rd[k] >> 8, wpk); }
            if constexpr (R == 1) { pk_mac<HALF, false>(F13[k], rd[k], wpk); pk_mac<HALF, false>(H13[k], rd[k] >> 8, wpk); }
            if constexpr (R == 2) { const uint32_t rv = slab_rev_bytes(rd[k]); pk_mac<HALF, false>(F02[3 - k], rv, wpk); pk_mac<HALF, false>(H02[3 - k], rv >> 8, wpk); }
            if constexpr (R == 3) { const uint32_t rv = slab_rev_bytes(rd[k]); pk_mac<HALF, false>(F13[3 - k], rv, wpk); pk_mac<HALF, false>(H13[3 - k], rv >> 8, wpk); }
        });
    }
    __device__ __forceinline__ void to_fields(RotAcc<4> &r) const {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            slab_split_sums(F02[k], H02[k], r.lo02[k], r.hi02[k]);
            slab_split_sums(F13[k], H13[k], r.lo13[k], r.hi13[k]);
        }
    }
};

// dst = (16-bit half HALF of x) * 16 + acc: one v_mad_u32_u16, the half picked by op_sel
template <int HALF>
__device__ __forceinline__ uint32_t mad16_half(uint32_t x, uint32_t acc) {
    uint32_t r;
    if constexpr (HALF == 0) asm("v_mad_u32_u16 %0, %1, 16, %2 op_sel:[0,0,0,0]" : "=v"(r) : "v"(x), "v"(acc));
    else asm("v_mad_u32_u16 %0, %1, 16, %2 op_sel:[1,0,0,0]" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

// LDS byte addresses of rows 0..3 of the pass in half HALF: 16 * (running sum of the path's unit steps); row 4 = row 0 + kSlabAll * 16
template <int HALF>
__device__ __forceinline__ void slab_row_addrs(const SlabPair &sp, uint32_t (&ad)[4]) {
    // one v_mad_u32_u16 per row: (the pass's half of the packed unit step) * 16 + the previous row's address -- extract, scale and add at once
    ad[0] = mad16_half<HALF>(sp.base, 0u);
#pragma unroll
    for (int j = 0; j < 3; ++j) ad[j + 1] = mad16_half<HALF>(sp.step[j], ad[j]);
}
template <int J>
__device__ __forceinline__ uint4 slab_row(const uint32_t (&ad)[4]) {
    return lds_u128(ad[J < 4 ? J : 0] + (uint32_t)(J < 4 ? 0 : kSlabAll * 16));
}

// Both passes of a rotation pair (R in the low halves of sp, R + 2 in the high halves) from the slab pair at LDS address 0.
// The second pass's rows are requested one by one as the first pass's rows are consumed -- into the registers those free --
// so the LDS latency of every second pass is covered by accumulation instead of being waited for.
template <int R>
__device__ __forceinline__ void slab_pair_rows(const SlabPair &sp, SlabAcc &acc) {
    uint32_t a0[4], a1[4];
    slab_row_addrs<0>(sp, a0);
    slab_row_addrs<1>(sp, a1);
    uint4 r0[5], r1[5];
    static_for<0, 5>([&](auto J) { r0[J] = slab_row<J>(a0); });
    static_for<0, 5>([&](auto J) {
        acc.template mac_row<R, 0>(r0[J], sp.w[J]);
        r1[J] = slab_row<J>(a1);
        __builtin_amdgcn_sched_barrier(0);      // keep the order: row J of the first pass consumed, row J of the second requested
    });
    static_for<0, 5>([&](auto J) { acc.template mac_row<R + 2, 1>(r1[J], sp.w[J]); });
}

// The part of a sample's 5x5 window a pattern touches.  Per window row: nothing, one dword from column x - 1
// (columns -1..2), or 8 bytes from column x - 2 (columns -2..5).  Row 0 is always 8 bytes (the anchor is byte 2 of it).
//   s: rows -1..1, columns -1..1;  d: rows -2/0/2, columns -2/0/2;  y: rows +-2 columns +-1, rows +-1 columns +-1, +-2
template <int PAT>
__host__ __device__ constexpr int slab_row_kind(int dy) {
    return dy == 0 ? 2 : PAT == 0 ? ((dy == 1 || dy == -1) ? 1 : 0) : PAT == 1 ? ((dy & 1) == 0 ? 2 : 0) : ((dy == 1 || dy == -1) ? 2 : 1);
}
struct SlabWin {
    uint32_t lo[5], hi[5];
    // the byte at window offset (dy, dx) of pattern PAT's loads: register and byte index
    template <int PAT, int DY, int DX>
    __device__ __forceinline__ uint32_t reg() const {
        constexpr int kind = slab_row_kind<PAT>(DY);
        static_assert(kind != 0 && (kind == 2 || (DX >= -1 && DX <= 2)), "window byte not loaded for this pattern");
        constexpr int idx = kind == 1 ? DX + 1 : DX + 2;
        return idx < 4 ? lo[DY + 2] : hi[DY + 2];
    }
    template <int PAT, int DY, int DX>
    static constexpr int byte_idx() { return (slab_row_kind<PAT>(DY) == 1 ? DX + 1 : DX + 2) & 3; }
};

// sample descriptor: byte offset of (n, c, y, x - 2) in the stage input (< 2^28) | min(y - ylo, 2) << 28 | min(yhi - y, 2) << 30
template <int PAT>
__device__ __forceinline__ void slab_load_window_t(const StageArgs &a, uint32_t desc, SlabWin &w) {
    const int top = (int)((desc >> 28) & 3u), bot = (int)(desc >> 30);
    const uint8_t *p0 = a.in.p + (desc & 0x0FFFFFFFu);
    static_for<0, 5>([&](auto RW) {
        constexpr int r = RW, dy = r - 2, kind = slab_row_kind<PAT>(dy);
        // every register of the window is assigned on every pattern's path (rows the pattern does not touch: zero): stores
        // to different elements in the three branches would be merged into one indexed store, i.e. the window put in scratch
        w.lo[r] = 0;
        w.hi[r] = 0;
        if constexpr (kind != 0) {
            const int dyc = dy < 0 ? -imin(-dy, top) : imin(dy, bot);       // edge replication at the true image borders
            const uint8_t *p = p0 + dyc * a.in.sY;
            if constexpr (kind == 1) {
                uint32_t v;
                __builtin_memcpy(&v, p + 1, 4);
                w.lo[r] = v;
            } else {
                uint2 v;
                __builtin_memcpy(&v, p, 8);
                w.lo[r] = v.x;
                w.hi[r] = v.y;
            }
        }
    });
}
__device__ __forceinline__ void slab_load_window(const StageArgs &a, int pat, uint32_t desc, SlabWin &w) {
    asm volatile("" : "+v"(desc));      // opaque: the row addresses are rebuilt here every time (hoisted out of the mode loop they would be parked in scratch)
    if (pat == 0) slab_load_window_t<0>(a, desc, w);          // scalar branches
    else if (pat == 1) slab_load_window_t<1>(a, desc, w);
    else slab_load_window_t<2>(a, desc, w);
}

// neighbour K of pattern PAT: rotation R's byte in the low half, rotation R + 2's (the opposite offset) in the high half
template <int PAT, int R, int K>
__device__ __forceinline__ uint32_t slab_nb(const SlabWin &w) {
    constexpr int dy = rot_dy(R, kPatDi[PAT][K], kPatDj[PAT][K]), dx = rot_dx(R, kPatDi[PAT][K], kPatDj[PAT][K]);
    constexpr uint32_t sel = 0x0C000C00u | ((uint32_t)(4 + SlabWin::byte_idx<PAT, -dy, -dx>()) << 16) | (uint32_t)SlabWin::byte_idx<PAT, dy, dx>();
    return __builtin_amdgcn_perm(w.template reg<PAT, -dy, -dx>(), w.template reg<PAT, dy, dx>(), sel);
}
template <int PAT, int R>
__device__ __forceinline__ void slab_pair_index(const SlabWin &w, uint32_t k0, SlabPair &sp) {
    simplex4_slab_pair(k0, slab_nb<PAT, R, 0>(w), slab_nb<PAT, R, 1>(w), slab_nb<PAT, R, 2>(w), sp);
}

// All four passes of one sample and mode.  Only the index math is specific to the pattern (a scalar switch per rotation
// pair, ~45 instructions each); the row reads and the accumulation -- most of the code -- are shared by the patterns, which
// keeps the loop over an item (4 samples x 3 modes) inside the instruction cache: with the pattern as a template
// parameter of the whole body the item loop was 77 KB of code and ran 1.5x slower.
__device__ __forceinline__ void slab_sample(int pat, const SlabWin &w, SlabAcc &acc) {
    uint32_t k0 = slab_anchor_key((w.lo[2] >> 16) & 0xFFu);
    static_for<0, 2>([&](auto RR) {
        constexpr int R = RR;
        SlabPair sp;
        if (pat == 0) slab_pair_index<0, R>(w, k0, sp);
        else if (pat == 1) slab_pair_index<1, R>(w, k0, sp);
        else slab_pair_index<2, R>(w, k0, sp);
        slab_pair_rows<R>(sp, acc);
        // one pair at a time: the next pair's index math must not be scheduled into this one (VGPR budget)
        asm volatile("" : "+v"(acc.F02[0]), "+v"(acc.F13[0]), "+v"(k0));
    });
}

#define SLAB_CLK(i) do { } while (0)
#define SLAB_CLK_ARGS
#define SLAB_CLK_PASS
// one mode of one item: the slab pair into LDS (LDS-DMA: 154 pieces of 1 KiB, wave w takes pieces w, w + 16, ...), then the
// mode's four passes of the thread's samples; the window of the next sample is in flight while the current one is computed,
// the first one while the slab pair is copied
__device__ __forceinline__ void slab_mode(const StageArgs &a, int pat, uint32_t cnt, const uint8_t *pair, bool in_lds, uint8_t *smem, const uint32_t (&desc)[kSlabS], SlabAcc (&acc)[kSlabS] SLAB_CLK_ARGS) {
    SlabWin wa, wb;
    slab_load_window(a, pat, desc[0], wa);
    if (!in_lds) {                     // workgroup-uniform
    __syncthreads();                   // everyone is done with the previous slab pair
    SLAB_CLK(0);                       // waiting for the slowest wave of the previous mode
    {
        const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
        constexpr int kPieces = kSlabLdsBytes / 1024;
#pragma unroll
        for (int k = 0; k < (kPieces + 15) / 16; ++k) {
            const int piece = wave + 16 * k;
            if (piece < kPieces)       // wave-uniform
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(pair + piece * 1024 + lane * 16),
                                                 (__attribute__((address_space(3))) void *)(smem + piece * 1024), 16, 0, 0);
        }
    }
    __syncthreads();
    }
    SLAB_CLK(1);                       // slab pair copy
    static_for<0, kSlabS>([&](auto S) {
        constexpr int s = S;
        if ((uint32_t)s * kSlabNT < cnt) {          // workgroup-uniform: a short item leaves sample slots empty
            if constexpr (s + 1 < kSlabS) slab_load_window(a, pat, desc[s + 1], (s & 1) ? wa : wb);
            slab_sample(pat, (s & 1) ? wb : wa, acc[s]);
        }
    });
    SLAB_CLK(2);                       // the mode's passes
}

__global__ void __launch_bounds__(kSlabNT) stage_slab_kernel(StageArgs a, DetailArgs d) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    if (!d.dirty_list && d.ctl[kDetAny] == 0u) return;      // no tile was marked detailed (workgroup-uniform)
    if (lds_addr_of(smem) != 0u) __builtin_trap();      // the row reads assume the slab pair starts at LDS address 0 (no static LDS here): fail loudly, never leave blocks unwritten
    const uint32_t nitems = d.ctl[kDetItems];
    const uint8_t *resident = nullptr;      // the slab pair in LDS
    bool snake = false;
    for (uint32_t it = blockIdx.x; it < nitems; it += gridDim.x) {
        const uint32_t hdr = d.items[2 * it], first = d.items[2 * it + 1];
        const uint32_t h = hdr >> 28, cnt = hdr & 0x0FFFFFFFu;
        uint32_t desc[kSlabS];
#pragma unroll
        for (int s = 0; s < kSlabS; ++s) {
            const uint32_t i = (uint32_t)s * kSlabNT + threadIdx.x;
            desc[s] = d.desc[first + (i < cnt ? i : cnt - 1u)];          // surplus lanes recompute the last sample (never stored)
        }
        SlabAcc acc[kSlabS];
#pragma unroll
        for (int s = 0; s < kSlabS; ++s) acc[s].clear();
        // the modes in alternating order from item to item: consecutive items of a workgroup mostly share their anchor, and
        // the pair the last mode left in LDS then serves the next item's first mode (the sums do not depend on the order)
        for (int mv = 0; mv < a.M; ++mv) {
            const int m = __builtin_amdgcn_readfirstlane(snake ? a.M - 1 - mv : mv);
            const int pat = a.dj[m][0] == 2 ? 1 : a.di[m][0] == 1 ? 2 : 0;     // scalar
            const uint8_t *pair = d.slab[m] + (size_t)h * kSlabPairBytes;
            SLAB_CLK(3);                   // item set-up / epilogue / stores
            slab_mode(a, pat, cnt, pair, pair == resident, smem, desc, acc SLAB_CLK_PASS);
            resident = pair;
        }
        snake = !snake;
#pragma unroll
        for (int s = 0; s < kSlabS; ++s) {
            RotAcc<4> r;
            acc[s].to_fields(r);
            uint32_t o[4];
            tube_finish_rows(a, r, o);
            const uint32_t i = (uint32_t)s * kSlabNT + threadIdx.x;
            if (i < cnt) d.blocks[(desc[s] & 0x0FFFFFFFu) + 2u] = make_uint4(o[0], o[1], o[2], o[3]);      // indexed by the sample's byte offset in the stage input
        }
    }
}

template <int OUT>
__global__ void __launch_bounds__(KB_TW *KB_TH) detail_retile_kernel(StageArgs a, DetailArgs d) {
    if (!d.dirty_list && d.ctl[kDetAny] == 0u) return;      // no tile was marked detailed (workgroup-uniform)
    const uint32_t ndet = d.ctl[kDetTiles];
    const int tx = threadIdx.x % KB_TW, ty = threadIdx.x / KB_TW;
    for (uint32_t li = blockIdx.x; li < ndet; li += gridDim.x) {
    const int tile = (int)d.dlist[li];
    int n, y0, x0;
    decode_tile(a, tile, n, y0, x0, KB_TW, KB_TH);
    const int y = y0 + ty, x = x0 + tx;
    if (y >= a.oy1 || x >= a.W || x < kSlabXLo || x >= a.W - slab_x_hi(a)) continue;      // border columns: the fix-up kernel's
    const size_t id = (size_t)(view_addr(a.in, n, 0, y, x) - a.in.p), cs = (size_t)a.in.sC;      // block index = byte offset in the stage input
    if constexpr (OUT == kOutPackedRGBU4) {
        const uint4 r = d.blocks[id], g = d.blocks[id + cs], b = d.blocks[id + 2 * cs];
        const uint32_t oR[4] = {r.x, r.y, r.z, r.w}, oG[4] = {g.x, g.y, g.z, g.w}, oB[4] = {b.x, b.y, b.z, b.w};
        store_rgb<4>(a, n, y, x, oR, oG, oB);
    } else {
        for (int c = 0; c < a.C; ++c) {
            const uint4 v = d.blocks[id + (size_t)c * cs];
            const uint32_t o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int sy = 0; sy < 4; ++sy) {
                if constexpr (OUT == kOutPlanarU4) {
                    *(uint32_t *)const_cast<uint8_t *>(view_addr(a.out, n, c, y * 4 + sy, x * 4)) = o[sy];
                } else {
#pragma unroll
                    for (int sx = 0; sx < 4; ++sx)
                        *const_cast<uint8_t *>(view_addr(a.out, n, c, y * 4 + sy, x * 4 + sx)) = (uint8_t)(o[sy] >> (8 * sx));
                }
            }
        }
    }
    }
}

bool detail_slab_supported(const StageArgs &a) {
    const long long tiles = (long long)a.N * a.tiles_x * a.tiles_y;      // 64x16 verdict tiling
    const unsigned long long bytes = (unsigned long long)a.N * (unsigned long long)(a.in.sN < 0 ? -a.in.sN : a.in.sN);
    return a.C <= 3 && a.M <= 3 && a.in.sX == 1 && tiles > 0 && tiles < (1ll << 20) &&
           bytes < (1ull << 28) && (unsigned long long)a.N * a.H * a.W < (1ull << 32);
}
size_t detail_ids_count(const StageArgs &a) { return (size_t)a.N * a.tiles_x * a.tiles_y * 3 * KB_TW * KB_TH; }
size_t detail_items_max(const StageArgs &a) { return detail_ids_count(a) / kSlabItem + 16 + 8192; }      // + the small-item case of detail_plan_kernel (<= 2 x CUs x 4 items)
size_t detail_blocks_count(const StageArgs &a) { return (size_t)a.N * (size_t)(a.in.sN < 0 ? -a.in.sN : a.in.sN); }      // one per byte of the stage input

// the detailed tiles (a.verdict[tile] == 1, histograms in d.thist from launch_tile_stat) of a u == 4 final stage
hipError_t launch_detail_slab(const StageArgs &a, const DetailArgs &d, int out_mode, int num_cus, hipStream_t st) {
    if (!detail_slab_supported(a) || !a.verdict || !a.fix_list || !a.fix_count) return hipErrorInvalidValue;
    {
        const hipError_t e = raise_lds_limit((const void *)stage_slab_kernel, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    const unsigned tiles = (unsigned)((long long)a.N * a.tiles_x * a.tiles_y);
    const unsigned walk = tiles < (unsigned)(8 * num_cus) ? tiles : (unsigned)(8 * num_cus);      // workgroups walking the list of detailed tiles
    hipLaunchKernelGGL(detail_plan_kernel, dim3(1), dim3(1024), 0, st, d, (const uint32_t *)a.verdict, tiles, (uint32_t)num_cus);      // few samples: about one item per workgroup (an item's three slab copies make a second round dearer than longer items)
    hipLaunchKernelGGL(detail_fill_kernel, dim3(tiles < 2 * walk ? tiles : 2 * walk), dim3(256), 0, st, a, d);
    hipLaunchKernelGGL(stage_slab_kernel, dim3((unsigned)num_cus), dim3(kSlabNT), (size_t)kSlabLdsBytes, st, a, d);
    if (out_mode == kOutPlanarU4) hipLaunchKernelGGL(detail_retile_kernel<kOutPlanarU4>, dim3(walk), dim3(KB_TW * KB_TH), 0, st, a, d);
    else if (out_mode == kOutPackedRGBU4 && a.C == 3) hipLaunchKernelGGL(detail_retile_kernel<kOutPackedRGBU4>, dim3(walk), dim3(KB_TW * KB_TH), 0, st, a, d);
    else hipLaunchKernelGGL(detail_retile_kernel<kOutGeneric>, dim3(walk), dim3(KB_TW * KB_TH), 0, st, a, d);
    return hipGetLastError();
}

}  // namespace mulut
