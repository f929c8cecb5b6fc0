// ARCHIVED, NOT BUILT (round 4).  stage_u1s_kernel: the 1-byte-row tube stage as wave-owned 64 x 4 strips with no workgroup barrier in
// the steady state, the routing statistic as a kernel of its own (u1_route_kernel), per-wave LDS fix-up buffers.  Bit-exact (the GPU
// suite passed with it as the default).  It was written because the phase stamps of stage_u1t_kernel (profiles/r04k_k1_phases.txt) put
// only 23 % of a wave's life in the pixel bodies and 27 % + 14 % at the tile's barriers and behind its stores.  Measured (MI355X, us per
// frame of LR 1080x1920x3, first stage, D-natural; profiles/r04l_ab_k1_strips_*.jsonl, r04m_ab_k1_strips_ablation.jsonl):
//     stage_u1t_kernel, one 1024-thread workgroup per 64 x 64 tile (shipped)          69.1
//     the same kernel, persistent workgroups + next tile's pixels prefetched          71.9   (sites 93 % of a wave's life -- and no faster)
//     this kernel                                                                     72.1   (8 x 270x480 frames: 26.0 vs 19.0)
//     this kernel with every pixel body executed twice (timing-only)                 122.1   => the bodies cost 48.6 of the 73.5
//     this kernel without the neighbourhood test (timing-only)                        66.8
// i.e. the stage is bound by the issue of its pixel bodies (26.7 VALU + 4.3 LDS instructions per pass: 96 SIMD cycles per pass in situ,
// 64-69 as a microbenchmark of the same stream with conflict-free LDS addresses, profiles/r04_ubench_stream_k1.txt) whatever the waves do
// between them: barrier waits are covered by the other workgroup of the CU, and removing them buys nothing.
// Goes into mulut_k1.hip before stage_u1t_tile(); launcher at the end.

// ------------------------------------------------------------------------------------------
// K1-strips (round 4): the tube stage with NO workgroup barrier in its steady state.
// Phase stamps of stage_u1t_kernel (profiles/r04k_k1_phases.txt) put 23 % of a wave's life in the pixel bodies and 27 % + 14 % at
// the tile's barriers and behind its stores: a SIMD arbitrates oldest-first, the waves of a workgroup finish a tile far apart, and
// everyone waits for the slowest (what round 3 found in the final-stage kernel).  Here a WAVE owns its work: a strip of 64 x 4 pixels
// (one row of 16 four-pixel groups per 16 lanes -- the thread layout of the tile kernel) with a private 68 x 8 x C image of pixel
// codes in LDS.  A wave draws strips from its workgroup's counter (a workgroup owns an XCD-contiguous run of 64 x 64 tiles = 16
// strips each), requests the next strip's pixels (aligned dwords into registers) before it computes the current one, and collects the
// sites it flags in a private LDS buffer that goes to the fix-up list with one memory-side atomic when it is full.  The routing
// statistic (which 64 x 64 tiles are left to the full-table kernel) is taken by u1_route_kernel before this kernel starts.
// The price is halo work: 8 image rows are turned into codes per 4 computed (1.13 x in the tile kernel) -- about one instruction per
// pass more.
// LDS: [ band s | band d | band y ][ NW images: C x 8 x 68 pixel codes ][ NW fix-up buffers ][ work counter ]
// ------------------------------------------------------------------------------------------
constexpr int K1S_TH = 4, K1S_PH = K1S_TH + 2 * kHalo, K1S_PW = K1T_PW, K1S_GR = (K1T_TW + 8) / 4;      // 8 image rows of 68 codes; 18 aligned groups cover them
constexpr int kU1sImgBytes = 3 * K1S_PH * K1S_PW * 2;          // 3264
constexpr int kU1sFixCap = 128;                                // entries of a wave's fix-up buffer (>= 64: one pixel slot of a strip)
constexpr int kU1sStrips = K1T_TH / K1S_TH;                    // 16 strips per 64 x 64 tile
__host__ __device__ constexpr int u1s_threads(int U, int pats) { return U == 1 ? (pats != 0 ? 1024 : 512) : 768; }
__host__ __device__ constexpr int u1s_waves(int U, int pats) { return U == 1 && pats != 0 ? 8 : 6; }
template <int U> __host__ __device__ constexpr int u1s_lds_bytes(int threads) {
    return 3 * u1t_band_bytes<U>() + (threads / 64) * (kU1sImgBytes + kU1sFixCap * 4) + 16;
}

// one aligned group of four pixels of image row gy starting at column gx (clamped into the image by the caller), as packed pairs of
// pixel codes per channel; columns beyond the image replicate its edge column
__device__ __forceinline__ void u1_codes_hwc(int gx, int W, uint32_t d0, uint32_t d1, uint32_t d2, uint32_t (&bp)[6]) {
    bp[0] = __builtin_amdgcn_perm(0u, d0, 0x0C030C00u); bp[1] = __builtin_amdgcn_perm(d2, d1, 0x0C050C02u);      // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
    bp[2] = __builtin_amdgcn_perm(d1, d0, 0x0C040C01u); bp[3] = __builtin_amdgcn_perm(d2, d1, 0x0C060C03u);
    bp[4] = __builtin_amdgcn_perm(d1, d0, 0x0C050C02u); bp[5] = __builtin_amdgcn_perm(0u, d2, 0x0C030C00u);
    if (gx < 0) {                 // left of the image: every column replicates column 0
        bp[0] = bp[1] = pk_dup(bp[0] & 0xFFFFu); bp[2] = bp[3] = pk_dup(bp[2] & 0xFFFFu); bp[4] = bp[5] = pk_dup(bp[4] & 0xFFFFu);
    } else if (gx > W - 4) {      // right of it: column W-1
        bp[0] = bp[1] = pk_dup(bp[1] >> 16); bp[2] = bp[3] = pk_dup(bp[3] >> 16); bp[4] = bp[5] = pk_dup(bp[5] >> 16);
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) bp[k] = codes_of(bp[k]);
}
__device__ __forceinline__ void u1_codes_planar(int gx, int W, uint32_t d, uint32_t &p01, uint32_t &p23) {
    p01 = __builtin_amdgcn_perm(0u, d, 0x0C010C00u); p23 = __builtin_amdgcn_perm(0u, d, 0x0C030C02u);
    if (gx < 0) p01 = p23 = pk_dup(p01 & 0xFFFFu);
    else if (gx > W - 4) p01 = p23 = pk_dup(p23 >> 16);
    p01 = codes_of(p01); p23 = codes_of(p23);
}
__device__ __forceinline__ bool u1_aligned(const StageArgs &a, bool &hwc3, bool &planar) {
    const bool al4 = ((a.W | a.in.sY) & 3) == 0 && (a.in.sN & 3) == 0 && (((uintptr_t)a.in.p) & 3) == 0;
    hwc3 = al4 && a.C == 3 && a.in.sC == 1 && a.in.sX == 3;     // packed RGB rows: 12-byte groups of four pixels
    planar = al4 && a.in.sX == 1 && (a.in.sC & 3) == 0;          // planar rows: dwords of four pixels
    return hwc3 || planar;
}

// Routing statistic of the tube stage, one wave per 64 x 64 tile: on every fourth row of the tile (+ halo) the share of four-pixel
// groups that span more than one MSB step; a tile above detail_per_1024 is marked in a.tile_list and left to the full-table kernel
// (stage_u1w_kernel, list mode).  The statistic stage_u1t_kernel takes inside its tile loop, bit for bit.
__global__ void __launch_bounds__(256) u1_route_kernel(StageArgs a, uint32_t detail_per_1024) {
    constexpr int PH = K1T_PH, GR = K1S_GR;
    const int tile = (int)(blockIdx.x * 4u + (threadIdx.x >> 6)), lane = (int)(threadIdx.x & 63);
    if (blockIdx.x == 0 && threadIdx.x == 0 && a.tile_count) a.tile_count[2] = 1u;      // "the marks mean something" (for the final stage's statistic)
    if (tile >= a.N * a.tiles_x * a.tiles_y) return;      // wave-uniform
    bool hwc3, planar;
    u1_aligned(a, hwc3, planar);
    int n, y0, x0;
    decode_tile(a, tile, n, y0, x0, K1T_TW, K1T_TH);
    const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
    uint32_t far = 0, seen = 0;
    if (hwc3) {
        for (int i = lane; i < (PH / 4) * GR; i += 64) {
            const int row = 4 * (i / GR) + 1, g = i % GR;
            const int gy = imin(imax(y0 + row - kHalo, ylo), yhi), gx = x0 - 4 + 4 * g, cgx = imin(imax(gx, 0), a.W - 4);
            const uint32_t *src = (const uint32_t *)view_addr(a.in, n, 0, gy, cgx);
            uint32_t bp[6];
            u1_codes_hwc(gx, a.W, src[0], src[1], src[2], bp);
            far += far_apart(bp[0], bp[1]) + far_apart(bp[2], bp[3]) + far_apart(bp[4], bp[5]);
            seen += 3;
        }
    } else {
        for (int i = lane; i < a.C * (PH / 4) * GR; i += 64) {
            const int c = i / (GR * (PH / 4)), row = 4 * ((i / GR) % (PH / 4)) + 1, g = i % GR;
            const int gy = imin(imax(y0 + row - kHalo, ylo), yhi), gx = x0 - 4 + 4 * g, cgx = imin(imax(gx, 0), a.W - 4);
            uint32_t p01, p23;
            u1_codes_planar(gx, a.W, *(const uint32_t *)view_addr(a.in, n, c, gy, cgx), p01, p23);
            far += far_apart(p01, p23);
            seen += 1;
        }
    }
    for (int o = 32; o > 0; o >>= 1) { far += __shfl_down(far, o); seen += __shfl_down(seen, o); }
    if (lane == 0 && far * 1024u > detail_per_1024 * seen) {
        a.tile_list[tile] = 1u;
        // counted only while few: the list kernel asks "fewer than half the workgroups?", and on detailed content tens of thousands of
        // atomics on one address would be a cost of their own
        if (a.tile_count && __hip_atomic_load(a.tile_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 160u) atomicAdd(a.tile_count, 1u);
    }
}

template <int U, int PATS>
__global__ void __launch_bounds__(u1s_threads(U, PATS), u1s_waves(U, PATS)) stage_u1s_kernel(StageArgs a, BandArgs b) {
    constexpr int NT = u1s_threads(U, PATS), NW = NT / 64, TW = K1T_TW, PW = K1S_PW, PH = K1S_PH, GR = K1S_GR, BB = u1t_band_bytes<U>();
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    uint32_t *s_next = (uint32_t *)(smem + u1s_lds_bytes<U>(NT) - 16);      // the workgroup's next work item
    if (lds_addr_of(smem) != 0u) __builtin_trap();      // the band reads assume the dynamic LDS block starts at address 0 (no static LDS here)

    uint32_t pats_rt = 0u;       // pattern of mode m in bits 2m, 2m + 1 (scalar)
    for (int m = 0; m < a.M; ++m) {
        const int pat = a.dj[m][0] == 2 ? 1 : a.di[m][0] == 1 ? 2 : 0;
        pats_rt |= (uint32_t)pat << (2 * m);
        const uint32_t *src = (const uint32_t *)b.band[m];
        uint32_t *dst = (uint32_t *)(smem + pat * BB);
        for (int i = (int)threadIdx.x; i < BB / 4; i += NT) dst[i] = src[i];
    }
    pats_rt = (uint32_t)__builtin_amdgcn_readfirstlane((int)pats_rt);
    if (threadIdx.x == 0) *s_next = 0u;
    const int ntiles = a.N * a.tiles_x * a.tiles_y;
    const int G = gridDim.x;
    const bool by_xcd = (G & 7) == 0;
    const int per = (ntiles + 7) >> 3;
    const int first = by_xcd ? (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int last = by_xcd ? imin(((int)(blockIdx.x & 7) + 1) * per, ntiles) : ntiles;
    const int step = by_xcd ? (G >> 3) : G;
    const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
    bool hwc3, planar;
    const bool aligned = u1_aligned(a, hwc3, planar);
    const bool routed = a.verdict_take >= 0 && aligned;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint8_t *img = smem + 3 * BB + wave * kU1sImgBytes;                                           // this wave's image of pixel codes
    uint32_t *fix_buf = (uint32_t *)(smem + 3 * BB + NW * kU1sImgBytes) + wave * kU1sFixCap;      // ... and its share of the fix-up list in the making
    uint32_t fix_have = 0u;      // (scalar)
    __syncthreads();             // bands staged, counter zeroed: the only barrier of the kernel
    if (first >= last) return;   // workgroup-uniform

    auto opaque_lane = [&]() {   // per-lane index terms are re-derived wherever they are needed: nothing of them may live across the pixel loop
        int l = (int)(threadIdx.x & 63);
        asm volatile("" : "+v"(l));
        return l;
    };
    // work item j of the workgroup = strip j & 15 of its (j >> 4)-th tile; -1 = none left.  Lane-uniform.
    auto grab = [&]() {
        for (;;) {
            uint32_t j = 0;
            if ((threadIdx.x & 63) == 0) j = atomicAdd(s_next, 1u);
            j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
            const long long tile = (long long)first + (long long)(j / kU1sStrips) * step;
            if (tile >= last) return -1;
            if (routed && a.tile_list[tile] != 0u) continue;      // left to the full-table kernel
            int n, ty0, tx0;
            decode_tile(a, (int)tile, n, ty0, tx0, TW, K1T_TH);
            if (ty0 + K1S_TH * (int)(j % kU1sStrips) >= a.oy1) continue;
            return (int)j;
        }
    };
    auto origin = [&](int j, int &n, int &y0, int &x0) {
        decode_tile(a, first + (j / kU1sStrips) * step, n, y0, x0, TW, K1T_TH);
        y0 += K1S_TH * (j % kU1sStrips);
    };
    // the strip's 8 image rows as aligned dwords: 144 groups of four pixels, lane l takes groups l, l + 64 and (l < 16) l + 128
    constexpr int KH = (PH * GR + 63) / 64, KP = (3 * PH * GR + 63) / 64, NR = 3 * KH > KP ? 3 * KH : KP;
    uint32_t R[NR];
#pragma unroll
    for (int k = 0; k < NR; ++k) R[k] = 0u;
    auto fetch = [&](int j) {          // (the clamped index of a surplus slot reads a valid address; never converted)
        if (j < 0 || !aligned) return;
        int n, y0, x0;
        origin(j, n, y0, x0);
        const int l = opaque_lane();
        if (hwc3) {
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                const int i = imin(l + 64 * k, PH * GR - 1), g = i % GR, row = i / GR;
                const int gy = imin(imax(y0 + row - kHalo, ylo), yhi), cgx = imin(imax(x0 - 4 + 4 * g, 0), a.W - 4);
                const uint32_t *src = (const uint32_t *)view_addr(a.in, n, 0, gy, cgx);
                R[3 * k] = src[0]; R[3 * k + 1] = src[1]; R[3 * k + 2] = src[2];
            }
        } else {
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int i = imin(l + 64 * k, a.C * PH * GR - 1), g = i % GR, row = (i / GR) % PH, c = i / (GR * PH);
                const int gy = imin(imax(y0 + row - kHalo, ylo), yhi), cgx = imin(imax(x0 - 4 + 4 * g, 0), a.W - 4);
                R[k] = *(const uint32_t *)view_addr(a.in, n, c, gy, cgx);
            }
        }
    };
    // one group of four pixel codes (two packed pairs) of channel c: image columns 4g - 4 .. 4g - 1 (from x0) -> image columns 4g - 2 ...
    auto put4 = [&](int c, int row, int g, uint32_t c01, uint32_t c23) {
        uint32_t *dst = (uint32_t *)(img + 2 * ((c * PH + row) * PW + 4 * g - 2));
        if (g > 0) dst[0] = c01;
        if (4 * g + 1 < PW) dst[1] = c23;
    };
    auto stash = [&](int j) {
        if (j < 0) return;
        int n, y0, x0;
        origin(j, n, y0, x0);
        const int l = opaque_lane();
        if (hwc3) {
#pragma unroll
            for (int k = 0; k < KH; ++k) {
                const int i = l + 64 * k;
                if (i < PH * GR) {
                    const int g = i % GR, row = i / GR;
                    uint32_t bp[6];
                    u1_codes_hwc(x0 - 4 + 4 * g, a.W, R[3 * k], R[3 * k + 1], R[3 * k + 2], bp);
                    put4(0, row, g, bp[0], bp[1]); put4(1, row, g, bp[2], bp[3]); put4(2, row, g, bp[4], bp[5]);
                }
            }
        } else if (planar) {
#pragma unroll
            for (int k = 0; k < KP; ++k) {
                const int i = l + 64 * k;
                if (i < a.C * PH * GR) {
                    const int g = i % GR, row = (i / GR) % PH, c = i / (GR * PH);
                    uint32_t p01, p23;
                    u1_codes_planar(x0 - 4 + 4 * g, a.W, R[k], p01, p23);
                    put4(c, row, g, p01, p23);
                }
            }
        } else {
            for (int i = l; i < a.C * PH * PW; i += 64) {
                const int px = i % PW, row = (i / PW) % PH, c = i / (PW * PH);
                const int gy = imin(imax(y0 + row - kHalo, ylo), yhi), gx = imin(imax(x0 + px - kHalo, 0), a.W - 1);
                ((uint16_t *)img)[i] = (uint16_t)pixel_code1(*view_addr(a.in, n, c, gy, gx));
            }
        }
    };
    auto fix_flush = [&]() {
        if (fix_have == 0u) return;      // wave-uniform
        // (called where lanes outside the image are masked off: the copy is shared out among the ACTIVE lanes)
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

    constexpr bool PF = PATS != 0;      // instances with registers to spare keep the next strip's pixels in flight during the sites
    int item = grab();
    if (PF) fetch(item);
    while (item >= 0) {
        if (!PF) fetch(item);
#if defined(MULUT_VARIANT_s1nostash)     /* timing-only: the image is converted for the first strip only */
        if (fix_have == 0xFFFFFFFFu || s_next[1] == 0u) { stash(item); if ((threadIdx.x & 63) == 0) s_next[1] = 1u; }
#else
        stash(item);          // the wave's image is its own: the previous strip's reads have returned, LDS serves a wave in order
#endif
        const int nxt = grab();
        if (PF) fetch(nxt);   // in flight while this strip's sites are computed
        int n, y0, x0;
        origin(item, n, y0, x0);
        {
            const int l = opaque_lane();
            if (y0 + l / (TW / 4) >= a.oy1 || x0 + (l % (TW / 4)) * 4 >= a.W) { item = nxt; continue; }      // (per lane; the lanes that stay go on together)
        }
        auto coords = [&](int &ty_, int &tx_) {
            const int l = opaque_lane();
            tx_ = (l % (TW / 4)) * 4;
            ty_ = l / (TW / 4);
        };
#pragma clang loop unroll(disable)
        for (int c = 0; c < a.C; ++c) {
            uint32_t dirty;
            {   // the 5 x 8 window of the lane's four pixels, only for the neighbourhood test
                uint32_t win8[5][4];
                int ty, tx4;
                coords(ty, tx4);
                const uint2 *row = (const uint2 *)(img + 2 * ((c * PH + ty) * PW + tx4));
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    const uint2 lo = row[q * (PW / 4)], hi = row[q * (PW / 4) + 1];
                    win8[q][0] = lo.x; win8[q][1] = lo.y; win8[q][2] = hi.x; win8[q][3] = hi.y;
                }
#if defined(MULUT_VARIANT_s1nodirty)     /* timing-only */
                dirty = win8[0][0] == 0xdeadbeefu ? 1u : 0u;
#else
                dirty = u1t_dirty(win8);
#endif
                asm volatile("" : "+v"(dirty));     // computed HERE: sunk below the pixel loop, its 20 window registers would be parked in scratch
            }
            uint32_t packed = 0;
#pragma clang loop unroll(disable)
            for (int it = 0; it < 2; ++it) {       // two pixels per step of a real loop (register budget)
                int ty, tx4;
                coords(ty, tx4);
                const int y = y0 + ty, x = x0 + tx4;
                if (U == 2 && x + 2 * it >= a.W) break;
                uint32_t win[5][3];
                const uint32_t *row = (const uint32_t *)(img + 2 * ((c * PH + ty) * PW + tx4 + 2 * it));
#pragma unroll
                for (int q = 0; q < 5; ++q) {
                    win[q][0] = row[q * (PW / 2)]; win[q][1] = row[q * (PW / 2) + 1]; win[q][2] = row[q * (PW / 2) + 2];
                }
#if defined(MULUT_VARIANT_s1twice)       /* timing-only: every pixel body twice */
                {
                    uint32_t e0 = u1t_pixel<U, 0, PATS>(a, pats_rt, win);
                    asm volatile("" : "+v"(e0), "+v"(win[2][1]));
                    uint32_t e1 = u1t_pixel<U, 1, PATS>(a, pats_rt, win);
                    asm volatile("" : "+v"(e1), "+v"(win[2][0]));
                    win[0][0] ^= (e0 ^ e1) & 0x100u;      // (codes never have bit 8 set in a way that matters: keeps the first round alive)
                    win[0][0] &= ~0x100u;
                }
#endif
                uint32_t b0 = u1t_pixel<U, 0, PATS>(a, pats_rt, win);
                asm volatile("" : "+v"(b0), "+v"(win[2][1]));       // the second pixel starts after the first is done
                const uint32_t b1 = u1t_pixel<U, 1, PATS>(a, pats_rt, win);
                if constexpr (U == 1) {
                    packed |= (b0 | (b1 << 8)) << (16 * it);
                } else {
                    // two 2 x 2 blocks side by side: HR rows 2y and 2y + 1, columns 2 (x + 2 it) .. + 3
                    const int xo = 2 * (x + 2 * it);
                    const uint32_t top = (b0 & 0xFFFFu) | (b1 << 16), bot = (b0 >> 16) | (b1 & 0xFFFF0000u);
                    uint8_t *d0 = const_cast<uint8_t *>(view_addr(a.out, n, c, 2 * y, xo));
                    uint8_t *d1 = const_cast<uint8_t *>(view_addr(a.out, n, c, 2 * y + 1, xo));
                    if (a.out.sX == 1 && x + 2 * it + 1 < a.W && ((((uintptr_t)d0) | ((uintptr_t)d1)) & 3) == 0) {
                        *(uint32_t *)d0 = top;
                        *(uint32_t *)d1 = bot;
                    } else {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            if (x + 2 * it + i / 2 < a.W) {
                                d0[i * a.out.sX] = (uint8_t)(top >> (8 * i));
                                d1[i * a.out.sX] = (uint8_t)(bot >> (8 * i));
                            }
                    }
                }
            }
            int ty, tx4;
            coords(ty, tx4);
            const int y = y0 + ty, x = x0 + tx4;
            if constexpr (U == 1) {
                uint8_t *dst = const_cast<uint8_t *>(view_addr(a.out, n, c, y, x));
                if (a.out.sX == 1 && x + 3 < a.W && (((uintptr_t)dst) & 3) == 0) {
                    *(uint32_t *)dst = packed;
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        if (x + i < a.W) dst[i * a.out.sX] = (uint8_t)(packed >> (8 * i));
                }
            }
#if !defined(MULUT_VARIANT_nofixlist)    /* (timing-only variant: nothing is listed, flagged sites stay wrong) */
            // sites that may have left the tube: into the wave's buffer, pixel slot by pixel slot (an append is at most 64 entries)
            if (__ballot(dirty != 0u) != 0ull) {
                const uint32_t id0 = (uint32_t)(((n * a.C + c) * a.H + y) * a.W + x);
                const unsigned long long below = (1ull << (threadIdx.x & 63)) - 1ull;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool d = ((dirty >> i) & 1u) != 0u && x + i < a.W;
                    const unsigned long long dm = __ballot(d);
                    if (dm == 0ull) continue;
                    const uint32_t cnt = (uint32_t)__popcll(dm);
                    if (fix_have + cnt > (uint32_t)kU1sFixCap) fix_flush();
                    if (d) fix_buf[fix_have + (uint32_t)__popcll(dm & below)] = id0 + (uint32_t)i;
                    fix_have += cnt;
                }
            }
#endif
        }
        item = nxt;
    }
    fix_flush();
}


// the tube stage as wave-owned strips (stage_u1s_kernel) + the routing statistic as a kernel of its own; same lists, same results
template <int U>
static hipError_t launch_u1s_t(const StageArgs &a, const BandArgs &b, unsigned detail_per_1024, int num_cus, hipStream_t st) {
    const bool sdy = a.M == 3 && a.di[0][0] == 0 && a.dj[0][0] == 1 && a.dj[1][0] == 2 && a.di[2][0] == 1 && a.dj[2][0] == 1;
    auto kern = sdy ? stage_u1s_kernel<U, kU1tPatsSDY> : stage_u1s_kernel<U, 0>;
    const int threads = sdy ? u1s_threads(U, kU1tPatsSDY) : u1s_threads(U, 0);
    {
        const hipError_t e = raise_lds_limit((const void *)kern, 96 * 1024);
        if (e != hipSuccess) return e;
    }
    const long long ntiles = (long long)a.N * a.tiles_x * a.tiles_y;
    if (ntiles <= 0 || ntiles > 0x7fffffffLL / kU1sStrips) return hipErrorInvalidValue;
    const bool aligned = ((a.W | a.in.sY) & 3) == 0 && (a.in.sN & 3) == 0 && (((uintptr_t)a.in.p) & 3) == 0 &&
                         ((a.C == 3 && a.in.sC == 1 && a.in.sX == 3) || (a.in.sX == 1 && (a.in.sC & 3) == 0));
    if (a.verdict_take >= 0 && aligned)
        hipLaunchKernelGGL(u1_route_kernel, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, st, a, (uint32_t)detail_per_1024);
    // as many workgroups as stay resident (two of 1024 threads, three of 512, two of 768 per CU), but no more than there are quarter tiles:
    // a small launch spreads its strips over more workgroups (a workgroup's waves share its run of tiles)
    const int per_cu = threads == 512 ? 3 : 2;
    const long long want = (long long)per_cu * num_cus, quarter = (ntiles * kU1sStrips + 4 * (threads / 64) - 1) / (4 * (threads / 64));
    long long grid = ntiles < want ? ntiles : want;
    (void)quarter;
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(threads), (size_t)u1s_lds_bytes<U>(threads), st, a, b);
    return hipGetLastError();
}
// the instance with the shipped mode list compiled in fits 64 registers; a run-time list would spill in this kernel and stays with the tile kernel
bool stage_u1s_supported(const StageArgs &a) {
    return a.C <= 3 && a.M == 3 && a.di[0][0] == 0 && a.dj[0][0] == 1 && a.dj[1][0] == 2 && a.di[2][0] == 1 && a.dj[2][0] == 1;
}
hipError_t launch_stage_u1s(const StageArgs &a, const BandArgs &b, unsigned detail_per_1024, int num_cus, hipStream_t st) {
    if (!stage_u1s_supported(a) || !a.fix_list || !a.fix_count) return hipErrorInvalidValue;
    return launch_u1s_t<1>(a, b, detail_per_1024, num_cus, st);
}

