// ARCHIVED, NOT BUILT (round 4).  stage_up_fixb_kernel: the fix-up of the tube kernels' work list with the tube bands in LDS -- a pass of a
// listed sample that stays in the tube (84 % of them on the synthetic natural field: a listed sample has 1.9 dirty passes of 12 on average)
// takes its rows from the band, only the others gather from the full tables; two items in flight per 16-lane group; the 5 x 5 window of an
// item as five 8-byte rows through LDS.  Bit-exact (GPU suite green with it selected).  Measured (MI355X, final stage incl. fix-up, us per
// frame of LR 1080x1920x3 at 32 frames per launch, D-natural; profiles/r04r..r04t_ab_fixb.jsonl):
//     stage_up_fix2_kernel (shipped: 8 workgroups of 256 threads per CU, every row from the tables)      170.3 - 174.6
//     fixb, one item per group                                                                            179.4   (vs 172.8 in the same run)
//     fixb, two items in flight per group                                                                 173.5   (vs 170.3)
//     fixb, two items + cooperative window                                                                180.0   (vs 174.6)
// The bands cost 100 KB of LDS: one 1024-thread workgroup (16 waves) per CU where the shipped kernel keeps 32 waves resident, and an item is
// three dependent trips to memory (entry, pixels, rows): with the gathers gone the walk is latency-bound at half the occupancy.
// The cooperative window alone, inside the shipped kernel (five 8-byte gathers per item instead of 48 byte gathers): 175.5 vs 175.8
// (profiles/r04u_ab_fix2_window.jsonl) -- the pixel gathers are not what the shipped kernel waits for either.
// (simplex4_strided<SA, SB, SC, SD> = simplex4 of mulut_core.h with the four key strides as template parameters.)
// stage_up_fix2_kernel with the tube bands in LDS (round 4).  A sample is on the list because SOME pass of it left the tube: on the synthetic
// natural field 1.9 of its 12 passes on average (55 % of the listed samples have exactly one), and the kernel above still fetches the rows
// of all twelve from the full tables -- 60 scattered 16-byte gathers per entry, which is what it is bound by (the texture path retires
// about one lane of such a gather per cycle).  Here a pass that stays in the tube takes its five rows from the band in LDS (the expanded
// image the tube kernels use: LO plane e(4k) | e(4k+2) << 16, HI plane e(4k+1) | e(4k+3) << 16), only the passes that left it gather
// from the tables.  Same work decomposition (a 16-lane group per entry, a pass per lane, block sums in LDS), same integer sums.
// Mode lists of up to four modes (what the tube kernels take); one 1024-thread workgroup per CU.
constexpr int kFixbE = 2;                  // entries a 16-lane group has in flight (the walk is three dependent trips to memory per entry; one workgroup per CU)
constexpr int kFixbLdsBytes = 3 * kTubeBandBytes + 64 * kFixbE * 16 * 4 + 64 * kFixbE * 5 * 8;      // bands, block sums, 5 x 8-byte windows
__global__ void __launch_bounds__(1024) stage_up_fixb_kernel(StageArgs a, BandArgs b) {
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    int *s_sum = (int *)(smem + 3 * kTubeBandBytes);
    const uint32_t count = *a.fix_count;
    if (blockIdx.x * 64u >= count) return;      // workgroup-uniform: nothing for this workgroup, no band is staged
    for (int m = 0; m < a.M; ++m) {             // bands: slot = pattern id of the mode (s, d, y)
        const int pat = a.dj[m][0] == 2 ? 1 : a.di[m][0] == 1 ? 2 : 0;
        const uint4 *src = (const uint4 *)b.band[m];
        uint4 *dst = (uint4 *)(smem + pat * kTubeBandBytes);
        for (int i = threadIdx.x; i < kTubeBandBytes / 16; i += 1024) dst[i] = src[i];
    }
    __syncthreads();
    const int grp = (int)(threadIdx.x >> 4), ln = (int)(threadIdx.x & 15);
    int *sum = s_sum + grp * (kFixbE * 16);
    uint2 *win = (uint2 *)(smem + 3 * kTubeBandBytes + 64 * kFixbE * 16 * 4) + grp * (kFixbE * 5);      // per item: window rows y-2 .. y+2, bytes x-2 .. x+5
    const int ylo = imax(a.oy0 - kHalo, 0), yhi = imin(a.oy1 + kHalo, a.H) - 1;
    const int unbias = 128 * kQ * 4 * a.M - a.bias_num;
    const int m = ln >> 2, r = ln & 3;          // this lane's pass (lanes 4 M .. 15 idle in the pass phase; M <= 4)
    const bool has_pass = ln < 4 * a.M;
    const int mm = has_pass ? m : 0;
    const int pat = a.dj[mm][0] == 2 ? 1 : a.di[mm][0] == 1 ? 2 : 0;
    int ody[3], odx[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) sample_offset(r, a.di[mm][k], a.dj[mm][k], ody[k], odx[k]);
    const uint8_t *band = smem + pat * kTubeBandBytes;
    const uint4 *tab = (const uint4 *)a.lut[mm];
    // A group's 16 lanes sit in one wave and LDS serves a wave's operations in order: no workgroup barrier in the loop, the groups run
    // independently (the fences only keep the compiler from moving the LDS accesses across each other).  A group takes kFixbE samples
    // (entry, channel) per trip -- their pixel loads, then their row loads, are in flight together.
    // Work items are samples: an entry that names every channel (border columns of the detailed-tile path) yields one item per channel.
    const uint32_t stride = gridDim.x * 64u;
    uint32_t i = blockIdx.x * 64u + (uint32_t)grp;
    int c_next = -1;                            // >= 0: the current entry still has this channel (and the ones after it) to do
    while (i < count || c_next >= 0) {          // uniform in the group
        int X[kFixbE], Y[kFixbE], N[kFixbE], Cc[kFixbE];
        bool ok[kFixbE];
#pragma unroll
        for (int e = 0; e < kFixbE; ++e) {
            ok[e] = false;
            X[e] = Y[e] = N[e] = Cc[e] = 0;
            while (i < count) {
                const uint32_t ent = a.fix_list[i], id = ent & 0x3FFFFFFFu, only = ent >> 30;
                const int x = (int)(id % (uint32_t)a.W), y = (int)((id / (uint32_t)a.W) % (uint32_t)a.H), n = (int)(id / ((uint32_t)a.W * (uint32_t)a.H));
                const int c_hi = only == 3u ? imin(a.C, 3) : imin((int)only + 1, a.C);
                const int c = c_next >= 0 ? c_next : (only == 3u ? 0 : (int)only);
                if (n >= a.N || y < a.oy0 || y >= a.oy1 || c >= c_hi) { c_next = -1; i += stride; continue; }      // never follow an entry outside the launch
                X[e] = x; Y[e] = y; N[e] = n; Cc[e] = c; ok[e] = true;
                if (c + 1 < c_hi) c_next = c + 1; else { c_next = -1; i += stride; }
                break;
            }
            if (!ok[e]) c_next = -1;
        }
        // The 5 x 5 window of an item as five 8-byte rows (from column x - 2), one row per lane of the group, into the group's LDS slot: five
        // gathers per item where every pass fetching its own four pixels makes 48.  Border columns (and inputs that are not planar) read the
        // image directly, with edge replication.
        bool inner[kFixbE];
#pragma unroll
        for (int e = 0; e < kFixbE; ++e) {
            sum[e * 16 + ln] = 0;
            inner[e] = ok[e] && a.in.sX == 1 && X[e] >= kSlabXLo && X[e] < a.W - slab_x_hi(a);      // 8 bytes from x - 2 stay inside the row / the padding
            if (inner[e] && ln < 5) {
                uint2 t;
                __builtin_memcpy(&t, view_addr(a.in, N[e], Cc[e], imin(imax(Y[e] + ln - 2, ylo), yhi), X[e] - 2), 8);
                win[e * 5 + ln] = t;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        int va[kFixbE], v[kFixbE][3];
#pragma unroll
        for (int e = 0; e < kFixbE; ++e) {
            auto px = [&](int dy, int dx) {
                if (inner[e]) return (int)((const uint8_t *)&win[e * 5 + dy + 2])[dx + 2];
                const int gy = imin(imax(Y[e] + dy, ylo), yhi), gx = imin(imax(X[e] + dx, 0), a.W - 1);
                return (int)*view_addr(a.in, N[e], Cc[e], gy, gx);
            };
            va[e] = v[e][0] = v[e][1] = v[e][2] = 0;
            if (ok[e] && has_pass) {
                va[e] = px(0, 0);
#pragma unroll
                for (int k = 0; k < 3; ++k) v[e][k] = px(ody[k], odx[k]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        // rows: from the band in LDS when the pass stays in the tube, else from the full table
        int w[kFixbE][5];
        uint4 rowA[kFixbE][5], rowB[kFixbE][5];      // in the tube: LO / HI plane of a band row; else: the table row (value + 128 bytes) / unused
        bool tube[kFixbE];
#pragma unroll
        for (int e = 0; e < kFixbE; ++e) {
            const int ha = va[e] >> 4, hb = v[e][0] >> 4, hc = v[e][1] >> 4, hd = v[e][2] >> 4;
            const int mx = imax(imax(ha, hb), imax(hc, hd)), mn = imin(imin(ha, hb), imin(hc, hd));
            tube[e] = mx - mn <= 1;
            int idx[5];
            if (tube[e]) {
                simplex4_strided<kTubeSA, kTubeSB, kTubeSC, kTubeSD>(va[e], v[e][0], v[e][1], v[e][2], idx, w[e]);
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    rowA[e][j] = *(const uint4 *)(band + idx[j] * 16);
                    rowB[e][j] = *(const uint4 *)(band + kTubePlaneBytes + idx[j] * 16);
                }
            } else {
                simplex4(va[e], v[e][0], v[e][1], v[e][2], idx, w[e]);
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    rowA[e][j] = (ok[e] && has_pass) ? tab[idx[j]] : make_uint4(0u, 0u, 0u, 0u);
                    rowB[e][j] = make_uint4(0u, 0u, 0u, 0u);
                }
            }
        }
#pragma unroll
        for (int e = 0; e < kFixbE; ++e) {
            if (ok[e] && has_pass) {
                uint32_t lo[4] = {0u, 0u, 0u, 0u}, hi[4] = {0u, 0u, 0u, 0u};      // sums as 16-bit fields: lo[k] = e(4k) | e(4k+2) << 16, hi[k] = e(4k+1) | e(4k+3) << 16
#pragma unroll
                for (int j = 0; j < 5; ++j) {
                    const uint32_t wj = (uint32_t)w[e][j];
                    const uint32_t ra[4] = {rowA[e][j].x, rowA[e][j].y, rowA[e][j].z, rowA[e][j].w}, rb[4] = {rowB[e][j].x, rowB[e][j].y, rowB[e][j].z, rowB[e][j].w};
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        lo[k] += (tube[e] ? ra[k] : (ra[k] & 0x00FF00FFu)) * wj;
                        hi[k] += (tube[e] ? rb[k] : ((ra[k] >> 8) & 0x00FF00FFu)) * wj;
                    }
                }
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const uint32_t word = (q & 1) ? hi[q >> 2] : lo[q >> 2];
                    const int val = (int)((q & 2) ? (word >> 16) : (word & 0xFFFFu));
                    // block position (sy, sx) that rotation r gives row element q (the inverse of row_elem)
                    const int pos = r == 0 ? q : r == 1 ? (q & 3) * 4 + 3 - (q >> 2) : r == 2 ? 15 - q : (3 - (q & 3)) * 4 + (q >> 2);
                    atomicAdd(&sum[e * 16 + pos], val);
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int e = 0; e < kFixbE; ++e)
            if (ok[e]) {
                const uint32_t byte = rhe_clip_u8(sum[e * 16 + ln] - unbias, a.div);
                *const_cast<uint8_t *>(view_addr(a.out, N[e], Cc[e], Y[e] * 4 + (ln >> 2), X[e] * 4 + (ln & 3))) = (uint8_t)byte;
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // sums read before the next trip clears them
    }
}


// the band-assisted fix-up (tuning "fix_kernel" 3, the default where it applies): u == 4 stages of up to four modes, bm.band[m] = expanded tube band of mode m
hipError_t launch_stage_up_fix_band(const StageArgs &a, const BandArgs &bm, int num_cus, hipStream_t st) {
    if (a.C > 3 || a.M > 4 || !a.fix_list || !a.fix_count) return hipErrorInvalidValue;
    {
        const hipError_t e = raise_lds_limit((const void *)stage_up_fixb_kernel, 160 * 1024);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(stage_up_fixb_kernel, dim3((unsigned)num_cus), dim3(1024), (size_t)kFixbLdsBytes, st, a, bm);
    return hipGetLastError();
}

