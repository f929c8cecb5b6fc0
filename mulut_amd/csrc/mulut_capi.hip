// mulut_capi.hip -- implementation of the C ABI declared in include/mulut.h.
// Owns: the context (device id, model shape), device copies of the tables, the ping-pong
// workspace for intermediate stage images.  All image buffers belong to the caller.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "../../include/mulut.h"
#include "mulut_kernels.h"

using namespace mulut;

struct DevTable {
    void *dev = nullptr;    // full table image
    void *tube = nullptr;   // "tube" band (keys spanning <= 2 MSB steps): v_num 16: expanded to 16-bit fields, two planes of kTubeSlots x 16 B;
                            // v_num 1: one dword per slot (kTube1BandBytes)
    size_t tube_bytes = 0;
    uint8_t *slab = nullptr; // v_num 16: the table as 16 anchor slab pairs (mulut_core.h), kSlabTableBytes
    int vnum = 0;
    size_t bytes = 0;
};

struct mulut_ctx {
    int device = 0;
    bool configured = false;
    int stages = 0, n_modes = 0, scale = 0, interval = 0;
    char modes[MULUT_MAX_MODES + 1] = {0};
    signed char di[MULUT_MAX_MODES][3], dj[MULUT_MAX_MODES][3];
    int reach = 2;  // LR rows one stage looks beyond its output rows
    DevTable tab[MULUT_MAX_STAGES][3];  // [stage-1][pattern id s,d,y]
    uint8_t *ws[2] = {nullptr, nullptr};
    size_t ws_bytes = 0;
    std::string hip_err;
    int num_cus = 256;
    int final_kernel = 0;   // 0 auto (= 6 hybrid with the tube kernel), 1 full-table kernel, 2 compact LDS band, 3 expanded LDS band, 4 hybrid (band-x),
                            // 5 tube kernel (all bands resident), 6 hybrid (tube)
    int f32_ok[2] = {0, 0}; // float epilogue proven exact for the [non-final, final] divisor
    int fma_ok = 0;         // fused (biased-sum) float epilogue proven exact for the final stage
    float epi_c = 0.0f;
    uint32_t *verdict = nullptr;   // per-tile smooth/detailed verdicts of the hybrid final stage
    size_t verdict_tiles = 0;
    uint32_t *fix = nullptr;       // [0] = count, [16...] = entries of the fix-up list (samples recomputed from the full tables)
    size_t fix_cap = 0;            // capacity in ids
    unsigned long long *dbg = nullptr;   // probe buffer (mulut_debug_read), MULUT_DEBUG_WORDS words, allocated on first use
    uint32_t *det_ctl = nullptr;   // detailed-tile path of the final stage (launch_detail_slab): counters, items, sample ids, blocks
    uint32_t *det_items = nullptr, *det_desc = nullptr, *det_tpos = nullptr, *det_dlist = nullptr;
    uint16_t *det_thist = nullptr;
    uint4 *det_blocks = nullptr;
    size_t det_items_cap = 0, det_ids_cap = 0, det_blocks_cap = 0, det_tiles_cap = 0;
    bool k1_valid = false;         // ctx->tlist holds the marks of the first-stage launch that produced the next stage's input ...
    int k1_N = 0, k1_W = 0, k1_H = 0, k1_tiles_x = 0, k1_tiles_y = 0, k1_oy0 = 0, k1_oy1 = 0;   // ... of this shape ...
    const uint8_t *k1_out = nullptr;                                      // ... written to this buffer
    int stat_from_k1 = 1;          // tuning "stat_from_first_stage": the final stage's statistic looks only at tiles the first stage marked
    int fix_variant = 0;           // tuning "fix_kernel"
    int tube2 = 1;                 // tuning "tube_pipelined": 1 = stage_tube2_kernel (hand-scheduled LDS reads) where the mode list has one, 0 = stage_tube_kernel
    int u1t_persist = 0;           // tuning "u1t_persist": persistent workgroups per CU of the 1-byte-row tube kernel (0 = one workgroup per tile)
    int detail_kernel = 0;         // tuning "detail_kernel": 0 = anchor slabs in LDS (when the launch qualifies), 1 = full-table gather kernel
    int first_kernel = 0;   // 1-byte-row stages: 0 auto (tube kernel, detailed tiles to the window kernel), 2 window kernel (full table in
                            // LDS) on every tile, 3 tube kernel on every tile
    int up_detail_per_1024 = 8;    // the same threshold for the routed x2 / x3 final stages (their detailed tiles go to the gather kernel; profiles/r04y_scale_bench.jsonl)
    int u1_detail_per_1024 = 24;   // a tile goes to the full-table kernel when more than this share of its (sampled) 4-pixel groups spans > 1 MSB step
    uint32_t *tlist = nullptr;     // [16 + tile] = 1: the tube kernel left this tile to the full-table kernel
    size_t tlist_cap = 0;
    int fma1_ok = 0;               // fused float epilogue proven exact for non-final stages
    int hybrid_oob_per_1024 = 128; // a tile is "detailed" when more than 1/8 of its (sampled) sites leave the band
    bool timing = false;
    hipEvent_t ev[MULUT_MAX_STAGES + 1] = {};
    hipEvent_t evk[MULUT_MAX_STAGES][2] = {};   // around each stage's dominant kernel
    bool evk_set[MULUT_MAX_STAGES] = {};
    int timed_stages = 0;
};

static int pattern_id(char m) { return m == 's' ? 0 : m == 'd' ? 1 : m == 'y' ? 2 : -1; }

#define HIP_TRY(ctx, expr)                                                                  \
    do {                                                                                    \
        hipError_t e__ = (expr);                                                            \
        if (e__ != hipSuccess) {                                                            \
            if (ctx) (ctx)->hip_err = std::string(#expr) + ": " + hipGetErrorString(e__);   \
            return MULUT_EHIP;                                                              \
        }                                                                                   \
    } while (0)

extern "C" {

int mulut_version(void) { return MULUT_VERSION; }

const char *mulut_strerror(int err) {
    switch (err) {
        case MULUT_OK: return "ok";
        case MULUT_EINVAL: return "invalid argument";
        case MULUT_EMODE: return "Mode not implemented.";
        case MULUT_ENOLUT: return "LUT for (stage, mode) not set";
        case MULUT_ESHAPE: return "LUT shape does not match (83521, v_num) for this stage";
        case MULUT_EUNSUPPORTED: return "unsupported configuration (interval must be 4, scale 1..4)";
        case MULUT_EHIP: return "HIP runtime error";
        case MULUT_ENODEVICE: return "no usable HIP device (there is no CPU path)";
        case MULUT_ENOTCONFIGURED: return "mulut_configure() has not been called";
        case MULUT_EWORKSPACE: return "input rows do not cover the strip plus halo";
        default: return "unknown error";
    }
}

const char *mulut_last_hip_error(const mulut_ctx *ctx) { return ctx ? ctx->hip_err.c_str() : ""; }

int mulut_create(int device_id, mulut_ctx **out_ctx) {
    if (!out_ctx) return MULUT_EINVAL;
    *out_ctx = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return MULUT_ENODEVICE;
    if (device_id < 0 || device_id >= n) return MULUT_EINVAL;
    mulut_ctx *c = new (std::nothrow) mulut_ctx();
    if (!c) return MULUT_EINVAL;
    c->device = device_id;
    if (hipSetDevice(device_id) != hipSuccess) {
        delete c;
        return MULUT_ENODEVICE;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device_id) == hipSuccess && prop.multiProcessorCount > 0)
        c->num_cus = prop.multiProcessorCount;
    *out_ctx = c;
    return MULUT_OK;
}

int mulut_destroy(mulut_ctx *ctx) {
    if (!ctx) return MULUT_EINVAL;
    (void)hipSetDevice(ctx->device);
    for (auto &st : ctx->tab)
        for (auto &t : st)
        {
            if (t.dev) (void)hipFree(t.dev);
            if (t.tube) (void)hipFree(t.tube);
            if (t.slab) (void)hipFree(t.slab);
        }
    for (auto &w : ctx->ws)
        if (w) (void)hipFree(w);
    if (ctx->verdict) (void)hipFree(ctx->verdict);
    if (ctx->fix) (void)hipFree(ctx->fix);
    if (ctx->tlist) (void)hipFree(ctx->tlist);
    if (ctx->dbg) (void)hipFree(ctx->dbg);
    if (ctx->det_ctl) (void)hipFree(ctx->det_ctl);
    if (ctx->det_items) (void)hipFree(ctx->det_items);
    if (ctx->det_desc) (void)hipFree(ctx->det_desc);
    if (ctx->det_thist) (void)hipFree(ctx->det_thist);
    if (ctx->det_tpos) (void)hipFree(ctx->det_tpos);
    if (ctx->det_dlist) (void)hipFree(ctx->det_dlist);
    if (ctx->det_blocks) (void)hipFree(ctx->det_blocks);
    for (auto &e : ctx->ev)
        if (e) (void)hipEventDestroy(e);
    for (auto &p : ctx->evk)
        for (auto &e : p)
            if (e) (void)hipEventDestroy(e);
    delete ctx;
    return MULUT_OK;
}

int mulut_configure(mulut_ctx *ctx, int stages, const char *modes, int scale, int interval) {
    if (!ctx || !modes) return MULUT_EINVAL;
    const size_t M = strlen(modes);
    if (stages < 1 || stages > MULUT_MAX_STAGES || M < 1 || M > MULUT_MAX_MODES) return MULUT_EUNSUPPORTED;
    if (interval != kInterval || scale < 1 || scale > 4) return MULUT_EUNSUPPORTED;
    const int reach = 2;  // tiles always stage a 2-px halo (d / y patterns); s-only models use it too
    for (size_t m = 0; m < M; ++m) {
        int di[3], dj[3];
        if (!pattern_offsets(modes[m], di, dj)) return MULUT_EMODE;
        for (int k = 0; k < 3; ++k) {
            ctx->di[m][k] = (signed char)di[k];
            ctx->dj[m][k] = (signed char)dj[k];
        }
    }
    ctx->stages = stages;
    ctx->n_modes = (int)M;
    ctx->scale = scale;
    ctx->interval = interval;
    memcpy(ctx->modes, modes, M + 1);
    ctx->reach = reach;
    for (int last = 0; last < 2; ++last) {
        const DivMagic dm = make_div_magic((uint32_t)stage_divisor((int)M, last != 0));
        const int span = 128 * kQ * 4 * (int)M;      // |q * sum| <= 128 * 16 * 4M
        ctx->f32_ok[last] = rhe_f32_valid(-span + stage_bias_num((int)M, last != 0), span + stage_bias_num((int)M, last != 0),
                                          dm, 1.0f / (float)dm.d) ? 1 : 0;
    }
    {   // final stage on value+128 rows: sums are biased by 128 per weight unit, S in [0, 255 * 16 * 4M]
        const DivMagic dm = make_div_magic((uint32_t)stage_divisor((int)M, true));
        const int unbias = 128 * kQ * 4 * (int)M - stage_bias_num((int)M, true);
        const float inv_d = 1.0f / (float)dm.d;
        ctx->epi_c = -(float)unbias * inv_d;
        ctx->fma_ok = rhe_fma_valid((uint32_t)(255 * kQ * 4 * (int)M), unbias, dm, inv_d, ctx->epi_c) ? 1 : 0;
    }
    {   // non-final stages: clip(rhe((K + 127 d) / d)) = cvt_u8(fma(K, 1/d, 127)), K in [-128 * 16 * 4M, 127 * 16 * 4M]
        const DivMagic dm = make_div_magic((uint32_t)stage_divisor((int)M, false));
        const int span = 128 * kQ * 4 * (int)M;
        ctx->fma1_ok = rhe_fma_valid_i(-span, span, stage_bias_num((int)M, false), dm, 1.0f / (float)dm.d, 127.0f) ? 1 : 0;
    }
    ctx->configured = true;
    return MULUT_OK;
}

int mulut_set_lut(mulut_ctx *ctx, int stage, char mode, const int8_t *host_rows, int64_t rows, int vnum) {
    if (!ctx || !host_rows) return MULUT_EINVAL;
    if (stage < 1 || stage > MULUT_MAX_STAGES) return MULUT_EINVAL;
    const int pid = pattern_id(mode);
    if (pid < 0) return MULUT_EMODE;
    if (rows != kRows) return MULUT_ESHAPE;
    int u = 0;
    for (int k = 1; k <= 4; ++k)
        if (k * k == vnum) u = k;
    if (!u) return MULUT_ESHAPE;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    DevTable &t = ctx->tab[stage - 1][pid];
    std::vector<uint8_t> img;
    if (u == 1) {
        img.assign(kU1TableBytes, 0);
        memcpy(img.data(), host_rows, kRows);
    } else {
        const int rb = row_dwords(u) * 4;
        img.assign((size_t)kRows * rb, 128);
        for (int64_t i = 0; i < kRows; ++i)
            for (int e = 0; e < vnum; ++e) img[(size_t)i * rb + e] = (uint8_t)((int)host_rows[i * vnum + e] + 128);
    }
    if (t.dev && t.bytes != img.size()) {
        HIP_TRY(ctx, hipFree(t.dev));
        t.dev = nullptr;
    }
    if (!t.dev) HIP_TRY(ctx, hipMalloc(&t.dev, img.size()));
    HIP_TRY(ctx, hipMemcpy(t.dev, img.data(), img.size(), hipMemcpyHostToDevice));
    t.vnum = vnum;
    t.bytes = img.size();
    if (u == 1) {
        // tube band of a 1-byte-row table: one dword per slot, the value as int16 in both halves
        std::vector<uint32_t> tb((size_t)kTube1BandBytes / 4, 0u);
        for (int A = 0; A < kL; ++A)
            for (int B = imax(0, A - 2); B <= imin(kL - 1, A + 2); ++B)
                for (int C = imax(0, A - 2); C <= imin(kL - 1, A + 2); ++C)
                    for (int D = imax(0, A - 2); D <= imin(kL - 1, A + 2); ++D) {
                        if (!tube_contains(A, B, C, D)) continue;
                        const uint32_t v = (uint32_t)(uint16_t)(int16_t)host_rows[(size_t)A * kStrideA + B * kStrideB + C * kStrideC + D];
                        tb[(size_t)tube_slot(A, B, C, D)] = v | (v << 16);
                    }
        if (t.tube && t.tube_bytes != tb.size() * 4) {
            HIP_TRY(ctx, hipFree(t.tube));
            t.tube = nullptr;
        }
        if (!t.tube) HIP_TRY(ctx, hipMalloc(&t.tube, tb.size() * 4));
        t.tube_bytes = tb.size() * 4;
        HIP_TRY(ctx, hipMemcpy(t.tube, tb.data(), tb.size() * 4, hipMemcpyHostToDevice));
    } else if (u == 2) {
        // tube band of a u == 2 table: 8 bytes per slot, (e0 | e1 << 16), (e2 | e3 << 16) as value + 128
        std::vector<uint32_t> tb((size_t)kTube2BandBytes / 4, 0x00800080u);
        for (int A = 0; A < kL; ++A)
            for (int B = imax(0, A - 2); B <= imin(kL - 1, A + 2); ++B)
                for (int C = imax(0, A - 2); C <= imin(kL - 1, A + 2); ++C)
                    for (int D = imax(0, A - 2); D <= imin(kL - 1, A + 2); ++D) {
                        if (!tube_contains(A, B, C, D)) continue;
                        const uint8_t *e = &img[((size_t)A * kStrideA + B * kStrideB + C * kStrideC + D) * 4];
                        tb[(size_t)tube_slot(A, B, C, D) * 2] = (uint32_t)e[0] | ((uint32_t)e[1] << 16);
                        tb[(size_t)tube_slot(A, B, C, D) * 2 + 1] = (uint32_t)e[2] | ((uint32_t)e[3] << 16);
                    }
        if (t.tube && t.tube_bytes != tb.size() * 4) {
            HIP_TRY(ctx, hipFree(t.tube));
            t.tube = nullptr;
        }
        if (!t.tube) HIP_TRY(ctx, hipMalloc(&t.tube, tb.size() * 4));
        t.tube_bytes = tb.size() * 4;
        HIP_TRY(ctx, hipMemcpy(t.tube, tb.data(), tb.size() * 4, hipMemcpyHostToDevice));
    } else if (u == 3) {
        // tube band of a u == 3 table: 24 bytes per slot, the nine values (+ 128) as ten 16-bit fields e0 e1 e2 e3 e4 e4 e5 e6 e7 e8
        std::vector<uint32_t> tb((size_t)kTube3BandBytes / 4, 0x00800080u);
        const int rb = row_dwords(3) * 4;
        for (int A = 0; A < kL; ++A)
            for (int B = imax(0, A - 2); B <= imin(kL - 1, A + 2); ++B)
                for (int C = imax(0, A - 2); C <= imin(kL - 1, A + 2); ++C)
                    for (int D = imax(0, A - 2); D <= imin(kL - 1, A + 2); ++D) {
                        if (!tube_contains(A, B, C, D)) continue;
                        const uint8_t *e = &img[((size_t)A * kStrideA + B * kStrideB + C * kStrideC + D) * rb];
                        uint32_t *d = &tb[(size_t)tube_slot(A, B, C, D) * (kTube3SlotBytes / 4)];
                        uint32_t f[10];
                        for (int q = 0; q < 9; ++q) f[tube3_field(q)] = e[q];
                        f[5] = e[4];
                        for (int k = 0; k < 5; ++k) d[k] = f[2 * k] | (f[2 * k + 1] << 16);
                    }
        if (t.tube && t.tube_bytes != tb.size() * 4) {
            HIP_TRY(ctx, hipFree(t.tube));
            t.tube = nullptr;
        }
        if (!t.tube) HIP_TRY(ctx, hipMalloc(&t.tube, tb.size() * 4));
        t.tube_bytes = tb.size() * 4;
        HIP_TRY(ctx, hipMemcpy(t.tube, tb.data(), tb.size() * 4, hipMemcpyHostToDevice));
    } else if (u == 4) {
        // tube band: rows with max - min of the keys <= 2 at tube_slot(), expanded to 16-bit fields in two planes: LO (lo_k = e(4k) | e(4k+2) << 16)
        // then HI (hi_k = e(4k+1) | e(4k+3) << 16)
        std::vector<uint32_t> tb((size_t)kTubeBandBytes / 4, 0x00800080u);
        for (int A = 0; A < kL; ++A)
            for (int B = imax(0, A - 2); B <= imin(kL - 1, A + 2); ++B)
                for (int C = imax(0, A - 2); C <= imin(kL - 1, A + 2); ++C)
                    for (int D = imax(0, A - 2); D <= imin(kL - 1, A + 2); ++D) {
                        if (!tube_contains(A, B, C, D)) continue;
                        const uint8_t *e = &img[((size_t)A * kStrideA + B * kStrideB + C * kStrideC + D) * 16];
                        const size_t s4 = (size_t)tube_slot(A, B, C, D) * 4;
                        for (int k = 0; k < 4; ++k) {
                            tb[s4 + k] = (uint32_t)e[4 * k] | ((uint32_t)e[4 * k + 2] << 16);
                            tb[(size_t)kTubePlaneBytes / 4 + s4 + k] = (uint32_t)e[4 * k + 1] | ((uint32_t)e[4 * k + 3] << 16);
                        }
                    }
        if (t.tube && t.tube_bytes != tb.size() * 4) {
            HIP_TRY(ctx, hipFree(t.tube));
            t.tube = nullptr;
        }
        if (!t.tube) HIP_TRY(ctx, hipMalloc(&t.tube, tb.size() * 4));
        t.tube_bytes = tb.size() * 4;
        HIP_TRY(ctx, hipMemcpy(t.tube, tb.data(), tb.size() * 4, hipMemcpyHostToDevice));
        // anchor slab pairs: pair A = rows (A, b, c, d) and (A + 1, b, c, d) interleaved, 32 bytes per (b, c, d)
        std::vector<uint8_t> sl((size_t)kSlabTableBytes + 1024, 128);      // the LDS copy of a pair moves whole KiB
        for (int A = 0; A < 16; ++A)
            for (int bcd = 0; bcd < kStrideA; ++bcd)
                for (int f = 0; f < 2; ++f)
                    memcpy(&sl[(size_t)A * kSlabPairBytes + (size_t)bcd * 32 + (size_t)f * 16], &img[((size_t)(A + f) * kStrideA + bcd) * 16], 16);
        if (!t.slab) HIP_TRY(ctx, hipMalloc((void **)&t.slab, sl.size()));
        HIP_TRY(ctx, hipMemcpy(t.slab, sl.data(), sl.size(), hipMemcpyHostToDevice));
    } else {
        if (t.tube) HIP_TRY(ctx, hipFree(t.tube));
        if (t.slab) HIP_TRY(ctx, hipFree(t.slab));
        t.tube = nullptr;
        t.slab = nullptr;
        t.tube_bytes = 0;
    }
    if (u != 4 && t.slab) {
        HIP_TRY(ctx, hipFree(t.slab));
        t.slab = nullptr;
    }
    return MULUT_OK;
}

// bracket the dominant kernel of a stage with events when timing is on (mulut_last_kernel_ms)
#define MAIN_KERNEL(ctx, stage, st, launch)                                                   \
    do {                                                                                      \
        if ((ctx)->timing) HIP_TRY(ctx, hipEventRecord((ctx)->evk[(stage) - 1][0], st));      \
        HIP_TRY(ctx, launch);                                                                 \
        if ((ctx)->timing) {                                                                  \
            HIP_TRY(ctx, hipEventRecord((ctx)->evk[(stage) - 1][1], st));                      \
            (ctx)->evk_set[(stage) - 1] = true;                                               \
        }                                                                                     \
    } while (0)

static int stage_u(const mulut_ctx *ctx, int stage) { return stage == ctx->stages ? ctx->scale : 1; }

// Tables of one stage in mode order, shape-checked against the stage's upscale.
static int stage_tables(const mulut_ctx *ctx, int stage, const void **lut) {
    const int vnum = stage_u(ctx, stage) * stage_u(ctx, stage);
    for (int m = 0; m < ctx->n_modes; ++m) {
        const DevTable &t = ctx->tab[stage - 1][pattern_id(ctx->modes[m])];
        if (!t.dev) return MULUT_ENOLUT;
        if (t.vnum != vnum) return MULUT_ESHAPE;
        lut[m] = t.dev;
    }
    return MULUT_OK;
}

int mulut_pass(mulut_ctx *ctx, int stage, char mode, int r, const uint8_t *in_chw, int H, int W, int C,
               int32_t *out_q, void *stream) {
    if (!ctx || !in_chw || !out_q || H <= 0 || W <= 0 || C <= 0 || r < 0 || r > 3) return MULUT_EINVAL;
    if (!ctx->configured) return MULUT_ENOTCONFIGURED;
    if (stage < 1 || stage > ctx->stages) return MULUT_EINVAL;
    const int pid = pattern_id(mode);
    if (pid < 0) return MULUT_EMODE;
    const DevTable &t = ctx->tab[stage - 1][pid];
    if (!t.dev) return MULUT_ENOLUT;
    const int u = stage_u(ctx, stage);
    if (t.vnum != u * u) return MULUT_ESHAPE;
    PassArgs a;
    a.in = in_chw;
    a.out = out_q;
    a.lut = t.dev;
    a.C = C; a.H = H; a.W = W; a.u = u; a.r = r;
    int di[3], dj[3];
    pattern_offsets(mode, di, dj);
    for (int k = 0; k < 3; ++k) {
        a.di[k] = (signed char)di[k];
        a.dj[k] = (signed char)dj[k];
    }
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    HIP_TRY(ctx, launch_pass(a, (hipStream_t)stream));
    return MULUT_OK;
}

static View make_view(const uint8_t *p, int layout, int rows, int W, int C, int row0) {
    View v;
    v.p = const_cast<uint8_t *>(p);
    v.row0 = row0;
    if (layout == MULUT_LAYOUT_HWC) {
        v.sX = C; v.sC = 1; v.sY = W * C;
    } else {
        v.sX = 1; v.sY = W; v.sC = rows * W;
    }
    v.sN = (long long)rows * W * C;
    return v;
}

static int ensure_verdict(mulut_ctx *ctx, size_t tiles);
static int ensure_fix(mulut_ctx *ctx, size_t ids);
static int ensure_tlist(mulut_ctx *ctx, size_t tiles);
static int ensure_detail(mulut_ctx *ctx, size_t tiles, size_t items, size_t ids, size_t blocks);

static hipError_t tube_launch(mulut_ctx *ctx, const StageArgs &a, const BandArgs &b, int mode, hipStream_t st) {
    if (ctx->tube2 && stage_tube2_supported(a)) return launch_stage_tube2(a, b, mode, ctx->num_cus, st);
    return launch_stage_tube(a, b, mode, ctx->num_cus, st);
}

// Launch one stage: input view holds LR rows [in.row0, ...), outputs for LR rows [oy0, oy1).
// C channels are processed (<= 3); they may be a group of an image with more (then the views carry that image's strides and
// packed_ok is false: the packed-RGB store needs a pixel stride of exactly 3)
// k1_ref / k1_n0: when this launch is a sub-launch of a larger one (run_stage below), the buffer and first image of the WHOLE launch --
// what the first stage's tile marks are recorded against
static int run_stage_one(mulut_ctx *ctx, int stage, const View &in, const View &out, int out_layout, int N, int H, int W,
                         int C, int oy0, int oy1, hipStream_t st, bool packed_ok, const uint8_t *k1_ref, int k1_n0) {
    StageArgs a;
    memset(&a, 0, sizeof(a));
    int rc = stage_tables(ctx, stage, a.lut);
    if (rc) return rc;
    const bool last = stage == ctx->stages;
    const int u = stage_u(ctx, stage);
    a.in = in; a.out = out;
    a.dbg = ctx->dbg;
    a.in_padded = (in.p == ctx->ws[0] || in.p == ctx->ws[1]) ? 1 : 0;
    a.N = N; a.C = C; a.H = H; a.W = W;
    a.oy0 = oy0; a.oy1 = oy1;
    a.M = ctx->n_modes;
    for (int m = 0; m < ctx->n_modes; ++m)
        for (int k = 0; k < 3; ++k) {
            a.di[m][k] = ctx->di[m][k];
            a.dj[m][k] = ctx->dj[m][k];
        }
    a.div = make_div_magic((uint32_t)stage_divisor(ctx->n_modes, last));
    a.bias_num = stage_bias_num(ctx->n_modes, last);
    a.inv_d = 1.0f / (float)a.div.d;
    a.use_f32 = ctx->f32_ok[last ? 1 : 0];
    a.epi_c = last ? ctx->epi_c : 127.0f;
    a.use_fma = last ? ctx->fma_ok : ctx->fma1_ok;
    // u == 4: the LDS kernels (tube bands resident) for up to 3 modes, and for longer lists that stage_tube2_kernel takes as a multiset of
    // its three patterns; final_kernel 5 = on every tile, 0 / 6 = hybrid with the per-tile statistic
    const bool tube = u == 4 && ctx->final_kernel != 1 && (ctx->n_modes <= 3 || (ctx->tube2 && C <= 3 && stage_tube2_supported(a)));
    const bool hybrid = tube && ctx->final_kernel != 5;
    a.verdict = nullptr;
    a.verdict_take = -1;
    int tw, th;
    if (u == 1) stage_u1_tile(tw, th); else if (tube) stage_band_tile(tw, th); else stage_up_tile(tw, th);
    a.tiles_x = (W + tw - 1) / tw;
    a.tiles_y = (oy1 - oy0 + th - 1) / th;
    if (u == 1 && last) a.use_fma = 0;     // (a final stage with 1-byte rows -- scale 1 -- takes the integer epilogue)
    // marks of the first-stage launch that produced this stage's input (same buffer, same shape); consumed here, never kept
    const bool k1_marks = ctx->k1_valid && ctx->k1_out == k1_ref && k1_n0 + N <= ctx->k1_N && ctx->k1_W == W && ctx->k1_H == H && ctx->k1_oy0 <= oy0 &&
                          oy1 <= ctx->k1_oy1;      // the marks cover exactly the images and rows that launch wrote: never index past its tile grid
    ctx->k1_valid = false;
    if (u == 1) {
        // (any mode list: the bands live in LDS per PATTERN, a repeated pattern is simply computed again into the int32 sum)
        const bool tube1 = (ctx->first_kernel == 0 || ctx->first_kernel == 3) &&
                           (unsigned long long)N * C * H * W < (1ull << 32);
        if (!tube1) {
            MAIN_KERNEL(ctx, stage, st, launch_stage_u1(a, st, 0));
            return MULUT_OK;
        }
        // tube kernel on the smooth tiles; the sites it flags are recomputed from the full tables, the tiles it leaves go
        // to the full-table kernel -- both through device-side lists (no host synchronisation, hipGraph-capturable)
        rc = ensure_fix(ctx, (size_t)N * C * (oy1 - oy0) * W);
        if (rc) return rc;
        rc = ensure_tlist(ctx, (size_t)N * a.tiles_x * a.tiles_y);
        if (rc) return rc;
        BandArgs b1;
        for (int m = 0; m < ctx->n_modes; ++m) b1.band[m] = ctx->tab[stage - 1][pattern_id(ctx->modes[m])].tube;
        a.fix_count = ctx->fix;
        a.fix_list = ctx->fix + 16;
        HIP_TRY(ctx, hipMemsetAsync(ctx->fix, 0, sizeof(uint32_t), st));
        if (ctx->first_kernel == 0) HIP_TRY(ctx, hipMemsetAsync(ctx->tlist, 0, (16 + (size_t)N * a.tiles_x * a.tiles_y) * sizeof(uint32_t), st));
        const bool route = ctx->first_kernel == 0;
        a.tile_count = ctx->tlist;
        a.tile_list = ctx->tlist + 16;
        a.verdict_take = route ? 0 : -1;
        MAIN_KERNEL(ctx, stage, st, launch_stage_u1t(a, b1, (unsigned)ctx->u1_detail_per_1024, ctx->num_cus, ctx->u1t_persist, st));
        if (route) HIP_TRY(ctx, launch_stage_u1w_list(a, ctx->num_cus, st));
        HIP_TRY(ctx, launch_stage_u1_fix(a, ctx->num_cus, st));
        ctx->k1_valid = route;
        ctx->k1_N = N; ctx->k1_W = W; ctx->k1_tiles_x = a.tiles_x; ctx->k1_tiles_y = a.tiles_y; ctx->k1_oy0 = oy0; ctx->k1_oy1 = oy1; ctx->k1_H = H;
        ctx->k1_out = out.p;
        return MULUT_OK;
    }
    int mode = kOutGeneric;
    if (u == 4 && (out_layout == MULUT_LAYOUT_CHW || (C == 1 && packed_ok))) mode = kOutPlanarU4;
    else if (u == 4 && out_layout == MULUT_LAYOUT_HWC && C == 3 && packed_ok) mode = kOutPackedRGBU4;
    if ((u == 2 || u == 3) && ctx->final_kernel != 1 && (unsigned long long)N * C * H * W < (1ull << 32)) {
        // u == 2 / u == 3 final stage on the tube band (the 1-byte-row kernel family with 4- / 9-value rows); flagged sites recomputed from the full table
        rc = ensure_fix(ctx, (size_t)N * C * (oy1 - oy0) * W);
        if (rc) return rc;
        BandArgs b2;
        for (int m = 0; m < ctx->n_modes; ++m) b2.band[m] = ctx->tab[stage - 1][pattern_id(ctx->modes[m])].tube;
        a.fix_count = ctx->fix;
        a.fix_list = ctx->fix + 16;
        HIP_TRY(ctx, hipMemsetAsync(ctx->fix, 0, sizeof(uint32_t), st));
        int t2w, t2h;
        stage_u1t_tile(t2w, t2h);
        StageArgs g = a;                // the gather kernel's launch on the tiles the tube kernel leaves out (its own tiling)
        a.tiles_x = (W + t2w - 1) / t2w;
        a.tiles_y = (oy1 - oy0 + t2h - 1) / t2h;
        // final_kernel 5: the tube kernel on every tile; otherwise routed by the 1-byte-row family's local-detail statistic -- detailed
        // 64 x 64 tiles (where most sites would end up on the fix-up list) go to the gather kernel, through device-side marks
        const bool route = ctx->final_kernel != 5;
        if (route) {
            rc = ensure_tlist(ctx, (size_t)N * a.tiles_x * a.tiles_y);
            if (rc) return rc;
            HIP_TRY(ctx, hipMemsetAsync(ctx->tlist, 0, (16 + (size_t)N * a.tiles_x * a.tiles_y) * sizeof(uint32_t), st));
            a.tile_count = ctx->tlist;
            a.tile_list = ctx->tlist + 16;
            a.verdict_take = 0;
        }
        if (u == 2) MAIN_KERNEL(ctx, stage, st, launch_stage_u2t(a, b2, (unsigned)ctx->up_detail_per_1024, ctx->num_cus, ctx->u1t_persist, st));
        else MAIN_KERNEL(ctx, stage, st, launch_stage_u3t(a, b2, (unsigned)ctx->up_detail_per_1024, ctx->num_cus, ctx->u1t_persist, st));
        if (route) {
            g.tile_list = a.tile_list;
            HIP_TRY(ctx, launch_stage_up(g, u, kOutGeneric, st));
        }
        return MULUT_OK;
    }
    if (!tube) {
        if (u == 4 && ctx->n_modes > 4) MAIN_KERNEL(ctx, stage, st, launch_stage_up_wide4(a, st));   // merged 16-bit fields hold 4 modes at most
        else MAIN_KERNEL(ctx, stage, st, launch_stage_up(a, u, mode, st));
        return MULUT_OK;
    }
    // every sample of the launch may end up on the fix-up list (entries: 30-bit pixel id + channel)
    if ((unsigned long long)N * H * W >= (1ull << 30)) return MULUT_EUNSUPPORTED;
    rc = ensure_fix(ctx, (size_t)N * (oy1 - oy0) * W * 3);
    if (rc) return rc;
    a.fix_count = ctx->fix;
    a.fix_list = ctx->fix + 16;
    HIP_TRY(ctx, hipMemsetAsync(ctx->fix, 0, sizeof(uint32_t), st));
    BandArgs b;
    for (int m = 0; m < ctx->n_modes; ++m) b.band[m] = ctx->tab[stage - 1][pattern_id(ctx->modes[m])].tube;
    if (!hybrid) {
        MAIN_KERNEL(ctx, stage, st, tube_launch(ctx, a, b, mode, st));
        HIP_TRY(ctx, launch_stage_up_fix(a, mode, ctx->num_cus, st, ctx->fix_variant));
        return MULUT_OK;
    }
    // per-tile choice on the device: smooth tiles -> tube kernel, detailed tiles -> anchor slabs in LDS (samples grouped by anchor
    // MSB), or the full-table gather kernel where that path does not apply
    rc = ensure_verdict(ctx, (size_t)N * a.tiles_x * a.tiles_y);
    if (rc) return rc;
    const bool slab = ctx->detail_kernel == 0 && detail_slab_supported(a);
    if (slab) {
        rc = ensure_detail(ctx, (size_t)N * a.tiles_x * a.tiles_y, detail_items_max(a), detail_ids_count(a), detail_blocks_count(a));
        if (rc) return rc;
        // the control words of the detailed-tile path are cleared before the statistic: it raises ctl[kDetAny] when it marks a tile
        HIP_TRY(ctx, hipMemsetAsync(ctx->det_ctl, 0, kDetCtlDwords * sizeof(uint32_t), st));
    }
    if (ctx->stat_from_k1 && k1_marks) {
        a.k1_hdr = ctx->tlist;
        a.k1_tiles_x = ctx->k1_tiles_x; a.k1_tiles_y = ctx->k1_tiles_y; a.k1_oy0 = ctx->k1_oy0; a.k1_n0 = k1_n0;
    }
    HIP_TRY(ctx, launch_tile_stat(a, ctx->verdict, (uint32_t)ctx->hybrid_oob_per_1024, st, slab ? ctx->det_thist : nullptr, slab ? ctx->det_ctl + kDetAny : nullptr));
    a.k1_hdr = nullptr;
    a.verdict = ctx->verdict;
    a.vt_x = a.tiles_x;
    a.vt_y = a.tiles_y;
    a.verdict_take = 0;
    MAIN_KERNEL(ctx, stage, st, tube_launch(ctx, a, b, mode, st));
    if (slab) {
        DetailArgs d;
        memset(&d, 0, sizeof(d));
        d.ctl = ctx->det_ctl; d.items = ctx->det_items; d.desc = ctx->det_desc; d.blocks = ctx->det_blocks;
        d.thist = ctx->det_thist; d.tpos = ctx->det_tpos; d.dlist = ctx->det_dlist;
        for (int m = 0; m < 3; ++m) d.slab[m] = m < ctx->n_modes ? ctx->tab[stage - 1][pattern_id(ctx->modes[m])].slab : nullptr;
        HIP_TRY(ctx, launch_detail_slab(a, d, mode, ctx->num_cus, st));
    } else {
        StageArgs g = a;
        int gw, gh;
        stage_up_tile(gw, gh);
        g.tiles_x = (W + gw - 1) / gw;
        g.tiles_y = (oy1 - oy0 + gh - 1) / gh;
        g.verdict_take = 1;
        if (ctx->n_modes > 4) HIP_TRY(ctx, launch_stage_up_wide4(g, st));      // (merged 16-bit fields hold 4 modes at most)
        else HIP_TRY(ctx, launch_stage_up(g, u, mode, st));
    }
    HIP_TRY(ctx, launch_stage_up_fix(a, mode, ctx->num_cus, st, ctx->fix_variant));
    return MULUT_OK;
}

// Index widths of the device work lists, per launch: site ids of the 1-byte-row / u == 2 tube kernels 32 bits (N C H W), pixel ids of the
// u == 4 fix-up list 30 bits (N H W), sample descriptors of the detailed-tile path 28 bits of byte offset into the stage input.  Images
// are independent (sr/4_test_lut.py:257-259 fans them out one by one), so a launch beyond a width runs as sub-launches of whole images
// that fit -- every entry point (mulut_stage, mulut_pipeline_rows, mulut_pipeline) comes through here, none falls to a slower kernel
// because of its batch size.  How many images of the launch fit one sub-launch (>= 1; N when nothing binds):
static int stage_fit_images(const mulut_ctx *ctx, int stage, const View &in, int N, int H, int W, int C) {
    const int u = stage_u(ctx, stage);
    unsigned long long fit = (unsigned long long)N;
    auto cap = [&](unsigned long long limit, unsigned long long per_image) {
        const unsigned long long f = per_image ? (limit - 1) / per_image : fit;
        if (f < fit) fit = f;
    };
    if (u == 1 || u == 2 || u == 3) cap(1ull << 32, (unsigned long long)C * H * W);
    if (u == 4 && ctx->final_kernel != 1) {
        cap(1ull << 30, (unsigned long long)H * W);
        if (ctx->final_kernel != 5 && ctx->detail_kernel == 0 && ctx->n_modes <= 3 && in.sX == 1)      // (what detail_slab_supported() asks of a launch)
            cap(1ull << 28, (unsigned long long)(in.sN < 0 ? -in.sN : in.sN));
    }
    return fit < 1 ? 1 : (int)fit;
}

static int run_stage(mulut_ctx *ctx, int stage, const View &in, const View &out, int out_layout, int N, int H, int W,
                     int C, int oy0, int oy1, hipStream_t st, bool packed_ok = true) {
    const int fit = stage_fit_images(ctx, stage, in, N, H, W, C);
    if (N <= fit) return run_stage_one(ctx, stage, in, out, out_layout, N, H, W, C, oy0, oy1, st, packed_ok, in.p, 0);
    // the first stage's tile marks (if this stage reads what it wrote) serve every sub-launch: kept across the calls that consume them
    const bool k1_valid = ctx->k1_valid;
    for (int n0 = 0; n0 < N; n0 += fit) {
        View vin = in, vout = out;
        vin.p += (long long)n0 * in.sN;
        vout.p += (long long)n0 * out.sN;
        ctx->k1_valid = k1_valid;
        const int rc = run_stage_one(ctx, stage, vin, vout, out_layout, imin(fit, N - n0), H, W, C, oy0, oy1, st, packed_ok, in.p, n0);
        if (rc) return rc;
    }
    // a split first stage leaves the marks of its last sub-launch only: the next stage must look at every tile itself
    ctx->k1_valid = false;
    return MULUT_OK;
}

int mulut_halo(const mulut_ctx *ctx) { return (ctx && ctx->configured) ? ctx->reach * ctx->stages : 0; }

static int ensure_workspace(mulut_ctx *ctx, size_t bytes) {
    if (bytes <= ctx->ws_bytes) return MULUT_OK;
    for (auto &w : ctx->ws) {
        if (w) HIP_TRY(ctx, hipFree(w));
        w = nullptr;
    }
    ctx->ws_bytes = 0;
    for (auto &w : ctx->ws) HIP_TRY(ctx, hipMalloc((void **)&w, bytes + 64));      // + padding: kernels may read whole dwords / 8 bytes at the very end (StageArgs::in_padded)
    ctx->ws_bytes = bytes;
    return MULUT_OK;
}

static int ensure_fix(mulut_ctx *ctx, size_t ids) {
    if (ids <= ctx->fix_cap) return MULUT_OK;
    if (ctx->fix) HIP_TRY(ctx, hipFree(ctx->fix));
    ctx->fix = nullptr;
    ctx->fix_cap = 0;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->fix, (ids + 16) * sizeof(uint32_t)));
    ctx->fix_cap = ids;
    return MULUT_OK;
}

static int ensure_tlist(mulut_ctx *ctx, size_t tiles) {
    if (tiles <= ctx->tlist_cap) return MULUT_OK;
    if (ctx->tlist) HIP_TRY(ctx, hipFree(ctx->tlist));
    ctx->tlist = nullptr;
    ctx->tlist_cap = 0;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->tlist, (tiles + 16) * sizeof(uint32_t)));
    ctx->tlist_cap = tiles;
    return MULUT_OK;
}

static int ensure_detail(mulut_ctx *ctx, size_t tiles, size_t items, size_t ids, size_t blocks) {
    if (!ctx->det_ctl) {
        HIP_TRY(ctx, hipMalloc((void **)&ctx->det_ctl, kDetCtlDwords * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMemset(ctx->det_ctl, 0, kDetCtlDwords * sizeof(uint32_t)));
    }
    if (tiles > ctx->det_tiles_cap) {
        if (ctx->det_thist) HIP_TRY(ctx, hipFree(ctx->det_thist));
        if (ctx->det_dlist) HIP_TRY(ctx, hipFree(ctx->det_dlist));
        if (ctx->det_tpos) HIP_TRY(ctx, hipFree(ctx->det_tpos));
        ctx->det_thist = nullptr;
        ctx->det_dlist = ctx->det_tpos = nullptr;
        ctx->det_tiles_cap = 0;
        HIP_TRY(ctx, hipMalloc((void **)&ctx->det_thist, tiles * 16 * sizeof(uint16_t)));
        HIP_TRY(ctx, hipMalloc((void **)&ctx->det_tpos, tiles * 16 * sizeof(uint32_t)));
        HIP_TRY(ctx, hipMalloc((void **)&ctx->det_dlist, tiles * sizeof(uint32_t)));
        ctx->det_tiles_cap = tiles;
    }
    if (items > ctx->det_items_cap) {
        if (ctx->det_items) HIP_TRY(ctx, hipFree(ctx->det_items));
        ctx->det_items = nullptr;
        ctx->det_items_cap = 0;
        HIP_TRY(ctx, hipMalloc((void **)&ctx->det_items, items * 2 * sizeof(uint32_t)));
        ctx->det_items_cap = items;
    }
    if (ids > ctx->det_ids_cap) {
        if (ctx->det_desc) HIP_TRY(ctx, hipFree(ctx->det_desc));
        ctx->det_desc = nullptr;
        ctx->det_ids_cap = 0;
        HIP_TRY(ctx, hipMalloc((void **)&ctx->det_desc, ids * sizeof(uint32_t)));
        ctx->det_ids_cap = ids;
    }
    if (blocks > ctx->det_blocks_cap) {
        if (ctx->det_blocks) HIP_TRY(ctx, hipFree(ctx->det_blocks));
        ctx->det_blocks = nullptr;
        ctx->det_blocks_cap = 0;
        HIP_TRY(ctx, hipMalloc((void **)&ctx->det_blocks, blocks * sizeof(uint4)));
        ctx->det_blocks_cap = blocks;
    }
    return MULUT_OK;
}

static int ensure_verdict(mulut_ctx *ctx, size_t tiles) {
    if (tiles <= ctx->verdict_tiles) return MULUT_OK;
    if (ctx->verdict) HIP_TRY(ctx, hipFree(ctx->verdict));
    ctx->verdict = nullptr;
    ctx->verdict_tiles = 0;
    HIP_TRY(ctx, hipMalloc((void **)&ctx->verdict, tiles * sizeof(uint32_t)));
    ctx->verdict_tiles = tiles;
    return MULUT_OK;
}

int mulut_reserve(mulut_ctx *ctx, int N, int H, int W, int C) {
    if (!ctx || N <= 0 || H <= 0 || W <= 0 || C <= 0) return MULUT_EINVAL;
    if (!ctx->configured) return MULUT_ENOTCONFIGURED;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    {
        int tw, th;
        stage_band_tile(tw, th);
        int rc = ensure_verdict(ctx, (size_t)N * ((W + tw - 1) / tw) * ((H + th - 1) / th));
        if (rc) return rc;
        {
            const bool u1 = ctx->stages > 1 || ctx->scale == 1;
            rc = ensure_fix(ctx, (size_t)N * H * W * ((u1 || ctx->scale == 2 || ctx->scale == 3 || ctx->scale == 4) ? (size_t)(C > 3 ? C : 3) : 1));
            if (rc) return rc;
            if (ctx->scale == 4) {
                stage_band_tile(tw, th);
                StageArgs t;
                memset(&t, 0, sizeof(t));
                // the final stage of a batch beyond the 28-bit sample descriptors runs as sub-launches (run_stage): the detailed-tile path's
                // buffers are sized for the largest of those, so that a captured call never allocates
                t.in.sN = (long long)H * W * imin(C, 3);       // (a cascade's final stage reads the planar workspace: groups of <= 3 channels)
                t.in.sX = 1; t.C = imin(C, 3); t.M = ctx->n_modes; t.H = H; t.W = W;
                t.N = imin(N, stage_fit_images(ctx, ctx->stages, t.in, N, H, W, t.C));
                t.tiles_x = (W + tw - 1) / tw; t.tiles_y = (H + th - 1) / th;
                if (detail_slab_supported(t)) {       // launches of this size that the anchor-slab path would take
                    rc = ensure_detail(ctx, (size_t)t.N * t.tiles_x * t.tiles_y, detail_items_max(t), detail_ids_count(t), detail_blocks_count(t));
                    if (rc) return rc;
                }
            }
            if (u1 || ctx->scale == 2 || ctx->scale == 3) {      // tile marks of the routed 1-byte-row family (first stages; x2 / x3 final stages)
                stage_u1_tile(tw, th);
                rc = ensure_tlist(ctx, (size_t)N * ((W + tw - 1) / tw) * ((H + th - 1) / th));
                if (rc) return rc;
            }
        }
    }
    if (ctx->stages < 2) return MULUT_OK;
    return ensure_workspace(ctx, (size_t)N * H * W * C);
}

int mulut_stage(mulut_ctx *ctx, int stage, const uint8_t *in, int in_layout, uint8_t *out, int out_layout, int N,
                int H, int W, int C, void *stream) {
    if (!ctx || !in || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0) return MULUT_EINVAL;
    if (!ctx->configured) return MULUT_ENOTCONFIGURED;
    if (stage < 1 || stage > ctx->stages) return MULUT_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int u = stage_u(ctx, stage);
    // channels are independent planes through the same tables (sr/4_test_lut.py:14-237 is channel-count agnostic): more than three
    // run in groups of three, each group a view into the caller's buffers
    for (int c0 = 0; c0 < C; c0 += 3) {
        View vin = make_view(in, in_layout, H, W, C, 0), vout = make_view(out, out_layout, H * u, W * u, C, 0);
        vin.p += (long long)c0 * vin.sC;
        vout.p += (long long)c0 * vout.sC;
        const int rc = run_stage(ctx, stage, vin, vout, out_layout, N, H, W, imin(3, C - c0), 0, H, (hipStream_t)stream, C <= 3);
        if (rc) return rc;
    }
    return MULUT_OK;
}

int mulut_pipeline_rows(mulut_ctx *ctx, const uint8_t *in, int in_row0, int in_rows, uint8_t *out, int y0, int y1,
                        int N, int H, int W, int C, int layout, void *stream) {
    if (!ctx || !in || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0) return MULUT_EINVAL;
    if (!ctx->configured) return MULUT_ENOTCONFIGURED;
    if (y0 < 0 || y1 > H || y0 >= y1 || in_row0 < 0 || in_rows <= 0 || in_row0 + in_rows > H) return MULUT_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int S = ctx->stages, reach = ctx->reach;
    // rows of each stage's output that the cascade needs
    int lo[MULUT_MAX_STAGES + 1], hi[MULUT_MAX_STAGES + 1];
    for (int s = 1; s <= S; ++s) {
        lo[s] = imax(0, y0 - reach * (S - s));
        hi[s] = imin(H, y1 + reach * (S - s));
    }
    // the caller's band must cover stage 1's reads
    if (in_row0 > imax(0, lo[1] - reach) || in_row0 + in_rows < imin(H, hi[1] + reach)) return MULUT_EWORKSPACE;
    if (S > 1) {
        size_t need = 0;
        for (int s = 1; s < S; ++s) {
            const size_t b = (size_t)N * imin(C, 3) * (hi[s] - lo[s]) * W;
            if (b > need) need = b;
        }
        int rc = ensure_workspace(ctx, need);
        if (rc) return rc;
    }
    ctx->timed_stages = 0;
    // channels are independent planes through the same tables (the reference function is channel-count agnostic): more than three
    // run as groups of three, each a view into the caller's buffers (stage timing: the last group's)
    for (int c0 = 0; c0 < C; c0 += 3) {
        const int Cg = imin(3, C - c0);
        View cur = make_view(in, layout, in_rows, W, C, in_row0);
        cur.p += (long long)c0 * cur.sC;
        if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev[0], (hipStream_t)stream));
        for (int s = 1; s <= S; ++s) {
            const int u = stage_u(ctx, s);
            View dst;
            int dst_layout;
            if (s == S) {
                dst = make_view(out, layout, (y1 - y0) * u, W * u, C, y0 * u);
                dst.p += (long long)c0 * dst.sC;
                dst_layout = layout;
            } else {
                dst = make_view(ctx->ws[s & 1], MULUT_LAYOUT_CHW, hi[s] - lo[s], W, Cg, lo[s]);
                dst_layout = MULUT_LAYOUT_CHW;
            }
            int rc = run_stage(ctx, s, cur, dst, dst_layout, N, H, W, Cg, lo[s], hi[s], (hipStream_t)stream, C <= 3);
            if (rc) return rc;
            if (ctx->timing) HIP_TRY(ctx, hipEventRecord(ctx->ev[s], (hipStream_t)stream));
            cur = dst;
        }
    }
    if (ctx->timing) ctx->timed_stages = S;
    return MULUT_OK;
}

int mulut_pipeline(mulut_ctx *ctx, const uint8_t *in, uint8_t *out, int N, int H, int W, int C, int layout,
                   void *stream) {
    if (!ctx || !in || !out || N <= 0 || H <= 0 || W <= 0 || C <= 0) return MULUT_EINVAL;
    if (!ctx->configured) return MULUT_ENOTCONFIGURED;
    // (batches beyond the index widths of the device work lists run as sub-launches per stage: run_stage)
    return mulut_pipeline_rows(ctx, in, 0, H, out, 0, H, N, H, W, C, layout, stream);
}

int mulut_set_stage_timing(mulut_ctx *ctx, int enable) {
    if (!ctx) return MULUT_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (enable) {
        for (auto &e : ctx->ev)
            if (!e) HIP_TRY(ctx, hipEventCreate(&e));
        for (auto &p : ctx->evk)
            for (auto &e : p)
                if (!e) HIP_TRY(ctx, hipEventCreate(&e));
    }
    for (auto &f : ctx->evk_set) f = false;
    ctx->timing = enable != 0;
    ctx->timed_stages = 0;
    return MULUT_OK;
}

int mulut_last_stage_ms(mulut_ctx *ctx, float *ms, int cap) {
    if (!ctx || !ms || cap <= 0) return MULUT_EINVAL;
    const int n = ctx->timed_stages < cap ? ctx->timed_stages : cap;
    if (n > 0) HIP_TRY(ctx, hipEventSynchronize(ctx->ev[ctx->timed_stages]));
    for (int s = 0; s < n; ++s) HIP_TRY(ctx, hipEventElapsedTime(&ms[s], ctx->ev[s], ctx->ev[s + 1]));
    return n;
}

int mulut_last_kernel_ms(mulut_ctx *ctx, float *ms, int cap) {
    if (!ctx || !ms || cap <= 0) return MULUT_EINVAL;
    const int n = ctx->timed_stages < cap ? ctx->timed_stages : cap;
    if (n > 0) HIP_TRY(ctx, hipEventSynchronize(ctx->ev[ctx->timed_stages]));
    for (int s = 0; s < n; ++s) {
        ms[s] = 0.0f;
        if (ctx->evk_set[s]) HIP_TRY(ctx, hipEventElapsedTime(&ms[s], ctx->evk[s][0], ctx->evk[s][1]));
    }
    return n;
}

int mulut_last_detail_counters(mulut_ctx *ctx, uint32_t *out, int cap, void *stream) {
    if (!ctx || !out || cap <= 0) return MULUT_EINVAL;
    if (!ctx->det_ctl || !ctx->fix) return 0;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    uint32_t ctl[kDetCtlDwords], fixn = 0;
    HIP_TRY(ctx, hipMemcpyAsync(ctl, ctx->det_ctl, sizeof(ctl), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(ctx, hipMemcpyAsync(&fixn, ctx->fix, sizeof(fixn), hipMemcpyDeviceToHost, (hipStream_t)stream));
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    int n = 0;
    for (int k = 0; k < 16 && n < cap; ++k) out[n++] = ctl[k];
    if (n < cap) out[n++] = ctl[kDetItems];
    if (n < cap) out[n++] = fixn;
    for (int k = 48; k < 56 && n < cap; ++k) out[n++] = ctl[k];      // phase clocks of the slabclk probe build (zero otherwise)
    return n;
}

int mulut_debug_read(mulut_ctx *ctx, unsigned long long *out, int cap, int reset, void *stream) {
    if (!ctx || (cap > 0 && !out) || cap < 0) return MULUT_EINVAL;
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!ctx->dbg) {
        HIP_TRY(ctx, hipMalloc((void **)&ctx->dbg, MULUT_DEBUG_WORDS * sizeof(unsigned long long)));
        HIP_TRY(ctx, hipMemset(ctx->dbg, 0, MULUT_DEBUG_WORDS * sizeof(unsigned long long)));
    }
    const int n = cap < MULUT_DEBUG_WORDS ? cap : MULUT_DEBUG_WORDS;
    if (n > 0) HIP_TRY(ctx, hipMemcpyAsync(out, ctx->dbg, (size_t)n * sizeof(unsigned long long), hipMemcpyDeviceToHost, (hipStream_t)stream));
    if (reset) HIP_TRY(ctx, hipMemsetAsync(ctx->dbg, 0, MULUT_DEBUG_WORDS * sizeof(unsigned long long), (hipStream_t)stream));
    HIP_TRY(ctx, hipStreamSynchronize((hipStream_t)stream));
    return n;
}

int mulut_set_tuning(mulut_ctx *ctx, const char *key, int value) {
    if (!ctx || !key) return MULUT_EINVAL;
    if (!strcmp(key, "final_stage_kernel")) {
        if (value != 0 && value != 1 && value != 5 && value != 6) return MULUT_EINVAL;      // (2-4: generations retired in round 3)
        ctx->final_kernel = value;
        return MULUT_OK;
    }
    if (!strcmp(key, "first_stage_kernel")) {
        if (value != 0 && value != 2 && value != 3) return MULUT_EINVAL;      // (1: retired in round 3)
        ctx->first_kernel = value;
        return MULUT_OK;
    }
    if (!strcmp(key, "stat_from_first_stage")) {
        if (value < 0 || value > 1) return MULUT_EINVAL;
        ctx->stat_from_k1 = value;
        return MULUT_OK;
    }
    if (!strcmp(key, "fix_kernel")) {      // fix-up of the u == 4 tube kernels: 0 = one pass per lane, 1 = one entry per thread, 2 = one pass per lane with the list walk pipelined
        if (value < 0 || value > 2) return MULUT_EINVAL;
        ctx->fix_variant = value;
        return MULUT_OK;
    }
    if (!strcmp(key, "tube_pipelined")) {
        if (value < 0 || value > 1) return MULUT_EINVAL;
        ctx->tube2 = value;
        return MULUT_OK;
    }
    if (!strcmp(key, "detail_kernel")) {     // final-stage tiles the statistic marks detailed: 0 anchor slabs in LDS, 1 full-table gathers
        if (value < 0 || value > 1) return MULUT_EINVAL;
        ctx->detail_kernel = value;
        return MULUT_OK;
    }
    if (!strcmp(key, "u1t_persist")) {      // experiment: persistent workgroups per CU of the 1-byte-row tube kernel (0 = one per tile)
        if (value < 0 || value > 8) return MULUT_EINVAL;
        ctx->u1t_persist = value;
        return MULUT_OK;
    }
    if (!strcmp(key, "first_stage_detail_per_1024")) {
        if (value < 0 || value > 1024) return MULUT_EINVAL;
        ctx->u1_detail_per_1024 = value;
        return MULUT_OK;
    }
    if (!strcmp(key, "final_stage_detail_per_1024")) {
        if (value < 0 || value > 1024) return MULUT_EINVAL;
        ctx->up_detail_per_1024 = value;
        return MULUT_OK;
    }
    if (!strcmp(key, "hybrid_oob_per_1024")) {
        if (value < 0 || value > 1024) return MULUT_EINVAL;
        ctx->hybrid_oob_per_1024 = value;
        return MULUT_OK;
    }
    return MULUT_EINVAL;
}

const char *mulut_kernel_name(const mulut_ctx *ctx, int is_final) {
    if (!ctx || !ctx->configured) return "";
    if (!is_final || ctx->scale == 1) return stage_u1_name(ctx->first_kernel);
    if (ctx->scale == 2 && ctx->final_kernel != 1) return "stage_u1t_kernel<2> + stage_up_fix_site_kernel<2>";
    if (ctx->scale == 3 && ctx->final_kernel != 1) return "stage_u1t_kernel<3> + stage_up_fix_site_kernel<3>";
    // (as run_stage decides: the pipelined kernel takes every list that uses all of s, d, y, up to kMaxTube2Modes modes, when the float
    // epilogue is exact for the divisor)
    const bool all3 = strchr(ctx->modes, 's') && strchr(ctx->modes, 'd') && strchr(ctx->modes, 'y');
    const bool t2 = ctx->tube2 && all3 && ctx->n_modes <= kMaxTube2Modes && (ctx->n_modes > 4 || ctx->f32_ok[1]);
    if (ctx->scale == 4 && (ctx->n_modes <= 3 || t2) && ctx->final_kernel != 1) {
        if (ctx->final_kernel == 5) return t2 ? "stage_tube2_kernel<rgb> + stage_up_fix2_kernel" : "stage_tube_kernel<rgb> + stage_up_fix2_kernel";
        if (ctx->detail_kernel == 0)
            return t2 ? "hybrid: tile_stat_kernel + stage_tube2_kernel<rgb> (smooth tiles; hand-scheduled LDS pipeline, one 16x4 tile per wave) + stage_slab_kernel (detailed tiles, anchor slabs in LDS)"
                      : "hybrid: tile_stat_kernel + stage_tube_kernel<rgb> (smooth tiles) + stage_slab_kernel (detailed tiles, anchor slabs in LDS)";
        return t2 ? "hybrid: tile_stat_kernel + stage_tube2_kernel<rgb> (smooth tiles) + stage_up_kernel<4,rgb> (detailed tiles)"
                  : "hybrid: tile_stat_kernel + stage_tube_kernel<rgb> (smooth tiles) + stage_up_kernel<4,rgb> (detailed tiles)";
    }
    return stage_up_name(ctx->scale, ctx->scale == 4 ? kOutPackedRGBU4 : kOutGeneric);
}

}  // extern "C"
