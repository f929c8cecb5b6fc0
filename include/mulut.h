/*
 * mulut.h -- C ABI of libmulut_hip.so: MuLUT LUT inference (4D LUT retrieval + 4-simplex
 * interpolation over rotated s/d/y patches, cascaded stages) on AMD MI355X (gfx950).
 *
 * This is the drop-in boundary for the hot path of the reference's `sr/4_test_lut.py`.  The
 * reference is pure Python and has no FFI layer; each entry point below states which reference
 * lines it replaces (paths relative to the reference repo root), and INTEGRATION.md shows the
 * ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - plain C, no exceptions: every call returns MULUT_OK (0) or a negative MULUT_E* code;
 *     mulut_strerror() names it.
 *   - all image buffers are CALLER-OWNED DEVICE pointers (e.g. torch tensor data_ptr()); the
 *     library owns only its context: device copies of the tables and an intermediate-stage
 *     workspace.  LUT rows passed to mulut_set_lut() are HOST pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); all work is
 *     stream-ordered and asynchronous with respect to the host.
 *   - one context per device; calls on one context must be serialised by the caller.
 *   - there is NO CPU fallback: without a usable HIP device mulut_create() fails.
 *
 * Image layouts
 *   MULUT_LAYOUT_CHW : planar  [N][C][H][W]   (what FourSimplexInterpFaster receives, :296)
 *   MULUT_LAYOUT_HWC : packed  [N][H][W][C]   (what PIL / the driver loop holds, :265-270,:301)
 */
#ifndef MULUT_H_
#define MULUT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MULUT_VERSION 100 /* 0.1.0 */

enum {
    MULUT_OK = 0,
    MULUT_EINVAL = -1,      /* bad argument (NULL pointer, non-positive size, ...)             */
    MULUT_EMODE = -2,       /* mode not in {s,d,y}: reference raises ValueError, 4_test_lut.py:54 */
    MULUT_ENOLUT = -3,      /* table (stage,mode) not set: reference raises from np.load, :333  */
    MULUT_ESHAPE = -4,      /* table shape does not match (rows, v_num) expected for the stage */
    MULUT_EUNSUPPORTED = -5,/* interval != 4, scale not in 1..4, stages/modes beyond limits    */
    MULUT_EHIP = -6,        /* a HIP runtime call failed (mulut_last_hip_error() has the text) */
    MULUT_ENODEVICE = -7,   /* no usable gfx950 device: there is no CPU path                   */
    MULUT_ENOTCONFIGURED = -8,
    MULUT_EWORKSPACE = -9   /* strip/halo bookkeeping inconsistent with the buffers given     */
};

enum { MULUT_LAYOUT_CHW = 0, MULUT_LAYOUT_HWC = 1 };

#define MULUT_MAX_STAGES 8
#define MULUT_MAX_MODES 8

typedef struct mulut_ctx mulut_ctx;

int mulut_version(void);
const char *mulut_strerror(int err);
/* text of the last failing HIP call on this context (empty string if none) */
const char *mulut_last_hip_error(const mulut_ctx *ctx);

/* Bind a context to HIP device `device_id`. */
int mulut_create(int device_id, mulut_ctx **out_ctx);
int mulut_destroy(mulut_ctx *ctx);

/* The model shape: replaces the options the reference reads from TestOptions
 * (common/option.py:21-23,17: --stages --modes --interval --scale) at sr/4_test_lut.py:279-287.
 * `modes` is a NUL-terminated string iterated character-wise exactly like `opt.modes` (:287).
 * Only interval == 4 (q=16, L=17, 83521 rows) is supported -- the only value for which the
 * reference's reader and writers agree on file names. */
int mulut_configure(mulut_ctx *ctx, int stages, const char *modes, int scale, int interval);

/* Upload one table: replaces
 *   lutDict["s{stage}_{mode}"] = np.load(path).astype(np.float32).reshape(-1, v_num)
 * (sr/4_test_lut.py:322-333).  `host_rows` is the int8 C-order content of the .npy file,
 * rows = 17^4 = 83521, vnum = scale*scale for the last stage, 1 otherwise.  stage is 1-based. */
int mulut_set_lut(mulut_ctx *ctx, int stage, char mode, const int8_t *host_rows, int64_t rows, int vnum);

/* One (table, mode, rotation) pass: replaces FourSimplexInterpFaster(weight, img_in, h, w,
 * interval, rot=4-r, upscale, mode) (sr/4_test_lut.py:14-237) TOGETHER WITH the caller's
 * np.rot90(img, r) + edge pad (:294-296).  in_chw: device uint8 planar [C][H][W], un-rotated,
 * un-padded.  out_q: device int32 planar [C][H*u][W*u] = q * (the float64 array the reference
 * returns), exact.  u = scale if `stage` is the last configured stage, else 1. */
int mulut_pass(mulut_ctx *ctx, int stage, char mode, int r, const uint8_t *in_chw, int H, int W, int C,
               int32_t *out_q, void *stream);

/* One whole stage (all modes x 4 rotations, average, +bias, round-half-even, clip): replaces one
 * iteration of the `for s in range(stages)` body (sr/4_test_lut.py:280-306).  N images.
 * Any C >= 1 (here and in mulut_pipeline / mulut_pipeline_rows): the kernels take up to three channels per
 * launch, images with more are run as groups of three (channels are independent in the reference too). */
int mulut_stage(mulut_ctx *ctx, int stage, const uint8_t *in, int in_layout, uint8_t *out, int out_layout, int N,
                int H, int W, int C, void *stream);

/* The whole cascade for N images: replaces sr/4_test_lut.py:279-306 (= sr/5_test_lut.py:271-307).
 * in: N x (H,W,C) uint8, out: N x (H*scale, W*scale, C) uint8, both in `layout`.
 * Batch size: the device work lists index one launch with fixed widths (28 bits of byte offset into the stage input on the
 * detailed-tile path of x4 final stages, 30-bit pixel ids, 32-bit site ids).  Every entry point -- this one, mulut_pipeline_rows,
 * mulut_stage -- runs a STAGE whose launch would exceed a width as sub-launches of whole images that fit (43 frames of 1080p RGB
 * per final-stage launch): same result, same kernels; no batch size falls to a slower path.  (The timing helpers then report the
 * last sub-launch of a stage.)  A single image beyond 2^28 bytes takes the gather kernels for its detailed tiles. */
int mulut_pipeline(mulut_ctx *ctx, const uint8_t *in, uint8_t *out, int N, int H, int W, int C, int layout,
                   void *stream);

/* Strip form for tile sharding (one strip per GPU).  The logical image is H_full x W; `in` holds
 * its rows [in_row0, in_row0 + in_rows) and `out` receives output rows for LR rows [y0, y1), i.e.
 * HR rows [y0*scale, y1*scale), stored from row 0 of `out`.  The input must cover the halo
 * [y0 - mulut_halo(ctx), y1 + mulut_halo(ctx)) clipped to the image; edge replication happens only
 * at true image borders, so strips tile bit-exactly. */
int mulut_pipeline_rows(mulut_ctx *ctx, const uint8_t *in, int in_row0, int in_rows, uint8_t *out, int y0, int y1,
                        int N, int H_full, int W, int C, int layout, void *stream);

/* LR rows of context needed above/below a strip for the configured cascade (2 per stage: the
 * reach of the d / y patterns over the four rotations). */
int mulut_halo(const mulut_ctx *ctx);

/* Pre-size the intermediate-stage workspace and every device work list so that later pipeline calls of at most this shape
 * allocate nothing (required before capturing a pipeline call into a hipGraph).  Buffers that only a stage's sub-launches
 * use (see mulut_pipeline, "Batch size") are sized for the largest sub-launch. */
int mulut_reserve(mulut_ctx *ctx, int N, int H, int W, int C);

/* Per-stage device timing for bench.py's roofline leg: when enabled, every mulut_pipeline[_rows]
 * call brackets each stage's kernel launch with hipEvents recorded on the caller's stream;
 * mulut_last_stage_ms() waits for the last call's events and returns the elapsed milliseconds of
 * each stage (n = number of stages written, <= cap).  Off by default (events cost a few us). */
int mulut_set_stage_timing(mulut_ctx *ctx, int enable);
int mulut_last_stage_ms(mulut_ctx *ctx, float *ms, int cap);
/* The same call's milliseconds of each stage's DOMINANT kernel alone (the stage's helper launches -- tile statistic,
 * fix-up lists, the kernel that takes the detailed tiles -- are in mulut_last_stage_ms only): the launch duration
 * bench.py prices against the roofline. */
int mulut_last_kernel_ms(mulut_ctx *ctx, float *ms, int cap);
/* Work counters of the detailed-tile path of the last final-stage launch (scale 4; device -> host copy, synchronises with
 * `stream`): out[0..15] = samples (pixel x channel) per anchor MSB that went through the anchor-slab kernel,
 * out[16] = work items, out[17] = entries (samples, or border pixels with all their channels) on the fix-up list of that launch.  Returns the number of values written
 * (0 when the path has not run).  For tests and the bench report; the reference has no counterpart. */
int mulut_last_detail_counters(mulut_ctx *ctx, uint32_t *out, int cap, void *stream);
/* Probe buffer of the context: MULUT_DEBUG_WORDS 64-bit words of device memory that only probe builds of the kernels
 * (-DMULUT_VARIANT_...prof: in-kernel clock stamps per phase) write to -- never an output buffer, and no output value is computed
 * from it.  Copies min(cap, MULUT_DEBUG_WORDS) words to `out` (synchronises with `stream`), then zeroes the buffer when
 * `reset` != 0.  Returns the number of words copied.  The shipped library leaves the buffer at zero. */
#define MULUT_DEBUG_WORDS 4096
int mulut_debug_read(mulut_ctx *ctx, unsigned long long *out, int cap, int reset, void *stream);

/* ---- LUT-aware fine-tuning (the differentiable twin; stateless, float32) -----------------------------
 * One stage of MuLUT.forward (sr/model.py:289-312) = InterpTorchBatch (:69-287) over all modes x 4 rotations
 * with the per-pass BPDA rounding (:308) and the stage's clamp/round (:309).
 *   weights_q : M device pointers, the QUANTISED tables clamp(round(w*127),-127,127) as float32 [83521][u*u]
 *               (sr/model.py:74-76 -- done by the caller, which also applies that step's backward)
 *   x         : device float32 [B][C][H][W] in 0..255 (the module multiplies its input by 255, :290)
 *   out       : device float32 [B][C][H*u][W*u] in 0..255
 * backward: grad_out = dL/d out; accumulates dL/d weights_q into grad_wq[m] (atomic adds; zero them first) and
 * dL/dx into grad_x (zero it first).  u = upscale for the last stage, else 1. */
/* The module's quantisation of its float parameters and that step's backward (sr/model.py:74-76), for the M tables of a stage
 * (n floats each) in one launch:
 *   mulut_ft_quantize          : weights_q[m][i] = clamp(round(weights[m][i] * 127), -127, 127), round = half to even (torch.round)
 *   mulut_ft_quantize_backward : grad[m][i] = grad[m][i] * inside * 127 in place, inside = round(weights[m][i] * 127) within [-127, 127]
 *                                (the rounding is a BPDA identity, :59-67; the clamp passes gradient on its closed interval) */
int mulut_ft_quantize(int device, const float *const *weights, float *const *weights_q, int M, long long n, void *stream);
int mulut_ft_quantize_backward(int device, const float *const *weights, float *const *grad, int M, long long n, void *stream);
int mulut_ft_stage_forward(int device, const float *const *weights_q, const char *modes, int is_last, int u, const float *x,
                           int B, int C, int H, int W, float *out, void *stream);
int mulut_ft_stage_backward(int device, const float *const *weights_q, const char *modes, int is_last, int u, const float *x,
                            const float *grad_out, int B, int C, int H, int W, float *const *grad_wq, float *grad_x,
                            void *stream);
/* The same pair with the clamp's mask handed from the forward to the backward: inside[(b*C + c)*H*W + y*W + x] bit sy*u+sx is set where
 * 0 <= pred/avg + bias <= 255 at that block position (:309, the closed interval on which the clamp passes gradient).  The plain backward
 * recomputes the stage forward to get it; with the mask it does not (one sixth of the final-stage backward at bs 256 x 48 x 48). */
int mulut_ft_stage_forward_mask(int device, const float *const *weights_q, const char *modes, int is_last, int u, const float *x,
                                int B, int C, int H, int W, float *out, unsigned short *inside, void *stream);
int mulut_ft_stage_backward_mask(int device, const float *const *weights_q, const char *modes, int is_last, int u, const float *x,
                                 const float *grad_out, const unsigned short *inside, int B, int C, int H, int W,
                                 float *const *grad_wq, float *grad_x, void *stream);

/* ---- device-side evaluation (not on the inference path) ---------------------------------------------------
 * Y-channel PSNR and SSIM of a super-resolved frame against its ground truth, exactly as the test script scores
 * them (sr/4_test_lut.py:313-315): y = _rgb2ycbcr(img)[:,:,0] (common/utils.py:42-60), PSNR with `shave` border
 * pixels removed and a float32 difference (:63-72), SSIM with the 11x11 sigma-1.5 Gaussian over 'valid' windows
 * in float64 (:75-101).  gt_hwc / out_hwc: device uint8 [H][W][3]; ws: device scratch of at least
 * mulut_eval_ws_doubles(H, W) doubles.  Synchronises `stream` and writes the two host doubles.
 * Errors: MULUT_ESHAPE (image smaller than the window or the shave), MULUT_EWORKSPACE. */
long long mulut_eval_ws_doubles(int H, int W);
int mulut_eval_y(int device, const void *gt_hwc, const void *out_hwc, int H, int W, int shave, double *ws, long long ws_doubles,
                 double *psnr, double *ssim, void *stream);

/* Tuning knobs (never change results).
 * "final_stage_kernel" (scale 4, <= 3 modes): 0 = auto (= 6), 1 = full-table gather kernel, 5 = tube kernel on every tile (the tube
 *   bands of all modes resident in LDS; samples with a pass outside the tube are recomputed from the full table through a device
 *   work list), 6 = hybrid: a per-tile statistic sends smooth 64x16 tiles to the tube kernel and detailed ones to the
 *   detailed-tile path.  (2-4: the band / expanded-band kernels of rounds 1-2, retired: MULUT_EINVAL.)
 *   Scales 2 and 3: 0 = the tube-band kernel of the 1-byte-row family with 4- / 9-value rows (stage_u1t_kernel<2>, <3>) on the
 *   64x64 tiles its local-detail statistic calls smooth, the gather kernel on the others (device-side tile marks; threshold
 *   "final_stage_detail_per_1024", as "first_stage_detail_per_1024" below); 5 = the tube-band kernel on every tile; 1 = the gather kernel on every tile.
 * "tube_pipelined": 1 (default) = stage_tube2_kernel where the mode list uses all of s, d, y (any order, repeats, up to 8 modes:
 *   the common "sdy" and e.g. "sdysd"; every LDS read hand-scheduled, the next pass's
 *   rows in flight under the current pass's multiply-adds, one 16x4 tile per wave, no workgroup barrier), 0 = stage_tube_kernel.
 * "detail_kernel": the detailed tiles of the hybrid: 0 (default) = anchor slabs in LDS (samples grouped by anchor MSB on the
 *   device, stage_slab_kernel; taken when the stage input is planar, < 2^28 bytes, <= 3 modes), 1 = full-table gather kernel.
 * "fix_kernel": the fix-up of the tube kernels' work list: 0 (default) = one pass per lane, 1 = one entry per thread,
 *               2 = one pass per lane with the list walk software-pipelined.
 * "stat_from_first_stage": 1 (default) = when the final stage reads what a content-routing first-stage launch of the same
 *   call wrote, its per-tile statistic looks only at the tiles that launch marked detailed; 0 = at every tile.
 * "hybrid_oob_per_1024": tile threshold of the hybrid (sites out of band per 1024, default 128).
 * "first_stage_kernel" (stages with 1-byte rows): 0 = auto (tube kernel; tiles its statistic calls detailed go to the
 *   window kernel, flagged sites are recomputed through a device work list), 2 = window kernel (full table in LDS) on every
 *   tile, 3 = tube kernel on every tile.  (1: the first one-read-per-neighbour kernel, retired: MULUT_EINVAL.)
 * "first_stage_detail_per_1024": tile threshold of first_stage_kernel 0 (default 24).
 * "u1t_persist" (experiments): 0 (default) = one workgroup per tile of the 1-byte-row tube kernel, 1..8 = that many persistent
 *   workgroups per CU walking XCD-contiguous tile ranges.  Per context, like every other key.
 * Unknown key or value: MULUT_EINVAL.
 * hipGraph capture: call mulut_reserve() for the largest (N, H, W, C) first -- the context's workspace, verdict and work-list
 * buffers are then never reallocated by smaller calls; a LARGER later call reallocates them and invalidates graphs captured
 * before it.  Run the call to be captured once outside capture first: the first launch of each kernel raises that kernel's
 * dynamic-LDS limit (hipFuncSetAttribute), which is not a capturable operation.  These one-time per-device set-ups are
 * serialised inside the library: contexts may be created and first used from several host threads. */
int mulut_set_tuning(mulut_ctx *ctx, const char *key, int value);

/* Name of the kernel variant used for the final / non-final stage (for profiles). */
const char *mulut_kernel_name(const mulut_ctx *ctx, int is_final);

#ifdef __cplusplus
}
#endif
#endif /* MULUT_H_ */
