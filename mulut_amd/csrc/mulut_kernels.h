// mulut_kernels.h -- launch interface between the C ABI (mulut_capi.hip) and the gfx950 kernels
// (mulut_kernels.hip).  Internal; the public boundary is include/mulut.h.
#ifndef MULUT_KERNELS_H_
#define MULUT_KERNELS_H_

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mulut_core.h"

namespace mulut {

constexpr int kMaxModes = 8;
constexpr int kU1TableBytes = (kRows + 15) & ~15;  // 83536: int8 rows of a v_num==1 table, padded to 16 B

// Byte view of a uint8 image batch restricted to a band of rows:
//   addr(n, c, y, x) = p + n*sN + c*sC + (y - row0)*sY + x*sX      (y in logical image rows)
struct View {
    uint8_t *p;
    long long sN;
    int sC, sY, sX;
    int row0;
};

// Device-side table formats (chosen at mulut_set_lut time):
//   v_num == 1 : int8 values as stored in the .npy, kU1TableBytes long (staged whole into LDS)
//   v_num  > 1 : value + 128 as uint8, row stride row_dwords(u)*4 bytes (one vector load per row)
__host__ __device__ constexpr int row_dwords(int u) { return (u * u + 3) / 4; }

struct StageArgs {
    int in_padded;   // the input buffer is followed by >= 8 readable bytes (the context's workspace)
    View in, out;
    int N, C, H, W;         // logical (full-image) size of the stage input
    int oy0, oy1;           // LR rows whose outputs this launch produces
    int tiles_x, tiles_y;   // tile grid over [oy0,oy1) x [0,W)
    int M;                  // number of modes
    const void *lut[kMaxModes];
    int di[kMaxModes][3], dj[kMaxModes][3];  // pattern offsets of keys b,c,d (rotation 0); dwords so that
                                             // a runtime mode index is served by scalar loads
    DivMagic div;           // epilogue divisor (16*M final, 64*M non-final)
    int bias_num;           // numerator bias (127*64*M non-final, 0 final)
    float inv_d;            // fl(1/d) for the float epilogue
    int use_f32;            // float epilogue proven exact for this divisor (rhe_f32_valid)
    float epi_c;            // fl(-unbias * fl(1/d)): addend of the fused epilogue on biased u*u-byte-row sums
    int use_fma;            // fused form proven exact for every reachable sum (rhe_fma_valid)
    // hybrid final stage: per 64x16-tile verdict written by tile_stat_kernel (0 smooth, 1 detailed).
    // verdict_take < 0: ignore; otherwise a kernel processes only tiles whose verdict == verdict_take.
    const uint32_t *verdict;
    int verdict_take;
    int vt_x, vt_y;         // verdict grid (64x16 tiles) per image
    // final stage, optional: the tile list header of the first-stage tube kernel that produced this stage's input ([2] != 0: it
    // routed by content, [16 + tile] = marked 64x64 tile of its grid): tile_stat_kernel calls the tiles it left unmarked smooth
    // without looking at them
    const uint32_t *k1_hdr;
    int k1_tiles_x, k1_tiles_y, k1_oy0;
    int k1_n0;              // image 0 of this launch is image k1_n0 of that first-stage launch (sub-launches of a large batch)
    // tube kernel: pixels with a pass outside the tube are appended here (id = (n H + y) W + x) and recomputed by
    // stage_up_fix_kernel; *fix_count is zeroed by the host side before the stage
    uint32_t *fix_list;
    uint32_t *fix_count;
    // 1-byte-row tube kernel: tile_list[tile] = 1 for the tiles it leaves to the full-table kernel (stage_u1w_kernel in
    // list mode); zeroed by the host side before the stage
    uint32_t *tile_list;
    uint32_t *tile_count;     // number of tiles marked in tile_list (the tube kernel counts, the list kernel sizes its work units by it)
    unsigned long long *dbg;  // the context's probe buffer (MULUT_DEBUG_WORDS words; written by probe builds only, see include/mulut.h)
};

struct PassArgs {
    const uint8_t *in;   // planar [C][H][W]
    int32_t *out;        // planar [C][H*u][W*u]
    const void *lut;
    int C, H, W, u, r;
    signed char di[3], dj[3];
};

struct BandArgs {
    const void *band[kMaxModes];   // device images of the tube band of each mode's table (mulut_core.h: kTubeBandBytes / kTube2BandBytes / kTube1BandBytes)
    uint32_t scale[3];     // stage_tube2_kernel only (its launcher fills it): band[p] is PATTERN p's band (s, d, y) and scale[p] how many
                           // modes of the list have that pattern, in both 16-bit halves: the band is multiplied by it while it is staged
};

enum K2Out { kOutGeneric = 0, kOutPlanarU4 = 1, kOutPackedRGBU4 = 2 };

// device buffers of the detailed-tile path of the final stage (launch_detail_slab)
// Pixels in the first 2 columns (and the last 2) need edge replication inside the 8 bytes a window row is read as (from
// column x - 2) and take the pixel fix-up list; so do the 4 columns before those unless the stage input is followed by
// readable padding (StageArgs::in_padded: the pipeline's own intermediate images are), because the 8 bytes of the image's
// very last rows would otherwise end beyond the buffer.
constexpr int kSlabXLo = 2, kSlabXHi = 6, kSlabXHiPadded = 2;
constexpr int kDetCount = 0, kDetStart = 16, kDetDirty = 32, kDetAny = 61, kDetTiles = 62, kDetItems = 63, kDetDirtyCursor = 64, kDetDirtyBase = 80, kDetCtlDwords = 128;      // dword offsets in DetailArgs::ctl
__host__ __device__ inline int slab_x_hi(const StageArgs &a) { return a.in_padded ? kSlabXHiPadded : kSlabXHi; }
struct DetailArgs {
    uint32_t *ctl;             // kDetCtlDwords: [0..15] samples per anchor MSB, [16..31] list starts, [32..47] dirty samples per anchor MSB (zero on entry),
                               // [62] detailed tiles, [63] work items, [64..79] dirty cursors (zero on entry), [80..95] where the dirty samples start
    uint32_t *dirty_count, *dirty_list;   // optional: the tube kernel's dirty samples (pixel id | channel << 30), to be computed here too
    uint16_t *thist;           // 16 per tile: anchor-MSB histogram of a detailed tile (tile_stat_kernel)
    uint32_t *tpos;            // 16 per tile: where the tile's samples of each anchor MSB start in ids / desc (detail_plan_kernel)
    uint32_t *dlist;           // the detailed tiles
    uint32_t *items;           // two dwords per item: anchor MSB << 28 | samples, first index into desc
    uint32_t *desc;            // samples grouped by anchor MSB: byte offset of (n, c, y, x - 2) in the stage input | row clamps (stage_slab_kernel)
    uint4 *blocks;             // finished 4x4 blocks (four packed rows), indexed by the sample's byte offset in the stage input
    const uint8_t *slab[3];    // per mode: the table as 16 slab pairs (mulut_core.h), kSlabTableBytes (+ 1 KiB of padding: the copy moves whole KiB)
};

// Raises a kernel's dynamic-LDS limit once per (device, kernel); safe to call from several host threads (engines may be created
// and first used concurrently).  Not a capturable operation: the first launch of a kernel must happen outside hipGraph capture.
hipError_t raise_lds_limit(const void *kernel, int bytes);

hipError_t launch_pass(const PassArgs &a, hipStream_t st);
// non-final (or u == 1 final) stage: tables staged in LDS, one byte out per site
// variant 0: window kernel (four adjacent pixels per thread, neighbours from registers), 1: one site per LDS read
hipError_t launch_stage_u1(const StageArgs &a, hipStream_t st, int variant);
// the same stage on the tube band (b.band[m] = dword-per-slot band of mode m, kTube1BandBytes): computes every tile
// whose local-detail statistic is at most detail_per_1024 (all tiles when a.tile_list is null), lists the others in
// a.tile_list and the sites that may have left the tube in a.fix_list
hipError_t launch_stage_u1t(const StageArgs &a, const BandArgs &b, unsigned detail_per_1024, int num_cus, int persist_per_cu, hipStream_t st);
// final stage with u == 2 on the tube band (8 bytes per slot: four 16-bit fields), + its site fix-up
// (a.verdict_take >= 0: routed -- tiles above detail_per_1024 are marked in a.tile_list and left to launch_stage_up, which then
// computes only marked tiles)
hipError_t launch_stage_u2t(const StageArgs &a, const BandArgs &b, unsigned detail_per_1024, int num_cus, int persist_per_cu, hipStream_t st);
// final stage with u == 3 on the tube band (24 bytes per slot: the nine values as ten 16-bit fields, mulut_core.h), + its site fix-up
hipError_t launch_stage_u3t(const StageArgs &a, const BandArgs &b, unsigned detail_per_1024, int num_cus, int persist_per_cu, hipStream_t st);
// full-table kernel over the tiles marked in a.tile_list[]
hipError_t launch_stage_u1w_list(const StageArgs &a, int num_cus, hipStream_t st);
// recompute the sites in a.fix_list[0 .. *a.fix_count) from the full tables (1-byte rows)
hipError_t launch_stage_u1_fix(const StageArgs &a, int num_cus, hipStream_t st);
void stage_u1t_tile(int &tw, int &th);
// final stage with u in {2,3,4}: u*u bytes out per site
hipError_t launch_stage_up(const StageArgs &a, int u, int out_mode, hipStream_t st);
// u == 4 and more than four modes (per-rotation accumulators)
hipError_t launch_stage_up_wide4(const StageArgs &a, hipStream_t st);
// final stage, u == 4, M <= 3, with the "tube" bands (mulut_core.h) of all modes resident in LDS: b.band[m] = expanded tube image of
// mode m (LO plane then HI plane, kTubeBandBytes); no band swaps, channel-outer loops
hipError_t launch_stage_tube(const StageArgs &a, const BandArgs &b, int out_mode, int num_cus, hipStream_t st);
const char *stage_tube_name(int out_mode);
// the same kernel with every LDS read hand-scheduled (rows of the next pass in flight under the current pass's MACs); built for
// the mode lists stage_tube2_supported() accepts
constexpr int kMaxTube2Modes = 8;
bool stage_tube2_supported(const StageArgs &a);
hipError_t launch_stage_tube2(const StageArgs &a, const BandArgs &b, int out_mode, int num_cus, hipStream_t st);
// recompute the pixels listed in a.fix_list[0 .. *a.fix_count) from the full tables (u == 4)
hipError_t launch_stage_up_fix(const StageArgs &a, int out_mode, int num_cus, hipStream_t st, int variant = 0);
// detailed tiles (a.verdict[tile] == 1, 64x16 tiling) of a u == 4 final stage from anchor slabs in LDS: bucket, plan,
// fill, slab and retile kernels; border columns are appended to a.fix_list.  d.thist comes from launch_tile_stat.
bool detail_slab_supported(const StageArgs &a);
size_t detail_ids_count(const StageArgs &a);
size_t detail_items_max(const StageArgs &a);
size_t detail_blocks_count(const StageArgs &a);
hipError_t launch_detail_slab(const StageArgs &a, const DetailArgs &d, int out_mode, int num_cus, hipStream_t st);
// per-tile smooth/detailed verdict for the hybrid final stage (tiles of stage_band_tile())
// thist (optional): [tile][16] anchor-MSB histograms of the detailed tiles for launch_detail_slab
// any (optional): set to 1 when some tile is marked detailed (DetailArgs::ctl + kDetAny, zeroed by the caller BEFORE this launch): the
// kernels of the detailed-tile path leave at once while it is 0
hipError_t launch_tile_stat(const StageArgs &a, uint32_t *verdict, uint32_t max_oob_per_1024, hipStream_t st, uint16_t *thist = nullptr, uint32_t *any = nullptr);
void stage_band_tile(int &tw, int &th);
void stage_u1_tile(int &tw, int &th);
void stage_up_tile(int &tw, int &th);
const char *stage_u1_name(int variant);
const char *stage_up_name(int u, int out_mode);

}  // namespace mulut
#endif
