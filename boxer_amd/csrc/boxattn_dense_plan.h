// Plan (host-computed, passed by value) and launch interface of the window-staged encoder kernels
// (boxattn_dense.h).  The kernels live in a translation unit of their own (boxattn_dense.hip, see
// boxer_amd/_lib.py); boxattn_capi.hip only sees this header.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "boxattn_combine.h"
#include "boxattn_binplan.h"
#include "boxattn_binpass.h"
#include "boxattn_scan_tail.h"

namespace boxattn {

constexpr int kDenseMaxLevels = 4;
constexpr int kDenseTile = 8;              // queries per tile side: one workgroup = 8x8 queries x one head
constexpr int kDenseSub = 4;               // ... = 2x2 sub-tiles of 4x4 queries, one wavefront each
constexpr int kDenseWinMax = 16;           // window rows / columns at most (one wave-load per window row)
// The windows in LDS.  A window row is written by ONE direct-to-LDS load (buffer_load_dwordx4 ... lds: 4 lanes
// per pixel, destination = a wave-uniform base + lane x 16 bytes), so its pixels are 64 bytes apart with no
// padding between them; consecutive ROWS are skewed by 16 bytes instead.  A pixel's 16-byte pieces then fall
// into bank group (4 column + row + piece) mod 16: the 2x2 footprint of a point and the pixels of neighbouring
// lanes spread over the 16 groups as well as a padded slot would, at 4/5 of the bytes.
constexpr int kDenseSlotBytes = 64;        // one staged pixel: 32 bf16 channels
constexpr int kDenseRowSkew = 16;          // bytes between the end of a window row and the start of the next
#ifndef BOXATTN_DENSE_LDS
#define BOXATTN_DENSE_LDS 28160
#endif
constexpr int kDenseLdsBytes = BOXATTN_DENSE_LDS;  // per workgroup: 27.5 KB -- five workgroups per CU also with the point-gradient
                                                   // kernel's 4 KB stash (157.5 of 160 KB); BoxeR-R50 tiles need 25.4 KB,
                                                   // odd map sizes (10 x 10 instead of 9 x 9 on the next level) 26.7
constexpr int kDenseZeroOff = kDenseLdsBytes - kDenseSlotBytes;    // the forward's row of zeros (make_dense_plan leaves it free)
constexpr int kDenseStatSlots = 64;        // pairs of 64-bit locality counters in the caller's state buffer (power of two)

struct DenseLevel {
    int H, W, start;         // map size, first row of the level in `value`
    int ntx, ntiles;         // tiles per tile row, tiles per image
    unsigned n_all;          // B * ntiles
    unsigned mag_ntx, mag_ntiles;    // floor(2^32 / d) (0xFFFFFFFF for d = 1): divmod_magic()
};
// window of level l for a tile of level lq: first column floor(tx * ax + bx) (tx = tile column),
// first row floor(ty * ay + by), clamped into the map; rows == 0: not staged (global path only).
// Staged pixel (r, c) of the window lives at byte 16 (off16 + r * pitch16) + 64 c of the workgroup's LDS.
struct DenseWin {
    int ax, bx, ay, by;      // 16.16 fixed point: first column (tx * ax + bx) >> 16, first row alike
    unsigned geo;            // rows | cols << 5 | pitch16 << 10 | off16 << 17 (one scalar register, see DenseWinPos)
};
inline int dense_win_pitch16(int cols) { return (cols * kDenseSlotBytes + kDenseRowSkew) / 16; }
inline unsigned dense_win_pack(int rows, int cols, int off16)
{
    return (unsigned)rows | ((unsigned)cols << 5) | ((unsigned)dense_win_pitch16(cols) << 10) | ((unsigned)off16 << 17);
}
struct DensePlan {
    int L, B, Lq, S, H;
    unsigned mag_h;          // floor(2^32 / H)
    DenseLevel lv[kDenseMaxLevels];
    DenseWin win[kDenseMaxLevels][kDenseMaxLevels];      // [query level][sampled level]
    float *dbg;              // debugging aid (builds with BOXATTN_DENSE_DEBUG; boxattn_set_debug_buffer)
};

// workgroups of the kernels: 8 XCD queues x heads x the longest queue (dense_tile_of_block)
inline unsigned dense_blocks(const DensePlan &p)
{
    unsigned longest = 0;
    for (unsigned x = 0; x < 8; ++x) {
        unsigned n_x = 0;
        for (int l = 0; l < p.L; ++l) {
            const unsigned n = (unsigned)p.B * (unsigned)p.lv[l].ntiles;
            n_x += (((x + 1) * n) >> 3) - ((x * n) >> 3);
        }
        longest = longest > n_x ? longest : n_x;
    }
    return 8u * (unsigned)p.H * longest;
}

// grad_loc / grad_attn of bf16 box attention on a query grid (+ the backward's fill riders, if any:
// ride.grid.n_riders > 0; the caller sets ride.grid.n_riders / .shift, the launcher places them: boxattn_ride.h)
void launch_pointgrad_dense(const uint16_t *value, const float *loc, const float *attn,
                            const uint16_t *grad_out, const DensePlan &dp, float *grad_loc,
                            float *grad_attn, unsigned value_bytes, hipStream_t st, const BinRide &ride);

// out of bf16 box attention on a query grid (boxattn_dense_fwd.h) (+ the training forward's count riders
// and the scans chained behind them, if any; stats: kDenseStatSlots pairs of locality counters or null)
void launch_fwd_dense(const uint16_t *value, const float *loc, const float *attn, uint16_t *out,
                      const DensePlan &dp, unsigned value_bytes, const BinRide &ride,
                      unsigned long long *stats, hipStream_t st);

// float32 storage (boxattn_dense_f32.h): the plan's window geometry in 32-byte units (make_dense_plan with 4-byte elements)
void launch_pointgrad_dense_f32(const float *value, const float *loc, const float *attn, const float *grad_out,
                                const DensePlan &dp, float *grad_loc, float *grad_attn, unsigned value_bytes,
                                hipStream_t st, const BinRide &ride);
void launch_fwd_dense_f32(const float *value, const float *loc, const float *attn, float *out, const DensePlan &dp,
                          unsigned value_bytes, const BinRide &ride, unsigned long long *stats, hipStream_t st);
#ifndef BOXATTN_DENSE_F32_LDS
#define BOXATTN_DENSE_F32_LDS 53248
#endif

// The matrix-core accumulate of bf16 box attention (boxattn_binned_tr.h; lives in this translation unit
// because it mixes float32 VALU work with MFMAs, see boxattn_dense.hip).  C = 16, 32 or 64 channels per
// head, grad_out below 2 GB (32-bit row offsets, and an out-of-range offset for idle lanes).
constexpr size_t kAccTrMaxBytes = (size_t)1 << 31;
void launch_accumulate_tr(int C, const uint16_t *grad_out, size_t grad_out_bytes, const BinPlan &plan, int S,
                          int H, int Lq, const int4 *items, const int *n_items, const int *records,
                          uint16_t *grad_value, float *partials, int wg_per_slice, int ns8, const ChunkCombine &cc,
                          const ZeroRole &zr, hipStream_t st);

// float32 storage, C = 32: the accumulate on the bf16 matrix cores with exact three-term splits (boxattn_binned_tr.h:
// binned_accumulate_split_kernel); grad_out below 2 GB
void launch_accumulate_split(const float *grad_out, size_t grad_out_bytes, const BinPlan &plan, int S, int H, int Lq,
                             const int4 *items, const int *n_items, const int *records, float *grad_value,
                             float *partials, int wg_per_slice, int ns8, const ChunkCombine &cc, const ZeroRole &zr,
                             hipStream_t st, const float *grad_mask = nullptr, size_t grad_mask_bytes = 0,
                             const float *w_lv = nullptr, int P = 1);

// float32 storage, C = 32: the accumulate on v_mfma_f32_32x32x2_f32 (boxattn_binned_tr.h); grad_out below 2 GB
void launch_accumulate_f32(const float *grad_out, size_t grad_out_bytes, const BinPlan &plan, int S, int H, int Lq,
                           const int4 *items, const int *n_items, const int *records, float *grad_value,
                           float *partials, int wg_per_slice, int ns8, const ChunkCombine &cc, const ZeroRole &zr,
                           hipStream_t st);

}  // namespace boxattn
