// Plan (host-computed, passed by value) and launch interface of the window-staged encoder kernels
// (boxattn_dense.h).  The kernels live in a translation unit of their own (boxattn_dense.hip, see
// boxer_amd/_lib.py); boxattn_capi.hip only sees this header.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "boxattn_combine.h"

namespace boxattn {

constexpr int kDenseMaxLevels = 4;
constexpr int kDenseTile = 8;              // queries per tile side: one workgroup = 8x8 queries x one head
constexpr int kDenseSub = 4;               // ... = 2x2 sub-tiles of 4x4 queries, one wavefront each
constexpr int kDenseWinMax = 16;           // window rows / columns at most (one wave-load per window row)
constexpr int kDenseSlotBytes = 80;        // one staged pixel: 64 bytes of bf16 channels + 16 bytes of padding
                                           // (bank = 20 slot + 4 chunk mod 64: 16 consecutive slots, 16 bank groups)
constexpr int kDenseSlots = 480;           // staged pixels per workgroup (38 400 bytes)

struct DenseLevel {
    int H, W, start;         // map size, first row of the level in `value`
    int ntx, ntiles;         // tiles per tile row, tiles per image
    float rcp_ntx, rcp_ntiles;
};
// window of level l for a tile of level lq: first column floor(tx * ax + bx) (tx = tile column),
// first row floor(ty * ay + by), clamped into the map; rows == 0: not staged (global path only).
// Staged pixel (r, c) of the window lives in slot off + r * pitch + c.
struct DenseWin {
    float ax, bx, ay, by;
    unsigned geo;            // rows | cols << 5 | pitch << 10 | off << 16 (one scalar register, see dense_win_*)
};
inline unsigned dense_win_pack(int rows, int cols, int pitch, int off)
{
    return (unsigned)rows | ((unsigned)cols << 5) | ((unsigned)pitch << 10) | ((unsigned)off << 16);
}
struct DensePlan {
    int L, B, Lq, S, H;
    float rcp_h;
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

// grad_loc / grad_attn of bf16 box attention on a query grid (+ the combine step's workers, if any)
void launch_pointgrad_dense(const uint16_t *value, const float *loc, const float *attn,
                            const uint16_t *grad_out, const DensePlan &dp, float *grad_loc,
                            float *grad_attn, unsigned value_bytes, hipStream_t st,
                            const CombineTail &tail);

}  // namespace boxattn
