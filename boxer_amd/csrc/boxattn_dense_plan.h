// Plan (host-computed, passed by value) and launch interface of the dense encoder kernels
// (boxattn_dense.h).  The kernels live in a translation unit of their own (boxattn_dense.hip, built
// with -fno-slp-vectorize, see boxer_amd/_lib.py); boxattn_capi.hip only sees this header.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "boxattn_combine.h"

namespace boxattn {

#ifndef BOXATTN_DENSE_STRIDE
#define BOXATTN_DENSE_STRIDE 12   // pixels per stored window row: 12 (lanes of columns 12-15 do not store) or 16
#endif
constexpr int kDenseStride = BOXATTN_DENSE_STRIDE;
constexpr int kDenseMaxLevels = 4;
constexpr int kDenseWin = 12;                         // window rows / columns (at most)
constexpr int kDenseTile = 4;                         // queries per tile side
constexpr int kDensePix = kDenseWin * kDenseStride;   // stored pixels of a full window

struct DenseLevel {
    int H, W, start;         // map size, first row of the level in `value`
    int ntx, ntiles;         // tiles per tile row, tiles per image
    float rcp_ntx, rcp_ntiles;
};
// window of level l for a tile of level lq: first column floor(tx * ax + bx) (tx = tile column),
// first row floor(ty * ay + by), clamped into the map; rows == 0: no window (slow path only)
struct DenseWin {
    float ax, bx, ay, by;
    int rows, cols;
};
struct DensePlan {
    int L, B, Lq, S, H;
    int hg;                  // head groups of 4 (one workgroup = one tile x 4 heads)
    float rcp_hg;
    DenseLevel lv[kDenseMaxLevels];
    DenseWin win[kDenseMaxLevels][kDenseMaxLevels];      // [query level][sampled level]
    float *dbg;              // debugging aid (builds with BOXATTN_DENSE_DEBUG; boxattn_set_debug_buffer):
                             // 8 floats per sample point
};


// workgroups of the dense kernels: 8 XCD queues x head groups x the longest queue (dense_tile_of_block)
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
    return 8u * (unsigned)p.hg * longest;
}

// grad_loc / grad_attn of bf16 box attention on a query grid (+ the combine step's workers, if any)
void launch_pointgrad_dense(const uint16_t *value, const float *loc, const float *attn,
                            const uint16_t *grad_out, const DensePlan &dp, float *grad_loc,
                            float *grad_attn, unsigned value_bytes, hipStream_t st,
                            const CombineTail &tail);

}  // namespace boxattn
