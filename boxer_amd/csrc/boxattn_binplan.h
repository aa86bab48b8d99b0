// Plan of the destination-binned backward (boxattn_binned.h) and the device helpers that map a sample
// point to the destination blocks it touches.  A header of its own: the window-staged encoder kernels
// (boxattn_dense.hip, a separate translation unit) count and write bin records too.
#pragma once
#include "boxattn_device.h"
#include "boxattn_combine.h"

namespace boxattn {

struct BinLevel {
    int H, W, start;      // level geometry, first pixel row inside S
    int nbx, nby;         // blocks along x / y
    int blk0;             // first block id of this level inside a slice
    unsigned mw, mh;      // floor(2^32 / W) + 1, floor(2^32 / H) + 1 (0 for sizes <= 1): blk_of()
    unsigned mnx, mny;    // floor(2^32 / nbx) + 1, floor(2^32 / nby) + 1: divisions by the block counts in
                          // pack_block_geo() (0: the host found the multiply-high inexact for this level -> divide)
};

// Blocks are a BALANCED partition of the map: block column c covers
// x in [ceil(c W / nbx), ceil((c + 1) W / nbx)) with nbx = ceil(W / 8), i.e. widths differ by at
// most one and never exceed 8 (rows alike with 4).  Fixed 8x4 tiles would leave a sliver at the
// right / bottom edge (25 = 8 + 8 + 8 + 1): a block with 1-5 live pixels whose lanes each walk
// lists 6-30x the usual length, and those few work items set the length of the whole kernel.
//   block of coordinate x = floor(x nb / size), by multiply-high with the precomputed magic
//   (exact for x nb < 2^32 / size, which the host checks).
__device__ __forceinline__ int blk_of(int x, int nb, unsigned magic)
{
    return (int)__umulhi((unsigned)__mul24(x, nb), magic);
}
// first coordinate of block c: ceil(c size / nb)
__device__ __forceinline__ int blk_lo(int c, int size, int nb) { return (c * size + nb - 1) / nb; }
// n / nb by multiply-high (the host checked every n the block geometry can ask for), or the division
__device__ __forceinline__ int div_nb(int n, int nb, unsigned magic)
{
    return magic ? (int)__umulhi((unsigned)n, magic) : n / nb;
}

struct BinPlan {
    int L;
    int nblk;             // blocks per (image, head) slice
    int rec_cap;          // record capacity per slice (worst case: every point in 4 blocks)
    int item_cap;         // work-item capacity per slice
    int chunk;            // records per work item
    int lp_bits;          // record = (query << lp_bits) | (level*P + point)
    int n_slices;         // B * H
    int pslot_cap;        // partial-tile slots per slice (chunks of blocks cut into several items)
    int min_items;        // work items of a block without records: 1 -- its item stores the zeros -- or 0 for sparse
                          // maps (far more blocks than records): the empty blocks get no item, zero workers in the
                          // accumulate launch store their zeros (boxattn_scan_tail.h: zero_empty_blocks)
    int zero_workers;     // ... that many per slice (0 unless min_items == 0), kZeroPer blocks each
    BinLevel lv[kMaxBinLevels];
};

// Which slices (image, head) an XCD's accumulate workers take: CONSECUTIVE ones (1), i.e. with
// two slices per XCD the heads 2j and 2j + 1 of one image -- the two 64-byte (bf16) halves of
// every 128-byte line of grad_out they read and of grad_value they write then meet in ONE L2 --
// or every 8th (0: head x of every image, the round-1 mapping: half lines in two L2s).
#ifndef BOXATTN_TUNE_SLICE_MAP
#define BOXATTN_TUNE_SLICE_MAP 1
#endif
__device__ __forceinline__ int slice_on_xcd(int xcd, int i, int per_xcd)
{
    return BOXATTN_TUNE_SLICE_MAP ? xcd * per_xcd + i : xcd + 8 * i;
}

// Blocks touched by the (valid part of the) 2x2 footprint of a sample: up to 2 block rows x 2
// block columns, -1 for the unused candidates.  The valid rows of the footprint are exactly
// {max(y0, 0), min(y0 + 1, H - 1)} (a point that passes the window test has y0 in [-1, H - 1]),
// columns alike, so clamping replaces the per-corner validity logic.
__device__ __forceinline__ void touched_blocks(float x, float y, const BinLevel &lv, int (&blk)[4])
{
    float h_im, w_im;
    {
#pragma clang fp contract(off)                   // two roundings, as in locate()
        h_im = y * (float)lv.H - 0.5f;
        w_im = x * (float)lv.W - 0.5f;
    }
    const bool inside = h_im > -1.f && w_im > -1.f && h_im < (float)lv.H && w_im < (float)lv.W &&
                        lv.H > 0 && lv.W > 0;
    const int y0 = (int)floorf(inside ? h_im : 0.f), x0 = (int)floorf(inside ? w_im : 0.f);
    const int ra = blk_of(max(y0, 0), lv.nby, lv.mh), rb = blk_of(min(y0 + 1, lv.H - 1), lv.nby, lv.mh);
    const int ca = blk_of(max(x0, 0), lv.nbx, lv.mw), cb = blk_of(min(x0 + 1, lv.W - 1), lv.nbx, lv.mw);
    const int base_a = lv.blk0 + ra * lv.nbx, base_b = lv.blk0 + rb * lv.nbx;
    blk[0] = inside ? base_a + ca : -1;
    blk[1] = inside && cb != ca ? base_a + cb : -1;
    blk[2] = inside && rb != ra ? base_b + ca : -1;
    blk[3] = inside && rb != ra && cb != ca ? base_b + cb : -1;
}

}  // namespace boxattn
