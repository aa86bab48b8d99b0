// The combine step of the binned backward (boxattn_binned.h step 6: sum the partial tiles of the blocks
// that were cut into chunks) as a device function, so that it can ride along as extra workgroups in
// whichever kernel is the last of the backward (pointgrad2_kernel, pointgrad_dense_kernel).  Kept in
// a header of its own: the dense kernels are a separate translation unit (other compiler flags).
#pragma once
#include "boxattn_device.h"

namespace boxattn {

constexpr int kMaxBinLevels = 8;   // levels the binned backward plans for (BoxeR uses 2-5)

struct BlockGeo { int oy, ox, bh, bw, level; };
__device__ __forceinline__ BlockGeo unpack_block_geo(unsigned g)
{
    BlockGeo r;
    r.oy = (int)(g & 0xFFFu); r.ox = (int)((g >> 12) & 0xFFFu);
    r.bh = (int)((g >> 24) & 3u) + 1; r.bw = (int)((g >> 26) & 7u) + 1;
    r.level = (int)(g >> 29);
    return r;
}


// What the combine step needs of the plan (small enough to ride along in another kernel's arguments).
struct CombinePlan {
    int nblk, pslot_cap, n_slices;
    int start[kMaxBinLevels], W[kMaxBinLevels];
};
// One wavefront (lane = threadIdx & 63) as worker `worker` of `n_workers` of slice s.
template <typename ST, int C>
__device__ __forceinline__ void combine_partials_body(const int4 *__restrict__ combos,
                                                      const int *__restrict__ n_items,
                                                      const float *__restrict__ partials,
                                                      const CombinePlan &plan, int S, int H,
                                                      ST *__restrict__ grad_value, int s, int worker,
                                                      int n_workers, int lane)
{
    constexpr int BW = 8, PB = 32, CH = C / 2, EPL = 16 / (int)sizeof(ST);
    const int b = s / H, h = s % H;
    const int n_comb = n_items[2 * s + 1];
    const int mypix = lane >> 1, half = lane & 1;
    for (int ci = worker; ci < n_comb; ci += n_workers) {
        const int4 cb = combos[(size_t)s * plan.nblk + ci];          // {block geometry, first slot, nch}
        const BlockGeo bg = unpack_block_geo((unsigned)cb.x);
        int lv_start = plan.start[0], lv_W = plan.W[0];
#pragma unroll
        for (int k = 1; k < kMaxBinLevels; ++k)
            if (k == bg.level) { lv_start = plan.start[k]; lv_W = plan.W[k]; }
        const int oy = bg.oy, ox = bg.ox, bh = bg.bh, bw = bg.bw;
        if (mypix / BW >= bh || mypix % BW >= bw) continue;
        const int yy = oy + mypix / BW, xx = ox + mypix % BW;
        float acc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = 0.f;
        const float *p0 = partials + (((size_t)s * plan.pslot_cap + cb.y) * PB + mypix) * C +
                          half * CH;
        for (int j0 = 0; j0 < cb.z; j0 += 4) {                  // 4 partial tiles in flight
            float4 t[4][CH / 4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4 *src = reinterpret_cast<const float4 *>(
                    p0 + (size_t)min(j0 + u, cb.z - 1) * PB * C);
#pragma unroll
                for (int c = 0; c < CH / 4; ++c) t[u][c] = src[c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (j0 + u < cb.z) {
#pragma unroll
                    for (int c = 0; c < CH / 4; ++c) {
                        acc[4 * c] += t[u][c].x; acc[4 * c + 1] += t[u][c].y;
                        acc[4 * c + 2] += t[u][c].z; acc[4 * c + 3] += t[u][c].w;
                    }
                }
            }
        }
        ST *dst = grad_value + (((size_t)b * S + lv_start + (size_t)yy * lv_W + xx) * H + h) * C +
                  half * CH;
#pragma unroll
        for (int c = 0; c < CH; c += EPL) {
            float t[EPL];
#pragma unroll
            for (int i = 0; i < EPL; ++i) t[i] = acc[c + i];
            VecIO<ST, EPL>::st(dst + c, t);
        }
    }
}


// The combine step of the binned backward (sum the partial tiles of the chunked blocks,
// boxattn_binned.h step 6) riding along in the point-gradient launch: `workers` single-wave workers
// per slice are appended to the grid as extra workgroups.  The two have nothing to do with each
// other except that the point gradients are the LAST kernel of the backward once they are launched
// after the accumulate kernel -- and a launch of its own for a few hundred waves of work is 5-7 us
// of every step (a tenth of a decoder-shaped one).
struct CombineTail {
    const int4 *combos;
    const int *n_items;
    const float *partials;
    void *grad_value;
    CombinePlan plan;
    int workers;          // per slice; 0: no combine work in this launch
};

}  // namespace boxattn
