// The combine step of the binned backward (boxattn_binned.h step 6: sum the fp32 partial tiles of the
// blocks that were cut into chunks and store the block's rows) as device functions:
//   chunk_finish        inside the accumulate kernels: a chunk item publishes its partial tile, takes a
//                       ticket on its block, and the block's LAST arriver sums the tiles -- no launch and
//                       no second pass for the combine step (boxattn_ride.h);
//   combine_partials_body   the same sums as a loop over a slice's chunked blocks, for the stand-alone
//                       combine_partials_kernel (plans too big for the ticket encoding).
// Kept in a header of its own: the matrix-core accumulate kernels are a separate translation unit.
#pragma once
#include "boxattn_device.h"
#include "boxattn_ride.h"

namespace boxattn {

constexpr int kMaxBinLevels = 8;   // levels the binned backward plans for (BoxeR uses 2-5)

// A work item of the accumulate kernels: int4 {block geometry, first record, end record, partial}.  `partial`
// is -1 for a block handled by ONE item (it stores its rows itself); a block cut into chunks gives every
// chunk a partial-tile slot and the block's ordinal among the slice's chunked blocks (its combine ticket
// and its entry in `combos`):  partial = slot | ordinal << kItemSlotBits  (the host checks that both fit)
constexpr int kItemSlotBits = 18;

typedef unsigned int combine_u32x4 __attribute__((ext_vector_type(4)));

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

// Sum the cb.z partial tiles of one chunked block (cb = {block geometry, first slot, chunks}) and store its
// rows in the storage type.  One wavefront, lane = (pixel, channel half).  AGENT: the tiles were published
// inside this launch (agent-scope loads, past the L1); else they come from an earlier launch.
template <typename ST, int C, bool AGENT, int DEPTH = 4>
__device__ __forceinline__ void combine_block(const int4 cb, const float *__restrict__ partials, int pslot_cap,
                                              int lv_start, int lv_W, int S, int H,
                                              ST *__restrict__ grad_value, int s, int lane)
{
    constexpr int BW = 8, PB = 32, CH = C / 2, EPL = 16 / (int)sizeof(ST);
    const int b = s / H, h = s % H;
    const int mypix = lane >> 1, half = lane & 1;
    const BlockGeo bg = unpack_block_geo((unsigned)cb.x);
    const int oy = bg.oy, ox = bg.ox, bh = bg.bh, bw = bg.bw;
    if (mypix / BW >= bh || mypix % BW >= bw) return;
    const int yy = oy + mypix / BW, xx = ox + mypix % BW;
    float acc[CH];
#pragma unroll
    for (int c = 0; c < CH; ++c) acc[c] = 0.f;
    // (the block's tiles are consecutive slots: one buffer over them, 32-bit offsets -- a chunked block has
    // at most rec_cap / chunk + 1 chunks of 4 C-float rows x 32 pixels, far below 4 GB)
    const float *tile0 = partials + ((size_t)s * pslot_cap + cb.y) * PB * C;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float *>(tile0), 0, (unsigned)cb.z * (unsigned)(PB * C * 4), 0x00020000);
    const unsigned off0 = (unsigned)((mypix * C + half * CH) * 4);
    for (int j0 = 0; j0 < cb.z; j0 += DEPTH) {              // DEPTH partial tiles in flight
        float4 t[DEPTH][CH / 4];
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            const unsigned off = off0 + (unsigned)min(j0 + u, cb.z - 1) * (unsigned)(PB * C * 4);
#pragma unroll
            for (int c = 0; c < CH / 4; ++c) {
                // AGENT: sc1 (aux 16) -- served past the L1, the counterpart of partial_store's write-through
                const combine_u32x4 w = AGENT ? __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16u * c, 0, 16)
                                              : __builtin_amdgcn_raw_buffer_load_b128(rs, off + 16u * c, 0, 0);
                t[u][c] = make_float4(__uint_as_float(w.x), __uint_as_float(w.y), __uint_as_float(w.z),
                                      __uint_as_float(w.w));
            }
        }
#pragma unroll
        for (int u = 0; u < DEPTH; ++u) {
            if (j0 + u < cb.z) {
#pragma unroll
                for (int c = 0; c < CH / 4; ++c) {
                    acc[4 * c] += t[u][c].x; acc[4 * c + 1] += t[u][c].y;
                    acc[4 * c + 2] += t[u][c].z; acc[4 * c + 3] += t[u][c].w;
                }
            }
        }
    }
    ST *dst = grad_value + (((size_t)b * S + lv_start + (size_t)yy * lv_W + xx) * H + h) * C + half * CH;
#pragma unroll
    for (int c = 0; c < CH; c += EPL) {
        float t[EPL];
#pragma unroll
        for (int i = 0; i < EPL; ++i) t[i] = acc[c + i];
        VecIO<ST, EPL>::st(dst + c, t);
    }
}

// One wavefront (lane = threadIdx & 63) as worker `worker` of `n_workers` of slice s.
template <typename ST, int C>
__device__ __forceinline__ void combine_partials_body(const int4 *__restrict__ combos,
                                                      const int *__restrict__ n_items,
                                                      const float *__restrict__ partials,
                                                      const CombinePlan &plan, int S, int H,
                                                      ST *__restrict__ grad_value, int s, int worker,
                                                      int n_workers, int lane)
{
    const int n_comb = n_items[2 * s + 1];
    for (int ci = worker; ci < n_comb; ci += n_workers) {
        const int4 cb = combos[(size_t)s * plan.nblk + ci];
        const int level = (int)((unsigned)cb.x >> 29);
        int lv_start = plan.start[0], lv_W = plan.W[0];
#pragma unroll
        for (int k = 1; k < kMaxBinLevels; ++k)
            if (k == level) { lv_start = plan.start[k]; lv_W = plan.W[k]; }
        // (a launch of its own is all latency -- the heaviest block's tiles in ONE round trip where the registers allow)
        combine_block<ST, C, false, (C <= 32 ? 8 : 4)>(cb, partials, plan.pslot_cap, lv_start, lv_W, S, H, grad_value, s, lane);
    }
}

// What the accumulate kernels need for the in-launch combine.  tickets == nullptr: off (the partial
// tiles are summed by combine_partials_kernel behind the accumulate launch).
struct ChunkCombine {
    int *tickets;            // [slice][nblk], zero on entry (the fill riders clear them), left zero
    const int4 *combos;      // [slice][nblk] {block geometry, first partial slot, chunks} of the chunked blocks
    int nblk, pslot_cap;
};

// A chunk item's partial tile (32 pixels x C floats) as a buffer; 16-byte pieces of it are written through
// (sc1, aux 16) when the block's last arriver -- possibly on another XCD -- will read them inside this launch
__device__ __forceinline__ __amdgpu_buffer_rsrc_t partial_tile(float *partials, int s, int pslot_cap, int slot, int C)
{
    return __builtin_amdgcn_make_buffer_rsrc(partials + ((size_t)s * pslot_cap + slot) * 32 * C, 0,
                                             (unsigned)(32 * C * 4), 0x00020000);
}
__device__ __forceinline__ void partial_store(__amdgpu_buffer_rsrc_t tile, unsigned byte_off, float4 v, bool publish)
{
    const combine_u32x4 w = {__float_as_uint(v.x), __float_as_uint(v.y), __float_as_uint(v.z), __float_as_uint(v.w)};
    if (publish) __builtin_amdgcn_raw_buffer_store_b128(w, tile, byte_off, 0, 16);
    else __builtin_amdgcn_raw_buffer_store_b128(w, tile, byte_off, 0, 0);
}

// A chunk item (single-wave workgroup) has stored its partial tile: ticket on the block; the last arriver
// of the block's chunks sums them.  item_w = slot | ordinal << kItemSlotBits (boxattn_scan_tail.h).
template <typename ST, int C>
__device__ __forceinline__ void chunk_finish(const ChunkCombine cc, const float *__restrict__ partials,
                                             int lv_start, int lv_W, int S, int H, ST *__restrict__ grad_value,
                                             int s, int item_w, int lane, int *lds_flag)
{
    stores_left();
    asm volatile("" : "+v"(lane));       // (the caller's loops get no hoisted per-lane addresses of this rarely taken tail)
    const int ordinal = item_w >> kItemSlotBits;
    const int4 cb = cc.combos[(size_t)s * cc.nblk + ordinal];
    if (!last_arriver<64>(cc.tickets + (size_t)s * cc.nblk + ordinal, cb.z, lds_flag)) return;
    combine_block<ST, C, true>(cb, partials, cc.pslot_cap, lv_start, lv_W, S, H, grad_value, s, lane);
}

}  // namespace boxattn
