// Destination-binned backward for grad_value: no floating-point atomics on the hot path.
//
// Why: the reference scatters 4*C atomicAdds per sample point into grad_value
// (box_attn_kernel.cuh:100-184; 436 M atomics at BoxeR-R50 COCO shapes).  On MI355X global
// fp32 atomics retire ~10 G 128-byte row-ops/s whatever their scope or locality, and LDS
// ds_add_f32 ~3 clk per LANE (profiles/r01_microbench_*.log), so any "atomic per
// contribution" design is 1.3-5 ms.  Here every grad_value row is instead produced by ONE
// owner that gathers its contributions:
//
//   1. bin_kernel<count>   one pass over the sampling locations: every sample point is
//                          assigned to the 32-pixel blocks (8x4, per image, head, level)
//                          its 2x2 footprint touches; per-workgroup LDS histograms, one
//                          global integer atomic per (workgroup, block).
//   2. bin_scan_kernel     per (image, head) slice: exclusive scan -> record offsets and
//                          the work-item list (blocks with many points are cut into chunks).
//   3. bin_kernel<fill>    same pass again, now writing the point ids into their bins.
//   4. bwd_fast_kernel<SCATTER=false> (boxattn_fast.h): grad_loc / grad_weight, query-major.
//   5. binned_accumulate_kernel  one workgroup per work item: records -> geometry ->
//                          per-pixel entry lists in LDS (integer LDS atomics for the ranks);
//                          the records' upstream-gradient rows (times the attention weight)
//                          are staged once in LDS; each 8-lane group owns one destination
//                          pixel and sums w * row over its list in registers; one plain
//                          coalesced 128-byte row store per pixel (fp32 atomics only for the
//                          few blocks that were cut into chunks, i.e. the coarse levels).
//
// The result is the same sum as the reference's, in a different (still unspecified) order.
#pragma once
#include "boxattn_device.h"

namespace boxattn {

struct BinLevel {
    int H, W, start;      // level geometry, first pixel row inside S
    int nbx, nby;         // blocks along x / y
    int blk0;             // first block id of this level inside a slice
};

struct BinPlan {
    int L;
    int nblk;             // blocks per (image, head) slice
    int rec_cap;          // record capacity per slice (worst case: every point in 4 blocks)
    int item_cap;         // work-item capacity per slice
    int chunk;            // records per work item
    BinLevel lv[kMaxLevels];
};

template <int G> struct BlockShape;             // G lanes per pixel -> 256/G pixels per block
template <> struct BlockShape<4>  { static constexpr int W = 8, H = 8; };
template <> struct BlockShape<8>  { static constexpr int W = 8, H = 4; };
template <> struct BlockShape<16> { static constexpr int W = 4, H = 4; };

// Blocks touched by the (valid part of the) 2x2 footprint of a sample; at most 2x2.
template <int BW, int BH>
__device__ __forceinline__ int touched_blocks(const Sample<float> &s, const BinLevel &lv,
                                              int (&blk)[4])
{
    if (!s.inside) return 0;
    const bool ya = s.ok[0] || s.ok[1], yb = s.ok[2] || s.ok[3];
    const bool xa = s.ok[0] || s.ok[2], xb = s.ok[1] || s.ok[3];
    int rows[2], cols[2], nr = 0, nc = 0;
    if (ya) rows[nr++] = s.y0 / BH;
    if (yb) { const int r = (s.y0 + 1) / BH; if (nr == 0 || r != rows[0]) rows[nr++] = r; }
    if (xa) cols[nc++] = s.x0 / BW;
    if (xb) { const int c = (s.x0 + 1) / BW; if (nc == 0 || c != cols[0]) cols[nc++] = c; }
    int n = 0;
    for (int i = 0; i < nr; ++i)
        for (int j = 0; j < nc; ++j) blk[n++] = lv.blk0 + rows[i] * lv.nbx + cols[j];
    return n;
}

// ---------------------------------------------------------------------------------------
// 1 + 3: count / fill.  grid = (query chunks, slices), block 256, dynamic LDS 2*nblk ints.
// ---------------------------------------------------------------------------------------
template <int BW, int BH, bool FILL>
__global__ __launch_bounds__(256) void bin_kernel(const float *__restrict__ loc, BinPlan plan,
                                                  int H, int Lq, int P, int q_per_wg,
                                                  int *__restrict__ counts,
                                                  int *__restrict__ cursors,
                                                  const int *__restrict__ offsets,
                                                  int *__restrict__ records)
{
    extern __shared__ int sh_bins[];
    int *hist = sh_bins;
    int *base = sh_bins + plan.nblk;
    const int s = blockIdx.y, b = s / H, h = s % H;
    const int LP = plan.L * P;
    const int q0 = blockIdx.x * q_per_wg;
    const int q1 = min(q0 + q_per_wg, Lq);
    const int n_pts = (q1 - q0) * LP;
    for (int k = threadIdx.x; k < plan.nblk; k += blockDim.x) hist[k] = 0;
    __syncthreads();

    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);
    for (int i = threadIdx.x; i < n_pts; i += blockDim.x) {
        const int q = q0 + i / LP, lp = i % LP, l = lp / P;
        const size_t pid = (((size_t)b * Lq + q) * H + h) * LP + lp;
        const float2 xy = loc2[pid];
        const Sample<float> sm = locate<float>(xy.x, xy.y, plan.lv[l].H, plan.lv[l].W);
        int blk[4];
        const int n = touched_blocks<BW, BH>(sm, plan.lv[l], blk);
        for (int j = 0; j < n; ++j) atomicAdd(&hist[blk[j]], 1);
    }
    __syncthreads();
    const size_t sb = (size_t)s * plan.nblk;
    if (!FILL) {
        for (int k = threadIdx.x; k < plan.nblk; k += blockDim.x)
            if (hist[k]) atomicAdd(&counts[sb + k], hist[k]);
        return;
    }
    for (int k = threadIdx.x; k < plan.nblk; k += blockDim.x) {
        const int c = hist[k];
        base[k] = c ? atomicAdd(&cursors[sb + k], c) + offsets[(size_t)s * (plan.nblk + 1) + k]
                    : 0;
        hist[k] = 0;
    }
    __syncthreads();
    int *rec = records + (size_t)s * plan.rec_cap;
    for (int i = threadIdx.x; i < n_pts; i += blockDim.x) {
        const int q = q0 + i / LP, lp = i % LP, l = lp / P;
        const size_t pid = (((size_t)b * Lq + q) * H + h) * LP + lp;
        const float2 xy = loc2[pid];
        const Sample<float> sm = locate<float>(xy.x, xy.y, plan.lv[l].H, plan.lv[l].W);
        int blk[4];
        const int n = touched_blocks<BW, BH>(sm, plan.lv[l], blk);
        for (int j = 0; j < n; ++j) {
            const int slot = base[blk[j]] + atomicAdd(&hist[blk[j]], 1);
            rec[slot] = q * LP + lp;                     // point id inside the slice
        }
    }
}

// ---------------------------------------------------------------------------------------
// 2: per-slice scan.  grid = slices, block 256.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bin_scan_kernel(const int *__restrict__ counts,
                                                       int *__restrict__ offsets,
                                                       int4 *__restrict__ items,
                                                       int *__restrict__ n_items, BinPlan plan)
{
    __shared__ int wsum_c[4], wsum_n[4];
    __shared__ int carry_c, carry_n;
    const int s = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) { carry_c = 0; carry_n = 0; }
    __syncthreads();
    for (int k0 = 0; k0 < plan.nblk; k0 += 256) {
        const int k = k0 + threadIdx.x;
        const int c = k < plan.nblk ? counts[(size_t)s * plan.nblk + k] : 0;
        const int nch = (c + plan.chunk - 1) / plan.chunk;
        int ic = c, in = nch;                                 // inclusive wave scans
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int tc = __shfl_up(ic, o, 64), tn = __shfl_up(in, o, 64);
            if (lane >= o) { ic += tc; in += tn; }
        }
        if (lane == 63) { wsum_c[wv] = ic; wsum_n[wv] = in; }
        __syncthreads();
        int pc = carry_c, pn = carry_n;
        for (int w = 0; w < wv; ++w) { pc += wsum_c[w]; pn += wsum_n[w]; }
        const int ec = pc + ic - c, en = pn + in - nch;       // exclusive
        if (k < plan.nblk) {
            offsets[(size_t)s * (plan.nblk + 1) + k] = ec;
            for (int j = 0; j < nch; ++j)
                items[(size_t)s * plan.item_cap + en + j] =
                    make_int4(k, j * plan.chunk, min(c, (j + 1) * plan.chunk), nch);
        }
        __syncthreads();
        if (threadIdx.x == 255) { carry_c = pc + ic; carry_n = pn + in; }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        offsets[(size_t)s * (plan.nblk + 1) + plan.nblk] = carry_c;
        n_items[s] = carry_n;
    }
}

// ---------------------------------------------------------------------------------------
// 5: accumulate.  grid = (workgroups per slice, slices), block 256; persistent over items.
// ---------------------------------------------------------------------------------------
template <typename ST, int G, bool INST>
__global__ __launch_bounds__(256) void binned_accumulate_kernel(
    const ST *__restrict__ grad_out, const ST *__restrict__ grad_mask,
    const float *__restrict__ loc, const float *__restrict__ w_sp,
    const float *__restrict__ w_lv, BinPlan plan, int S, int H, int Lq, int P,
    const int *__restrict__ offsets, const int4 *__restrict__ items,
    const int *__restrict__ n_items, int *__restrict__ qhead, const int *__restrict__ records,
    float *__restrict__ grad_value)
{
    constexpr int VEC = 4, C = VEC * G, PB = 256 / G;
    constexpr int BW = BlockShape<G>::W, BH = BlockShape<G>::H;
    constexpr int R = (G == 16) ? 128 : 256;          // records per round
    constexpr int TS = C + 4;                         // padded row stride (floats), 16-B aligned
    __shared__ __attribute__((aligned(16))) float tstage[R * TS];
    __shared__ float2 ent[4 * R];                     // {bilinear weight, record slot}
    __shared__ int pcnt[64], poff[65];
    __shared__ int rec_row[R], rec_mrow[INST ? R : 1];
    __shared__ float rec_as[R], rec_al[INST ? R : 1];
    __shared__ int cur_item;

    const int s = blockIdx.y, b = s / H, h = s % H;
    const int LP = plan.L * P;
    const int tid = threadIdx.x, m = tid % G, mypix = tid / G;
    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);
    const int n_it = n_items[s];

    for (;;) {
        if (tid == 0) cur_item = atomicAdd(&qhead[s], 1);
        __syncthreads();
        const int it = cur_item;
        if (it >= n_it) break;
        const int4 item = items[(size_t)s * plan.item_cap + it];
        const int blk = item.x;
        int l = 0;
        while (l + 1 < plan.L && blk >= plan.lv[l + 1].blk0) ++l;
        const BinLevel lv = plan.lv[l];
        const int by = (blk - lv.blk0) / lv.nbx, bx = (blk - lv.blk0) % lv.nbx;
        const int oy = by * BH, ox = bx * BW;
        const int *rec = records + (size_t)s * plan.rec_cap +
                         offsets[(size_t)s * (plan.nblk + 1) + blk];
        float acc[VEC] = {0.f, 0.f, 0.f, 0.f};

        for (int rr = item.y; rr < item.z; rr += R) {
            const int n = min(R, item.z - rr);
            if (tid < 64) pcnt[tid] = 0;
            __syncthreads();
            // ---- phase 1: one thread per record: geometry, ranks inside the pixel lists
            int rank[4], pixk[4];
            float wk[4];
            bool use[4] = {false, false, false, false};
            if (tid < n) {
                const int lpid = rec[rr + tid];
                const int q = lpid / LP, lp = lpid % LP;
                const size_t row = ((size_t)b * Lq + q) * H + h;
                const size_t pid = row * LP + lp;
                const float2 xy = loc2[pid];
                const Sample<float> sm = locate<float>(xy.x, xy.y, lv.H, lv.W);
                wk[0] = sm.hh * sm.hw; wk[1] = sm.hh * sm.lw;
                wk[2] = sm.lh * sm.hw; wk[3] = sm.lh * sm.lw;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int yy = sm.y0 + (k >> 1), xx = sm.x0 + (k & 1);
                    if (sm.ok[k] && yy / BH == by && xx / BW == bx) {
                        use[k] = true;
                        pixk[k] = (yy - oy) * BW + (xx - ox);
                        rank[k] = atomicAdd(&pcnt[pixk[k]], 1);
                    }
                }
                rec_row[tid] = (int)row;
                rec_as[tid] = w_sp[pid];
                if constexpr (INST) {
                    const int p = lp % P;
                    rec_mrow[tid] = (int)((((size_t)b * Lq + q) * P + p) * H + h);
                    rec_al[tid] = w_lv[pid];
                }
            }
            __syncthreads();
            if (tid < 64) {                                   // exclusive scan of the PB counts
                const int c = tid < PB ? pcnt[tid] : 0;
                int ic = c;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_up(ic, o, 64);
                    if (tid >= o) ic += t;
                }
                poff[tid + 1] = ic;
                if (tid == 0) poff[0] = 0;
            }
            __syncthreads();
            if (tid < n) {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (use[k])
                        ent[poff[pixk[k]] + rank[k]] = make_float2(wk[k], __int_as_float(tid));
            }
            // ---- stage t = a_s * g (+ a_l * g_mask) rows of this round's records
            for (int j = mypix; j < n; j += PB) {
                float g[VEC], t[VEC];
                VecIO<ST, VEC>::ld(grad_out + (size_t)rec_row[j] * C + m * VEC, g);
                const float as = rec_as[j];
#pragma unroll
                for (int c = 0; c < VEC; ++c) t[c] = g[c] * as;
                if constexpr (INST) {
                    float gm[VEC];
                    VecIO<ST, VEC>::ld(grad_mask + (size_t)rec_mrow[j] * C + m * VEC, gm);
                    const float al = rec_al[j];
#pragma unroll
                    for (int c = 0; c < VEC; ++c) t[c] += gm[c] * al;
                }
                *reinterpret_cast<float4 *>(&tstage[j * TS + m * VEC]) =
                    make_float4(t[0], t[1], t[2], t[3]);
            }
            __syncthreads();
            // ---- phase 2: every G-lane group owns one destination pixel
            const int e1 = poff[mypix + 1];
            for (int e = poff[mypix]; e < e1; ++e) {
                const float2 en = ent[e];
                const int j = __float_as_int(en.y);
                const float4 tv = *reinterpret_cast<const float4 *>(&tstage[j * TS + m * VEC]);
                acc[0] += en.x * tv.x; acc[1] += en.x * tv.y;
                acc[2] += en.x * tv.z; acc[3] += en.x * tv.w;
            }
            __syncthreads();
        }
        // ---- one row store per destination pixel
        const int yy = oy + mypix / BW, xx = ox + mypix % BW;
        if (yy < lv.H && xx < lv.W) {
            float *dst = grad_value +
                         (((size_t)b * S + lv.start + (size_t)yy * lv.W + xx) * H + h) * C +
                         m * VEC;
            if (item.w == 1) {
                *reinterpret_cast<float4 *>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
            } else {
#pragma unroll
                for (int c = 0; c < VEC; ++c) atomic_add(dst + c, acc[c]);
            }
        }
        __syncthreads();                                       // cur_item is reused
    }
}

}  // namespace boxattn
