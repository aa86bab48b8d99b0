// Dense (matrix-core) kernels for the ENCODER case of box attention: one query per pixel
// (Lq == S, packed levels), bf16 storage, C = 32 channels per head, 2x2 points per level.
//
// The gather kernels (boxattn_gather2.h) pull 4 corner rows per sample point through the vector
// L1 -- 64 rows per (query, head) -- and sit at the rate the L1 can deliver rows.
// In the encoder the queries are the pixels, a query's boxes lie around its own position on every
// level, so the rows a 4x4 TILE of queries touches on one level form a small window (12x12 pixels
// on the tile's own level, 7x7 / 5x5 / 4x4 on the coarser ones).  These kernels work on
// (tile, head) pairs, one wavefront each, and replace the per-point row gathers by products on the
// matrix cores over the window:
//
//   point gradients   S[q][pix] = sum_c G[q][c] V[pix][c]     (v_mfma_f32_16x16x32_bf16: M = the 16
//                     queries, N = 16 pixels of one window row, K = the 32 channels; both operands
//                     are natural 16-byte pieces of a grad_out / value row, straight from memory),
//                     S goes to wave-private LDS and every sample point picks its four corners:
//                     grad_w = sum_k w_k S_k, grad_x / grad_y from the corner differences
//                     (reference box_attn_kernel.cuh:145-183 with the channel sum pulled out).
//
// Lane = (query of the tile, point of the 2x2 grid); the levels are walked one after the other.
// The window of (tile, level) is placed by geometry alone (wave-uniform, no reductions): the tile's
// position projected onto the level plus a margin for the box size and its predicted offset.  That
// placement is a performance heuristic only: a point whose footprint is not inside the window takes
// the per-lane slow path (its four rows fetched and multiplied out by the lane itself), levels whose
// window would not fit (a coarse tile looking at a fine level) take it for every point -- results
// are the same either way, for any input.
#pragma once
#include "boxattn_device.h"
#include "boxattn_combine.h"      // CombineTail: the combine step's workers ride in this launch too
#include "boxattn_dense_plan.h"

namespace boxattn {

#ifndef BOXATTN_DENSE_DEBUG
#define BOXATTN_DENSE_DEBUG 0     // 1: the point-gradient kernel dumps its corner sums (DensePlan::dbg)
#endif
typedef unsigned int dense_u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 dense_bf16x8 __attribute__((ext_vector_type(8)));
typedef float dense_f32x4 __attribute__((ext_vector_type(4)));

// n / d, n % d for 0 <= n < 2^24 (float estimate + correction), d > 0
__device__ __forceinline__ void dense_divmod(unsigned n, int d, float rcp, unsigned &q, unsigned &r)
{
    int qi, ri;
    divmod_small((int)n, d, rcp, qi, ri);
    // the operands are wave-uniform, the float estimate runs on the VALU: tell the compiler the
    // results are uniform again (everything derived from them then stays in scalar registers)
    q = (unsigned)__builtin_amdgcn_readfirstlane(qi);
    r = (unsigned)__builtin_amdgcn_readfirstlane(ri);
}

// one 64-byte row (32 bf16 channels) as 16 words
__device__ __forceinline__ void dense_load_row(const bf16_t *p, unsigned (&w)[16])
{
    const dense_u32x4 *q = reinterpret_cast<const dense_u32x4 *>(p);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const dense_u32x4 t = q[i];
        w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w;
    }
}

// Which (tile, head group) a workgroup works on.  Workgroup w runs on XCD w % 8 (observed placement;
// only speed depends on it): every XCD gets one contiguous eighth of every level's tiles -- its L2
// then holds one spatial band of the maps -- and walks the levels coarsest first, because the tiles
// of the coarse levels are the slow ones (their fine-level points all take the slow path).
struct DenseTileId {
    int lq;                  // query level, -1: no tile
    unsigned b;
    int ty, tx, hg;
};
template <int L>
__device__ __forceinline__ DenseTileId dense_tile_of_block(const DensePlan &pl, unsigned block)
{
    DenseTileId t;
    t.lq = -1;
    const unsigned x = block & 7u;
    unsigned j, hg;
    dense_divmod(block >> 3, pl.hg, pl.rcp_hg, j, hg);
    t.hg = (int)hg;
    unsigned ti = 0;
#pragma unroll
    for (int l = L - 1; l >= 0; --l) {
        const unsigned n = (unsigned)pl.B * (unsigned)pl.lv[l].ntiles;
        const unsigned lo = (x * n) >> 3, hi = ((x + 1) * n) >> 3;
        const unsigned c = hi - lo;
        if (t.lq < 0) {
            if (j < c) {
                t.lq = l;
                ti = lo + j;
            } else {
                j -= c;
            }
        }
    }
    if (t.lq < 0) return t;
    const DenseLevel &Q = pl.lv[t.lq];
    unsigned tr, ty, tx;
    dense_divmod(ti, Q.ntiles, Q.rcp_ntiles, t.b, tr);
    dense_divmod(tr, Q.ntx, Q.rcp_ntx, ty, tx);
    t.ty = (int)ty;
    t.tx = (int)tx;
    return t;
}

// One workgroup = 4 wavefronts = 4 heads of one tile; the waves do not talk to each other (wave-
// private LDS, no workgroup barrier).
//   lds: S[plane 4][pixel 144][4 queries] float, S[q][pix] at plane q / 4, slot q % 4: the MFMA's
//   result registers (4 consecutive queries of one pixel per lane) go out as one 16-byte write per
//   lane, lanes of one instruction on consecutive 16-byte slots.
constexpr int kDenseLdsFloats = 4 * kDensePix * 4;

template <int L>
__global__ __launch_bounds__(256) void pointgrad_dense_kernel(
    const bf16_t *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ attn,
    const bf16_t *__restrict__ grad_out, float *__restrict__ grad_loc, float *__restrict__ grad_attn,
    DensePlan pl, unsigned value_bytes, unsigned tile_blocks, CombineTail ct)
{
    constexpr int C = 32, P = 4, LP = L * P;
    __shared__ __attribute__((aligned(16))) float lds_all[4][kDenseLdsFloats];
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    if (blockIdx.x >= tile_blocks) {                  // the appended combine workgroups (pointgrad2_kernel)
        if (ct.workers > 0) {
            const int w = (int)((blockIdx.x - tile_blocks) * 4 + wv);
            const int s = w / ct.workers;
            if (s < ct.plan.n_slices)
                combine_partials_body<bf16_t, C>(ct.combos, ct.n_items, ct.partials, ct.plan, pl.S, pl.H,
                                                 static_cast<bf16_t *>(ct.grad_value), s, w % ct.workers,
                                                 ct.workers, lane);
        }
        return;
    }
    float *lds = lds_all[wv];
#if BOXATTN_DENSE_DEBUG == 2
    const unsigned long long t_start = __builtin_amdgcn_s_memtime();
#endif

    const DenseTileId t = dense_tile_of_block<L>(pl, blockIdx.x);
    if (t.lq < 0) return;
    const int H = pl.H, h = t.hg * 4 + wv;
    if (h >= H) return;                                            // wave-uniform
    const DenseLevel Q = pl.lv[t.lq];

    // ---- lane -> (query of the tile, point)
    const int qi = lane >> 2, p = lane & 3;
    const int qy = t.ty * kDenseTile + (qi >> 2), qx = t.tx * kDenseTile + (qi & 3);
    const bool vq = qy < Q.H && qx < Q.W;
    const unsigned q = (unsigned)(Q.start + min(qy, Q.H - 1) * Q.W + min(qx, Q.W - 1));
    const unsigned qh = (t.b * (unsigned)pl.Lq + q) * (unsigned)H + (unsigned)h;
    const unsigned pt0 = qh * (unsigned)LP;
    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);
    float2 xy[L];
    float a[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        xy[l] = loc2[pt0 + l * P + p];
        a[l] = attn[pt0 + l * P + p];
    }
    // ---- the A operand: row i = lane & 15 (query i of the tile), channels 8 (lane >> 4) ..
    dense_bf16x8 gfrag;
    {
        const int i = lane & 15;
        const int yi = min(t.ty * kDenseTile + (i >> 2), Q.H - 1);
        const int xi = min(t.tx * kDenseTile + (i & 3), Q.W - 1);
        const unsigned qhi =
            (t.b * (unsigned)pl.Lq + (unsigned)(Q.start + yi * Q.W + xi)) * (unsigned)H + (unsigned)h;
        gfrag = __builtin_bit_cast(dense_bf16x8, *reinterpret_cast<const dense_u32x4 *>(
                                                     grad_out + (size_t)qhi * C + (lane >> 4) * 8));
    }
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(value), 0, value_bytes, 0x00020000);

    // The window of level l: wave-uniform placement, its rows fetched in chunks of 4 (rows past the
    // window's last one repeat it: their results land in unused rows of the LDS tile).  The loads of
    // level l + 1 are issued before the look-ups of level l, so that a wave always has one level's
    // rows in flight behind its arithmetic.
    struct Win { int rows, cols, x0, y0; };
    Win win[L];
    dense_u32x4 rowreg[kDenseWin];
    auto issue = [&](int l) {
        const DenseLevel T = pl.lv[l];
        const DenseWin w = pl.win[t.lq][l];
        Win &o = win[l];
        o.rows = w.rows;
        o.cols = w.cols;
        const int x0 = (int)floorf((float)t.tx * w.ax + w.bx), y0 = (int)floorf((float)t.ty * w.ay + w.by);
        o.x0 = __builtin_amdgcn_readfirstlane(max(0, min(x0, T.W - w.cols)));
        o.y0 = __builtin_amdgcn_readfirstlane(max(0, min(y0, T.H - w.rows)));
        const int jx = min(o.x0 + (lane & 15), T.W - 1);
        const unsigned voff =
            ((((t.b * (unsigned)pl.S + (unsigned)(T.start + o.y0 * T.W + jx)) * (unsigned)H + (unsigned)h) * C) +
             (unsigned)(lane >> 4) * 8u) * 2u;
        const unsigned row_bytes = (unsigned)T.W * (unsigned)H * (C * 2u);
#pragma unroll
        for (int c = 0; c < kDenseWin / 4; ++c)
            if (c * 4 < o.rows) {                                  // wave-uniform
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int n = min(c * 4 + k, o.rows - 1);
                    const unsigned soff = (unsigned)(min(o.y0 + n, T.H - 1) - o.y0) * row_bytes;
                    rowreg[c * 4 + k] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff, soff, 0);
                }
            }
    };
    auto consume = [&](int l) {                                    // rows -> S tile in LDS
        const Win &o = win[l];
        const int j = lane & 15;
        float *dst = lds + (((lane >> 4) * kDensePix + j) << 2);
        const bool wr_lane = j < kDenseStride;
#pragma unroll
        for (int c = 0; c < kDenseWin / 4; ++c)
            if (c * 4 < o.rows) {
                dense_f32x4 acc[4];
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    acc[k] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(
                        gfrag, __builtin_bit_cast(dense_bf16x8, rowreg[c * 4 + k]),
                        dense_f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                if (wr_lane) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        *reinterpret_cast<dense_f32x4 *>(dst + (c * 4 + k) * (kDenseStride * 4)) = acc[k];
                }
            }
    };

    // BOXATTN_DENSE_DEBUG == 2: s_memtime stamps of every wave (tools/gpu_dense_trace.py): 0 start,
    // 1 prologue loads issued, then per level {rows consumed, look-ups done, level done}, last: end
#if BOXATTN_DENSE_DEBUG == 2
    unsigned long long ts[16];
    int ts_n = 0;
#define DENSE_STAMP() do { __builtin_amdgcn_sched_barrier(0); ts[ts_n++] = __builtin_amdgcn_s_memtime(); \
                           __builtin_amdgcn_sched_barrier(0); } while (0)
    ts[ts_n++] = t_start;
#else
#define DENSE_STAMP() do { } while (0)
#endif
    float ga[L], gx[L], gy[L];
    issue(0);
    DENSE_STAMP();
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const DenseLevel T = pl.lv[l];
        const Sample<float> s = locate<float>(xy[l].x, xy[l].y, T.H, T.W);
        const int wr = win[l].rows, wc = win[l].cols, wx0 = win[l].x0, wy0 = win[l].y0;
        if (wr > 0) {
            wave_lds_sync();                                       // the previous level's look-ups are done
            consume(l);
            wave_lds_sync();
        }
        DENSE_STAMP();
        if (l + 1 < L) issue(l + 1);
        // ---- the point's four corners
        float s1, s2, s3, s4;
        bool slow = false;
        {
            const int r0 = s.y0 - wy0, c0 = s.x0 - wx0;            // window coordinates of corner 1
            const bool in_r0 = r0 >= 0 && r0 < wr, in_r1 = r0 + 1 >= 0 && r0 + 1 < wr;
            const bool in_c0 = c0 >= 0 && c0 < wc, in_c1 = c0 + 1 >= 0 && c0 + 1 < wc;
            const bool i1 = in_r0 && in_c0, i2 = in_r0 && in_c1, i3 = in_r1 && in_c0, i4 = in_r1 && in_c1;
            slow = vq && ((s.ok[0] && !i1) || (s.ok[1] && !i2) || (s.ok[2] && !i3) || (s.ok[3] && !i4));
            const int rr0 = min(max(r0, 0), kDenseWin - 1), rr1 = min(max(r0 + 1, 0), kDenseWin - 1);
            const int cc0 = min(max(c0, 0), kDenseWin - 1), cc1 = min(max(c0 + 1, 0), kDenseWin - 1);
            const float *src = lds + (qi >> 2) * (kDensePix * 4) + (qi & 3);
            const float v1 = src[(rr0 * kDenseStride + cc0) * 4], v2 = src[(rr0 * kDenseStride + cc1) * 4];
            const float v3 = src[(rr1 * kDenseStride + cc0) * 4], v4 = src[(rr1 * kDenseStride + cc1) * 4];
            s1 = (s.ok[0] && i1) ? v1 : 0.f;
            s2 = (s.ok[1] && i2) ? v2 : 0.f;
            s3 = (s.ok[2] && i3) ? v3 : 0.f;
            s4 = (s.ok[3] && i4) ? v4 : 0.f;
        }
#if BOXATTN_DENSE_DEBUG == 2
        asm volatile("" ::"v"(s1), "v"(s2), "v"(s3), "v"(s4));
#endif
        DENSE_STAMP();
        if (__builtin_amdgcn_ballot_w64(slow) != 0ull) {           // wave-uniform
            if (slow) {
                // the lane's own dot products: its query's grad_out row against the four corner rows
                // (the words go through plain arrays: a bit_cast of element i of an ext_vector
                // selects element 0 for every i with this compiler)
                unsigned gw[16];
                dense_load_row(grad_out + (size_t)qh * C, gw);
                const unsigned row0 = t.b * (unsigned)pl.S + (unsigned)T.start;
                float sk[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    unsigned vw[16];
                    dense_load_row(value + ((size_t)(row0 + (unsigned)s.pix[k]) * H + h) * C, vw);
                    float d = 0.f;
#pragma unroll
                    for (int i = 0; i < 16; ++i)
                        d = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, gw[i]),
                                                            __builtin_bit_cast(bf16x2_t, vw[i]), d, false);
                    sk[k] = s.ok[k] ? d : 0.f;
                }
                s1 = sk[0]; s2 = sk[1]; s3 = sk[2]; s4 = sk[3];
            }
        }
        if (BOXATTN_DENSE_DEBUG == 1 && pl.dbg && vq) {
            float *o = pl.dbg + (size_t)(pt0 + l * P + p) * 8;
            o[0] = s1; o[1] = s2; o[2] = s3; o[3] = s4;
            o[4] = slow ? 1.f : 0.f;
            o[5] = xy[l].x; o[6] = (float)s.x0; o[7] = (float)wx0;
        }
        // ---- finish
        const float w1 = s.hh * s.hw, w2 = s.hh * s.lw, w3 = s.lh * s.hw, w4 = s.lh * s.lw;
        const float gs_ = w1 * s1 + w2 * s2 + w3 * s3 + w4 * s4;
        const float gx_ = (float)T.W * a[l] * (s.hh * (s2 - s1) + s.lh * (s4 - s3));
        const float gy_ = (float)T.H * a[l] * (s.hw * (s3 - s1) + s.lw * (s4 - s2));
        ga[l] = s.inside ? gs_ : 0.f;
        gx[l] = s.inside ? gx_ : 0.f;
        gy[l] = s.inside ? gy_ : 0.f;
#if BOXATTN_DENSE_DEBUG == 2
        asm volatile("" ::"v"(ga[l]), "v"(gx[l]), "v"(gy[l]));
#endif
        DENSE_STAMP();
    }
    // ---- results: lane (q, p) holds its point on every level; memory wants, per (query, head),
    //      [level][point] runs -- transposed through LDS so that lane (q, j) writes level j's 4 points
    //      as 16 + 32 contiguous bytes (whole 64- / 128-byte runs per query)
    wave_lds_sync();
    float *res_a = lds + qi * LP, *res_xy = lds + 16 * LP + qi * LP * 2;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        res_a[l * P + p] = ga[l];
        *reinterpret_cast<float2 *>(res_xy + (l * P + p) * 2) = make_float2(gx[l], gy[l]);
    }
    wave_lds_sync();
    if (vq && p < L) {
        const float4 o_a = *reinterpret_cast<const float4 *>(res_a + p * P);
        const float4 o_0 = *reinterpret_cast<const float4 *>(res_xy + p * P * 2);
        const float4 o_1 = *reinterpret_cast<const float4 *>(res_xy + p * P * 2 + 4);
        *reinterpret_cast<float4 *>(grad_attn + pt0 + p * P) = o_a;
        float4 *gl = reinterpret_cast<float4 *>(grad_loc + 2 * (size_t)(pt0 + p * P));
        gl[0] = o_0;
        gl[1] = o_1;
    }
#if BOXATTN_DENSE_DEBUG == 2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DENSE_STAMP();
    if (pl.dbg && lane == 0) {
        float *o = pl.dbg + ((size_t)blockIdx.x * 4 + wv) * 20;
        o[0] = (float)t.lq;
        o[1] = (float)(unsigned)(ts[0] & 0xffffffu);
        for (int i = 1; i < ts_n; ++i) o[1 + i] = (float)(unsigned)(ts[i] - ts[0]);
        o[19] = (float)ts_n;
    }
#endif
}

}  // namespace boxattn
