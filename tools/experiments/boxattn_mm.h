// EXPERIMENT, NOT PART OF THE LIBRARY (round 2).  Kept for the record:
//   * measured at BoxeR-R50 COCO shapes (C2, bf16, model-like inputs): 68 us against 50 us of
//     pointgrad2_kernel (uniformly random locations: 119 against 80 us) -- the per-tile work
//     (window address arithmetic, result tile through LDS, 16 corner look-ups per tile) costs more
//     instructions than the 4-lane gathers it replaces, at 2 waves per SIMD (176 registers);
//   * results matched the gather kernel except for a rare, non-reproducible mismatch of the first
//     point of the levels owned by lanes 32-63 (~100 of 3.4 M points per run, different ones every
//     run) that extra waits / nops / a shuffle-only reduction did not remove -- unresolved;
//   * compiler note (hipcc 7.2): __builtin_bit_cast(bf16x2_t, v[i]) on an ELEMENT of an
//     ext_vector_type(4) vector selects element 0 for every i (the loads shrink to one dword);
//     copy the vector into a plain array first.
//
// Matrix-core kernels for the encoder case (one query per pixel of the packed multi-level map,
// Lq = S; box attention, bf16 storage, 32 channels per head, 2x2 points on 2 or 4 levels).
//
// The gather kernels (boxattn_gather2.h) spend their time on instruction issue and on the vector
// L1's miss path: per sample point four 64-byte rows are fetched (871 MB through the L1 per pass
// at BoxeR-R50 shapes) and 4 lanes x 48 VALU operations multiply-accumulate them.  Here a
// wavefront owns a PATCH of 8 x 4 neighbouring queries of one head, and the channel contraction
// is done once per (query, window pixel) pair on the matrix cores instead of once per sample
// corner on the VALU:
//
//   point gradients     S[q][px] = sum_c grad_out[q][c] * value[px][c]      (32 x K x 32)
//
// for the K pixels of the patch's BOUNDING WINDOW at a level (the exact bounding box of the
// valid footprints of its 32 x P points there, from the data: typically 14 x 10 = 140 pixels on
// the query's own level).  Both MFMA operands are rows in their natural layout -- the lane of
// (query | pixel, k-half) loads 16 bytes of its row straight from global memory, no LDS, no
// transposes -- so a window pixel's row is fetched once per patch and level (48 bytes per sample
// point instead of 256).  The 32 x 32 result tile goes to a wave-private LDS tile, from which
// every point picks the entries of its four corners, S_k = S[q][corner_k], and finishes
//   grad_w = sum_k w_k S_k,  grad_x = W a (hh (S2 - S1) + lh (S4 - S3)),  grad_y = H a (...)
// exactly like pointgrad2_kernel (reference box_attn_kernel.cuh:145-183 with the channel sum
// pulled out).  bf16 x bf16 products are exact in fp32 and the MFMA accumulates in fp32, so the
// sums differ from the VALU kernels' only by the fp32 summation order.
//
// The dense product pays while the window is small: a patch of a coarse query level looking at a
// fine value level has a window of thousands of pixels for its 128 points there (and such waves
// were a 70 us tail).  Windows of more than `max_tiles` 32-pixel tiles are therefore not
// multiplied out: the two lanes of a query (they hold the two halves of its upstream row as MFMA
// operand) fetch the halves of each corner row and reduce the dot products with v_dot2c -- the
// gather formulation with two lanes per (query, head) instead of four.
//
// Any input is handled exactly (the window is the exact bounding box; big windows take the direct
// path).  A non-finite value or upstream element only reaches the (query, pixel) entries it
// belongs to.
#pragma once
#include "boxattn_qgrid.h"

#ifndef BOXATTN_MM_EXTRA_NOPS
#define BOXATTN_MM_EXTRA_NOPS 0
#endif

namespace boxattn {

constexpr int kMmMaxLevels = 4;

struct MmLevel {
    int H, W, start;          // the level as value map and as query grid
    int npx, patch0;          // 8x4 query patches: per row, first patch id of the level
};
struct MmPlan {
    int L, n_patches;         // levels; patches per image
    int max_tiles;            // windows of more 32-pixel tiles than this use the direct path
    MmLevel lv[kMmMaxLevels];
};

typedef unsigned int mm_u32x4 __attribute__((ext_vector_type(4)));

// min / max of packed unsigned 16-bit pairs over each 32-lane HALF of the wave (result uniform
// inside a half)
template <bool MAX> __device__ __forceinline__ unsigned mm_half_pk(unsigned v)
{
    auto op = [](unsigned a, unsigned b) -> unsigned {
        const qg_u16x2 x = __builtin_bit_cast(qg_u16x2, a), y = __builtin_bit_cast(qg_u16x2, b);
        return __builtin_bit_cast(unsigned, MAX ? __builtin_elementwise_max(x, y)
                                                : __builtin_elementwise_min(x, y));
    };
#if BOXATTN_MM_EXTRA_NOPS == 3
    for (int o = 1; o < 32; o <<= 1) v = op(v, (unsigned)__shfl_xor((int)v, o, 64));
    return v;
#else
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true));
    return op(v, (unsigned)__shfl_xor((int)v, 16, 64));
#endif
}

// ---------------------------------------------------------------------------------------
// point gradients.  LH = levels per half-wave (L = 2 LH), P = 4 points per level.
// grid: H == 8: workgroup g = head g % 8 (= XCD), patches 4 (g / 8) + wave; else flat.
// ---------------------------------------------------------------------------------------
template <int LH>
__global__ __launch_bounds__(256) void pointgrad_mm_kernel(
    const bf16_t *__restrict__ value, const float *__restrict__ loc,
    const float *__restrict__ attn, const bf16_t *__restrict__ grad_out, MmPlan plan, int B, int S,
    int H, int Lq, float *__restrict__ grad_loc, float *__restrict__ grad_attn)
{
    constexpr int P = 4, L = 2 * LH, NPL = LH * P, LP = L * P, C = 32;
    constexpr int TS = 36;                                   // floats per row of the result tile
    __shared__ MmLevel lvs[kMmMaxLevels];
    __shared__ __attribute__((aligned(16))) float tile_all[4][32 * TS];
#pragma unroll
    for (int l = 0; l < kMmMaxLevels; ++l)
        if (threadIdx.x == l) lvs[l] = plan.lv[l];
    __syncthreads();

    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    float *tile = tile_all[wv];
    int h, pb;
    if (H == 8) {
        h = blockIdx.x % 8;
        pb = (blockIdx.x / 8) * 4 + wv;
    } else {
        const unsigned flat = blockIdx.x * 4u + wv;
        h = (int)(flat % (unsigned)H);
        pb = (int)(flat / (unsigned)H);
    }
    if (pb >= B * plan.n_patches) return;                    // wave-uniform
    const int b = pb / plan.n_patches, patch = pb % plan.n_patches;
    int lq = 0;
#pragma unroll
    for (int l = 1; l < L; ++l)
        if (patch >= lvs[l].patch0) lq = l;
    const int pp = patch - lvs[lq].patch0;
    const int j = lane & 31, half = lane >> 5;               // query of the patch; which levels
    const int qx = (pp % lvs[lq].npx) * 8 + (j & 7), qy = (pp / lvs[lq].npx) * 4 + (j >> 3);
    const bool active = qx < lvs[lq].W && qy < lvs[lq].H;
    const int q = lvs[lq].start + min(qy, lvs[lq].H - 1) * lvs[lq].W + min(qx, lvs[lq].W - 1);
    const size_t row = ((size_t)b * Lq + q) * H + h;         // (query, head) row of grad_out / out
    const size_t pt0 = row * LP + (size_t)half * NPL;        // the lane's first point

    // ---- my points: locations, weights (contiguous per lane), geometry
    float xs[NPL], ys[NPL], aw[NPL];
    {
        const float4 *l4 = reinterpret_cast<const float4 *>(loc + 2 * pt0);
        const float4 *a4 = reinterpret_cast<const float4 *>(attn + pt0);
#pragma unroll
        for (int i = 0; i < NPL / 2; ++i) {
            const float4 t = l4[i];
            xs[2 * i] = t.x; ys[2 * i] = t.y; xs[2 * i + 1] = t.z; ys[2 * i + 1] = t.w;
        }
#pragma unroll
        for (int i = 0; i < NPL / 4; ++i) {
            const float4 t = a4[i];
            aw[4 * i] = t.x; aw[4 * i + 1] = t.y; aw[4 * i + 2] = t.z; aw[4 * i + 3] = t.w;
        }
    }
    // the upstream row as MFMA A operand: lane (query j, k-half) holds channels 8 half + 0..7
    // (K-step 0) and 16 + 8 half + 0..7 (K-step 1)
    const mm_u32x4 g0 = *reinterpret_cast<const mm_u32x4 *>(grad_out + row * C + 8 * half);
    const mm_u32x4 g1 = *reinterpret_cast<const mm_u32x4 *>(grad_out + row * C + 16 + 8 * half);

    float lw[NPL], lh[NPL];
    int xy0[NPL];                                            // (x0 + 1) | (y0 + 1) << 16 | inside << 31
    unsigned lo[LH], hi[LH];                                 // my levels' bounding boxes
#pragma unroll
    for (int li = 0; li < LH; ++li) {
        const MmLevel vl = lvs[half * LH + li];
        unsigned mn = 0xFFFFFFFFu, mx = 0u;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int i = li * P + p;
            const Sample<float> sm = locate<float>(xs[i], ys[i], vl.H, vl.W);
            const bool in = active && sm.inside;
            lw[i] = sm.lw;
            lh[i] = sm.lh;
            xy0[i] = (sm.x0 + 1) | ((sm.y0 + 1) << 16) | (in ? (int)0x80000000 : 0);
            const unsigned xa = (unsigned)max(sm.x0, 0), xb = (unsigned)min(sm.x0 + 1, vl.W - 1);
            const unsigned ya = (unsigned)max(sm.y0, 0), yb = (unsigned)min(sm.y0 + 1, vl.H - 1);
            if (in) {
                const qg_u16x2 a = __builtin_bit_cast(qg_u16x2, mn), c = __builtin_bit_cast(qg_u16x2, xa | (ya << 16));
                const qg_u16x2 d = __builtin_bit_cast(qg_u16x2, mx), e = __builtin_bit_cast(qg_u16x2, xb | (yb << 16));
                mn = __builtin_bit_cast(unsigned, __builtin_elementwise_min(a, c));
                mx = __builtin_bit_cast(unsigned, __builtin_elementwise_max(d, e));
            }
        }
        lo[li] = mm_half_pk<false>(mn);
        hi[li] = mm_half_pk<true>(mx);
    }

    float gx[NPL], gy[NPL], ga[NPL];
#pragma unroll
    for (int i = 0; i < NPL; ++i) gx[i] = gy[i] = ga[i] = 0.f;

    // ---- level by level: S = G V^T over the window, 32 pixels at a time
#pragma unroll
    for (int l = 0; l < L; ++l) {
        constexpr int dummy = 0;
        (void)dummy;
        const int oh = l / LH, li = l % LH;                  // owning half, its level slot
        const unsigned blo = (unsigned)__builtin_amdgcn_readlane((int)lo[li], 32 * oh);
        const unsigned bhi = (unsigned)__builtin_amdgcn_readlane((int)hi[li], 32 * oh);
        const int x_lo = (int)(blo & 0xffffu), y_lo = (int)(blo >> 16);
        const int x_hi = (int)(bhi & 0xffffu), y_hi = (int)(bhi >> 16);
        if (x_lo > x_hi || y_lo > y_hi) continue;            // no point of the patch on this level
        const MmLevel vl = lvs[l];
        const int ww = x_hi - x_lo + 1, K = ww * (y_hi - y_lo + 1);
        const float rcp_ww = 1.0f / (float)ww;
        const bool mine = half == oh;
        // my points of this level: window index of the four corners (-1: corner outside the map)
        int cidx[P][4];
        float s[P][4];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int i = li * P + p;
            const int x0 = (xy0[i] & 0xffff) - 1, y0 = ((xy0[i] >> 16) & 0x7fff) - 1;
            const bool in = mine && xy0[i] < 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int yy = y0 + (k >> 1), xx = x0 + (k & 1);
                const bool ok = in && yy >= 0 && yy < vl.H && xx >= 0 && xx < vl.W;
                cidx[p][k] = ok ? (yy - y_lo) * ww + (xx - x_lo) : -1;
                s[p][k] = 0.f;
            }
        }
        const bf16_t *vbase = value + (((size_t)b * S + vl.start) * H + h) * C + 8 * half;
        const int n_tiles = (K + 31) / 32;
        if (n_tiles > plan.max_tiles) {                      // wave-uniform: direct path
#pragma unroll
            for (int p = 0; p < P; ++p) {
                mm_u32x4 r0[4], r1[4];
                int pix[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // the owner's corner pixel (window index -> map pixel), shared with its partner lane
                    int ky, kx;
                    divmod_small(max(cidx[p][k], 0), ww, rcp_ww, ky, kx);
                    const int mine_pix = cidx[p][k] >= 0 ? (y_lo + ky) * vl.W + x_lo + kx : -1;
                    const int other = __shfl_xor(mine_pix, 32, 64);
                    pix[k] = mine ? mine_pix : other;
                    const bf16_t *pr = vbase + (size_t)max(pix[k], 0) * H * C;
                    r0[k] = *reinterpret_cast<const mm_u32x4 *>(pr);
                    r1[k] = *reinterpret_cast<const mm_u32x4 *>(pr + 16);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // (plain arrays: __builtin_bit_cast of an ext_vector ELEMENT picks element 0 for
                    // every index with this compiler -- hipcc 7.2, checked in isolation)
                    const unsigned ga[4] = {g0.x, g0.y, g0.z, g0.w}, gb[4] = {g1.x, g1.y, g1.z, g1.w};
                    const unsigned ra[4] = {r0[k].x, r0[k].y, r0[k].z, r0[k].w};
                    const unsigned rb[4] = {r1[k].x, r1[k].y, r1[k].z, r1[k].w};
                    float acc = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, ga[i]),
                                                              __builtin_bit_cast(bf16x2_t, ra[i]), acc, false);
                        acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, gb[i]),
                                                              __builtin_bit_cast(bf16x2_t, rb[i]), acc, false);
                    }
                    acc = pix[k] >= 0 ? acc : 0.f;
                    s[p][k] = acc + __shfl_xor(acc, 32, 64);
                }
            }
        } else {
            auto v_rows = [&](int t, mm_u32x4 &v0, mm_u32x4 &v1) {
                const int k = min(32 * t + j, K - 1);            // pixel of this lane (clamped: unused columns)
                int ky, kx;
                divmod_small(k, ww, rcp_ww, ky, kx);
                const bf16_t *p = vbase + (size_t)((y_lo + ky) * vl.W + x_lo + kx) * H * C;
                v0 = *reinterpret_cast<const mm_u32x4 *>(p);
                v1 = *reinterpret_cast<const mm_u32x4 *>(p + 16);
            };
            mm_u32x4 v0, v1;
            v_rows(0, v0, v1);
            for (int t = 0; t < n_tiles; ++t) {
                mm_u32x4 n0, n1;
                v_rows(min(t + 1, n_tiles - 1), n0, n1);         // next tile in flight
                mfma_f32x16 d;
    #pragma unroll
                for (int r = 0; r < 16; ++r) d[r] = 0.f;
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, g0),
                                                            __builtin_bit_cast(mfma_bf16x8, v0), d, 0, 0, 0);
                d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(mfma_bf16x8, g1),
                                                            __builtin_bit_cast(mfma_bf16x8, v1), d, 0, 0, 0);
                // D[query i][pixel j]: lane = pixel j, registers = queries (r & 3) + 8 (r >> 2) + 4 half
                // -> tile[pixel][query], four consecutive queries per 16-byte store
#if BOXATTN_MM_EXTRA_NOPS == 1
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
#endif
                wave_lds_sync();                                 // previous tile's reads are done
    #pragma unroll
                for (int g4 = 0; g4 < 4; ++g4)
                    *reinterpret_cast<float4 *>(&tile[j * TS + 8 * g4 + 4 * half]) =
                        make_float4(d[4 * g4], d[4 * g4 + 1], d[4 * g4 + 2], d[4 * g4 + 3]);
#if BOXATTN_MM_EXTRA_NOPS == 6
                __builtin_amdgcn_s_waitcnt(0xc07f);              // lgkmcnt(0): the stores have landed
#endif
                wave_lds_sync();
    #pragma unroll
                for (int p = 0; p < P; ++p)
    #pragma unroll
                    for (int k = 0; k < 4; ++k)
#if BOXATTN_MM_EXTRA_NOPS == 2
                    {
                        const float tv = tile[(max(cidx[p][k], 0) & 31) * TS + j];
                        s[p][k] = (cidx[p][k] >> 5) == t ? tv : s[p][k];
                    }
#else
                        if ((cidx[p][k] >> 5) == t) s[p][k] = tile[(cidx[p][k] & 31) * TS + j];
#endif
                v0 = n0;
                v1 = n1;
#if BOXATTN_MM_EXTRA_NOPS == 4
                __builtin_amdgcn_s_waitcnt(0);
#endif
            }
        }
#if BOXATTN_MM_EXTRA_NOPS == 5
        __builtin_amdgcn_s_waitcnt(0);
#endif
        // finish my points of this level
        const float Wl = (float)vl.W, Hl = (float)vl.H;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const int i = li * P + p;
            const float hw = 1.f - lw[i], hh = 1.f - lh[i], a = aw[i];
            const float w1 = hh * hw, w2 = hh * lw[i], w3 = lh[i] * hw, w4 = lh[i] * lw[i];
            const float s1 = s[p][0], s2 = s[p][1], s3 = s[p][2], s4 = s[p][3];
            const bool in = mine && xy0[i] < 0;
            const float vs = w1 * s1 + w2 * s2 + w3 * s3 + w4 * s4;
            const float vx = Wl * a * (hh * (s2 - s1) + lh[i] * (s4 - s3));
            const float vy = Hl * a * (hw * (s3 - s1) + lw[i] * (s4 - s2));
            ga[i] = in ? vs : ga[i];
            gx[i] = in ? vx : gx[i];
            gy[i] = in ? vy : gy[i];
        }
    }

    if (active) {
        float4 *gl4 = reinterpret_cast<float4 *>(grad_loc + 2 * pt0);
        float4 *ga4 = reinterpret_cast<float4 *>(grad_attn + pt0);
#pragma unroll
        for (int i = 0; i < NPL / 2; ++i)
            gl4[i] = make_float4(gx[2 * i], gy[2 * i], gx[2 * i + 1], gy[2 * i + 1]);
#pragma unroll
        for (int i = 0; i < NPL / 4; ++i)
            ga4[i] = make_float4(ga[4 * i], ga[4 * i + 1], ga[4 * i + 2], ga[4 * i + 3]);
    }
}

}  // namespace boxattn
