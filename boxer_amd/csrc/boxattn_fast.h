// Fast kernels for the BoxeR head geometry: C = VEC * G channels per head, VEC channels per
// lane (one 8- or 16-byte access), G consecutive lanes per (query, head) pair, so a wave64
// covers 64/G consecutive (query, head) pairs -- for BoxeR (C=32, H=8, fp32: VEC=4, G=8)
// exactly one query with all of its heads: the four corner reads of a sample point are
// 128-byte row segments per head, the output row (H*C = 256 channels) is written as one
// contiguous 1 KiB (fp32) / 512 B (bf16) wave store, and loc / weights / their gradients of
// the wave's pairs are contiguous in memory.
//
// Differences from the generic kernels (same results up to fp32 summation order):
//   * corner reads are unconditional from a clamped address and discarded by select, so no
//     divergent branches sit between the loads (keeps many gathers in flight);
//   * the cross-channel sums of the backward are DPP row operations inside the G-lane
//     group -- no LDS, no barriers (the reference: smem + __syncthreads + thread-0 serial
//     sum per point, box_attn_kernel.cuh:443-463);
//   * d(val)/dx and d(val)/dy are factored as hh*(v2-v1)+lh*(v4-v3) and
//     hw*(v3-v1)+lw*(v4-v2); the common factors W_l*a are applied after the reduction.
#pragma once
#include "boxattn_device.h"

namespace boxattn {

// Level table -> LDS once per workgroup (the reference re-reads the int64 tables from
// global memory per thread per level, box_attn_kernel.cuh:313-316).
struct LevelTable {
    int h[kMaxLevels];
    int w[kMaxLevels];
    int start[kMaxLevels];
};

__device__ __forceinline__ void load_levels(LevelTable &t, const int64_t *shapes,
                                            const int64_t *lsi, int L)
{
    if ((int)threadIdx.x < L) {
        t.h[threadIdx.x] = (int)shapes[2 * threadIdx.x];
        t.w[threadIdx.x] = (int)shapes[2 * threadIdx.x + 1];
        t.start[threadIdx.x] = (int)lsi[threadIdx.x];
    }
    __syncthreads();
}

// The same in two steps, so that a kernel can put its own first loads BETWEEN the request of the
// level table and the barrier that publishes it: one global round trip at the start of a wave
// instead of two in a row (s_memtime stamps in the point-gradient kernel: 6 600 of a wave's
// 21 000 cycles passed before its loop started, tools/gpu_pg_trace.py).
struct LevelRegs { int h, w, start; };
__device__ __forceinline__ LevelRegs levels_request(const int64_t *shapes, const int64_t *lsi, int L)
{
    // unconditional loads from a clamped index: a load under a divergent branch makes the
    // compiler wait for it at the join (s_waitcnt right there: the round trip this split avoids)
    // ... and only the LOW dwords of the int64 entries (little endian; sizes < 2^31): loading the
    // whole words leaves dead high halves whose registers the compiler hands out again at once
    // -- with an s_waitcnt for the outstanding load in front of the first reuse
    const int i = min((int)threadIdx.x, L - 1);
    const int *sh32 = reinterpret_cast<const int *>(shapes), *ls32 = reinterpret_cast<const int *>(lsi);
    LevelRegs r;
    r.h = sh32[4 * i];
    r.w = sh32[4 * i + 2];
    r.start = ls32[2 * i];
    return r;
}
__device__ __forceinline__ void levels_commit(LevelTable &t, const LevelRegs &r, int L)
{
    if ((int)threadIdx.x < L) {
        t.h[threadIdx.x] = r.h;
        t.w[threadIdx.x] = r.w;
        t.start[threadIdx.x] = r.start;
    }
    __syncthreads();
}

// Workgroups are dispatched round-robin over the 8 XCDs (block b -> XCD b % 8, each with a
// private 4 MiB L2).  Consecutive queries sample neighbouring pixels, so give every XCD one
// contiguous chunk of the query range instead of every 8th block.  Bijective for any grid.
__device__ __forceinline__ unsigned xcd_chunked_block(unsigned bid, unsigned nblk)
{
    constexpr unsigned kXcd = 8;
    const unsigned q = nblk / kXcd, r = nblk % kXcd;
    const unsigned xcd = bid % kXcd, k = bid / kXcd;
    const unsigned first = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return first + k;
}

template <typename ST, int VEC>
__device__ __forceinline__ void gather_corner(const ST *lvl_base, int pix, size_t HC, bool ok,
                                              float (&v)[VEC])
{
    VecIO<ST, VEC>::ld(lvl_base + (size_t)pix * HC, v);
#pragma unroll
    for (int c = 0; c < VEC; ++c) v[c] = ok ? v[c] : 0.f;
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
template <typename ST, int VEC, int G, bool INST>
__global__ __launch_bounds__(256) void fwd_fast_kernel(
    const ST *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc,
    const float *__restrict__ w_sp, const float *__restrict__ w_lv, int S, int H, int L, int Lq,
    int P, ST *__restrict__ out, ST *__restrict__ mask, size_t n_qh)
{
    constexpr int C = VEC * G;
    constexpr int PAIRS = kWave / G;
    __shared__ LevelTable lv;
    load_levels(lv, shapes, lsi, L);

    const unsigned bid = xcd_chunked_block(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x & (kWave - 1);
    const size_t wave = (size_t)bid * (blockDim.x / kWave) + threadIdx.x / kWave;
    size_t qh = wave * PAIRS + lane / G;
    const bool active = qh < n_qh;
    if (!active) qh = n_qh - 1;                 // keep the lane's addresses valid; no stores
    const int cl = (lane % G) * VEC;
    const int m = (int)(qh % H);
    const size_t bq = qh / H;
    const size_t b = bq / Lq;
    const size_t HC = (size_t)H * C;
    const ST *vb = value + b * S * HC + (size_t)m * C + cl;
    const size_t pt0 = qh * L * P;

    float acc[VEC];
#pragma unroll
    for (int c = 0; c < VEC; ++c) acc[c] = 0.f;

    if constexpr (!INST) {
        for (int l = 0; l < L; ++l) {
            const int Hl = lv.h[l], Wl = lv.w[l];
            const ST *vl = vb + (size_t)lv.start[l] * HC;
            const float2 *lp = reinterpret_cast<const float2 *>(loc) + pt0 + (size_t)l * P;
            const float *ap = w_sp + pt0 + (size_t)l * P;
#pragma unroll 4
            for (int p = 0; p < P; ++p) {
                const float2 xy = lp[p];
                const float a = ap[p];
                const Sample<float> s = locate<float>(xy.x, xy.y, Hl, Wl);
                float v1[VEC], v2[VEC], v3[VEC], v4[VEC];
                gather_corner<ST, VEC>(vl, s.pix[0], HC, s.ok[0], v1);
                gather_corner<ST, VEC>(vl, s.pix[1], HC, s.ok[1], v2);
                gather_corner<ST, VEC>(vl, s.pix[2], HC, s.ok[2], v3);
                gather_corner<ST, VEC>(vl, s.pix[3], HC, s.ok[3], v4);
                const float w1 = s.hh * s.hw * a, w2 = s.hh * s.lw * a, w3 = s.lh * s.hw * a,
                            w4 = s.lh * s.lw * a;
#pragma unroll
                for (int c = 0; c < VEC; ++c)
                    acc[c] += w1 * v1[c] + w2 * v2[c] + w3 * v3[c] + w4 * v4[c];
            }
        }
    } else {
        ST *mk = mask + bq * P * HC + (size_t)m * C + cl;
        for (int p = 0; p < P; ++p) {
            float macc[VEC];
#pragma unroll
            for (int c = 0; c < VEC; ++c) macc[c] = 0.f;
#pragma unroll 4
            for (int l = 0; l < L; ++l) {
                const int Hl = lv.h[l], Wl = lv.w[l];
                const ST *vl = vb + (size_t)lv.start[l] * HC;
                const size_t i = pt0 + (size_t)l * P + p;
                const float2 xy = reinterpret_cast<const float2 *>(loc)[i];
                const float as = w_sp[i], al = w_lv[i];
                const Sample<float> s = locate<float>(xy.x, xy.y, Hl, Wl);
                float v1[VEC], v2[VEC], v3[VEC], v4[VEC];
                gather_corner<ST, VEC>(vl, s.pix[0], HC, s.ok[0], v1);
                gather_corner<ST, VEC>(vl, s.pix[1], HC, s.ok[1], v2);
                gather_corner<ST, VEC>(vl, s.pix[2], HC, s.ok[2], v3);
                gather_corner<ST, VEC>(vl, s.pix[3], HC, s.ok[3], v4);
                const float w1 = s.hh * s.hw, w2 = s.hh * s.lw, w3 = s.lh * s.hw,
                            w4 = s.lh * s.lw;
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    const float val = w1 * v1[c] + w2 * v2[c] + w3 * v3[c] + w4 * v4[c];
                    acc[c] += val * as;
                    macc[c] += val * al;
                }
            }
            if (active) VecIO<ST, VEC>::st(mk + (size_t)p * HC, macc);
        }
    }
    if (active) VecIO<ST, VEC>::st(out + qh * C + cl, acc);
}

// ---------------------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------------------
// SCATTER = false turns the kernel into the "point gradients" half of the binned backward
// (grad_loc / grad_weight only; grad_value comes from boxattn_binned.h without atomics).
template <typename ST, int VEC, int G, bool INST, bool SCATTER = true>
__global__ __launch_bounds__(256) void bwd_fast_kernel(
    const ST *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc,
    const float *__restrict__ w_sp, const float *__restrict__ w_lv,
    const ST *__restrict__ grad_out, const ST *__restrict__ grad_mask, int S, int H, int L,
    int Lq, int P, float *__restrict__ grad_value, float *__restrict__ grad_loc,
    float *__restrict__ grad_sp, float *__restrict__ grad_lv, size_t n_qh)
{
    constexpr int C = VEC * G;
    constexpr int PAIRS = kWave / G;
    __shared__ LevelTable lv;
    load_levels(lv, shapes, lsi, L);

    const unsigned bid = xcd_chunked_block(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x & (kWave - 1);
    const size_t wave = (size_t)bid * (blockDim.x / kWave) + threadIdx.x / kWave;
    size_t qh = wave * PAIRS + lane / G;
    const bool active = qh < n_qh;
    if (!active) qh = n_qh - 1;
    const int cl = (lane % G) * VEC;
    const bool writer = active && (lane % G) == 0;
    const int m = (int)(qh % H);
    const size_t bq = qh / H;
    const size_t b = bq / Lq;
    const size_t HC = (size_t)H * C;
    const size_t voff = b * S * HC + (size_t)m * C + cl;
    const size_t pt0 = qh * L * P;

    float g[VEC];
    VecIO<ST, VEC>::ld(grad_out + qh * C + cl, g);

    for (int l = 0; l < L; ++l) {
        const int Hl = lv.h[l], Wl = lv.w[l];
        const size_t lo = voff + (size_t)lv.start[l] * HC;
        const ST *vl = value + lo;
        float *gvl = grad_value + lo;
        for (int p = 0; p < P; ++p) {
            const size_t i = pt0 + (size_t)l * P + p;
            const float2 xy = reinterpret_cast<const float2 *>(loc)[i];
            const float as = w_sp[i];
            const float al = INST ? w_lv[i] : 0.f;
            const Sample<float> s = locate<float>(xy.x, xy.y, Hl, Wl);
            float v1[VEC], v2[VEC], v3[VEC], v4[VEC], gm[VEC];
            gather_corner<ST, VEC>(vl, s.pix[0], HC, s.ok[0], v1);
            gather_corner<ST, VEC>(vl, s.pix[1], HC, s.ok[1], v2);
            gather_corner<ST, VEC>(vl, s.pix[2], HC, s.ok[2], v3);
            gather_corner<ST, VEC>(vl, s.pix[3], HC, s.ok[3], v4);
            if constexpr (INST)
                VecIO<ST, VEC>::ld(grad_mask + (bq * P + p) * HC + (size_t)m * C + cl, gm);
            const float w1 = s.hh * s.hw, w2 = s.hh * s.lw, w3 = s.lh * s.hw, w4 = s.lh * s.lw;
            float ps = 0.f, pl = 0.f, px = 0.f, py = 0.f;
            float t[VEC];
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                t[c] = INST ? g[c] * as + gm[c] * al : g[c] * as;
                const float val = w1 * v1[c] + w2 * v2[c] + w3 * v3[c] + w4 * v4[c];
                ps += g[c] * val;
                if constexpr (INST) pl += gm[c] * val;
                const float dw = s.hh * (v2[c] - v1[c]) + s.lh * (v4[c] - v3[c]);
                const float dh = s.hw * (v3[c] - v1[c]) + s.lw * (v4[c] - v2[c]);
                px += dw * t[c];
                py += dh * t[c];
            }
            if (SCATTER && active) {
                if (s.ok[0]) {
                    float *d = gvl + (size_t)s.pix[0] * HC;
#pragma unroll
                    for (int c = 0; c < VEC; ++c) atomic_add(d + c, w1 * t[c]);
                }
                if (s.ok[1]) {
                    float *d = gvl + (size_t)s.pix[1] * HC;
#pragma unroll
                    for (int c = 0; c < VEC; ++c) atomic_add(d + c, w2 * t[c]);
                }
                if (s.ok[2]) {
                    float *d = gvl + (size_t)s.pix[2] * HC;
#pragma unroll
                    for (int c = 0; c < VEC; ++c) atomic_add(d + c, w3 * t[c]);
                }
                if (s.ok[3]) {
                    float *d = gvl + (size_t)s.pix[3] * HC;
#pragma unroll
                    for (int c = 0; c < VEC; ++c) atomic_add(d + c, w4 * t[c]);
                }
            }
            ps = group_sum<G>(ps);
            px = group_sum<G>(px);
            py = group_sum<G>(py);
            if constexpr (INST) pl = group_sum<G>(pl);
            if (writer) {
                grad_sp[i] = s.inside ? ps : 0.f;
                if constexpr (INST) grad_lv[i] = s.inside ? pl : 0.f;
                reinterpret_cast<float2 *>(grad_loc)[i] =
                    s.inside ? make_float2((float)Wl * px, (float)Hl * py) : make_float2(0.f, 0.f);
            }
        }
    }
}

}  // namespace boxattn
