// Window-staged kernels for the encoder case of FLOAT32 box attention -- the reference's own arithmetic
// (box_attention_func.py:11: the CUDA op computes in float32 whatever autocast says).
//
// Rounds 1-4 ran float32 on the row-gather kernels (boxattn_gather2.h: forward 64 us, point gradients 82 us at C2)
// and kept the window-staged formulation of boxattn_dense.h for bf16 storage only: the staged bf16 FORWARD is a
// matrix-core kernel (transposing 16-bit LDS reads, v_mfma_f32_4x4x4_16B_bf16) that has no float32 counterpart
// worth having.  The staged POINT-GRADIENT kernel, however, is plain VALU work on rows read from
// LDS -- lane = one sample point, four corner rows, dot products with the query's upstream row -- and so is a
// forward written the same way (lane = one sample point, four corner rows, weighted sum into 32 accumulators,
// the quad's four points summed at the end).  Both carry over to float32 with 128-byte pixels: a tile's windows are
// 2 x 25.4 KB at BoxeR-R50 shapes, three workgroups per CU instead of five, and the kernels trade the gather
// kernels' 64 L1 row requests per (query, head) for LDS reads.
//
// Same tiles, same window placement (geometry only, results never depend on it), same riders, same global path for
// points whose footprint is not inside a staged window as the bf16 kernels (boxattn_dense.h); the plan's window
// geometry is in units of a quarter pixel (32 bytes here, 16 there).  float32 throughout: products and sums in the
// order of the reference's kernel up to the association of the per-point sums (within 1e-4 of the float64 goldens,
// as the gather kernels).
#pragma once
#include "boxattn_dense.h"

namespace boxattn {

constexpr int kDenseF32Slot = 128;                 // one staged pixel: 32 float32 channels
constexpr int kDenseF32Unit = 32;                  // bytes per unit of DenseWin::geo's pitch / offset fields
// (BOXATTN_DENSE_F32_LDS -- 52 KB: three workgroups per CU, BoxeR-R50 tiles need 50.8 KB -- is defined ONCE, in
// boxattn_dense_plan.h, next to the host's window budget that must agree with it)
#ifndef BOXATTN_DENSE_F32_SKEW
#define BOXATTN_DENSE_F32_SKEW 32
#endif
constexpr int kDenseF32LdsBytes = BOXATTN_DENSE_F32_LDS;
constexpr int kDenseF32ZeroOff = kDenseF32LdsBytes - kDenseF32Slot;    // the forward's row of zeros (make_dense_plan leaves it free)

struct DenseWinPosF32 {
    unsigned geo;
    int x0, y0;
    __device__ __forceinline__ int rows() const { return (int)(geo & 31u); }
    __device__ __forceinline__ int cols() const { return (int)((geo >> 5) & 31u); }
    // Bytes between window rows: the row's pixels + a skew of 32 bytes (one unit of the host's pitch field).  A lane reads
    // a corner row as eight 16-byte pieces, every lane of an instruction piece i of ITS row, and rows and pixels start at
    // multiples of 32 bytes: only 8 of the 16 bank groups ever hold a piece i (half of the kernels' LDS-busy time is bank
    // conflicts, profiles/r05_pmc_sq_fp32.txt).  A skew of 16 bytes -- consecutive rows then step through all 16 groups --
    // was measured in round 6 (-DBOXATTN_DENSE_F32_SKEW=16, same box, C2 / C2' float32): 210.0 / 342.9 us a step against
    // 210.7 / 342.5: nothing (profiles/r06_f32_skew.log) -- the lanes of a wave read few distinct ROWS (a 4 x 4 query
    // sub-tile), what collides are pixels of one row, 128 bytes apart, and a direct-to-LDS load cannot pad between the
    // pixels it writes (lane-linear destination).
    __device__ __forceinline__ int pitchb() const { return cols() * kDenseF32Slot + BOXATTN_DENSE_F32_SKEW; }
    __device__ __forceinline__ int offb() const { return (int)(geo >> 17) * kDenseF32Unit; }
};

// Placement + cooperative fetch (dense_stage_issue for 128-byte pixels): a wave-load of 64 x 16 bytes is 8 pixels, a
// window row of up to 16 pixels two of them; wave w takes the rows w, w + 4, ...
template <int L>
__device__ __forceinline__ void dense_f32_stage_issue(const DenseHot<L> &hot, const DenseWin (&wrow)[L], const DenseTileId &t,
                                                      int lane, int wv, __amdgpu_buffer_rsrc_t rs, unsigned char *lds,
                                                      DenseWinPosF32 (&win)[L])
{
    constexpr int C = 32, RPW = kDenseWinMax / 4;
    const int j = lane >> 3, chunk = lane & 7;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const DenseMap T = hot.lv[l];
        const DenseWin w = wrow[l];
        DenseWinPosF32 &o = win[l];
        o.geo = w.geo;
        const int x0 = (t.tx * w.ax + w.bx) >> 16, y0 = (t.ty * w.ay + w.by) >> 16;
        o.x0 = max(0, min(x0, T.W - o.cols()));
        o.y0 = max(0, min(y0, T.H - o.rows()));
        const unsigned row_bytes = (unsigned)T.W * (unsigned)hot.H * (C * 4u);
        const int rows = min(o.rows(), T.H - o.y0);
#pragma unroll
        for (int half = 0; half < 2; ++half) {                     // pixels 0-7 / 8-15 of the window row
            const int jj = j + 8 * half;
            const int jx = min(o.x0 + jj, T.W - 1);
            const unsigned voff =
                ((((t.b * (unsigned)hot.S + (unsigned)(T.start + jx)) * (unsigned)hot.H + (unsigned)t.h) * C) +
                 (unsigned)chunk * 4u) * 4u;
            unsigned soff = (unsigned)(o.y0 + wv) * row_bytes;
            int dst = o.offb() + wv * o.pitchb() + half * 1024;
            const int step = 4 * o.pitchb();
            if (8 * half < o.cols() && jj < o.cols()) {            // (first condition wave-uniform)
#pragma unroll
                for (int k = 0; k < RPW; ++k) {
                    if (wv + 4 * k < rows)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (dense_lds_void *)(lds + dst), 16, voff, soff, 0, 0);
                    soff += 4u * row_bytes;
                    dst += step;
                }
            }
        }
    }
}

// one 128-byte row (32 float32 channels)
template <typename PTR> __device__ __forceinline__ void dense_f32_load_row(PTR p, float (&w)[32])
{
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float4 t = reinterpret_cast<const float4 *>(p)[i];
        w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w;
    }
}
__device__ __forceinline__ float dense_f32_dot_row(const float (&g)[32], const float (&v)[32])
{
    float d0 = 0.f, d1 = 0.f;                      // two chains
#pragma unroll
    for (int i = 0; i < 32; i += 2) {
        d0 = fmaf(g[i], v[i], d0);
        d1 = fmaf(g[i + 1], v[i + 1], d1);
    }
    return d0 + d1;
}

// which corners of a located point count, and how far its footprint is inside the staged window (d < 0: not)
struct DenseFoot {
    unsigned m[4];           // all-ones where the corner lies inside the map (and the point inside the window test)
    int ra, rb, ca, cb;      // rows / columns of the map the footprint needs (clamped into the map)
    int d;
};
__device__ __forceinline__ DenseFoot dense_f32_foot(const DensePoint &s, const DenseMap &T, const DenseWinPosF32 &o)
{
    DenseFoot f;
    const int Hm1 = T.H - 1, Wm1 = T.W - 1;
    const unsigned mi = s.inside ? 0xffffffffu : 0u;
    const unsigned mr0 = ~(unsigned)(s.y0 >> 31) & mi, mr1 = (unsigned)((s.y0 - Hm1) >> 31) & mi;
    const unsigned mc0 = ~(unsigned)(s.x0 >> 31), mc1 = (unsigned)((s.x0 - Wm1) >> 31);
    f.m[0] = mr0 & mc0; f.m[1] = mr0 & mc1; f.m[2] = mr1 & mc0; f.m[3] = mr1 & mc1;
    f.ra = max(s.y0, 0); f.rb = min(s.y0 + 1, Hm1); f.ca = max(s.x0, 0); f.cb = min(s.x0 + 1, Wm1);
    const int rows = o.rows(), cols = o.cols();
    f.d = min(min(f.ra - o.y0, o.y0 + rows - 1 - f.rb), min(f.ca - o.x0, o.x0 + cols - 1 - f.cb));
    return f;
}
// LDS byte offsets of the four corner rows of a footprint that lies inside the window (clamped for lanes whose does not)
__device__ __forceinline__ void dense_f32_slots(const DensePoint &s, const DenseWinPosF32 &o, int (&slot)[4])
{
    const int rows = o.rows(), cols = o.cols(), pitchb = o.pitchb(), offb = o.offb();
    const int rr0 = min(max(s.y0 - o.y0, 0), rows - 1), rr1 = min(max(s.y0 + 1 - o.y0, 0), rows - 1);
    const int cc0 = min(max(s.x0 - o.x0, 0), cols - 1), cc1 = min(max(s.x0 + 1 - o.x0, 0), cols - 1);
    const int rowb0 = offb + __mul24(rr0, pitchb), rowb1 = offb + __mul24(rr1, pitchb);
    const int colb0 = cc0 * kDenseF32Slot, colb1 = cc1 * kDenseF32Slot;
    slot[0] = rowb0 + colb0; slot[1] = rowb0 + colb1; slot[2] = rowb1 + colb0; slot[3] = rowb1 + colb1;
}

// ---------------------------------------------------------------------------------------------------------
// point gradients
// ---------------------------------------------------------------------------------------------------------
template <int L>
__global__ __launch_bounds__(256, 3) void pointgrad_dense_f32_kernel(
    const float *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ attn,
    const float *__restrict__ grad_out, float *__restrict__ grad_loc, float *__restrict__ grad_attn,
    DensePlan pl, unsigned value_bytes, BinRide ride)
{
    constexpr int C = 32, P = 4, LP = L * P;
    __shared__ __attribute__((aligned(16))) unsigned char win_lds[kDenseF32LdsBytes];
    static_assert(4 * kDenseResFloats * sizeof(float) <= sizeof(win_lds), "the result tiles reuse the window buffer");
    static_assert(kRideLdsInts * sizeof(int) <= sizeof(win_lds), "so do the riders");
    const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const RideRole role = ride_role(blockIdx.x, ride.grid);
    if (role.rider) {
        bin_fill_ride<256>(ride, role.id, reinterpret_cast<int *>(win_lds));
        return;
    }
    DenseHot<L> hot;
    DenseMap Q;
    DenseWin wrow[L];
    const DenseTileId t = dense_tile_of_block<L>(pl, role.id, hot, Q, wrow);
    if (t.lq < 0) return;                                          // workgroup-uniform
    const int H = hot.H, h = t.h;
    const int qi = lane >> 2, p = lane & 3;
    const int qy = t.ty * kDenseTile + (wv >> 1) * kDenseSub + (qi >> 2);
    const int qx = t.tx * kDenseTile + (wv & 1) * kDenseSub + (qi & 3);
    const bool vq = qy < Q.H && qx < Q.W;
    const unsigned q = (unsigned)(Q.start + min(qy, Q.H - 1) * Q.W + min(qx, Q.W - 1));
    const unsigned qh = (t.b * (unsigned)hot.Lq + q) * (unsigned)H + (unsigned)h;
    const unsigned pt0 = qh * (unsigned)LP;
    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(value), 0, value_bytes, 0x00020000);
    DenseWinPosF32 win[L];
    dense_f32_stage_issue<L>(hot, wrow, t, lane, wv, rs, win_lds, win);
    float2 xy[L];
    float a[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        xy[l] = loc2[pt0 + l * P + p];
        a[l] = attn[pt0 + l * P + p];
    }
    float gw[32];                                                  // the query's grad_out row
    dense_f32_load_row(grad_out + (size_t)qh * C, gw);
    dense_stage_wait();                                            // windows complete

    float ga[L], gx[L], gy[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const DenseMap T = hot.lv[l];
        const DenseWinPosF32 &o = win[l];
        const DensePoint s = dense_locate(xy[l].x, xy[l].y, T.H, T.W);
        const DenseFoot f = dense_f32_foot(s, T, o);
        const bool act = vq && s.inside;
        const bool slow = act && f.d < 0, fast = act && f.d >= 0;
        float sk[4];
        if (__builtin_amdgcn_ballot_w64(fast) != 0ull) {           // wave-uniform: somebody reads the window
            int slot[4];
            dense_f32_slots(s, o, slot);
#pragma unroll
            for (int k = 0; k < 4; ++k) {                          // a corner at a time: 8 LDS reads, then the 32 products
                float va[32];
                dense_f32_load_row(win_lds + slot[k], va);
                sk[k] = __uint_as_float(__float_as_uint(dense_f32_dot_row(gw, va)) & f.m[k]);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) sk[k] = 0.f;
        }
        if (__builtin_amdgcn_ballot_w64(slow) != 0ull) {           // wave-uniform: the global path
            // one point of every quad at a time: the quad's lanes fetch the four corner rows of point tp together,
            // 32 bytes each, and sum their partial dot products with DPP adds (as the bf16 kernel)
            float gch[8];                                          // this lane's 8 channels of the grad_out row
#pragma unroll
            for (int i = 0; i < 8; ++i) gch[i] = p == 0 ? gw[i] : p == 1 ? gw[8 + i] : p == 2 ? gw[16 + i] : gw[24 + i];
            const unsigned row0 = t.b * (unsigned)hot.S + (unsigned)T.start;
            const int pra = __mul24(f.ra, T.W), prb = __mul24(f.rb, T.W);
            const int pix[4] = {pra + f.ca, pra + f.cb, prb + f.ca, prb + f.cb};
            unsigned off[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) off[k] = (unsigned)(((row0 + (unsigned)pix[k]) * H + h) * (C * 4));
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                float4 v[4][2];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 *src = reinterpret_cast<const float4 *>(
                        reinterpret_cast<const char *>(value) + quad_bcast_u32(off[k], tp) + (unsigned)p * 32u);
                    v[k][0] = src[0];
                    v[k][1] = src[1];
                }
                float part[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float dd = gch[0] * v[k][0].x;
                    dd = fmaf(gch[1], v[k][0].y, dd); dd = fmaf(gch[2], v[k][0].z, dd); dd = fmaf(gch[3], v[k][0].w, dd);
                    dd = fmaf(gch[4], v[k][1].x, dd); dd = fmaf(gch[5], v[k][1].y, dd);
                    dd = fmaf(gch[6], v[k][1].z, dd); dd = fmaf(gch[7], v[k][1].w, dd);
                    part[k] = group_sum<4>(dd);
                }
                if (p == tp && slow) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) sk[k] = __uint_as_float(__float_as_uint(part[k]) & f.m[k]);
                }
            }
        }
        const float w1 = s.hh * s.hw, w2 = s.hh * s.lw, w3 = s.lh * s.hw, w4 = s.lh * s.lw;
        const float gs_ = w1 * sk[0] + w2 * sk[1] + w3 * sk[2] + w4 * sk[3];
        const float gx_ = (float)T.W * a[l] * (s.hh * (sk[1] - sk[0]) + s.lh * (sk[3] - sk[2]));
        const float gy_ = (float)T.H * a[l] * (s.hw * (sk[2] - sk[0]) + s.lw * (sk[3] - sk[1]));
        ga[l] = s.inside ? gs_ : 0.f;
        gx[l] = s.inside ? gx_ : 0.f;
        gy[l] = s.inside ? gy_ : 0.f;
        asm volatile("" : "+v"(ga[l]), "+v"(gx[l]), "+v"(gy[l]));
    }
    // ---- results through (wave-private) LDS: lane (q, j) writes level j's 4 points as 16 + 32 contiguous bytes
    __syncthreads();                                               // every wave is done with the windows
    float *res = reinterpret_cast<float *>(win_lds) + wv * kDenseResFloats;
    float *res_a = res + qi * LP, *res_xy = res + 16 * LP + qi * LP * 2;
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
        float4 *ga4 = reinterpret_cast<float4 *>(grad_attn + pt0 + p * P);
        float4 *gl = reinterpret_cast<float4 *>(grad_loc + 2 * (size_t)(pt0 + p * P));
        typedef float pg_f32x4 __attribute__((ext_vector_type(4)));
        __builtin_nontemporal_store(pg_f32x4{o_a.x, o_a.y, o_a.z, o_a.w}, reinterpret_cast<pg_f32x4 *>(ga4));
        __builtin_nontemporal_store(pg_f32x4{o_0.x, o_0.y, o_0.z, o_0.w}, reinterpret_cast<pg_f32x4 *>(gl));
        __builtin_nontemporal_store(pg_f32x4{o_1.x, o_1.y, o_1.z, o_1.w}, reinterpret_cast<pg_f32x4 *>(gl + 1));
    }
}

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
template <int L>
__global__ __launch_bounds__(256, 3) void fwd_dense_f32_kernel(
    const float *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ attn,
    float *__restrict__ out, DensePlan pl, unsigned value_bytes, BinRide ride,
    unsigned long long *__restrict__ stats)
{
    constexpr int C = 32, P = 4, LP = L * P;
    __shared__ __attribute__((aligned(16))) unsigned char win_lds[kDenseF32LdsBytes];
    static_assert(256 * 33 * sizeof(float) <= sizeof(win_lds), "the result tiles reuse the window buffer");
    const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const RideRole role = ride_role(blockIdx.x, ride.grid);
    if (role.rider) {
        bin_count_ride<256>(ride, role.id, reinterpret_cast<int *>(win_lds));
        return;
    }
    DenseHot<L> hot;
    DenseMap Q;
    DenseWin wrow[L];
    const DenseTileId t = dense_tile_of_block<L>(pl, role.id, hot, Q, wrow);
    if (t.lq < 0) return;
    const int H = hot.H, h = t.h;
    const int qi = lane >> 2, p = lane & 3;
    const int qy = t.ty * kDenseTile + (wv >> 1) * kDenseSub + (qi >> 2);
    const int qx = t.tx * kDenseTile + (wv & 1) * kDenseSub + (qi & 3);
    const bool vq = qy < Q.H && qx < Q.W;
    const unsigned q = (unsigned)(Q.start + min(qy, Q.H - 1) * Q.W + min(qx, Q.W - 1));
    const unsigned qh = (t.b * (unsigned)hot.Lq + q) * (unsigned)H + (unsigned)h;
    const unsigned pt0 = qh * (unsigned)LP;
    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(value), 0, value_bytes, 0x00020000);
    DenseWinPosF32 win[L];
    dense_f32_stage_issue<L>(hot, wrow, t, lane, wv, rs, win_lds, win);
    float2 xy[L];
    float a[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        xy[l] = loc2[pt0 + l * P + p];
        a[l] = attn[pt0 + l * P + p];
    }
    if (threadIdx.x < 8)
        *reinterpret_cast<float4 *>(win_lds + kDenseF32ZeroOff + 16 * threadIdx.x) = make_float4(0.f, 0.f, 0.f, 0.f);
    dense_stage_wait();

    unsigned n_slow = 0, n_act = 0;                                // locality statistics of this wave (stats != nullptr)
    float acc[32];                                                 // this lane's point(s): all 32 channels
    float accs[8];                                                 // global path: channels 8 p .. 8 p + 7 of the QUERY
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = 0.f;
#pragma unroll
    for (int c = 0; c < 8; ++c) accs[c] = 0.f;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const DenseMap T = hot.lv[l];
        const DenseWinPosF32 &o = win[l];
        const DensePoint s = dense_locate(xy[l].x, xy[l].y, T.H, T.W);
        const DenseFoot f = dense_f32_foot(s, T, o);
        const bool act = vq && s.inside;
        const bool slow = act && f.d < 0, fast = act && f.d >= 0;
        // the reference's products: (hh hw) v a, summed per point, then over the points
        const float wk[4] = {s.hh * s.hw, s.hh * s.lw, s.lh * s.hw, s.lh * s.lw};
        if (__builtin_amdgcn_ballot_w64(fast) != 0ull) {
            int slot[4];
            dense_f32_slots(s, o, slot);
            const float af = fast ? a[l] : 0.f;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // (a corner outside the map -- or a lane that is not served from the window -- reads the row of ZEROS:
                // whatever pixel its clamped slot names, finite or not, never meets its (zero) weight: 0 * Inf is NaN)
                const bool counts = fast && f.m[k] != 0u;
                float va[32];
                dense_f32_load_row(win_lds + (counts ? slot[k] : kDenseF32ZeroOff), va);
                const float wa = counts ? wk[k] * af : 0.f;
#pragma unroll
                for (int c = 0; c < 32; ++c) acc[c] = fmaf(wa, va[c], acc[c]);
            }
        }
        const unsigned long long slow_lanes = __builtin_amdgcn_ballot_w64(slow);
        n_slow += (unsigned)__builtin_popcountll(slow_lanes);
        n_act += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(act));
        if (slow_lanes != 0ull) {                                  // wave-uniform: the global path, quad-cooperative
            constexpr unsigned kNoRow = 0x80000000u;               // outside the buffer: the load returns zeros
            const unsigned row0 = t.b * (unsigned)hot.S + (unsigned)T.start;
            const int pra = __mul24(f.ra, T.W), prb = __mul24(f.rb, T.W);
            const int pix[4] = {pra + f.ca, pra + f.cb, prb + f.ca, prb + f.cb};
            unsigned goff[4];
            float wsl[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                goff[k] = slow && f.m[k] ? (unsigned)(((row0 + (unsigned)pix[k]) * H + h) * (C * 4)) : kNoRow;
                wsl[k] = slow ? wk[k] * a[l] : 0.f;
            }
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                if ((slow_lanes & (0x1111111111111111ull << tp)) == 0ull) continue;     // nobody's point tp
                dense_u32x4 rw[4][2];
                float wv_[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned oo = quad_bcast_u32(goff[k], tp) + (unsigned)p * 32u;
                    rw[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rs, oo, 0, 0);
                    rw[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rs, oo + 16u, 0, 0);
                    wv_[k] = __uint_as_float(quad_bcast_u32(__float_as_uint(wsl[k]), tp));
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned w8[8] = {rw[k][0].x, rw[k][0].y, rw[k][0].z, rw[k][0].w,
                                            rw[k][1].x, rw[k][1].y, rw[k][1].z, rw[k][1].w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) accs[i] = fmaf(wv_[k], __uint_as_float(w8[i]), accs[i]);
                }
            }
        }
    }
    // how local were this wave's points (one tile in 61 reports, as the bf16 forward: boxattn_dense_fwd.h)
    if (stats && lane == 0 && role.id % 61u == 0u) {
        unsigned long long *slot = stats + 2 * (((role.id / 61u) * 4u + (unsigned)wv) & (kDenseStatSlots - 1));
        __hip_atomic_fetch_add(slot, (unsigned long long)n_slow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(slot + 1, (unsigned long long)n_act, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- the quad's four points -> the query's row: through LDS (the window buffer, after a barrier); lane p of the
    //      quad sums channels 8 p .. 8 p + 7 over the four points and adds the global path's sums
    __syncthreads();
    float *res = reinterpret_cast<float *>(win_lds) + (wv * kWave + lane) * 33;        // (33: conflict-free columns)
#pragma unroll
    for (int c = 0; c < 32; ++c) res[c] = acc[c];
    wave_lds_sync();
    const float *quad = reinterpret_cast<float *>(win_lds) + (wv * kWave + (lane & ~3)) * 33 + 8 * p;
    float o8[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) o8[i] = ((quad[i] + quad[33 + i]) + (quad[66 + i] + quad[99 + i])) + accs[i];
    if (vq) {
        float4 *dst = reinterpret_cast<float4 *>(out + (size_t)qh * C + 8 * p);
        dst[0] = make_float4(o8[0], o8[1], o8[2], o8[3]);
        dst[1] = make_float4(o8[4], o8[5], o8[6], o8[7]);
    }
}

}  // namespace boxattn
