// Second-generation gather kernels (forward, and the point-gradient half of the backward).
//
// Measured on the first fast kernels (profiles/r01_step2_*): ~134 (fwd) / ~190 (bwd) VALU
// instructions per sample point and lane, half of them the bilinear geometry that all G lanes
// of a (query, head) group recomputed redundantly, plus 64-bit address arithmetic and
// select-to-zero of out-of-map corners.  Here:
//
//   step A  lane (pair j, slot t) does the geometry of ONE sample point of pair j -- G points
//           per pair at a time -- and leaves {4 byte offsets, 4 weights, ...} in a wave-private
//           LDS tile (no workgroup barrier: producer and consumers are the same wavefront);
//   step B  lane (pair j, channel chunk m) walks the G points of its pair: one 32-byte LDS
//           broadcast read per point, four BUFFER loads (32-bit offset, the hardware range
//           check returns 0 for the out-of-map marker, so no address select and no
//           select-to-zero), 16 FMAs.
//
// The backward flavour accumulates S_k = sum_c g_c * v_k,c for the four corners (4 FMAs per
// channel instead of 16 ops), reduces the four sums over the G lanes with DPP, hands them to
// lane (j, t) and lets that lane finish grad_loc / grad_weight from its own step-A registers:
//   grad_w   = sum_k w_k S_k
//   grad_x   = W_l * a * (hh (S2 - S1) + lh (S4 - S3))
//   grad_y   = H_l * a * (hw (S3 - S1) + lw (S4 - S2))
// (reference formulas box_attn_kernel.cuh:145-183 with the channel sum pulled out).
#pragma once
#include "boxattn_device.h"
#include "boxattn_fast.h"

namespace boxattn {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

constexpr unsigned kOobOffset = 0x80000000u;     // >= any valid byte offset (tensor < 2 GiB)

template <typename ST> struct BufLd;
template <> struct BufLd<float> {                 // 4 channels = 16 bytes
    static __device__ __forceinline__ void ld(__amdgpu_buffer_rsrc_t r, unsigned off, float (&v)[4]) {
        const u32x4_t t = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
        v[0] = __uint_as_float(t.x); v[1] = __uint_as_float(t.y);
        v[2] = __uint_as_float(t.z); v[3] = __uint_as_float(t.w);
    }
};
template <> struct BufLd<bf16_t> {                // 4 channels = 8 bytes
    static __device__ __forceinline__ void ld(__amdgpu_buffer_rsrc_t r, unsigned off, float (&v)[4]) {
        const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    }
};

struct GeoBox { unsigned off[4]; float w[4]; };                     // 32 bytes
struct GeoInst { unsigned off[4]; float w[4]; float as, al, pad0, pad1; };   // 48 bytes

// Step A for one point: byte offsets of the four corners (row of head h, channel 0) or the
// out-of-map marker, and the bilinear weights.
template <typename ST>
__device__ __forceinline__ void corner_offsets(const Sample<float> &s, unsigned row0, int H, int h,
                                               int C, unsigned (&off)[4])
{
#pragma unroll
    for (int k = 0; k < 4; ++k)
        off[k] = s.ok[k] ? ((row0 + (unsigned)s.pix[k]) * H + h) * (unsigned)(C * sizeof(ST))
                         : kOobOffset;
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
template <typename ST, int G, bool INST, int U = G>
__global__ __launch_bounds__(256) void fwd2_kernel(
    const ST *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc,
    const float *__restrict__ w_sp, const float *__restrict__ w_lv, int S, int H, int L, int Lq,
    int P, ST *__restrict__ out, ST *__restrict__ mask, size_t n_qh, unsigned value_bytes)
{
    constexpr int VEC = 4, C = VEC * G, PAIRS = kWave / G;
    typedef typename std::conditional<INST, GeoInst, GeoBox>::type Geo;
    __shared__ LevelTable lv;
    // one pad slot per G-lane group: the groups' broadcast reads then fall on different banks
    __shared__ __attribute__((aligned(16))) Geo geo_all[4][kWave + kWave / G];
    load_levels(lv, shapes, lsi, L);

    const unsigned bid = xcd_chunked_block(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    Geo *geo = geo_all[wv];
    const size_t wave = (size_t)bid * (blockDim.x / kWave) + wv;
    size_t qh = wave * PAIRS + lane / G;
    const bool active = qh < n_qh;
    if (!active) qh = n_qh - 1;
    const int slot = lane % G;                      // step A: point slot; step B: channel chunk
    const int grp0 = (lane & ~(G - 1)) + lane / G;  // padded tile index of the group's slot 0
    const int myslot = grp0 + lane % G;
    const int h = (int)(qh % H);
    const size_t bq = qh / H;
    const unsigned b = (unsigned)(bq / Lq);
    const size_t HC = (size_t)H * C;
    const int LP = L * P;
    const size_t pt0 = qh * LP;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<ST *>(value), 0, value_bytes, 0x00020000);
    const unsigned lane_off = (unsigned)(slot * VEC * sizeof(ST));
    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);

    float acc[VEC] = {0.f, 0.f, 0.f, 0.f};

    if constexpr (!INST) {
        for (int t0 = 0; t0 < LP; t0 += G) {
            {   // ---- step A
                const int lp = t0 + slot;
                Geo g;
                if (lp < LP) {
                    const int l = lp / P;
                    const float2 xy = loc2[pt0 + lp];
                    const float a = w_sp[pt0 + lp];
                    const Sample<float> s = locate<float>(xy.x, xy.y, lv.h[l], lv.w[l]);
                    corner_offsets<ST>(s, b * (unsigned)S + (unsigned)lv.start[l], H, h, C, g.off);
                    g.w[0] = s.hh * s.hw * a; g.w[1] = s.hh * s.lw * a;
                    g.w[2] = s.lh * s.hw * a; g.w[3] = s.lh * s.lw * a;
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k) { g.off[k] = kOobOffset; g.w[k] = 0.f; }
                }
                wave_lds_sync();                   // previous tile fully consumed
                geo[myslot] = g;
                wave_lds_sync();
            }
            // ---- step B (U points' loads in flight at a time)
#pragma unroll 1
            for (int tb = 0; tb < G; tb += U) {
#pragma unroll
                for (int t = tb; t < tb + U; ++t) {
                    const Geo g = geo[grp0 + t];
                    float v[4][VEC];
#pragma unroll
                    for (int k = 0; k < 4; ++k) BufLd<ST>::ld(rs, g.off[k] + lane_off, v[k]);
#pragma unroll
                    for (int c = 0; c < VEC; ++c)
                        acc[c] += g.w[0] * v[0][c] + g.w[1] * v[1][c] + g.w[2] * v[2][c] +
                                  g.w[3] * v[3][c];
                }
            }
        }
    } else {
        ST *mk = mask + bq * P * HC + (size_t)h * C + slot * VEC;
        // few queries x many points (decoder, 14x14 grids): gridDim.y workgroups share the
        // point tiles of a (query, head) pair; out is then accumulated with atomics
        const int tiles = (P + G - 1) / G, tps = (tiles + (int)gridDim.y - 1) / (int)gridDim.y;
        const int p_begin = (int)blockIdx.y * tps * G, p_end = min(P, p_begin + tps * G);
        for (int p0 = p_begin; p0 < p_end; p0 += G) {
            float macc[G][VEC];
#pragma unroll
            for (int t = 0; t < G; ++t)
#pragma unroll
                for (int c = 0; c < VEC; ++c) macc[t][c] = 0.f;
            for (int l = 0; l < L; ++l) {
                {   // ---- step A: points (l, p0 .. p0+G-1)
                    const int p = p0 + slot;
                    Geo g;
                    if (p < P) {
                        const size_t i = pt0 + (size_t)l * P + p;
                        const float2 xy = loc2[i];
                        g.as = w_sp[i];
                        g.al = w_lv[i];
                        const Sample<float> s = locate<float>(xy.x, xy.y, lv.h[l], lv.w[l]);
                        corner_offsets<ST>(s, b * (unsigned)S + (unsigned)lv.start[l], H, h, C,
                                           g.off);
                        g.w[0] = s.hh * s.hw; g.w[1] = s.hh * s.lw;
                        g.w[2] = s.lh * s.hw; g.w[3] = s.lh * s.lw;
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k) { g.off[k] = kOobOffset; g.w[k] = 0.f; }
                        g.as = 0.f; g.al = 0.f;
                    }
                    g.pad0 = 0.f; g.pad1 = 0.f;
                    wave_lds_sync();
                    geo[myslot] = g;
                    wave_lds_sync();
                }
#pragma unroll
                for (int t = 0; t < G; ++t) {
                    const Geo g = geo[grp0 + t];
                    float v[4][VEC];
#pragma unroll
                    for (int k = 0; k < 4; ++k) BufLd<ST>::ld(rs, g.off[k] + lane_off, v[k]);
#pragma unroll
                    for (int c = 0; c < VEC; ++c) {
                        const float val = g.w[0] * v[0][c] + g.w[1] * v[1][c] + g.w[2] * v[2][c] +
                                          g.w[3] * v[3][c];
                        acc[c] += val * g.as;
                        macc[t][c] += val * g.al;
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < G; ++t)
                if (active && p0 + t < P) VecIO<ST, VEC>::st(mk + (size_t)(p0 + t) * HC, macc[t]);
        }
    }
    if (active) {
        if constexpr (INST && std::is_same<ST, float>::value) {
            if (gridDim.y > 1) {                   // split points: out was zero-filled by the host
#pragma unroll
                for (int c = 0; c < VEC; ++c) atomic_add(out + qh * C + slot * VEC + c, acc[c]);
                return;
            }
        }
        VecIO<ST, VEC>::st(out + qh * C + slot * VEC, acc);
    }
}

// ---------------------------------------------------------------------------------------
// backward, point gradients only (grad_loc, grad_weight[s]); grad_value is boxattn_binned.h
// ---------------------------------------------------------------------------------------
template <typename ST, int G, bool INST>
__global__ __launch_bounds__(256) void pointgrad2_kernel(
    const ST *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc,
    const float *__restrict__ w_sp, const float *__restrict__ w_lv,
    const ST *__restrict__ grad_out, const ST *__restrict__ grad_mask, int S, int H, int L,
    int Lq, int P, float *__restrict__ grad_loc, float *__restrict__ grad_sp,
    float *__restrict__ grad_lv, size_t n_qh, unsigned value_bytes)
{
    constexpr int VEC = 4, C = VEC * G, PAIRS = kWave / G;
    __shared__ LevelTable lv;
    __shared__ __attribute__((aligned(16))) GeoBox geo_all[4][kWave + kWave / G];
    load_levels(lv, shapes, lsi, L);

    const unsigned bid = xcd_chunked_block(blockIdx.x, gridDim.x);
    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    GeoBox *geo = geo_all[wv];
    const size_t wave = (size_t)bid * (blockDim.x / kWave) + wv;
    size_t qh = wave * PAIRS + lane / G;
    const bool active = qh < n_qh;
    if (!active) qh = n_qh - 1;
    const int slot = lane % G;
    const int grp0 = (lane & ~(G - 1)) + lane / G;  // padded tile index of the group's slot 0
    const int myslot = grp0 + lane % G;
    const int h = (int)(qh % H);
    const size_t bq = qh / H;
    const unsigned b = (unsigned)(bq / Lq);
    const size_t HC = (size_t)H * C;
    const int LP = L * P;
    const size_t pt0 = qh * LP;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<ST *>(value), 0, value_bytes, 0x00020000);
    const unsigned lane_off = (unsigned)(slot * VEC * sizeof(ST));
    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);

    float g[VEC];
    VecIO<ST, VEC>::ld(grad_out + qh * C + slot * VEC, g);

    // gridDim.y workgroups share the point tiles of a pair (few queries x many points)
    const int tiles = (LP + G - 1) / G, tps = (tiles + (int)gridDim.y - 1) / (int)gridDim.y;
    const int t_begin = (int)blockIdx.y * tps * G, t_end = min(LP, t_begin + tps * G);
    for (int t0 = t_begin; t0 < t_end; t0 += G) {
        // ---- step A (the lane keeps its point's geometry in registers for the finish)
        const int lp = t0 + slot;
        const bool have = lp < LP;
        const int lq = have ? lp : LP - 1;
        const int l = lq / P;
        const float2 xy = loc2[pt0 + lq];
        const float as = w_sp[pt0 + lq];
        const float al = INST ? w_lv[pt0 + lq] : 0.f;
        const Sample<float> s = locate<float>(xy.x, xy.y, lv.h[l], lv.w[l]);
        GeoBox gg;
        corner_offsets<ST>(s, b * (unsigned)S + (unsigned)lv.start[l], H, h, C, gg.off);
        gg.w[0] = s.hh * s.hw; gg.w[1] = s.hh * s.lw; gg.w[2] = s.lh * s.hw; gg.w[3] = s.lh * s.lw;
        if (!have) {
#pragma unroll
            for (int k = 0; k < 4; ++k) gg.off[k] = kOobOffset;
        }
        wave_lds_sync();
        geo[myslot] = gg;
        wave_lds_sync();

        // ---- step B: corner sums of the G points of this lane's pair
        float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;         // S_k of "my" point (slot)
        float m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f;         // instance: sums with grad_mask
#pragma unroll
        for (int t = 0; t < G; ++t) {
            const GeoBox q = geo[grp0 + t];
            float v[4][VEC];
#pragma unroll
            for (int k = 0; k < 4; ++k) BufLd<ST>::ld(rs, q.off[k] + lane_off, v[k]);
            float a1 = 0.f, a2 = 0.f, a3 = 0.f, a4 = 0.f;
#pragma unroll
            for (int c = 0; c < VEC; ++c) {
                a1 += g[c] * v[0][c]; a2 += g[c] * v[1][c];
                a3 += g[c] * v[2][c]; a4 += g[c] * v[3][c];
            }
            a1 = group_sum<G>(a1); a2 = group_sum<G>(a2);
            a3 = group_sum<G>(a3); a4 = group_sum<G>(a4);
            const bool mine = slot == t;
            s1 = mine ? a1 : s1; s2 = mine ? a2 : s2; s3 = mine ? a3 : s3; s4 = mine ? a4 : s4;
            if constexpr (INST) {
                // grad_mask row of point (t0 + t): its p is uniform inside the group
                const int lpt = min(t0 + t, LP - 1);
                const int pt = lpt % P;
                float gm[VEC];
                VecIO<ST, VEC>::ld(grad_mask + (bq * P + pt) * HC + (size_t)h * C + slot * VEC, gm);
                float b1 = 0.f, b2 = 0.f, b3 = 0.f, b4 = 0.f;
#pragma unroll
                for (int c = 0; c < VEC; ++c) {
                    b1 += gm[c] * v[0][c]; b2 += gm[c] * v[1][c];
                    b3 += gm[c] * v[2][c]; b4 += gm[c] * v[3][c];
                }
                b1 = group_sum<G>(b1); b2 = group_sum<G>(b2);
                b3 = group_sum<G>(b3); b4 = group_sum<G>(b4);
                m1 = mine ? b1 : m1; m2 = mine ? b2 : m2; m3 = mine ? b3 : m3; m4 = mine ? b4 : m4;
            }
        }
        // ---- finish: lane (pair, slot) owns point t0 + slot
        if (active && have) {
            const size_t i = pt0 + lp;
            const float Wl = (float)lv.w[l], Hl = (float)lv.h[l];
            float gs, gx, gy;
            gs = gg.w[0] * s1 + gg.w[1] * s2 + gg.w[2] * s3 + gg.w[3] * s4;
            if constexpr (!INST) {
                gx = Wl * as * (s.hh * (s2 - s1) + s.lh * (s4 - s3));
                gy = Hl * as * (s.hw * (s3 - s1) + s.lw * (s4 - s2));
            } else {
                const float t1 = as * s1 + al * m1, t2 = as * s2 + al * m2;
                const float t3 = as * s3 + al * m3, t4 = as * s4 + al * m4;
                gx = Wl * (s.hh * (t2 - t1) + s.lh * (t4 - t3));
                gy = Hl * (s.hw * (t3 - t1) + s.lw * (t4 - t2));
                grad_lv[i] = s.inside ? gg.w[0] * m1 + gg.w[1] * m2 + gg.w[2] * m3 + gg.w[3] * m4
                                      : 0.f;
            }
            grad_sp[i] = s.inside ? gs : 0.f;
            reinterpret_cast<float2 *>(grad_loc)[i] =
                s.inside ? make_float2(gx, gy) : make_float2(0.f, 0.f);
        }
    }
}

}  // namespace boxattn
