// Window-staged FORWARD for the encoder case of bf16 box attention (one query per pixel, C = 32, 2x2 points,
// <= 4 levels): the staging of boxattn_dense.h (the value windows of an 8x8 query tile and head in LDS) with the
// arithmetic on the matrix cores.
//
// out[q][c] = sum over the query's 16 points and their 4 corners of w * v[corner][c], w a float32 weight.  On the
// VALU that is an unpack + a multiply-add per bf16 element: 2/3 of fwd2_kernel's instructions.  Here one
// v_mfma_f32_4x4x4_16B_bf16 (16 independent 4x4x4 products, one per lane QUAD) does 4 channels x 4 corners of one
// point for 16 queries at once:
//
//     D[i][j] += sum_k A[i][k] B[k][j]      k = corner, j = channel (4 per instruction), i = term of the weight
//
//   A (lane i of the quad holds row i): the point's four corner weights split into two bf16 terms, w = hi + lo
//     (|w - hi - lo| <= 2^-17 |w|): row 0 = hi, row 1 = lo (rows 2, 3 repeat them; their results are not used);
//     broadcast inside the quad from the lane that located the point (DPP quad_perm), hi or lo by the lane's parity;
//   B (lane j holds column j = 4 corners of one channel): exactly what ds_read_b64_tr_b16 delivers -- in a 16-lane
//     group lane 4 k + q supplies the address of corner k's row for the query of quad q, and lane 4 q + r receives
//     element r (channel c0 + r) of the four rows.  Eight reads (c0 = 0, 4, .. 28) and eight MFMAs per point;
//   D: lane r of a quad ends with channels r, 4 + r, .. 28 + r of its query (term rows 0 and 1 are added at the end).
//
// A quad = one query; its four lanes locate the four points of a level (lane r: point r), then the quad walks the
// four points together.  The supplier lane of (corner k, quad q) gets the point's packed {slot of its top-left
// corner, which corners count} from the locating lane with one ds_bpermute and derives its own corner's slot.
// Corners that do not count (outside the map) and points that are not served from the window read a zero row.
//
// Points whose footprint is not inside the staged window (and every point of a level that is not staged) take the
// global path of the gather kernels inside the same kernel: the quad fetches the four corner rows together, 16
// bytes per lane (out-of-map corners and the other quads of the wave: an offset outside the buffer, for which the
// hardware returns zeros), unpack + multiply-add on the VALU into a second, channel-contiguous accumulator.
// Entered per (level, point step) only if some quad of the wave needs it.  Same results either way.
//
// Epilogue: the matrix-core sums (channel 4 m + r in lane r) are brought into the channel-contiguous order through
// LDS (the window buffer, after a barrier), added to the VALU sums and stored as one 16-byte piece per lane.
#pragma once
#include "boxattn_dense.h"

namespace boxattn {

// lanes (0, 1, 2, 3) of every quad <- lanes (2 u, 2 u, 2 u + 1, 2 u + 1)
__device__ __forceinline__ unsigned quad_pairs_u32(unsigned v, int u)       // u is a constant after unrolling
{
    return u == 0 ? (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x50, 0xF, 0xF, true)      // quad_perm [0,0,1,1]
                  : (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xFA, 0xF, 0xF, true);     // quad_perm [2,2,3,3]
}
// lanes (0, 1, 2, 3) <- lanes (2 u, 2 u + 1, 2 u, 2 u + 1)
__device__ __forceinline__ unsigned quad_evenodd_u32(unsigned v, int u)
{
    return u == 0 ? (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x44, 0xF, 0xF, true)      // quad_perm [0,1,0,1]
                  : (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xEE, 0xF, 0xF, true);     // quad_perm [2,3,2,3]
}

typedef short fwd_i16x4 __attribute__((ext_vector_type(4)));
typedef float fwd_f32x4 __attribute__((ext_vector_type(4)));

template <int L>
__global__ __launch_bounds__(256, BOXATTN_DENSE_WPE) void fwd_dense_kernel(
    const bf16_t *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ attn,
    bf16_t *__restrict__ out, DensePlan pl, unsigned value_bytes, BinRide ride,
    unsigned long long *__restrict__ stats)
{
    constexpr int C = 32, P = 4, LP = L * P;
    constexpr int kBias = 4096;                         // keeps the packed slot offset non-negative
    constexpr int kZeroOff = kDenseZeroOff;
    __shared__ __attribute__((aligned(16))) unsigned char win_lds[kDenseLdsBytes];
    const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    // training forward: the backward's count pass and the scans chained behind it ride in this launch
    // (boxattn_ride.h, bin_count_ride)
    const RideRole role = ride_role(blockIdx.x, ride.grid);
    if (role.rider) {
        bin_count_ride<256>(ride, role.id, reinterpret_cast<int *>(win_lds));
        return;
    }
#ifdef BOXATTN_DEBUG_NO_TILES          // timing experiments only: the riders alone
    return;
#endif
    DenseHot<L> hot;
    DenseMap Q;
    DenseWin wrow[L];
    const DenseTileId t = dense_tile_of_block<L>(pl, role.id, hot, Q, wrow);
    if (t.lq < 0) return;                                          // workgroup-uniform
    const int H = hot.H, h = t.h;
    // ---- lane -> (query of the wave's 4x4 sub-tile, lane of its quad)
    const int qi = lane >> 2, r = lane & 3;
    const int qy = t.ty * kDenseTile + (wv >> 1) * kDenseSub + (qi >> 2);
    const int qx = t.tx * kDenseTile + (wv & 1) * kDenseSub + (qi & 3);
    const bool vq = qy < Q.H && qx < Q.W;
    const unsigned q = (unsigned)(Q.start + min(qy, Q.H - 1) * Q.W + min(qx, Q.W - 1));
    const unsigned qh = (t.b * (unsigned)hot.Lq + q) * (unsigned)H + (unsigned)h;
    const unsigned pt0 = qh * (unsigned)LP;
    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(value), 0, value_bytes, 0x00020000);
    DenseWinPos win[L];
    dense_stage_issue<L>(hot, wrow, t, lane, wv, rs, win_lds, win);
    float2 xy[L];
    float a[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {                                  // lane r of the quad: point r of every level
        xy[l] = loc2[pt0 + l * P + r];
        a[l] = attn[pt0 + l * P + r];
    }
    if (threadIdx.x < 4)
        *reinterpret_cast<dense_u32x4 *>(win_lds + kZeroOff + 16 * threadIdx.x) = dense_u32x4{0u, 0u, 0u, 0u};
    dense_stage_wait();                                            // windows complete

    // supplier role inside the 16-lane group: the row of corner jc for the query of quad qs
    const int jc = (lane >> 2) & 3, qs = lane & 3;
    const unsigned odd_mask = 0u - (unsigned)(lane & 1);
    const int src_lane4 = ((lane & 48) + 4 * qs) * 4;              // ds_bpermute address of lane 0 of that quad
    unsigned n_slow = 0, n_act = 0;                                // locality statistics of this wave (stats != nullptr)
    fwd_f32x4 acc[8];                                              // matrix cores: channel 4 m + r, rows = weight terms
    float accv[8];                                                 // VALU (global path): channels 8 r .. 8 r + 7
#pragma unroll
    for (int m = 0; m < 8; ++m) { acc[m] = fwd_f32x4{0.f, 0.f, 0.f, 0.f}; accv[m] = 0.f; }
#pragma unroll
    for (int l = 0; l < L; ++l) {
        if (BOXATTN_DEBUG_NO_MATH) {                               // (timing experiments: loads kept, arithmetic gone)
            asm volatile("" ::"v"(xy[l].x), "v"(xy[l].y), "v"(a[l]));
            continue;
        }
        const DenseMap T = hot.lv[l];
        const DenseWinPos &o = win[l];
        const DensePoint s = dense_locate(xy[l].x, xy[l].y, T.H, T.W);
        const int rows = o.rows(), cols = o.cols();
        const int Hm1 = T.H - 1, Wm1 = T.W - 1;
        // which corners count (bit k), and is the footprint inside the staged window
        const unsigned mr0 = ~(unsigned)(s.y0 >> 31), mr1 = (unsigned)((s.y0 - Hm1) >> 31);
        const unsigned mc0 = ~(unsigned)(s.x0 >> 31), mc1 = (unsigned)((s.x0 - Wm1) >> 31);
        const unsigned bits = ((mr0 & mc0) & 1u) | ((mr0 & mc1) & 2u) | ((mr1 & mc0) & 4u) | ((mr1 & mc1) & 8u);
        const int ra = max(s.y0, 0), rb = min(s.y0 + 1, Hm1), ca = max(s.x0, 0), cb = min(s.x0 + 1, Wm1);
        const int d = min(min(ra - o.y0, o.y0 + rows - 1 - rb), min(ca - o.x0, o.x0 + cols - 1 - cb));
        const bool act = vq && s.inside;
        const bool fast = act && d >= 0, slow = act && d < 0;
        const int pitchb = o.pitchb(), offb = o.offb();
        const int slot0 = offb + __mul24(s.y0 - o.y0, pitchb) + __mul24(s.x0 - o.x0, kDenseSlotBytes) + kBias;
        const unsigned pack = fast ? ((unsigned)slot0 | (bits << 20)) : 0u;
        // the four corner weights x attention weight; for the matrix cores as hi + lo bf16 terms
        const float aa = s.inside ? a[l] : 0.f;
        const float ha = s.hh * aa, la = s.lh * aa;
        const float wk[4] = {ha * s.hw, ha * s.lw, la * s.hw, la * s.lw};
        const unsigned hi01 = pack_bf16x2(wk[0], wk[1]), hi23 = pack_bf16x2(wk[2], wk[3]);
        const unsigned lo01 = pack_bf16x2(wk[0] - __uint_as_float(hi01 << 16), wk[1] - __uint_as_float(hi01 & 0xffff0000u));
        const unsigned lo23 = pack_bf16x2(wk[2] - __uint_as_float(hi23 << 16), wk[3] - __uint_as_float(hi23 & 0xffff0000u));
        const int dj = (jc & 1) * kDenseSlotBytes + (jc >> 1) * pitchb - kBias;       // my corner relative to the packed slot
        // A operand: lane i of a quad holds row i -- rows 0, 2: the hi terms, rows 1, 3: the lo terms of the point
        // the quad is working on.  Arranged once per level so that ONE quad permute per register and point
        // delivers it: z_[0] = {hi(0), lo(0), hi(1), lo(1)} by lane, z_[1] = {hi(2), lo(2), hi(3), lo(3)}
        // (bitwise merges, not ?: -- the compiler turns a select of two DPP moves into a branch and runs each move
        // with half of the lanes switched off, where a DPP read of a disabled lane returns 0)
        unsigned z01[2], z23[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            z01[u] = (quad_pairs_u32(lo01, u) & odd_mask) | (quad_pairs_u32(hi01, u) & ~odd_mask);
            z23[u] = (quad_pairs_u32(lo23, u) & odd_mask) | (quad_pairs_u32(hi23, u) & ~odd_mask);
        }
        if (rows > 0 && __builtin_amdgcn_ballot_w64(fast) != 0ull) {   // (wave-uniform: staged, and somebody reads it)
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                const unsigned pk = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane4 + 4 * tp, (int)pack);
                const bool counts = ((pk >> (20 + jc)) & 1u) != 0u;
                const int addr = counts ? (int)(pk & 0xfffffu) + dj : kZeroOff;
                const unsigned a0 = quad_evenodd_u32(z01[tp >> 1], tp & 1), a1 = quad_evenodd_u32(z23[tp >> 1], tp & 1);
                const fwd_i16x4 av = __builtin_bit_cast(fwd_i16x4, uint2{a0, a1});
                typedef __attribute__((address_space(3))) fwd_i16x4 lds_vec;
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const fwd_i16x4 bv = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_vec *)(win_lds + addr + 8 * m));
                    acc[m] = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(av, bv, acc[m], 0, 0, 0);
                }
            }
        }
        const unsigned long long slow_lanes = __builtin_amdgcn_ballot_w64(slow);
        n_slow += (unsigned)__builtin_popcountll(slow_lanes);
        n_act += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(act));
        if (slow_lanes != 0ull) {                                  // wave-uniform: the global path
            constexpr unsigned kNoRow = 0x80000000u;               // outside the buffer: the load returns zeros
            const unsigned row0 = t.b * (unsigned)hot.S + (unsigned)T.start;
            const int pra = __mul24(ra, T.W), prb = __mul24(rb, T.W);
            const int pix[4] = {pra + ca, pra + cb, prb + ca, prb + cb};
            unsigned goff[4];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                goff[k] = slow && ((bits >> k) & 1u) ? (unsigned)(((row0 + (unsigned)pix[k]) * H + h) * (C * 2)) : kNoRow;
#pragma unroll
            for (int tp = 0; tp < 4; ++tp) {
                if ((slow_lanes & (0x1111111111111111ull << tp)) == 0ull) continue;     // nobody's point tp
                dense_u32x4 rw[4];
                float wv_[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    rw[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, quad_bcast_u32(goff[k], tp) + (unsigned)r * 16u, 0, 0);
                    wv_[k] = __uint_as_float(quad_bcast_u32(__float_as_uint(wk[k]), tp));
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned w4[4] = {rw[k].x, rw[k].y, rw[k].z, rw[k].w};
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        accv[2 * i] = fmaf(wv_[k], __uint_as_float(w4[i] << 16), accv[2 * i]);
                        accv[2 * i + 1] = fmaf(wv_[k], __uint_as_float(w4[i] & 0xffff0000u), accv[2 * i + 1]);
                    }
                }
            }
        }
    }
    // How local were this wave's points?  {points served from global memory, points inside the window test}
    // into one of kDenseStatSlots pairs of monotonic 64-bit counters of the caller's state buffer -- from one tile
    // in 61 only (a prime: the sample walks through the XCDs' bands of the map; every wave of every tile doing it
    // was 15 k same-line atomics, +20 us).  Fire and forget: nobody waits for these atomics.  The caller reads the
    // counters now and then and, when most points miss their windows, asks for the row-gather kernels instead
    // (hints, include/boxattn.h).
    if (stats && lane == 0 && role.id % 61u == 0u) {
        unsigned long long *slot = stats + 2 * (((role.id / 61u) * 4u + (unsigned)wv) & (kDenseStatSlots - 1));
        __hip_atomic_fetch_add(slot, (unsigned long long)n_slow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_fetch_add(slot + 1, (unsigned long long)n_act, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    // ---- lane r of the quad holds channels r, 4 + r, .. 28 + r from the matrix cores: through LDS into the
    //      channel-contiguous order of the VALU sums, then one 16-byte piece of the query's row per lane
    constexpr int kResPitch = 36;                                  // floats per query (16-byte aligned rows, 2-way banks)
    __syncthreads();                                               // every wave is done with the windows
    float *res = reinterpret_cast<float *>(win_lds) + (wv * 16 + qi) * kResPitch;
#pragma unroll
    for (int m = 0; m < 8; ++m) res[4 * m + r] = acc[m][0] + acc[m][1];
    wave_lds_sync();
    const float4 x0 = *reinterpret_cast<const float4 *>(res + 8 * r), x1 = *reinterpret_cast<const float4 *>(res + 8 * r + 4);
    if (vq) {
        dense_u32x4 o4;
        o4.x = pack_bf16x2(x0.x + accv[0], x0.y + accv[1]);
        o4.y = pack_bf16x2(x0.z + accv[2], x0.w + accv[3]);
        o4.z = pack_bf16x2(x1.x + accv[4], x1.y + accv[5]);
        o4.w = pack_bf16x2(x1.z + accv[6], x1.w + accv[7]);
        *reinterpret_cast<dense_u32x4 *>(out + (size_t)qh * C + 8 * r) = o4;
    }
}

}  // namespace boxattn
