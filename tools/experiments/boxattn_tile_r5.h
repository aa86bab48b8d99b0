// Window-staged tile kernels of the encoder case, round 5: a workgroup walks a RANGE of (tile, head) units, and
// the destination binning of the backward happens INSIDE the tiles.
//
// Rounds 3-4 ran one workgroup per (tile, head) and carried the backward's count / fill passes as rider
// workgroups (boxattn_ride.h).  Two things were paid for that: every wave repeated a ~300
// instruction scalar preamble -- tile decode, window placement -- that is the same for the 8 heads of a tile, and
// the riders re-loaded and re-located every sample point twice (count, fill: ~10 M wave instructions, 80 MB).
// Every kernel of the step retires one instruction per SIMD every ~4 cycles whatever its type, so instructions
// are what a step costs.  Here:
//
//   * UNITS.  The (tile, head) units of an XCD's queue are numbered tile-major; workgroup k of the XCD takes
//     units [k U, (k + 1) U) -- U consecutive heads of (at most two) tiles -- decodes a tile once and keeps its
//     placement in scalar registers while it walks the heads.  U is chosen so that the whole grid is resident at
//     once (5 workgroups per CU): no second, half-empty round of workgroups.
//   * COUNT (training forward).  The forward locates every point anyway; a lane adds its point to the <= 4
//     destination blocks of its footprint in an LDS histogram of the unit's (image, head) slice.  At the end
//     of the unit the histogram goes to the unit's row of the plan (rows[unit][block], plain stores) -- no
//     atomics, no tickets, nobody waits.  (First build: one returned global atomic per touched block + a ticket
//     per unit, the slice's last unit scanning: each unit then ended with two memory round trips during which
//     its workgroup held a fifth of a CU and issued nothing -- forward 33 -> 66 us.)  Two small launches behind
//     the forward (tile_colscan_kernel, bin_scan_kernel: boxattn_binned.h) turn the rows into first slots
//     (exclusive prefix over a slice's units per block) and the block totals into offsets + work-item list.
//   * FILL (point-gradient kernel).  A unit starts its LDS histogram at offsets[block] + rows[unit][block] and
//     every point gets its record slot from one LDS atomic; the 16-byte record {id, x, y, weight} is written by
//     the lane that holds the point in registers.  No second count, no claim, nobody waits.
//
// The record order inside a block: by unit, inside a unit by the order of the LDS atomics (lanes in order, waves
// as they come): the accumulate kernel's float32 summation order may differ from run to run, nothing else does.
//
// Arguments.  The kernels read their arguments through the ADDRESS of the kernel-argument segment (constant
// address space: scalar loads), made opaque once per unit, so that whatever a unit needs of the plan is loaded
// when it needs it and dies with the unit.  Named by-value arguments are loop-invariant to the compiler: it
// hoisted their loads -- and every per-lane value derived from them -- out of the unit loop and spilled them
// (640 bytes of scratch per lane in the first build of these kernels).
#pragma once
#include "boxattn_dense.h"
#include "boxattn_dense_fwd.h"

namespace boxattn {

#if defined(__HIP_DEVICE_COMPILE__)
#define BOXATTN_KARG __attribute__((address_space(4)))
#else
#define BOXATTN_KARG            // (the host pass only parses the kernels)
#endif
struct FwdTileArgs {
    const bf16_t *value;
    const float *loc, *attn;
    bf16_t *out;
    unsigned long long *stats;
    unsigned value_bytes;
    DensePlan pl;
    TileBin tb;
};
struct PgTileArgs {
    const bf16_t *value;
    const float *loc, *attn;
    const bf16_t *grad_out;
    float *grad_loc, *grad_attn;
    unsigned value_bytes;
    DensePlan pl;
    TileBin tb;
};

// ---------------------------------------------------------------------------------------------------------
// units
// ---------------------------------------------------------------------------------------------------------
// What a workgroup keeps of its current tile between the units (scalar registers).
template <int L> struct TileState {
    int lq;                  // query level, -1: past the end of the XCD's queue
    unsigned b, tr;          // image, index of the tile inside its level and image
    int ty, tx;
    int wx0[L], wy0[L];      // window origins on the sampled levels
};

// Tile j of XCD x's queue (levels coarsest first, every XCD one contiguous eighth of every level's tiles) and
// the placement of its windows: the scalar preamble, once per tile.
template <int L>
__device__ __forceinline__ void tile_decode(const BOXATTN_KARG DensePlan &pl, unsigned x, unsigned j, TileState<L> &t)
{
    int lq = -1;
    unsigned ti = 0, first = 0;
#pragma unroll
    for (int l = L - 1; l >= 0; --l) {
        const unsigned n_all = pl.lv[l].n_all;
        const unsigned lo = (x * n_all) >> 3, cnt = (((x + 1) * n_all) >> 3) - lo;
        const bool here = j - first < cnt;                 // first <= j < first + cnt
        lq = here ? l : lq;
        ti = here ? lo + (j - first) : ti;
        first += cnt;
    }
    t.lq = lq;
    const int lqc = max(lq, 0);
    const unsigned ntiles = (unsigned)pl.lv[lqc].ntiles, ntx = (unsigned)pl.lv[lqc].ntx;
    unsigned ty, tx;
    divmod_magic(ti, ntiles, pl.lv[lqc].mag_ntiles, t.b, t.tr);
    divmod_magic(t.tr, ntx, pl.lv[lqc].mag_ntx, ty, tx);
    t.ty = (int)ty;
    t.tx = (int)tx;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const DenseWin w = pl.win[lqc][l];
        const int rows = (int)(w.geo & 31u), cols = (int)((w.geo >> 5) & 31u);
        const int x0 = (t.tx * w.ax + w.bx) >> 16, y0 = (t.ty * w.ay + w.by) >> 16;
        t.wx0[l] = max(0, min(x0, pl.lv[l].W - cols));
        t.wy0[l] = max(0, min(y0, pl.lv[l].H - rows));
    }
}

// first tile of level lq among an image's tiles
template <int L> __device__ __forceinline__ unsigned tile_first(const BOXATTN_KARG TileBin &tb, int lq)
{
    int t0 = tb.tile0[0];
#pragma unroll
    for (int l = 1; l < L; ++l) t0 = l == lq ? tb.tile0[l] : t0;
    return (unsigned)t0;
}

// host: workgroups of a tile kernel launch for U units per workgroup
inline unsigned tile_grid(const DensePlan &p, int U)
{
    unsigned longest = 0;
    for (unsigned x = 0; x < 8; ++x) {
        unsigned n_x = 0;
        for (int l = 0; l < p.L; ++l) {
            const unsigned n = (unsigned)p.B * (unsigned)p.lv[l].ntiles;
            n_x += (((x + 1) * n) >> 3) - ((x * n) >> 3);
        }
        longest = longest > n_x ? longest : n_x;
    }
    return 8u * ((longest * (unsigned)p.H + (unsigned)U - 1u) / (unsigned)U);
}

// what a unit keeps of the plan (scalar registers, loaded per unit)
template <int L>
__device__ __forceinline__ void tile_hot(const BOXATTN_KARG DensePlan &pl, const TileState<L> &t, DenseHot<L> &hot,
                                         DenseWinPos (&win)[L])
{
    const int lqc = max(t.lq, 0);
#pragma unroll
    for (int l = 0; l < L; ++l) {
        hot.lv[l] = DenseMap{pl.lv[l].H, pl.lv[l].W, pl.lv[l].start};
        win[l].geo = pl.win[lqc][l].geo;
        win[l].x0 = t.wx0[l];
        win[l].y0 = t.wy0[l];
    }
    hot.H = pl.H; hot.Lq = pl.Lq; hot.S = pl.S;
}

// The cooperative fetch of one head's window rows: wave w takes the window rows w, w + 4, ... of every level; a row
// of up to 16 pixels x 64 bytes is ONE direct-to-LDS load (boxattn_dense.h: dense_stage_issue).
template <int L>
__device__ __forceinline__ void tile_stage_rows(const DenseHot<L> &hot, const DenseWinPos (&win)[L], unsigned b, unsigned h,
                                                int lane, int wv, __amdgpu_buffer_rsrc_t rs, unsigned char *lds)
{
    constexpr int C = 32, RPW = kDenseWinMax / 4;
    const int j = lane >> 2, chunk = lane & 3;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const DenseMap T = hot.lv[l];
        const DenseWinPos &o = win[l];
        const int jx = min(o.x0 + j, T.W - 1);
        const unsigned voff =
            ((((b * (unsigned)hot.S + (unsigned)(T.start + jx)) * (unsigned)hot.H + h) * C) + (unsigned)chunk * 8u) * 2u;
        const unsigned row_bytes = (unsigned)T.W * (unsigned)hot.H * (C * 2u);
        const int rows = min(o.rows(), T.H - o.y0);
        unsigned soff = (unsigned)(o.y0 + wv) * row_bytes;
        int dst = o.offb() + wv * o.pitchb();
        const int step = 4 * o.pitchb();
        if (j < o.cols()) {
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

// ---------------------------------------------------------------------------------------------------------
// destination blocks of a located point (touched_blocks of boxattn_binplan.h on the clamped rows / columns the
// tile kernels already hold): blk[0] always, the others -1 when the footprint does not cross that block edge
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ void tile_blocks(int ra, int rb, int ca, int cb, int nbx, int nby, int blk0, unsigned mw,
                                            unsigned mh, int (&blk)[4])
{
    const int bra = blk_of(ra, nby, mh), brb = blk_of(rb, nby, mh);
    const int bca = blk_of(ca, nbx, mw), bcb = blk_of(cb, nbx, mw);
    const int base_a = blk0 + __mul24(bra, nbx), base_b = blk0 + __mul24(brb, nbx);
    blk[0] = base_a + bca;
    blk[1] = bcb != bca ? base_a + bcb : -1;
    blk[2] = brb != bra ? base_b + bca : -1;
    blk[3] = brb != bra && bcb != bca ? base_b + bcb : -1;
}
__device__ __forceinline__ void lds_count(int *p)
{
    (void)__hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ int lds_take(int *p)
{
    return __hip_atomic_fetch_add(p, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// The unit's histogram is complete (barrier passed): the unit's row of the plan <- its per-block counts (ALL blocks
// of the slice, zeros included: the column scan behind the forward turns the rows into first slots, tile_colscan_kernel
// of boxattn_binned.h), the histogram left zero.  Plain stores, nobody waits for anything.
__device__ __forceinline__ void tile_dump(const BOXATTN_KARG TileBin &tb, int *hist, unsigned uid, int tid)
{
    const int nblk = tb.plan.nblk;
    int *row = tb.rows + (size_t)uid * nblk;
    for (int k = tid; k < nblk; k += 256) {
        row[k] = hist[k];
        hist[k] = 0;
    }
}

// ---------------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------------
template <int L, bool TRAIN>
__global__ __launch_bounds__(256, BOXATTN_DENSE_WPE) void fwd_tile_kernel(FwdTileArgs args_by_value)
{
    constexpr int C = 32, P = 4, LP = L * P;
    constexpr int kBias = 4096;                         // keeps the packed slot offset non-negative
    constexpr int kResPitch = 36;                       // floats per query of the result tiles
    extern __shared__ __attribute__((aligned(16))) unsigned char win_lds[];
    const BOXATTN_KARG FwdTileArgs *A =
        (const BOXATTN_KARG FwdTileArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const unsigned xcd = blockIdx.x & 7u, kq = blockIdx.x >> 3;
    if (TRAIN) {
        int *hist0 = reinterpret_cast<int *>(win_lds + A->tb.lds_hist);
        for (int k = threadIdx.x; k < A->tb.plan.nblk; k += 256) hist0[k] = 0;     // (the first unit's barrier covers it)
    }
    if (threadIdx.x < 4)                                           // the row of zeros (nothing else is ever written there)
        *reinterpret_cast<dense_u32x4 *>(win_lds + A->pl.zero_off + 16 * threadIdx.x) = dense_u32x4{0u, 0u, 0u, 0u};
    TileState<L> t;
    t.lq = -1;
    // per lane a workgroup carries from unit to unit: its thread index and its query (bit 31: outside the map)
    unsigned cur_j = 0xffffffffu, qv = 0;
    int tid = threadIdx.x;
    const int U = A->pl.units_per_wg;
    for (int i = 0; i < U; ++i) {
        asm volatile("" : "+s"(A));                                // (nothing of the plan is carried from unit to unit)
        asm volatile("" : "+v"(tid));                              // (... and nothing per lane that a unit can recompute)
        const int lane = tid & (kWave - 1);
        const int qi = lane >> 2, r = lane & 3;
        const unsigned u = kq * (unsigned)U + (unsigned)i;
        unsigned j, h;
        divmod_magic(u, (unsigned)A->pl.H, A->pl.mag_h, j, h);
        if (j != cur_j) {                                          // workgroup-uniform: a new tile
            cur_j = j;
            tile_decode<L>(A->pl, xcd, j, t);
            if (t.lq < 0) return;                                  // past the end of this XCD's queue
            const int lqc = t.lq;
            const int QH = A->pl.lv[lqc].H, QW = A->pl.lv[lqc].W, Qs = A->pl.lv[lqc].start;
            const int qy = t.ty * kDenseTile + (wv >> 1) * kDenseSub + (qi >> 2);
            const int qx = t.tx * kDenseTile + (wv & 1) * kDenseSub + (qi & 3);
            qv = (unsigned)(Qs + min(qy, QH - 1) * QW + min(qx, QW - 1)) | (qy < QH && qx < QW ? 0u : 0x80000000u);
        }
        const bool vq = (int)qv >= 0;
        const unsigned q = qv & 0x7fffffffu;
        DenseHot<L> hot;
        DenseWinPos win[L];
        tile_hot<L>(A->pl, t, hot, win);
        const int H = hot.H;
        const int kZeroOff = A->pl.zero_off;
        int *hist = reinterpret_cast<int *>(win_lds + A->tb.lds_hist);
        const unsigned qh = (t.b * (unsigned)hot.Lq + q) * (unsigned)H + h;
        const unsigned pt0 = qh * (unsigned)LP;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(A->value), 0, A->value_bytes, 0x00020000);
        const float2 *loc2 = reinterpret_cast<const float2 *>(A->loc);
        const float *attn = A->attn;
        float2 xy[L];
        float a[L];
        // (training: the locations are requested BEFORE the window rows -- they are counted while the rows fly)
        if (!TRAIN) tile_stage_rows<L>(hot, win, t.b, h, lane, wv, rs, win_lds);
#pragma unroll
        for (int l = 0; l < L; ++l) xy[l] = loc2[pt0 + l * P + r];  // lane r of the quad: point r of every level
        if (TRAIN) tile_stage_rows<L>(hot, win, t.b, h, lane, wv, rs, win_lds);
#pragma unroll
        for (int l = 0; l < L; ++l) a[l] = attn[pt0 + l * P + r];
        if (TRAIN) {
            // The backward's count pass: every point into the <= 4 destination blocks of its footprint, in the LDS
            // histogram of the unit's slice.  Here, in front of the barrier that waits for the window rows -- few
            // registers are live, and the wave has nothing else to do -- at the price of locating the points twice
            // (inside the level loop below the block arithmetic pushed two accumulators out to scratch at every level).
#pragma unroll
            for (int l = 0; l < L; ++l) {
                const DenseMap T = hot.lv[l];
                const DensePoint s = dense_locate(xy[l].x, xy[l].y, T.H, T.W);
                const int ra = max(s.y0, 0), rb = min(s.y0 + 1, T.H - 1), ca = max(s.x0, 0), cb = min(s.x0 + 1, T.W - 1);
                int blk[4];
                tile_blocks(ra, rb, ca, cb, A->tb.plan.lv[l].nbx, A->tb.plan.lv[l].nby, A->tb.plan.lv[l].blk0,
                            A->tb.plan.lv[l].mw, A->tb.plan.lv[l].mh, blk);
                if (vq && s.inside) {
                    lds_count(hist + blk[0]);
#pragma unroll
                    for (int k = 1; k < 4; ++k)
                        if (blk[k] >= 0) lds_count(hist + blk[k]);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        dense_stage_wait();                                        // windows complete

        unsigned n_slow = 0, n_act = 0;
        fwd_f32x4 acc[8];
        float accv[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) { acc[m] = fwd_f32x4{0.f, 0.f, 0.f, 0.f}; accv[m] = 0.f; }
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const DenseMap T = hot.lv[l];
            const DenseWinPos &o = win[l];
            // supplier role inside the 16-lane group: the row of corner jc for the query of quad qs (formed per level
            // from the thread index: kept across the levels, one of these per-lane constants went to scratch)
            int ll = tid;
            asm volatile("" : "+v"(ll));
            const int jc = (ll >> 2) & 3, qs = ll & 3;
            const unsigned odd_mask = 0u - (unsigned)(ll & 1);
            const int src_lane4 = ((ll & 48) + 4 * qs) * 4;        // ds_bpermute address of lane 0 of that quad
            const DensePoint s = dense_locate(xy[l].x, xy[l].y, T.H, T.W);
            const int rows = o.rows(), cols = o.cols();
            const int Hm1 = T.H - 1, Wm1 = T.W - 1;
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
            unsigned z01[2], z23[2];
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) {
                z01[uu] = (quad_pairs_u32(lo01, uu) & odd_mask) | (quad_pairs_u32(hi01, uu) & ~odd_mask);
                z23[uu] = (quad_pairs_u32(lo23, uu) & odd_mask) | (quad_pairs_u32(hi23, uu) & ~odd_mask);
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
                        for (int ii = 0; ii < 4; ++ii) {
                            accv[2 * ii] = fmaf(wv_[k], __uint_as_float(w4[ii] << 16), accv[2 * ii]);
                            accv[2 * ii + 1] = fmaf(wv_[k], __uint_as_float(w4[ii] & 0xffff0000u), accv[2 * ii + 1]);
                        }
                    }
                }
            }
        }
        // How local were this wave's points?  {points served from global memory, points inside the window test} into
        // the caller's locality counters, from one unit in 61 (hints, include/boxattn.h); nobody waits for these atomics
        unsigned long long *stats = A->stats;
        const unsigned un = blockIdx.x * (unsigned)U + (unsigned)i;
        if (stats && lane == 0 && un % 61u == 0u) {
            unsigned long long *slot = stats + 2 * (((un / 61u) * 4u + (unsigned)wv) & (kDenseStatSlots - 1));
            __hip_atomic_fetch_add(slot, (unsigned long long)n_slow, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(slot + 1, (unsigned long long)n_act, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        // ---- lane r of the quad holds channels r, 4 + r, .. 28 + r from the matrix cores: through LDS into the
        //      channel-contiguous order of the VALU sums, then one 16-byte piece of the query's row per lane
        __syncthreads();                                               // every wave is done with the windows (and has counted)
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
            unsigned qhe = qh;
            asm volatile("" : "+v"(qhe));                              // (the row's address is formed here, not carried)
            *reinterpret_cast<dense_u32x4 *>(A->out + (size_t)qhe * C + 8 * r) = o4;
        }
        if (TRAIN) {
            const unsigned uid = ((t.b * (unsigned)A->tb.tiles_per_image + tile_first<L>(A->tb, t.lq) + t.tr) * (unsigned)H + h);
            tile_dump(A->tb, hist, uid, tid);
        }
        __syncthreads();                                               // the result tiles have been read (the histogram is zero
                                                                       // again): the next unit's rows may land
    }
}

// ---------------------------------------------------------------------------------------------------------
// point gradients (+ the backward's fill pass)
// ---------------------------------------------------------------------------------------------------------
template <int L, bool FILL>
__global__ __launch_bounds__(256, BOXATTN_DENSE_WPE) void pointgrad_tile_kernel(PgTileArgs args_by_value)
{
    constexpr int C = 32, P = 4, LP = L * P;
    extern __shared__ __attribute__((aligned(16))) unsigned char win_lds[];
    const BOXATTN_KARG PgTileArgs *A =
        (const BOXATTN_KARG PgTileArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const unsigned xcd = blockIdx.x & 7u, kq = blockIdx.x >> 3;
    TileState<L> t;
    t.lq = -1;
    unsigned cur_j = 0xffffffffu, qv = 0;
    int tid = threadIdx.x;
    const int U = A->pl.units_per_wg;
    for (int i = 0; i < U; ++i) {
        asm volatile("" : "+s"(A));
        asm volatile("" : "+v"(tid));
        const int lane = tid & (kWave - 1);
        const int qi = lane >> 2, p = lane & 3;
        const unsigned u = kq * (unsigned)U + (unsigned)i;
        unsigned j, h;
        divmod_magic(u, (unsigned)A->pl.H, A->pl.mag_h, j, h);
        if (j != cur_j) {
            cur_j = j;
            tile_decode<L>(A->pl, xcd, j, t);
            if (t.lq < 0) return;
            const int lqc = t.lq;
            const int QH = A->pl.lv[lqc].H, QW = A->pl.lv[lqc].W, Qs = A->pl.lv[lqc].start;
            const int qy = t.ty * kDenseTile + (wv >> 1) * kDenseSub + (qi >> 2);
            const int qx = t.tx * kDenseTile + (wv & 1) * kDenseSub + (qi & 3);
            qv = (unsigned)(Qs + min(qy, QH - 1) * QW + min(qx, QW - 1)) | (qy < QH && qx < QW ? 0u : 0x80000000u);
        }
        const bool vq = (int)qv >= 0;
        const unsigned q = qv & 0x7fffffffu;
        DenseHot<L> hot;
        DenseWinPos win[L];
        tile_hot<L>(A->pl, t, hot, win);
        const int H = hot.H;
        int *hist = reinterpret_cast<int *>(win_lds + A->tb.lds_hist);
        // (the attention weights are needed last, for the location gradients: parked in a lane-private piece of LDS
        // meanwhile; an LDS pointer by type -- address spaces are not inferred for volatile accesses)
        typedef __attribute__((address_space(3))) float lds_float;
        volatile lds_float *a_stash = (volatile lds_float *)(win_lds + A->pl.stash_off) + 4 * tid;
        const unsigned qh = (t.b * (unsigned)hot.Lq + q) * (unsigned)H + h;
        const unsigned pt0 = qh * (unsigned)LP;
        const int s_id = (int)(t.b * (unsigned)H + h);
        const bf16_t *value = A->value;
        const __amdgpu_buffer_rsrc_t rs =
            __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(value), 0, A->value_bytes, 0x00020000);
        tile_stage_rows<L>(hot, win, t.b, h, lane, wv, rs, win_lds);
        const float2 *loc2 = reinterpret_cast<const float2 *>(A->loc);
        const float *attn = A->attn;
        float2 xy[L];
        float a[L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            xy[l] = loc2[pt0 + l * P + p];
            a[l] = attn[pt0 + l * P + p];
        }
        unsigned gw[16];                                               // the query's grad_out row
        dense_load_row(A->grad_out + (size_t)qh * C, gw);
        if (FILL) {
            // the unit's first slot in every block: the block's first record + what the forward's atomic returned
            // (entries of blocks the unit does not touch are never read)
            const int nblk = A->tb.plan.nblk;
            const unsigned t0 = tile_first<L>(A->tb, t.lq);
            const unsigned uid = ((t.b * (unsigned)A->tb.tiles_per_image + t0 + t.tr) * (unsigned)H + h);
            const int *row = A->tb.rows + (size_t)uid * nblk;
            const int *off = A->tb.out.offsets + (size_t)s_id * (nblk + 1);
            for (int k = tid; k < nblk; k += 256) hist[k] = off[k] + row[k];
            int *ctickets = A->tb.ctickets;
            if (ctickets && t0 + t.tr == 0u)                           // one unit per slice clears the combine tickets
                for (int k = tid; k < nblk; k += 256) ctickets[(size_t)s_id * nblk + k] = 0;
        }
#pragma unroll
        for (int l = 0; l < L; ++l) a_stash[l] = a[l];
        dense_stage_wait();                                            // windows (and the histogram) complete

        float ga[L], gx[L], gy[L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
            const DenseMap T = hot.lv[l];
            const DensePoint s = dense_locate(xy[l].x, xy[l].y, T.H, T.W);
            if (FILL) {                                                // the backward's fill pass, on the point in hand
                const int Hm1 = T.H - 1, Wm1 = T.W - 1;
                const int ra = max(s.y0, 0), rb = min(s.y0 + 1, Hm1), ca = max(s.x0, 0), cb = min(s.x0 + 1, Wm1);
                int blk[4];
                tile_blocks(ra, rb, ca, cb, A->tb.plan.lv[l].nbx, A->tb.plan.lv[l].nby, A->tb.plan.lv[l].blk0,
                            A->tb.plan.lv[l].mw, A->tb.plan.lv[l].mh, blk);
                if (vq && s.inside) {
                    int4 *rec = reinterpret_cast<int4 *>(A->tb.records) + (size_t)s_id * A->tb.plan.rec_cap;
                    const int4 rv = make_int4((int)((q << A->tb.plan.lp_bits) | (unsigned)(l * P + p)), __float_as_int(xy[l].x),
                                              __float_as_int(xy[l].y), __float_as_int(a_stash[l]));
#ifndef BOXATTN_EXP_FILL
#define BOXATTN_EXP_FILL 0          // experiments: 1 no record stores (atomics only), 2 no atomics (slot = lane), 3 neither
#endif
                    auto take = [&](int b) -> int { return (BOXATTN_EXP_FILL & 2) ? hist[b] + lane : lds_take(hist + b); };
                    auto put = [&](int slot) {
                        if (BOXATTN_EXP_FILL & 1) asm volatile("" ::"v"(slot));
                        else rec[slot] = rv;
                    };
                    put(take(blk[0]));
#pragma unroll
                    for (int k = 1; k < 4; ++k)
                        if (blk[k] >= 0) put(take(blk[k]));
                }
                __builtin_amdgcn_sched_barrier(0);         // (not interleaved with the corner rows below: registers)
            }
            float sk[4];
            dense_corner_sums(s, T, win[l], win_lds, gw, value, t.b * (unsigned)hot.S + (unsigned)T.start, H, (int)h, vq, p, sk);
            const float w1 = s.hh * s.hw, w2 = s.hh * s.lw, w3 = s.lh * s.hw, w4 = s.lh * s.lw;
            const float gs_ = w1 * sk[0] + w2 * sk[1] + w3 * sk[2] + w4 * sk[3];
            const float al = a_stash[l];
            const float gx_ = (float)T.W * al * (s.hh * (sk[1] - sk[0]) + s.lh * (sk[3] - sk[2]));
            const float gy_ = (float)T.H * al * (s.hw * (sk[2] - sk[0]) + s.lw * (sk[3] - sk[1]));
            ga[l] = s.inside ? gs_ : 0.f;
            gx[l] = s.inside ? gx_ : 0.f;
            gy[l] = s.inside ? gy_ : 0.f;
            // (finished HERE: left alone the compiler postpones this arithmetic to the epilogue and carries nine
            // values per level -- corner sums and weights -- instead of three)
            asm volatile("" : "+v"(ga[l]), "+v"(gx[l]), "+v"(gy[l]));
        }
        // ---- results: lane (q, p) holds its point on every level; memory wants, per (query, head), [level][point]
        //      runs -- transposed through (wave-private) LDS so that lane (q, j) writes level j's 4 points as 16 + 32
        //      contiguous bytes
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
            unsigned qe = q;
            asm volatile("" : "+v"(qe));                               // (the addresses are formed here, not carried)
            const unsigned pte = ((t.b * (unsigned)hot.Lq + qe) * (unsigned)H + h) * (unsigned)LP;
            float4 *ga4 = reinterpret_cast<float4 *>(A->grad_attn + pte + p * P);
            float4 *gl = reinterpret_cast<float4 *>(A->grad_loc + 2 * (size_t)(pte + p * P));
            // (non-temporal: nobody on the GPU reads the point gradients soon; the bin records should stay cached)
            typedef float pg_f32x4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(pg_f32x4{o_a.x, o_a.y, o_a.z, o_a.w}, reinterpret_cast<pg_f32x4 *>(ga4));
            __builtin_nontemporal_store(pg_f32x4{o_0.x, o_0.y, o_0.z, o_0.w}, reinterpret_cast<pg_f32x4 *>(gl));
            __builtin_nontemporal_store(pg_f32x4{o_1.x, o_1.y, o_1.z, o_1.w}, reinterpret_cast<pg_f32x4 *>(gl + 1));
        }
        __syncthreads();                                               // the result tiles have been read
    }
}

}  // namespace boxattn
