// Window-staged kernels for the ENCODER case of box attention: one query per pixel (Lq == S, packed
// levels), bf16 storage, C = 32 channels per head, 2x2 points per level.
//
// The gather kernels (boxattn_gather2.h) pull 4 corner rows per sample point through the vector L1
// -- 64 rows per (query, head) -- and sit at the rate at which the L1 gets rows out of the L2
// (DESIGN.md 4.1).  In the encoder the queries are the pixels and a query's boxes lie around its own
// position on every level, so the rows an 8x8 TILE of queries touches on one level form a small
// window: 16x16 pixels on the tile's own level, 9x9 / 6x6 / 4x4 on the coarser ones -- 389 rows for
// 64 queries x 64 corner rows.  One workgroup = one (tile, head): its four waves fetch the windows of
// all levels once, as whole coalesced rows, straight into LDS (26 KB), and after ONE barrier every lane works
// alone:
//
//   lane = one sample point of one query (wave = 4x4 sub-tile of queries x the 2x2 points), the
//   levels one after the other.  A lane reads the four corner rows of its point from LDS (64 bytes
//   each, 4 x ds_read_b128) and multiplies them with its query's grad_out row, which it keeps in 16
//   registers: 16 v_dot2c_f32_bf16 per corner, no cross-lane step at all (the gather kernels spread
//   a row over 4 lanes because a lane per row is poison for the vector L1; from LDS it is free).
//
// The window of (tile, level) is placed by geometry alone (wave-uniform): the tile's position
// projected onto the level plus a margin for the box size and its predicted offset.  Placement and
// staging are a performance matter only: a point whose footprint is not inside its staged window --
// and every point of a level whose window is not staged (a coarse tile looking at a fine level: too
// large) -- takes the global path, the same dot products on rows fetched by the lane itself.  The
// results are the same either way, for any input.
//
// (Round 3 first built this on the matrix cores -- S = G V^T per 4x4 tile and window row on
// v_mfma_f32_16x16x32_bf16, look-ups from an S tile in LDS: tools/experiments/boxattn_dense_mfma_v1.h.
// Parity-green, 61 us against the gather kernel's 48: with a wave-private 12x12 window per 4x4 tile
// the L2 -> L1 traffic is the gather kernel's, and that traffic is what bounds both.)
#pragma once
#include "boxattn_device.h"
#include "boxattn_binpass.h"      // the backward's fill riders sit in the point-gradient kernel's launch
#include "boxattn_dense_plan.h"

namespace boxattn {

#ifndef BOXATTN_DENSE_DEBUG
#define BOXATTN_DENSE_DEBUG 0     // 2: s_memtime stamps of every wave (tools/gpu_dense_trace.py, DensePlan::dbg)
#endif
#ifndef BOXATTN_TUNE_PG_ROWS
#define BOXATTN_TUNE_PG_ROWS 2    // corner rows a lane has in flight from LDS (2: 8 reads, 32 registers)
#endif
#ifndef BOXATTN_DENSE_WPE
#define BOXATTN_DENSE_WPE 5       // waves per SIMD the register allocation aims at (96 registers; 26 / 30 KB of LDS: 5 workgroups per CU)
#endif
typedef unsigned int dense_u32x4 __attribute__((ext_vector_type(4)));

// one 64-byte row (32 bf16 channels) as 16 words.  (The words go through a plain array: a bit_cast
// of element i of an ext_vector selects element 0 for every i with this compiler)
template <typename PTR> __device__ __forceinline__ void dense_load_row(PTR p, unsigned (&w)[16])
{
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const dense_u32x4 t = reinterpret_cast<const dense_u32x4 *>(p)[i];
        w[4 * i] = t.x; w[4 * i + 1] = t.y; w[4 * i + 2] = t.z; w[4 * i + 3] = t.w;
    }
}
__device__ __forceinline__ float dense_dot_row(const unsigned (&g)[16], const unsigned (&v)[16])
{
    // bf16 x bf16 products are exact in fp32; two chains: a v_dot2c waits for its own accumulator
    float d0 = 0.f, d1 = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i += 2) {
        d0 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, g[i]),
                                             __builtin_bit_cast(bf16x2_t, v[i]), d0, false);
        d1 = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, g[i + 1]),
                                             __builtin_bit_cast(bf16x2_t, v[i + 1]), d1, false);
    }
    return d0 + d1;
}

// Which (tile, head) a workgroup works on.  Workgroup w runs on XCD w % 8 (observed placement; only
// speed depends on it): every XCD gets one contiguous eighth of every level's tiles, all heads of a
// tile (the two 64-byte halves of a value line belong to neighbouring heads) -- its L2 then holds one
// spatial band of the maps -- and walks the levels coarsest first, because the tiles of the coarse
// levels are the slow ones (their fine-level points take the global path).
//
// All of it is scalar work at the top of every wave -- and scalar instructions are not free: the CU's
// one scalar unit serves its four SIMDs, an s_ instruction costs a wave as much issue time as a vector
// one (PMC: all kernels of the step retire ~1 instruction per 4 cycles and SIMD whatever the mix).  The
// decode is therefore integer-only (multiply-high divisions with host magic numbers, 16.16 fixed-point
// window origins: ~100 scalar instructions; with float estimates and v_readfirstlane it was ~450), and
// it reads the plan in two batches: what is addressed statically up front, the query level's entry and
// its L windows -- the only reads at a computed offset -- right after the level is known (read one by
// one behind branches they were a third of a wave's life: 6 600 of 17 700 cycles before the first row
// request left, s_memtime stamps).
struct DenseTileId {
    int lq;                  // query level, -1: no tile
    unsigned b;
    int ty, tx, h;
};
struct DenseMap { int H, W, start; };                   // what the kernel keeps of a level
template <int L> struct DenseHot {
    DenseMap lv[L];
    int H, Lq, S;
};
template <int L>
__device__ __forceinline__ DenseTileId dense_tile_of_block(const DensePlan &pl, unsigned block,
                                                           DenseHot<L> &hot, DenseMap &Q, DenseWin (&wrow)[L])
{
    unsigned n_all[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        hot.lv[l] = DenseMap{pl.lv[l].H, pl.lv[l].W, pl.lv[l].start};
        n_all[l] = pl.lv[l].n_all;
        // (opaque to the compiler from here on: it otherwise sinks these scalar loads to their first use,
        // one round trip per level in the middle of the row requests)
        asm volatile("" : "+s"(hot.lv[l].H), "+s"(hot.lv[l].W), "+s"(hot.lv[l].start));
    }
    hot.H = pl.H; hot.Lq = pl.Lq; hot.S = pl.S;
    asm volatile("" : "+s"(hot.H), "+s"(hot.Lq), "+s"(hot.S));
    DenseTileId t;
    const unsigned x = block & 7u;
    unsigned j, h;
    divmod_magic(block >> 3, (unsigned)hot.H, pl.mag_h, j, h);
    t.h = (int)h;
    int lq = -1;
    unsigned ti = 0, first = 0;
#pragma unroll
    for (int l = L - 1; l >= 0; --l) {
        const unsigned lo = (x * n_all[l]) >> 3, cnt = (((x + 1) * n_all[l]) >> 3) - lo;
        const bool here = j - first < cnt;                 // first <= j < first + cnt
        lq = here ? l : lq;
        ti = here ? lo + (j - first) : ti;
        first += cnt;
    }
    t.lq = lq;
    const int lqc = max(lq, 0);
    const DenseLevel q = pl.lv[lqc];                       // the reads at a computed offset: one batch
#pragma unroll
    for (int l = 0; l < L; ++l) wrow[l] = pl.win[lqc][l];
    Q = DenseMap{q.H, q.W, q.start};
    asm volatile("" : "+s"(Q.H), "+s"(Q.W), "+s"(Q.start));      // (loaded with the batch, not at first use)
    unsigned tr, ty, tx;
    divmod_magic(ti, (unsigned)q.ntiles, q.mag_ntiles, t.b, tr);
    divmod_magic(tr, (unsigned)q.ntx, q.mag_ntx, ty, tx);
    t.ty = (int)ty;
    t.tx = (int)tx;
    return t;
}

// Placement of a tile's windows (wave-uniform) and the cooperative fetch: wave w takes the window rows
// w, w + 4, ... of every level; a row of up to 16 pixels x 64 bytes is ONE direct-to-LDS load (4 lanes per
// pixel; the lanes of pixels past the window's last column are switched off and write nothing).  The data
// never passes through registers: round 3 staged through 64 VGPRs and a second pass of ds_write_b128,
// which held the kernels at 4 waves per SIMD.  The rows have landed after s_waitcnt vmcnt(0) in every
// wave + the workgroup's barrier (dense_stage_wait).
struct DenseWinPos {
    unsigned geo;            // DenseWin::geo
    int x0, y0;
    __device__ __forceinline__ int rows() const { return (int)(geo & 31u); }
    __device__ __forceinline__ int cols() const { return (int)((geo >> 5) & 31u); }
    __device__ __forceinline__ int pitchb() const { return (int)((geo >> 10) & 127u) * 16; }     // bytes per window row
    __device__ __forceinline__ int offb() const { return (int)(geo >> 17) * 16; }               // first byte
};

typedef __attribute__((address_space(3))) void dense_lds_void;
// timing experiments only (phase split of the tile kernels): the windows are not fetched (the arithmetic
// runs on whatever the LDS holds) / the arithmetic is skipped (windows fetched, locations loaded, zeros stored)
#ifndef BOXATTN_DEBUG_NO_STAGE
#define BOXATTN_DEBUG_NO_STAGE 0
#endif
#ifndef BOXATTN_DEBUG_NO_MATH
#define BOXATTN_DEBUG_NO_MATH 0
#endif

template <int L>
__device__ __forceinline__ void dense_stage_issue(const DenseHot<L> &hot, const DenseWin (&wrow)[L],
                                                  const DenseTileId &t, int lane, int wv,
                                                  __amdgpu_buffer_rsrc_t rs, unsigned char *lds, DenseWinPos (&win)[L])
{
    constexpr int C = 32, RPW = kDenseWinMax / 4;              // rows per wave and level at most
    const int j = lane >> 2, chunk = lane & 3;
#pragma unroll
    for (int l = 0; l < L; ++l) {
        const DenseMap T = hot.lv[l];
        const DenseWin w = wrow[l];
        DenseWinPos &o = win[l];
        o.geo = w.geo;
        const int x0 = (t.tx * w.ax + w.bx) >> 16, y0 = (t.ty * w.ay + w.by) >> 16;
        o.x0 = max(0, min(x0, T.W - o.cols()));
        o.y0 = max(0, min(y0, T.H - o.rows()));
        const int jx = min(o.x0 + j, T.W - 1);
        const unsigned voff =
            ((((t.b * (unsigned)hot.S + (unsigned)(T.start + jx)) * (unsigned)hot.H + (unsigned)t.h) * C) +
             (unsigned)chunk * 8u) * 2u;
        const unsigned row_bytes = (unsigned)T.W * (unsigned)hot.H * (C * 2u);
        // (rows past the bottom of the map -- a window taller than the map -- are never looked up: a
        // corner row that counts lies inside the map; they are simply not fetched)
        const int rows = min(o.rows(), T.H - o.y0);
        unsigned soff = (unsigned)(o.y0 + wv) * row_bytes;
        int dst = o.offb() + wv * o.pitchb();                       // wave-uniform: it becomes M0
        const int step = 4 * o.pitchb();
        if (j < o.cols()) {
#pragma unroll
            for (int k = 0; k < RPW; ++k) {
                if (wv + 4 * k < rows && !BOXATTN_DEBUG_NO_STAGE)      // wave-uniform
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (dense_lds_void *)(lds + dst), 16, voff, soff, 0, 0);
                soff += 4u * row_bytes;
                dst += step;
            }
        }
    }
}
// every wave's rows have landed, and everybody knows
__device__ __forceinline__ void dense_stage_wait()
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

__device__ __forceinline__ unsigned quad_bcast_u32(unsigned v, int t)      // t is a constant after unrolling
{
    switch (t & 3) {
    case 0: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x00, 0xF, 0xF, true);
    case 1: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x55, 0xF, 0xF, true);
    case 2: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xAA, 0xF, 0xF, true);
    default: return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xFF, 0xF, 0xF, true);
    }
}

// A sample point as the kernels need it (locate() of boxattn_device.h without the per-corner flags:
// the flag algebra is done on integers below -- lane masks combined in scalar registers were a third
// of this kernel's instructions, and scalar instructions are not free).
struct DensePoint {
    int y0, x0;              // top-left corner of the footprint, in [-1, H - 1] x [-1, W - 1] (0, 0 if !inside)
    float lh, lw, hh, hw;
    bool inside;             // the reference's window test (the levels of a dense plan are never empty)
};
__device__ __forceinline__ DensePoint dense_locate(float x, float y, int H, int W)
{
    DensePoint s;
    float h_im, w_im;
    {
#pragma clang fp contract(off)                   // two roundings, as in locate()
        h_im = y * (float)H - 0.5f;
        w_im = x * (float)W - 0.5f;
    }
    s.inside = (h_im > -1.f) && (w_im > -1.f) && (h_im < (float)H) && (w_im < (float)W);
    const float hs = s.inside ? h_im : 0.f, ws = s.inside ? w_im : 0.f;   // NaN / outside: finite index arithmetic
    const float hf = floorf(hs), wf = floorf(ws);
    s.y0 = (int)hf;
    s.x0 = (int)wf;
    s.lh = hs - hf;
    s.lw = ws - wf;
    s.hh = 1.f - s.lh;
    s.hw = 1.f - s.lw;
    return s;
}

// The four corner sums S_k = sum_c g_c v_k,c of one sample point: from the staged window if the part of
// its footprint that lies inside the map lies inside the window, else from global memory (`slow`,
// decided per lane, entered per wave).  Corners outside the map give exactly 0.
__device__ __forceinline__ void dense_corner_sums(const DensePoint &s, const DenseMap &T, const DenseWinPos &o,
                                                  const unsigned char *lds, const unsigned (&gw)[16],
                                                  const bf16_t *value, unsigned row0, int H, int h, bool vq,
                                                  int lane_p, float (&sk)[4])
{
    constexpr int C = 32;
    const int rows = o.rows(), cols = o.cols();
    const int Hm1 = T.H - 1, Wm1 = T.W - 1;
    // all-ones where the corner counts: inside the map (row y0 unless y0 == -1, row y0 + 1 unless y0 == H - 1,
    // columns alike) and the point inside the window test
    const unsigned mi = s.inside ? 0xffffffffu : 0u;
    const unsigned mr0 = ~(unsigned)(s.y0 >> 31) & mi, mr1 = (unsigned)((s.y0 - Hm1) >> 31) & mi;
    const unsigned mc0 = ~(unsigned)(s.x0 >> 31), mc1 = (unsigned)((s.x0 - Wm1) >> 31);
    const unsigned m[4] = {mr0 & mc0, mr0 & mc1, mr1 & mc0, mr1 & mc1};
    // rows / columns of the map the footprint needs, and how far they are inside the staged window
    // (negative: something is missing; an unstaged level has rows == 0, so d < 0 for every point)
    const int ra = max(s.y0, 0), rb = min(s.y0 + 1, Hm1), ca = max(s.x0, 0), cb = min(s.x0 + 1, Wm1);
    const int d = min(min(ra - o.y0, o.y0 + rows - 1 - rb), min(ca - o.x0, o.x0 + cols - 1 - cb));
    const bool act = vq && s.inside;
    const bool slow = act && d < 0, fast = act && d >= 0;
    if (__builtin_amdgcn_ballot_w64(fast) != 0ull) {               // wave-uniform: somebody reads the window
        // (lanes that do not: clamped slots, results masked or overwritten below)
        const int pitchb = o.pitchb(), offb = o.offb();
        const int rr0 = min(max(s.y0 - o.y0, 0), rows - 1), rr1 = min(max(s.y0 + 1 - o.y0, 0), rows - 1);
        const int cc0 = min(max(s.x0 - o.x0, 0), cols - 1), cc1 = min(max(s.x0 + 1 - o.x0, 0), cols - 1);
        const int rowb0 = offb + __mul24(rr0, pitchb), rowb1 = offb + __mul24(rr1, pitchb);
        const int colb0 = cc0 * kDenseSlotBytes, colb1 = cc1 * kDenseSlotBytes;
        const int slot[4] = {rowb0 + colb0, rowb0 + colb1, rowb1 + colb0, rowb1 + colb1};
        // two corners at a time: 8 LDS reads in flight, then the 32 dot products; the rows of corners
        // that do not count are read from a clamped slot and masked out bitwise -- a select on the
        // finished sum lets the compiler put every corner under its own branch (reads, wait, dots, four
        // times in a row), a multiplication by 0 would let a non-finite value of an unrelated pixel through
#if BOXATTN_TUNE_PG_ROWS == 2
#pragma unroll
        for (int k0 = 0; k0 < 4; k0 += 2) {
            unsigned va[16], vb[16];
            dense_load_row(lds + slot[k0], va);
            dense_load_row(lds + slot[k0 + 1], vb);
            const float da = dense_dot_row(gw, va), db = dense_dot_row(gw, vb);
            sk[k0] = __uint_as_float(__float_as_uint(da) & m[k0]);
            sk[k0 + 1] = __uint_as_float(__float_as_uint(db) & m[k0 + 1]);
        }
#else
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned va[16];
            dense_load_row(lds + slot[k], va);
            sk[k] = __uint_as_float(__float_as_uint(dense_dot_row(gw, va)) & m[k]);
        }
#endif
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) sk[k] = 0.f;
    }
    if (__builtin_amdgcn_ballot_w64(slow) != 0ull) {               // wave-uniform
        // The global path, one point of every quad at a time (a quad = the 4 points of one query): the
        // quad's lanes fetch the four corner rows of point t together, 16 bytes each -- one 64-byte
        // request per row, as the gather kernels do it; a lane fetching whole rows alone costs the
        // vector L1 four requests per row -- and sum their partial dot products with DPP adds.
        const int p = lane_p;
        unsigned gch[4];                                           // this lane's 8 channels of the grad_out row
#pragma unroll
        for (int i = 0; i < 4; ++i)
            gch[i] = p == 0 ? gw[i] : p == 1 ? gw[4 + i] : p == 2 ? gw[8 + i] : gw[12 + i];
        const int pra = __mul24(ra, T.W), prb = __mul24(rb, T.W);
        const int pix[4] = {pra + ca, pra + cb, prb + ca, prb + cb};          // (clamped into the map: always a valid row)
        unsigned off[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            off[k] = (unsigned)(((row0 + (unsigned)pix[k]) * H + h) * (C * 2));
        // two points of the quad at a time: their 8 row pieces are requested together and unconditionally
        // (the offsets are always inside the map) -- one point at a time, each under its own "anyone
        // slow?" branch, a level was four memory round trips in a row (s_memtime: 9 000 cycles a level)
#pragma unroll
        for (int t0 = 0; t0 < 4; t0 += 2) {
            dense_u32x4 v[2][4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    // lane t's row, my 16-byte piece of it
                    const unsigned oo = quad_bcast_u32(off[k], t0 + u) + (unsigned)p * 16u;
                    v[u][k] = *reinterpret_cast<const dense_u32x4 *>(reinterpret_cast<const char *>(value) + oo);
                }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                float part[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const unsigned vw[4] = {v[u][k].x, v[u][k].y, v[u][k].z, v[u][k].w};
                    float dd = 0.f;
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        dd = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, gch[i]),
                                                            __builtin_bit_cast(bf16x2_t, vw[i]), dd, false);
                    part[k] = group_sum<4>(dd);
                }
                if (p == t0 + u && slow) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) sk[k] = __uint_as_float(__float_as_uint(part[k]) & m[k]);
                }
            }
        }
    }
}

constexpr int kDenseResFloats = 16 * 16 * 3;       // per wave: the results of 16 queries x 16 points

template <int L>
__global__ __launch_bounds__(256, BOXATTN_DENSE_WPE) void pointgrad_dense_kernel(
    const bf16_t *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ attn,
    const bf16_t *__restrict__ grad_out, float *__restrict__ grad_loc, float *__restrict__ grad_attn,
    DensePlan pl, unsigned value_bytes, BinRide ride)
{
    constexpr int C = 32, P = 4, LP = L * P;
#ifndef BOXATTN_TUNE_PG_STASH
#define BOXATTN_TUNE_PG_STASH 1    // a lane parks its L attention weights in LDS until it needs them (4 registers)
#endif
    __shared__ __attribute__((aligned(16))) unsigned char win_lds[kDenseLdsBytes + (BOXATTN_TUNE_PG_STASH ? 256 * 16 : 0)];
    static_assert(4 * kDenseResFloats * sizeof(float) <= sizeof(win_lds), "the result tiles reuse the window buffer");
    static_assert(kRideLdsInts * sizeof(int) <= sizeof(win_lds), "so do the riders");
    // (the wave index as a scalar: everything per window row -- row index, clamps, byte offsets, the
    // "row exists" branches -- is then scalar code; derived from threadIdx in a vector register it was
    // ~25 vector instructions, a v_readfirstlane and an exec-mask branch per row)
    const int lane = threadIdx.x & (kWave - 1), wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    // the backward's fill pass rides in this launch (boxattn_ride.h): rider workgroups write the bin
    // records while the tiles' workgroups compute -- the one waits on memory, the other on issue slots
    const RideRole role = ride_role(blockIdx.x, ride.grid);
    if (role.rider) {
        bin_fill_ride<256>(ride, role.id, reinterpret_cast<int *>(win_lds));
        return;
    }
#ifdef BOXATTN_DEBUG_NO_TILES          // timing experiments only: the riders alone
    return;
#endif
#if BOXATTN_DENSE_DEBUG == 2
    unsigned long long ts[12];
    int ts_n = 0;
    const unsigned long long rt0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz, the same on every XCD
#define DENSE_STAMP() do { __builtin_amdgcn_sched_barrier(0); ts[ts_n++] = __builtin_amdgcn_s_memtime(); \
                           __builtin_amdgcn_sched_barrier(0); } while (0)
    DENSE_STAMP();
#else
#define DENSE_STAMP() do { } while (0)
#endif
    DenseHot<L> hot;
    DenseMap Q;
    DenseWin wrow[L];
    const DenseTileId t = dense_tile_of_block<L>(pl, role.id, hot, Q, wrow);
    if (t.lq < 0) return;                                          // workgroup-uniform
    const int H = hot.H, h = t.h;

    // ---- lane -> (query of the wave's 4x4 sub-tile, point)
    const int qi = lane >> 2, p = lane & 3;
    const int qy = t.ty * kDenseTile + (wv >> 1) * kDenseSub + (qi >> 2);
    const int qx = t.tx * kDenseTile + (wv & 1) * kDenseSub + (qi & 3);
    const bool vq = qy < Q.H && qx < Q.W;
    const unsigned q = (unsigned)(Q.start + min(qy, Q.H - 1) * Q.W + min(qx, Q.W - 1));
    const unsigned qh = (t.b * (unsigned)hot.Lq + q) * (unsigned)H + (unsigned)h;
    const unsigned pt0 = qh * (unsigned)LP;
    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t *>(value), 0, value_bytes, 0x00020000);
    // the window rows first: they are what the workgroup's barrier waits for; the lane's own inputs
    // (locations, weights, grad_out row) are requested behind them and land while the rows are staged
    DenseWinPos win[L];
    DENSE_STAMP();                                                 // tile decoded
    dense_stage_issue<L>(hot, wrow, t, lane, wv, rs, win_lds, win);
    DENSE_STAMP();                                                 // window rows requested
    float2 xy[L];
    float a[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        // (plain loads: the training forward read these lines a launch ago and most are still in the Infinity
        // Cache -- with non-temporal loads here and in the fill riders the launch took 81 us instead of 51)
        xy[l] = loc2[pt0 + l * P + p];
        a[l] = attn[pt0 + l * P + p];
    }
    unsigned gw[16];                                               // the query's grad_out row
    dense_load_row(grad_out + (size_t)qh * C, gw);
    DENSE_STAMP();                                                 // own inputs requested
#if BOXATTN_DENSE_DEBUG == 2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DENSE_STAMP();                                                 // everything has arrived
#endif
    // (the attention weights are needed last, for the location gradients: parked in a lane-private piece of LDS
    // meanwhile -- held in registers the kernel does not fit five waves per SIMD, and the compiler's own spill
    // goes to scratch with a full memory wait per value)
    // (an LDS pointer BY TYPE: address spaces are not inferred for volatile accesses -- through a generic pointer these
    // were flat stores / loads with system scope, each followed by s_waitcnt vmcnt(0): four serialised round trips in
    // front of the barrier, with every window row and operand load of the wave still in flight)
    typedef __attribute__((address_space(3))) float dense_lds_float;
    volatile dense_lds_float *a_stash = (volatile dense_lds_float *)(win_lds + kDenseLdsBytes) + 4 * threadIdx.x;
    if (BOXATTN_TUNE_PG_STASH) {
#pragma unroll
        for (int l = 0; l < L; ++l) a_stash[l] = a[l];
    }
    dense_stage_wait();                    // windows complete
    DENSE_STAMP();

    float ga[L], gx[L], gy[L];
#pragma unroll
    for (int l = 0; l < L; ++l) {
        if (BOXATTN_DEBUG_NO_MATH) {                               // (timing experiments: loads kept, arithmetic gone)
            asm volatile("" ::"v"(xy[l].x), "v"(xy[l].y), "v"(a[l]), "v"(gw[0]), "v"(gw[15]));
            ga[l] = gx[l] = gy[l] = 0.f;
            continue;
        }
        const DenseMap T = hot.lv[l];
        const DensePoint s = dense_locate(xy[l].x, xy[l].y, T.H, T.W);
        float sk[4];
        dense_corner_sums(s, T, win[l], win_lds, gw, value, t.b * (unsigned)hot.S + (unsigned)T.start, H, h, vq, p, sk);
        const float w1 = s.hh * s.hw, w2 = s.hh * s.lw, w3 = s.lh * s.hw, w4 = s.lh * s.lw;
        const float gs_ = w1 * sk[0] + w2 * sk[1] + w3 * sk[2] + w4 * sk[3];
        const float al = BOXATTN_TUNE_PG_STASH ? a_stash[l] : a[l];
        const float gx_ = (float)T.W * al * (s.hh * (sk[1] - sk[0]) + s.lh * (sk[3] - sk[2]));
        const float gy_ = (float)T.H * al * (s.hw * (sk[2] - sk[0]) + s.lw * (sk[3] - sk[1]));
        ga[l] = s.inside ? gs_ : 0.f;
        gx[l] = s.inside ? gx_ : 0.f;
        gy[l] = s.inside ? gy_ : 0.f;
        // (finished HERE: left alone the compiler postpones this arithmetic to the epilogue and carries nine
        // values per level -- corner sums and weights -- instead of three)
        asm volatile("" : "+v"(ga[l]), "+v"(gx[l]), "+v"(gy[l]));
    }
    DENSE_STAMP();
    // ---- results: lane (q, p) holds its point on every level; memory wants, per (query, head),
    //      [level][point] runs -- transposed through (wave-private) LDS so that lane (q, j) writes
    //      level j's 4 points as 16 + 32 contiguous bytes (whole 64- / 128-byte runs per query)
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
#ifndef BOXATTN_TUNE_PG_NT
#define BOXATTN_TUNE_PG_NT 1       // the point gradients leave with non-temporal stores: nobody on the GPU reads them soon, and
#endif                             // the bin records the fill riders of this launch write should stay in the Infinity Cache
        float4 *ga4 = reinterpret_cast<float4 *>(grad_attn + pt0 + p * P);
        float4 *gl = reinterpret_cast<float4 *>(grad_loc + 2 * (size_t)(pt0 + p * P));
        if (BOXATTN_TUNE_PG_NT) {
            typedef float pg_f32x4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(pg_f32x4{o_a.x, o_a.y, o_a.z, o_a.w}, reinterpret_cast<pg_f32x4 *>(ga4));
            __builtin_nontemporal_store(pg_f32x4{o_0.x, o_0.y, o_0.z, o_0.w}, reinterpret_cast<pg_f32x4 *>(gl));
            __builtin_nontemporal_store(pg_f32x4{o_1.x, o_1.y, o_1.z, o_1.w}, reinterpret_cast<pg_f32x4 *>(gl + 1));
        } else {
            *ga4 = o_a;
            gl[0] = o_0;
            gl[1] = o_1;
        }
    }
#if BOXATTN_DENSE_DEBUG == 2
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DENSE_STAMP();
    if (pl.dbg && lane == 0) {
        float *o = pl.dbg + ((size_t)role.id * 4 + wv) * 20;
        o[0] = (float)t.lq;
        o[1] = (float)(unsigned)(ts[0] & 0xffffffu);
        for (int i = 1; i < ts_n; ++i) o[1 + i] = (float)(unsigned)(ts[i] - ts[0]);
        o[19] = (float)ts_n;
        o[17] = (float)(unsigned)(rt0 & 0xffffffu);
        o[18] = (float)(unsigned)(__builtin_amdgcn_s_memrealtime() & 0xffffffu);
    }
#endif
}

}  // namespace boxattn
