// Query-grid kernels: the encoder case (one query per pixel of the multi-level map, Lq = S,
// box_transformer.py:346-354 `query = src + pos`, `value = src`).
//
// A workgroup = one head x one TILE of TX x TY neighbouring queries of one level.  Neighbouring
// queries sample neighbouring pixels, so the rows (pixel, head) the tile needs form a small
// window per value level.  The second-generation gather kernels fetched every corner row of
// every point through the vector L1 (64 rows per query, 871 MB per pass at BoxeR-R50 shapes for
// 27 MB of compulsory bytes) and were pinned at the L2 -> L1 request rate.
// Here the windows are staged ONCE per workgroup in LDS (~3-5 rows per query) and the corner
// reads are ds_read_b128:
//
//   1. every lane loads its points' locations / weights and locates them (lane (pair, slot)
//      owns the points 4 k + slot of its (query, head) pair, k = round);
//   2. how far a point's footprint reaches beyond the tile's own footprint at its level is
//      max-reduced per level (DPP inside a wave, LDS across waves): the window margin follows the
//      DATA -- trained box sizes, valid ratios and rotations need no host-side assumption;
//   3. the windows (clipped to the map, capped by the LDS budget, coarsest level first) are
//      copied global -> LDS, 16 bytes per lane;
//   4. per point: four LDS rows, packed FMAs -- the arithmetic of fwd2_kernel.  A point whose
//      footprint lies outside its window (outliers, levels whose window did not fit: coarse-level
//      queries looking at fine levels) reads the zero row there and gets its rows from global
//      memory in a second, wave-uniformly skipped, step.
//
// Nothing about the result depends on the query <-> pixel correspondence; it only decides
// where the windows are put.  Any input is handled exactly; local inputs are handled fast.
#pragma once
#include "boxattn_gather2.h"

namespace boxattn {

constexpr int kTileMaxLevels = 8;

struct TileLevel {
    int H, W, start;          // the level as value map / as query grid
    int ntx, tile0;           // query tiles of this level: tiles per row, first tile id
    float rw, rh;             // 1 / W, 1 / H
};
struct TilePlan {
    int L, n_tiles;           // levels; tiles per image
    int row_budget;           // LDS rows (of C channels) available for the windows
    int margin_cap;           // cap of the data-driven margin, in pixels of the window's level
    int static_q16;           // > 0: fixed margin of static_q16 / 16 query-level pixels (+1) instead
    int ablate;               // timing experiments only (wrong results): 1 no compute, 2 no staging
    TileLevel lv[kTileMaxLevels];
};

// max over the 64 lanes of a wave (non-negative ints), result wave-uniform
__device__ __forceinline__ int wave_max_nonneg(int v)
{
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xF, 0xF, true));   // row_half_mirror
    v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x128, 0xF, 0xF, true));   // row_ror:8
    const int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16);
    const int c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
    return max(max(a, b), max(c, d));
}

// One located sample point, kept in registers between the two phases of a tile kernel.
struct TilePoint {
    float lw, lh, a;          // bilinear fractions, attention weight (0 for padding points)
    int xy0;                  // (x0 + 1) | (y0 + 1) << 16, unclamped top-left corner (>= -1)
    unsigned flags;           // bit k: corner k inside the map; bit 4: passes the window test
};

// Pixel footprint [lo, hi] at a level of size n of queries lo_q..hi_q of a level of size nq
// (queries sit at pixel centres): the corner columns a sample AT the query centres touches.
__device__ __forceinline__ void tile_footprint(int lo_q, int hi_q, float rcp_nq, int n, int &lo,
                                               int &hi)
{
    lo = (int)floorf(((float)lo_q + 0.5f) * rcp_nq * (float)n - 0.5f);
    hi = (int)floorf(((float)hi_q + 0.5f) * rcp_nq * (float)n - 0.5f) + 1;
    lo = max(lo, 0);
    hi = min(hi, n - 1);
}

// 16 bytes LDS <-> registers
typedef unsigned int tile_u32x4 __attribute__((ext_vector_type(4)));

template <typename ST, int VEC>
__device__ __forceinline__ void row_load_lds(const unsigned char *p, Row<ST, VEC> &v)
{
    constexpr int NW = Row<ST, VEC>::NW;
    static_assert(NW % 4 == 0, "16-byte pieces");
#pragma unroll
    for (int i = 0; i < NW / 4; ++i) {
        const tile_u32x4 t = *reinterpret_cast<const tile_u32x4 *>(p + 64 * i);   // pieces 64 B apart (G = 4)
        v.w[4 * i] = t.x; v.w[4 * i + 1] = t.y; v.w[4 * i + 2] = t.z; v.w[4 * i + 3] = t.w;
    }
}

// Shared state of a tile workgroup.
template <int NWAVES> struct TileShared {
    TileLevel lv[kTileMaxLevels];        // the plan's levels (a kernel argument cannot be indexed dynamically)
    int4 win[kTileMaxLevels];            // {wx0, wy0, ww, byte offset of the window (-1: not staged)}
    int wh[kTileMaxLevels];
    int need[NWAVES][kTileMaxLevels];    // per wave: margin its points need, per level
    int rows[kTileMaxLevels];            // rows of the window with the data-driven margin
    int4 foot[kTileMaxLevels];           // tile footprint {x lo, x hi, y lo, y hi} per level
};

// Steps 1-3 of the header comment.  On return (after a workgroup barrier) the windows are in
// `rows_lds` (row 0 = zeros), `sh.win` describes them and pts[] holds the lane's points.
//   TX, TY   queries per tile;  NR = L * P / 4 rounds;  P % 4 == 0 (a round = one level)
template <typename ST, int NR, int TX, int TY>
__device__ __forceinline__ void tile_setup(
    const TilePlan &plan, const ST *__restrict__ value, const float *__restrict__ loc,
    const float *__restrict__ attn, int S, int H, int Lq, int P, unsigned char *rows_lds,
    TileShared<TX * TY * 4 / 64> &sh, TilePoint (&pts)[NR], int &b, int &h, unsigned &qh,
    bool &active)
{
    constexpr int C = 32, NT = TX * TY * 4, NWAVES = NT / 64;
    constexpr int ROWB = C * (int)sizeof(ST), PPR = ROWB / 16;    // 16-byte pieces per row
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, slot = tid & 3, pr = tid >> 2;
    const int L = plan.L;
#pragma unroll
    for (int l = 0; l < kTileMaxLevels; ++l)
        if (tid == l) sh.lv[l] = plan.lv[l];
    __syncthreads();

    // ---- which tile
    const unsigned bid = blockIdx.x;
    h = (int)(bid % (unsigned)H);
    const unsigned tf = bid / (unsigned)H;
    const int t = (int)(tf % (unsigned)plan.n_tiles);
    b = (int)(tf / (unsigned)plan.n_tiles);
    int lq = 0;
    for (int l = 1; l < L; ++l)
        if (t >= sh.lv[l].tile0) lq = l;
    const TileLevel ql = sh.lv[lq];
    const int tt = t - ql.tile0;
    const int qx0 = (tt % ql.ntx) * TX, qy0 = (tt / ql.ntx) * TY;
    const int qx1 = min(qx0 + TX - 1, ql.W - 1), qy1 = min(qy0 + TY - 1, ql.H - 1);
    const int qx = qx0 + pr % TX, qy = qy0 + pr / TX;
    active = qx <= qx1 && qy <= qy1;
    const int q = ql.start + min(qy, qy1) * ql.W + min(qx, qx1);
    qh = ((unsigned)b * (unsigned)Lq + (unsigned)q) * (unsigned)H + (unsigned)h;
    const size_t pt0 = (size_t)qh * (NR * 4);

    // ---- 1. locations / weights of my points: all loads first
    float2 xy[NR];
    float aw[NR];
    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        xy[k] = loc2[pt0 + 4 * k + slot];
        aw[k] = attn[pt0 + 4 * k + slot];
    }
    if (tid < L) {                                  // the tile's own footprint per level
        const TileLevel vl = sh.lv[tid];
        int x_lo, x_hi, y_lo, y_hi;
        tile_footprint(qx0, qx1, ql.rw, vl.W, x_lo, x_hi);
        tile_footprint(qy0, qy1, ql.rh, vl.H, y_lo, y_hi);
        sh.foot[tid] = make_int4(x_lo, x_hi, y_lo, y_hi);
    }
    if (tid < PPR) reinterpret_cast<tile_u32x4 *>(rows_lds)[tid] = tile_u32x4{0u, 0u, 0u, 0u};   // row 0
    __syncthreads();

    // ---- 2. locate; margin each level needs
    const float rcp_p = 1.0f / (float)P;
#pragma unroll
    for (int k = 0; k < NR; ++k) {
        const int l = (int)(((float)(4 * k) + 0.5f) * rcp_p);        // level of this round
        const TileLevel vl = sh.lv[l];
        const Sample<float> s = locate<float>(xy[k].x, xy[k].y, vl.H, vl.W);
        pts[k].lw = s.lw;
        pts[k].lh = s.lh;
        pts[k].a = active ? aw[k] : 0.f;
        pts[k].xy0 = (s.x0 + 1) | ((s.y0 + 1) << 16);
        pts[k].flags = (s.ok[0] ? 1u : 0u) | (s.ok[1] ? 2u : 0u) | (s.ok[2] ? 4u : 0u) |
                       (s.ok[3] ? 8u : 0u) | (s.inside ? 16u : 0u);
        const int4 f = sh.foot[l];
        const int xa = max(s.x0, 0), xb = min(s.x0 + 1, vl.W - 1);
        const int ya = max(s.y0, 0), yb = min(s.y0 + 1, vl.H - 1);
        if (plan.static_q16 <= 0) {
            int need = max(max(f.x - xa, xb - f.y), max(f.z - ya, yb - f.w));
            need = (s.inside && active) ? max(need, 0) : 0;
            const int m = wave_max_nonneg(need);
            if (lane == 0) sh.need[wv][k] = m;
        }
    }
    if (plan.static_q16 <= 0) __syncthreads();
    if (tid < L) {
        int m = 0;
        const TileLevel vl0 = sh.lv[tid];
        if (plan.static_q16 > 0) {
            m = (int)ceilf((float)plan.static_q16 * (1.f / 16.f) *
                           fmaxf((float)vl0.W * ql.rw, (float)vl0.H * ql.rh)) + 1;
        } else {
            for (int k = 0; k < NR; ++k) {
                const int l = (int)(((float)(4 * k) + 0.5f) * rcp_p);
                if (l == tid)
                    for (int w = 0; w < NWAVES; ++w) m = max(m, sh.need[w][k]);
            }
        }
        m = min(m, plan.margin_cap);
        const TileLevel vl = sh.lv[tid];
        const int4 f = sh.foot[tid];
        const int x0 = max(f.x - m, 0), x1 = min(f.y + m, vl.W - 1);
        const int y0 = max(f.z - m, 0), y1 = min(f.w + m, vl.H - 1);
        sh.win[tid] = make_int4(x0, y0, x1 - x0 + 1, -1);
        sh.wh[tid] = y1 - y0 + 1;
        sh.rows[tid] = (x1 - x0 + 1) * (y1 - y0 + 1);
    }
    __syncthreads();
    // budget: coarsest level first (small windows, many points per row); every thread computes
    // the same assignment, thread 0 publishes it
    {
        int used = 1;                                               // row 0 = zeros
        int base[kTileMaxLevels];
#pragma unroll
        for (int l = kTileMaxLevels - 1; l >= 0; --l) {
            base[l] = -1;
            if (l < L) {
                const int r = sh.rows[l];
                if (used + r <= plan.row_budget) {
                    base[l] = used * ROWB;
                    used += r;
                }
            }
        }
        if (tid == 0) {
#pragma unroll
            for (int l = 0; l < kTileMaxLevels; ++l)
                if (l < L) sh.win[l].w = base[l];
        }
        // ---- 3. stage the windows: thread -> (row r, 16-byte piece), 4 rows in flight
        const int piece = tid % PPR;
#pragma unroll
        for (int l = 0; l < kTileMaxLevels; ++l) {
            if (l >= L || base[l] < 0 || (plan.ablate & 2)) continue;   // uniform
            const TileLevel vl = sh.lv[l];
            const int4 w = sh.win[l];
            const int n = sh.rows[l], ww = w.z;
            const float rcp_ww = 1.0f / (float)ww;
            const unsigned char *src = reinterpret_cast<const unsigned char *>(value) +
                                       ((size_t)b * S + vl.start) * (size_t)H * ROWB +
                                       (size_t)h * ROWB + piece * 16;
            unsigned char *dst = rows_lds + base[l] + piece * 16;
            constexpr int RSTEP = NT / PPR, UN = 4;
            for (int r0 = tid / PPR; r0 < n; r0 += RSTEP * UN) {
                tile_u32x4 v[UN];
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int r = r0 + u * RSTEP;
                    if (r < n) {
                        int ry, rx;
                        divmod_small(r, ww, rcp_ww, ry, rx);
                        const size_t pix = (size_t)(w.y + ry) * vl.W + (w.x + rx);
                        v[u] = *reinterpret_cast<const tile_u32x4 *>(src + pix * (size_t)H * ROWB);
                    }
                }
#pragma unroll
                for (int u = 0; u < UN; ++u) {
                    const int r = r0 + u * RSTEP;
                    if (r < n) *reinterpret_cast<tile_u32x4 *>(dst + (size_t)r * ROWB) = v[u];
                }
            }
        }
    }
    __syncthreads();
}

// LDS byte offsets of the four corners of a point (0 = the zero row for corners outside the
// map), or -- fb -- its global byte offsets when the footprint is not inside the staged window.
template <typename ST>
__device__ __forceinline__ void tile_corner_offsets(const TilePoint &p, const int4 &w, int wh,
                                                    const TileLevel &vl, int b, int S, int H, int h,
                                                    tile_u32x4 &off, bool &fb)
{
    constexpr int C = 32, ROWB = C * (int)sizeof(ST);
    const int x0 = (p.xy0 & 0xffff) - 1, y0 = (p.xy0 >> 16) - 1;
    const int dx = x0 - w.x, dy = y0 - w.y;
    const bool okx0 = (unsigned)dx < (unsigned)w.z, okx1 = (unsigned)(dx + 1) < (unsigned)w.z;
    const bool oky0 = (unsigned)dy < (unsigned)wh, oky1 = (unsigned)(dy + 1) < (unsigned)wh;
    const bool v0 = p.flags & 1u, v1 = p.flags & 2u, v2 = p.flags & 4u, v3 = p.flags & 8u;
    const bool in = (!v0 || (okx0 && oky0)) && (!v1 || (okx1 && oky0)) &&
                    (!v2 || (okx0 && oky1)) && (!v3 || (okx1 && oky1));
    fb = (p.flags & 15u) != 0u && (w.w < 0 || !in);
    if (!fb) {
        const int o = w.w + (dy * w.z + dx) * ROWB;
        off.x = v0 ? (unsigned)o : 0u;
        off.y = v1 ? (unsigned)(o + ROWB) : 0u;
        off.z = v2 ? (unsigned)(o + w.z * ROWB) : 0u;
        off.w = v3 ? (unsigned)(o + w.z * ROWB + ROWB) : 0u;
    } else {
        const unsigned row0 = (unsigned)b * (unsigned)S + (unsigned)vl.start;
        const unsigned xx0 = (unsigned)max(x0, 0), xx1 = (unsigned)min(x0 + 1, vl.W - 1);
        const unsigned yy0 = (unsigned)max(y0, 0), yy1 = (unsigned)min(y0 + 1, vl.H - 1);
        const unsigned W = (unsigned)vl.W, HH = (unsigned)H, hh = (unsigned)h;
        off.x = v0 ? ((row0 + yy0 * W + xx0) * HH + hh) * ROWB : kOobOffset;
        off.y = v1 ? ((row0 + yy0 * W + xx1) * HH + hh) * ROWB : kOobOffset;
        off.z = v2 ? ((row0 + yy1 * W + xx0) * HH + hh) * ROWB : kOobOffset;
        off.w = v3 ? ((row0 + yy1 * W + xx1) * HH + hh) * ROWB : kOobOffset;
    }
}

// ---------------------------------------------------------------------------------------
// forward (box attention), G = 4 lanes per (query, head) pair, 8 channels per lane
// ---------------------------------------------------------------------------------------
template <typename ST, int NR, int TX, int TY>
__global__ __launch_bounds__(TX *TY * 4, TX *TY * 4 / 128) void fwd_tile_kernel(
    const ST *__restrict__ value, const float *__restrict__ loc, const float *__restrict__ attn,
    TilePlan plan, int S, int H, int Lq, int P, ST *__restrict__ out, unsigned value_bytes)
{
    constexpr int C = 32, VEC = 8, G = 4;
    typedef Row<ST, VEC> RowT;
    constexpr int PSB = RowGeom<ST, VEC, G>::kPieceStride;
    constexpr int LCH = RowT::kLaneBytes / (int)sizeof(ST);
    extern __shared__ __attribute__((aligned(16))) unsigned char rows_lds[];
    __shared__ TileShared<TX * TY * 4 / 64> sh;

    TilePoint pts[NR];
    int b, h;
    unsigned qh;
    bool active;
    tile_setup<ST, NR, TX, TY>(plan, value, loc, attn, S, H, Lq, P, rows_lds, sh, pts, b, h, qh,
                               active);

    const int slot = threadIdx.x & 3;
    const unsigned lane_off = (unsigned)(slot * RowT::kLaneBytes);
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<ST *>(value), 0, value_bytes, 0x00020000);
    const float rcp_p = 1.0f / (float)P;

    f32x2 acc[VEC / 2];
#pragma unroll
    for (int i = 0; i < VEC / 2; ++i) acc[i] = f32x2{0.f, 0.f};

#pragma unroll
    for (int k = 0; k < NR; ++k) {
        if (plan.ablate & 1) break;
        const int l = (int)(((float)(4 * k) + 0.5f) * rcp_p);
        const TilePoint p = pts[k];
        tile_u32x4 my_off;
        bool my_fb;
        tile_corner_offsets<ST>(p, sh.win[l], sh.wh[l], sh.lv[l], b, S, H, h, my_off, my_fb);
        const float hw = 1.f - p.lw, hh = 1.f - p.lh;
        const bool in = (p.flags & 16u) != 0u;
        const float a = in ? p.a : 0.f;
        const tile_u32x4 my_wt = as_u32x4(hh * hw * a, hh * p.lw * a, p.lh * hw * a, p.lh * p.lw * a);
        const unsigned fb_bits = my_fb ? 1u : 0u;
        const bool any_fb = __builtin_amdgcn_ballot_w64(my_fb) != 0ull;     // wave-uniform
        const tile_u32x4 my_lds = my_fb ? tile_u32x4{0u, 0u, 0u, 0u} : my_off;
        constexpr int U = 2;                  // points of a pair in flight
#pragma unroll
        for (int tb = 0; tb < 4; tb += U) {
            tile_u32x4 off[U], wt[U];
            RowT v[U][4];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                off[u] = quad_bcast(my_lds, tb + u);
                wt[u] = quad_bcast(my_wt, tb + u);
                row_load_lds<ST, VEC>(rows_lds + off[u].x + lane_off, v[u][0]);
                row_load_lds<ST, VEC>(rows_lds + off[u].y + lane_off, v[u][1]);
                row_load_lds<ST, VEC>(rows_lds + off[u].z + lane_off, v[u][2]);
                row_load_lds<ST, VEC>(rows_lds + off[u].w + lane_off, v[u][3]);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                row_axpy<ST, VEC>(acc, __uint_as_float(wt[u].x), v[u][0]);
                row_axpy<ST, VEC>(acc, __uint_as_float(wt[u].y), v[u][1]);
                row_axpy<ST, VEC>(acc, __uint_as_float(wt[u].z), v[u][2]);
                row_axpy<ST, VEC>(acc, __uint_as_float(wt[u].w), v[u][3]);
            }
        }
        // points outside their window: rows from global memory (their LDS offsets above pointed
        // at the zero row, so they have contributed nothing yet)
        if (any_fb) {                                                 // wave-uniform, rare
            for (int u = 0; u < 4; ++u) {
                const int src = (threadIdx.x & 60) + u;               // lane (pair, u)
                const bool fbu = __shfl((int)fb_bits, src, kWave) != 0;
                if (!fbu) continue;
                RowT g[4];
                row_load<ST, VEC, PSB>(rs, (unsigned)__shfl((int)my_off.x, src, kWave) + lane_off, g[0]);
                row_load<ST, VEC, PSB>(rs, (unsigned)__shfl((int)my_off.y, src, kWave) + lane_off, g[1]);
                row_load<ST, VEC, PSB>(rs, (unsigned)__shfl((int)my_off.z, src, kWave) + lane_off, g[2]);
                row_load<ST, VEC, PSB>(rs, (unsigned)__shfl((int)my_off.w, src, kWave) + lane_off, g[3]);
                row_axpy<ST, VEC>(acc, __shfl(__uint_as_float(my_wt.x), src, kWave), g[0]);
                row_axpy<ST, VEC>(acc, __shfl(__uint_as_float(my_wt.y), src, kWave), g[1]);
                row_axpy<ST, VEC>(acc, __shfl(__uint_as_float(my_wt.z), src, kWave), g[2]);
                row_axpy<ST, VEC>(acc, __shfl(__uint_as_float(my_wt.w), src, kWave), g[3]);
            }
        }
    }
    if (active) row_store<ST, VEC, PSB>(out + (size_t)qh * C + slot * LCH, acc);
}

}  // namespace boxattn
