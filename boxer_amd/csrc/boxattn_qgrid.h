// Backward for grad_value in the encoder case (one query per pixel of the packed multi-level
// map, Lq = S): destination-owned like the binned backward (boxattn_binned.h), but WITHOUT the
// global binning passes and their record stream.
//
// The binned backward first sorts all sample points by destination block (count + 2 scans +
// fill: 47 us and ~150 MB of records written and read back at BoxeR-R50 shapes)
// because, in general, any query may sample anywhere.  In the encoder neighbouring queries
// sample neighbouring pixels, so the points that can reach a destination block are found by
// looking at the queries around it:
//
//   1. qg_prep_kernel      per (image, head, value level) and 4x4 tile of queries: the bounding
//                          box of the tile's sample footprints at that level, 8 bytes (exact,
//                          from the data -- no assumption about box sizes), and the tile's 64
//                          points LOCATED, level-major: {upstream row, corner, fractions} +
//                          weight, 20 bytes a point, so that step 2 reads a tile's points of one
//                          level as whole lines (in the op's own layout they are 32-byte pieces
//                          1 KiB apart: 32 line requests per tile and level, and the step sat on
//                          the L2 -> L1 request rate).
//   2. qg_accumulate_kernel  one wavefront per work item = a group of 1, 2 or 4 destination
//                          blocks (8x4 pixels each) of one level x a share of the query tiles.
//                          It scans the tile boxes of its (image, head, level), visits the tiles
//                          whose box meets its pixels, locates their 64 points (lane = query x
//                          point), keeps those whose footprint meets the group, and queues them
//                          in LDS; every 64 queued records are one round of the dense MFMA
//                          scatter-product of boxattn_binned_mfma.h (grad_value^T += G^T A^T,
//                          A split into two bf16 terms), the records' upstream rows staged once
//                          per round for all blocks of the group.  Heavy (coarse-level) blocks
//                          are split over several items that take every cpb-th tile and write
//                          fp32 partial tiles.
//   3. qg_combine_kernel   sums the partial tiles of split blocks.
//
// Every grad_value row has exactly one owner and is stored once (no zero-fill, no atomics).
// Any input is handled exactly: a far-away sample only widens its tile's box, i.e. more items
// look at that tile.  Inputs without locality (uniformly random locations) make every item look
// at every tile; the host keeps the binned path for shapes that are not query grids.
#pragma once
#include "boxattn_binned_mfma.h"

namespace boxattn {

constexpr int kQgMaxLevels = 8;
constexpr int kQgQueue = 128;            // records in a work item's LDS queue (64 left over + 64 new)
constexpr int kQgList = 1024;            // query tiles listed per scan pass (ids < 65536)

struct QgLevel {
    int H, W, start;          // the level as value map and as query grid
    int ntx4, tile0;          // 4x4 query tiles: per row, first tile id of the level
    int gx, gy;               // blocks (8x4 pixels) per item group along x / y: 1 or 2
    int ngx, ngy;             // groups along x / y
    int cpb;                  // items per group (each takes every cpb-th query tile)
    int item0;                // first item of the level inside an (image, head) slice
    int part0;                // first partial-tile slot of the level inside a slice (cpb > 1)
};
struct QgPlan {
    int L, n_tiles4, n_items, n_parts;   // per image: query tiles; per slice: items, partial tiles
    int ablate;                          // timing experiments only (wrong results): 1 no rounds, 2 no candidates, 4 no scan
    QgLevel lv[kQgMaxLevels];
};

__device__ __forceinline__ void qg_load_plan(QgLevel *dst, const QgPlan &plan, int tid)
{
#pragma unroll
    for (int l = 0; l < kQgMaxLevels; ++l)        // constant indices: a kernel argument cannot be
        if (tid == l) dst[l] = plan.lv[l];        // indexed dynamically without a scratch copy
}

// 4x4 query tile t -> level, first query coordinates
__device__ __forceinline__ void qg_tile(const QgLevel *lv, int L, int t, int &lq, int &qx0, int &qy0)
{
    lq = 0;
    for (int l = 1; l < L; ++l)
        if (t >= lv[l].tile0) lq = l;
    const int tt = t - lv[lq].tile0;
    qx0 = (tt % lv[lq].ntx4) * 4;
    qy0 = (tt / lv[lq].ntx4) * 4;
}

typedef unsigned short qg_u16x2 __attribute__((ext_vector_type(2)));

// min / max over the wave of two packed unsigned 16-bit values
template <bool MAX> __device__ __forceinline__ unsigned qg_wave_pk(unsigned v)
{
    auto op = [](unsigned a, unsigned b) -> unsigned {
        const qg_u16x2 x = __builtin_bit_cast(qg_u16x2, a), y = __builtin_bit_cast(qg_u16x2, b);
        return __builtin_bit_cast(unsigned, MAX ? __builtin_elementwise_max(x, y)
                                                : __builtin_elementwise_min(x, y));
    };
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true));
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true));
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true));
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x128, 0xF, 0xF, true));
    const unsigned a = (unsigned)__builtin_amdgcn_readlane((int)v, 0), b = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned c = (unsigned)__builtin_amdgcn_readlane((int)v, 32), d = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return op(op(a, b), op(c, d));
}

// ---------------------------------------------------------------------------------------
// 1. tile boxes + located points.  One wave per (image, head, query tile); lane = (query j of
//    the tile, point s).  With i = ((b H + h) L + l) n_tiles4 + t:
//    bbox[i]          = {x_min | y_min << 16, x_max | y_max << 16} of the valid corner pixels of
//                       the tile's points at level l (x_min > x_max: no point in the map);
//    cand[64 i + lane]  = {upstream row (b Lq + q) H + h, or -1 for a point outside the map / a
//                       query outside the level, (x0 + 1) | (y0 + 1) << 16, lw, lh};
//    cand_w[64 i + lane] = attention weight.
// ---------------------------------------------------------------------------------------
template <int LV>      // levels (P == 4: point s of level k is element 4 k + s of the pair)
__global__ __launch_bounds__(256) void qg_prep_kernel(const float *__restrict__ loc,
                                                      const float *__restrict__ attn, QgPlan plan,
                                                      int B, int H, int Lq, uint2 *__restrict__ bbox,
                                                      int4 *__restrict__ cand, float *__restrict__ cand_w)
{
    __shared__ QgLevel lv[kQgMaxLevels];
    qg_load_plan(lv, plan, threadIdx.x);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const unsigned wid = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int h = (int)(wid % (unsigned)H);
    const unsigned r = wid / (unsigned)H;
    const int t = (int)(r % (unsigned)plan.n_tiles4), b = (int)(r / (unsigned)plan.n_tiles4);
    if (b >= B) return;                                            // wave-uniform
    int lq, qx0, qy0;
    qg_tile(lv, plan.L, t, lq, qx0, qy0);
    const int j = lane >> 2, s = lane & 3;
    const int qx = qx0 + (j & 3), qy = qy0 + (j >> 2);
    const bool active = qx < lv[lq].W && qy < lv[lq].H;
    const int q = lv[lq].start + min(qy, lv[lq].H - 1) * lv[lq].W + min(qx, lv[lq].W - 1);
    const int row = (b * Lq + q) * H + h;
    const size_t pt0 = (size_t)row * (size_t)(4 * LV);
    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);
    float2 xy[LV];
    float aw[LV];
#pragma unroll
    for (int k = 0; k < LV; ++k) {
        xy[k] = loc2[pt0 + 4 * k + s];
        aw[k] = attn[pt0 + 4 * k + s];
    }
    const size_t i0 = ((size_t)(b * H + h) * LV) * plan.n_tiles4 + t;
#pragma unroll
    for (int k = 0; k < LV; ++k) {
        const Sample<float> sm = locate<float>(xy[k].x, xy[k].y, lv[k].H, lv[k].W);
        const bool in = active && sm.inside;
        const unsigned xa = (unsigned)max(sm.x0, 0), xb = (unsigned)min(sm.x0 + 1, lv[k].W - 1);
        const unsigned ya = (unsigned)max(sm.y0, 0), yb = (unsigned)min(sm.y0 + 1, lv[k].H - 1);
        const unsigned lo = qg_wave_pk<false>(in ? (xa | (ya << 16)) : 0xFFFFFFFFu);
        const unsigned hi = qg_wave_pk<true>(in ? (xb | (yb << 16)) : 0u);
        const size_t i = i0 + (size_t)k * plan.n_tiles4;
        if (lane == 0) bbox[i] = make_uint2(lo, hi);
        cand[i * 64 + lane] = make_int4(in ? row : -1, (sm.x0 + 1) | ((sm.y0 + 1) << 16),
                                        __float_as_int(sm.lw), __float_as_int(sm.lh));
        cand_w[i * 64 + lane] = aw[k];
    }
}

// ---------------------------------------------------------------------------------------
// 2. accumulate.  PERSISTENT single-wave workgroups (a fresh wave per item cost ~2 us of launch
//    latency each: 17 of the kernel's first 129 us): grid = 8 x waves per XCD; workgroup g runs on
//    XCD g % 8 and takes the items of that XCD's slices (s % 8 == XCD: a slice only reads its own
//    head's rows) round-robin.
// ---------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(64) void qg_accumulate_kernel(
    const bf16_t *__restrict__ grad_out, const uint2 *__restrict__ bbox,
    const int4 *__restrict__ cand, const float *__restrict__ cand_w, QgPlan plan, int n_slices,
    int S, int H, bf16_t *__restrict__ grad_value, float *__restrict__ partials)
{
    static_assert(C == 32, "one 32-channel block");
    constexpr int BW = 8, BH = 4, PB = 32, R = 64, NBMAX = 4;
    constexpr int LPR = C * 2 / 16;                // lanes that fetch one upstream row, 16 B each
    constexpr int RPP = 64 / LPR, NPASS = R / RPP;
    constexpr int GS = 68, AS = 72;                // see boxattn_binned_mfma.h
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) unsigned short gt[C * GS];
    __shared__ __attribute__((aligned(16))) unsigned short at[PB * AS];
    __shared__ int4 queue[kQgQueue];               // {upstream row, corner, lw, lh}
    __shared__ float queue_w[kQgQueue];            // attention weight
    __shared__ unsigned short tlist[kQgList];      // query tiles whose box meets the group
    __shared__ QgLevel lvs[kQgMaxLevels];

    const int lane = threadIdx.x;
    qg_load_plan(lvs, plan, lane);
    for (int i = lane; i < PB * AS / 8; i += 64) reinterpret_cast<u32x4 *>(at)[i] = u32x4{0u, 0u, 0u, 0u};
    wave_lds_sync();
    const int L = plan.L, NT = plan.n_tiles4;
    const int xcd = blockIdx.x % 8, wave = blockIdx.x / 8, n_waves = gridDim.x / 8;
    const int my_slices = (n_slices - xcd + 7) / 8;                    // slices xcd, xcd + 8, ...
    const int col = lane & 31, kb = lane >> 5;

    for (int gi = wave; gi < my_slices * plan.n_items; gi += n_waves) {
    // items interleaved over the XCD's slices: neighbouring waves work on the same blocks of
    // different heads / images
    const int s = xcd + 8 * (gi % my_slices), it = gi / my_slices;
    const int b = s / H, h = s % H;
    int l = 0;
    for (int k = 1; k < L; ++k)
        if (it >= lvs[k].item0) l = k;
    const QgLevel lv = lvs[l];
    const int idx = it - lv.item0;
    const int chunk = idx % lv.cpb, grp = idx / lv.cpb;
    const int ox = (grp % lv.ngx) * lv.gx * BW, oy = (grp / lv.ngx) * lv.gy * BH;
    const int gw = min(lv.gx * BW, lv.W - ox), gh = min(lv.gy * BH, lv.H - oy);
    const int nb = lv.gx * lv.gy;

    mfma_f32x16 acc[NBMAX];
#pragma unroll
    for (int m = 0; m < NBMAX; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

    // ---- one round: 64 (or fewer) records popped from the queue
    int4 rec = make_int4(-1, 0, 0, 0);             // the pending round's record of this lane
    float rec_w = 0.f;
    u32x4 grow[NPASS];                             // ... and the upstream rows, in flight
    bool pending = false;
    auto fetch_rows = [&]() {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
            const int j = ps * RPP + lane / LPR, piece = lane % LPR;
            const int rj = max(__shfl(rec.x, j, 64), 0);     // idle lanes (-1): any row, zeroed when staged
            grow[ps] = *reinterpret_cast<const u32x4 *>(grad_out + (size_t)rj * C + piece * 8);
        }
    };
    auto complete = [&]() {
        if (plan.ablate & 1) return;
        // G^T[c][j]: the lanes of records j and j + 1 swap halves and write whole dwords
        {
            const int odd = (lane / LPR) & 1;
            const unsigned sel = odd ? 0x03020706u : 0x05040100u;
#pragma unroll
            for (int ps = 0; ps < NPASS; ++ps) {
                const int j = ps * RPP + lane / LPR, piece = lane % LPR;
                unsigned int *dst =
                    reinterpret_cast<unsigned int *>(&gt[(piece * 8 + odd) * GS + (j & ~1)]);
                // idle lanes stage a ZERO row: in a dense product 0 * Inf = NaN would leak
                const bool live = __shfl(rec.x, j, 64) >= 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const unsigned own = live ? grow[ps][i] : 0u;
                    const unsigned oth = pair_exchange<LPR>(own);
                    dst[i * GS] = __builtin_amdgcn_perm(oth, own, sel);
                }
            }
        }
        // lane = record: bilinear x attention weights as hi + lo bf16 (the fractions were
        // computed by locate() in qg_prep_kernel: same bits as every other kernel)
        const int x0 = (rec.y & 0xffff) - 1, y0 = (rec.y >> 16) - 1;
        const float lw = __int_as_float(rec.z), lh = __int_as_float(rec.w);
        const float hw = 1.f - lw, hh = 1.f - lh, a = rec_w;
        const float wk[4] = {hh * hw * a, hh * lw * a, lh * hw * a, lh * lw * a};
        const unsigned hi01 = pack_bf16x2(wk[0], wk[1]), hi23 = pack_bf16x2(wk[2], wk[3]);
        const unsigned lo01 = pack_bf16x2(wk[0] - __uint_as_float(hi01 << 16),
                                          wk[1] - __uint_as_float(hi01 & 0xffff0000u));
        const unsigned lo23 = pack_bf16x2(wk[2] - __uint_as_float(hi23 << 16),
                                          wk[3] - __uint_as_float(hi23 & 0xffff0000u));
        const unsigned short whi[4] = {(unsigned short)(hi01 & 0xffffu), (unsigned short)(hi01 >> 16),
                                       (unsigned short)(hi23 & 0xffffu), (unsigned short)(hi23 >> 16)};
        const unsigned short wlo[4] = {(unsigned short)(lo01 & 0xffffu), (unsigned short)(lo01 >> 16),
                                       (unsigned short)(lo23 & 0xffffu), (unsigned short)(lo23 >> 16)};
        int slot[4], tile_of[4];                   // A^T element / block of the group per corner
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int yy = y0 + (k >> 1), xx = x0 + (k & 1);
            const int ry = yy - oy, rx = xx - ox;
            // (a corner inside the group is inside the map)
            const bool use = rec.x >= 0 && (unsigned)ry < (unsigned)gh && (unsigned)rx < (unsigned)gw;
            tile_of[k] = use ? (ry >> 2) * lv.gx + (rx >> 3) : -1;
            slot[k] = ((ry & 3) * BW + (rx & 7)) * AS + lane;
        }
        wave_lds_sync();                           // G^T staged
        mfma_bf16x8 g[R / 16];
#pragma unroll
        for (int t = 0; t < R / 16; ++t) {
            const u32x2 *gp = reinterpret_cast<const u32x2 *>(&gt[col * GS + 16 * t + 8 * kb]);
            const u32x2 g0 = gp[0], g1 = gp[1];
            g[t] = __builtin_bit_cast(mfma_bf16x8, u32x4{g0.x, g0.y, g1.x, g1.y});
        }
#pragma unroll
        for (int m = 0; m < NBMAX; ++m) {
            if (m >= nb) break;                    // wave-uniform
#pragma unroll
            for (int pass = 0; pass < 2; ++pass) { // hi term, then lo term in place
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (tile_of[k] == m) at[slot[k]] = pass ? wlo[k] : whi[k];
                wave_lds_sync();
#pragma unroll
                for (int t = 0; t < R / 16; ++t) {
                    const mfma_bf16x8 p = __builtin_bit_cast(
                        mfma_bf16x8, *reinterpret_cast<const u32x4 *>(&at[col * AS + 16 * t + 8 * kb]));
                    acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(g[t], p, acc[m], 0, 0, 0);
                }
                wave_lds_sync();
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (tile_of[k] == m) at[slot[k]] = 0;
        }
        wave_lds_sync();
    };

    // ---- candidates: the query tiles of this item (every cpb-th) whose box meets the group.
    // Phase 1 scans the tile boxes (coalesced, four loads in flight) and lists the tiles that
    // meet the group in LDS; phase 2 walks the list, four tiles' located points in flight.
    int qhead = 0, qcount = 0;
    const size_t i0 = ((size_t)s * L + l) * NT;
    const uint2 *boxes = bbox + i0;
    const int4 *cnd = cand + i0 * 64 + lane;
    const float *cnw = cand_w + i0 * 64 + lane;
    const int n_mine = (NT - chunk + lv.cpb - 1) / lv.cpb;           // tiles chunk, chunk + cpb, ...
    const int gx1 = ox + gw - 1, gy1 = oy + gh - 1;
    struct Cand { int4 r; float a; };
    // (unconditional loads from clamped indices: a load under a branch makes the compiler wait
    // for everything in flight at every use)
    auto load_cand = [&](int n, int n_list) -> Cand {
        Cand c;
        const size_t t = tlist[min(n, n_list - 1)];
        c.r = cnd[t * 64];
        c.a = cnw[t * 64];
        if (n >= n_list) c.r.x = -1;
        return c;
    };
    auto filter = [&](const Cand &c) {
        const int x0 = (c.r.y & 0xffff) - 1, y0 = (c.r.y >> 16) - 1;
        // the footprint's valid pixels: [max(x0, 0), min(x0 + 1, W - 1)] x [max(y0, 0), ...];
        // the group lies inside the map, so clamping is not needed for the overlap test
        const bool hit = c.r.x >= 0 && x0 <= gx1 && x0 + 1 >= ox && y0 <= gy1 && y0 + 1 >= oy;
        const unsigned long long hm = __builtin_amdgcn_ballot_w64(hit);
        if (hit) {
            const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(hm >> 32),
                                                            __builtin_amdgcn_mbcnt_lo((unsigned)hm, 0u));
            const int at_ = (qhead + qcount + rank) & (kQgQueue - 1);
            queue[at_] = c.r;
            queue_w[at_] = c.a;
        }
        qcount += __builtin_popcountll(hm);
        if (qcount >= R) {                                            // wave-uniform
            if (pending) complete();
            wave_lds_sync();                                          // queue writes visible
            rec = queue[(qhead + lane) & (kQgQueue - 1)];
            rec_w = queue_w[(qhead + lane) & (kQgQueue - 1)];
            qhead = (qhead + R) & (kQgQueue - 1);
            qcount -= R;
            fetch_rows();
            pending = true;
        }
    };
    for (int scan = (plan.ablate & 4) ? n_mine : 0; scan < n_mine;) {
        int n_list = 0;
        while (scan < n_mine && n_list + 256 <= kQgList) {            // wave-uniform
            uint2 bx[4];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                bx[u] = boxes[chunk + min(scan + 64 * u + lane, n_mine - 1) * lv.cpb];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int x_lo = (int)(bx[u].x & 0xffffu), y_lo = (int)(bx[u].x >> 16);
                const int x_hi = (int)(bx[u].y & 0xffffu), y_hi = (int)(bx[u].y >> 16);
                const bool hit_t = scan + 64 * u + lane < n_mine && x_lo <= gx1 && x_hi >= ox &&
                                   y_lo <= gy1 && y_hi >= oy;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(hit_t);
                if (hit_t) {
                    const int rank = (int)__builtin_amdgcn_mbcnt_hi(
                        (unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    tlist[n_list + rank] = (unsigned short)(chunk + (scan + 64 * u + lane) * lv.cpb);
                }
                n_list += __builtin_popcountll(m);
            }
            scan += 256;
        }
        wave_lds_sync();
        if (plan.ablate & 2) n_list = 0;
        // four tiles' points in flight at a time
        for (int n = 0; n < n_list; n += 4) {
            const Cand c0 = load_cand(n, n_list), c1 = load_cand(n + 1, n_list);
            const Cand c2 = load_cand(n + 2, n_list), c3 = load_cand(n + 3, n_list);
            filter(c0);
            filter(c1);
            filter(c2);
            filter(c3);
        }
        wave_lds_sync();                                              // list consumed before it is refilled
    }
    if (pending) complete();
    if (qcount > 0) {
        wave_lds_sync();
        const int at_ = (qhead + lane) & (kQgQueue - 1);
        rec = lane < qcount ? queue[at_] : make_int4(-1, 0, 0, 0);
        rec_w = queue_w[at_];
        fetch_rows();
        complete();
    }

    // ---- store.  Lane = pixel `col` of a block; registers = channels 8 g + 4 kb + 0..3.
    const int py = col / BW, px = col % BW;
#pragma unroll
    for (int m = 0; m < NBMAX; ++m) {
        if (m >= nb) break;
        const int ty = oy + (m / lv.gx) * BH + py, tx = ox + (m % lv.gx) * BW + px;
        const bool live = ty < oy + gh && tx < ox + gw;
        if (lv.cpb == 1) {
            bf16_t *dst = grad_value + (((size_t)b * S + lv.start + (size_t)ty * lv.W + tx) * H + h) * C;
            unsigned pk[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) pk[i] = pack_bf16x2(acc[m][2 * i], acc[m][2 * i + 1]);
            const unsigned r0 = __shfl_xor(kb ? pk[0] : pk[2], 32, 64);
            const unsigned r1 = __shfl_xor(kb ? pk[1] : pk[3], 32, 64);
            const unsigned r2 = __shfl_xor(kb ? pk[4] : pk[6], 32, 64);
            const unsigned r3 = __shfl_xor(kb ? pk[5] : pk[7], 32, 64);
            const u32x4 lo_piece = kb ? u32x4{r0, r1, pk[2], pk[3]} : u32x4{pk[0], pk[1], r0, r1};
            const u32x4 hi_piece = kb ? u32x4{r2, r3, pk[6], pk[7]} : u32x4{pk[4], pk[5], r2, r3};
            if (live) {
                *reinterpret_cast<u32x4 *>(dst + 8 * kb) = lo_piece;
                *reinterpret_cast<u32x4 *>(dst + 16 + 8 * kb) = hi_piece;
            }
        } else {                                   // split block (gx = gy = 1): fp32 partial tile
            float *dst = partials +
                         (((size_t)s * plan.n_parts + lv.part0 + (size_t)grp * lv.cpb + chunk) * PB + col) * C;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                *reinterpret_cast<float4 *>(dst + 8 * g4 + 4 * kb) =
                    make_float4(acc[m][4 * g4], acc[m][4 * g4 + 1], acc[m][4 * g4 + 2], acc[m][4 * g4 + 3]);
        }
    }
    }   // item loop
}

// ---------------------------------------------------------------------------------------
// 3. split blocks: sum of the cpb partial tiles -> rows in the storage type.
//    grid = (split groups per slice, slices), 128 threads: (pixel, 8 channels).
// ---------------------------------------------------------------------------------------
template <int C>
__global__ __launch_bounds__(128) void qg_combine_kernel(const float *__restrict__ partials,
                                                         QgPlan plan, int S, int H,
                                                         bf16_t *__restrict__ grad_value)
{
    constexpr int BW = 8, BH = 4, PB = 32;
    __shared__ QgLevel lvs[kQgMaxLevels];
    qg_load_plan(lvs, plan, threadIdx.x);
    __syncthreads();
    // which split group: the levels with cpb > 1 in order
    int rest = blockIdx.x, l = -1;
    for (int k = 0; k < plan.L; ++k) {
        if (lvs[k].cpb <= 1) continue;
        const int n = lvs[k].ngx * lvs[k].ngy;
        if (l < 0 && rest < n) l = k;
        if (l < 0) rest -= n;
    }
    if (l < 0) return;
    const QgLevel lv = lvs[l];
    const int s = blockIdx.y, b = s / H, h = s % H, grp = rest;
    const int ox = (grp % lv.ngx) * BW, oy = (grp / lv.ngx) * BH;
    const int px = (threadIdx.x >> 2) % BW, py = (threadIdx.x >> 2) / BW, c0 = (threadIdx.x & 3) * 8;
    if (ox + px >= lv.W || oy + py >= lv.H) return;
    const float *src = partials +
                       (((size_t)s * plan.n_parts + lv.part0 + (size_t)grp * lv.cpb) * PB + (threadIdx.x >> 2)) * C + c0;
    float4 a0 = make_float4(0.f, 0.f, 0.f, 0.f), a1 = a0;
    for (int c = 0; c < lv.cpb; ++c) {
        const float4 u = *reinterpret_cast<const float4 *>(src + (size_t)c * PB * C);
        const float4 v = *reinterpret_cast<const float4 *>(src + (size_t)c * PB * C + 4);
        a0.x += u.x; a0.y += u.y; a0.z += u.z; a0.w += u.w;
        a1.x += v.x; a1.y += v.y; a1.z += v.z; a1.w += v.w;
    }
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
    bf16_t *dst = grad_value + (((size_t)b * S + lv.start + (size_t)(oy + py) * lv.W + ox + px) * H + h) * C + c0;
    *reinterpret_cast<u32x4 *>(dst) = u32x4{pack_bf16x2(a0.x, a0.y), pack_bf16x2(a0.z, a0.w),
                                            pack_bf16x2(a1.x, a1.y), pack_bf16x2(a1.z, a1.w)};
}

}  // namespace boxattn
