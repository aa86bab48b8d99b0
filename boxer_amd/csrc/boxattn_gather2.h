// Second-generation gather kernels (forward, and the point-gradient half of the backward).
//
// Both kernels are bound by VALU issue (PMC: 65-80 % of the VALU slots busy, HBM < 1.5 TB/s),
// so the design minimises wave instructions per sample point:
//
//   step A  lane (pair j, slot t) does the geometry of ONE sample point of pair j -- G points
//           per pair at a time -- and leaves {4 byte offsets, 4 weights, ...} in a wave-private
//           LDS tile (no workgroup barrier: producer and consumers are the same wavefront);
//   step B  lane (pair j, channel chunk m) walks the G points of its pair: one LDS broadcast
//           read per point, four BUFFER loads of VEC channels (32-bit offset, the hardware
//           range check returns 0 for the out-of-map marker, so no address select and no
//           select-to-zero), then packed math: v_pk_fma_f32 on channel pairs (two FMAs per
//           lane and instruction).
//
// VEC = 8 channels per lane (16-byte loads for bf16, 2 x 16 bytes for fp32) whenever C allows:
// G = C / VEC lanes per (query, head) pair, so the per-point overhead (tile read, offset adds)
// is spread over twice the channels of the VEC = 4 layout.
//
// The backward flavour accumulates S_k = sum_c g_c * v_k,c for the four corners (bf16: one
// v_dot2c_f32_bf16 per channel PAIR straight on the packed words, no unpack; fp32: v_pk_fma),
// reduces the four sums over the G lanes with DPP, hands them to lane (j, t) and lets that
// lane finish grad_loc / grad_weight from its own step-A registers:
//   grad_w   = sum_k w_k S_k
//   grad_x   = W_l * a * (hh (S2 - S1) + lh (S4 - S3))
//   grad_y   = H_l * a * (hw (S3 - S1) + lw (S4 - S2))
// (reference formulas box_attn_kernel.cuh:145-183 with the channel sum pulled out).
#pragma once
#include "boxattn_device.h"
#include "boxattn_fast.h"
#include "boxattn_binned.h"
#include "boxattn_binpass.h"

namespace boxattn {

typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2_t __attribute__((ext_vector_type(2)));

constexpr unsigned kOobOffset = 0x80000000u;     // >= any valid byte offset (tensor < 2 GiB)
// The gather kernels are launched with 256 threads (4 waves).  A constant, not blockDim.x: the
// block size lives in the dispatch packet, and reading it is a VECTOR global load plus
// s_waitcnt vmcnt(0) at the top of every wave -- a full memory round trip before the first
// useful load could be issued (s_memtime stamps: 6 200 of a wave's 21 000 cycles).
constexpr unsigned kGatherWaves = 4;

// Tuning switches (tools/build_variants.sh builds the alternatives side by side).
#ifndef BOXATTN_TUNE_SCHED
#define BOXATTN_TUNE_SCHED 1      // all row loads of a step are issued before the first FMA
#endif
#ifndef BOXATTN_TUNE_PREFETCH
#define BOXATTN_TUNE_PREFETCH 1   // locations / weights of tile t+1 are requested during tile t
#endif
#ifndef BOXATTN_TUNE_PRELOAD
#define BOXATTN_TUNE_PRELOAD 1    // L P == 4 G: the locations / weights of all tiles up front
#endif
constexpr bool kGatherPreload = BOXATTN_TUNE_PRELOAD != 0;
constexpr int kPreTiles = 4;
constexpr bool kGatherSched = BOXATTN_TUNE_SCHED != 0;
constexpr bool kGatherPrefetch = BOXATTN_TUNE_PREFETCH != 0;
__device__ __forceinline__ void loads_issued()
{
    // keep the compiler from sinking loads below the math to save registers: the point of the
    // step is to have them all in flight at once
    if constexpr (kGatherSched) __builtin_amdgcn_sched_barrier(0);
}

// The VEC channels a lane owns, as raw 32-bit words (fp32: VEC words, bf16: VEC / 2 words).
// They are fetched in 16-byte pieces (8 bytes for bf16 x 4).  A lane with two pieces (fp32,
// VEC = 8) owns channels [4 slot, 4 slot + 4) and [4 G + 4 slot, ...): piece i of the G lanes
// of a pair is then one contiguous 16 G-byte run, i.e. every load instruction touches whole
// 64-byte segments (two pieces side by side per lane would leave 16-byte holes in each
// instruction's footprint and cost twice the L1 tag lookups).
template <typename ST, int VEC> struct Row {
    static constexpr int NW = VEC * (int)sizeof(ST) / 4;
    static constexpr int NP = NW > 4 ? NW / 4 : 1;               // pieces per lane
    static constexpr int kLaneBytes = VEC * (int)sizeof(ST) / NP;  // bytes of one piece
    unsigned w[NW];
};
// byte distance between the pieces of a lane
template <typename ST, int VEC, int G> struct RowGeom {
    static constexpr int kPieceStride = Row<ST, VEC>::kLaneBytes * G;
};

template <typename ST, int VEC, int PSB>
__device__ __forceinline__ void row_load(__amdgpu_buffer_rsrc_t r, unsigned off, Row<ST, VEC> &v)
{
    constexpr int NW = Row<ST, VEC>::NW;
    if constexpr (NW == 2) {
        const u32x2_t t = __builtin_amdgcn_raw_buffer_load_b64(r, off, 0, 0);
        v.w[0] = t.x; v.w[1] = t.y;
    } else {
#pragma unroll
        for (int i = 0; i < NW / 4; ++i) {
            const u32x4_t t =
                __builtin_amdgcn_raw_buffer_load_b128(r, off + (unsigned)(PSB * i), 0, 0);
            v.w[4 * i] = t.x; v.w[4 * i + 1] = t.y; v.w[4 * i + 2] = t.z; v.w[4 * i + 3] = t.w;
        }
    }
}

template <typename ST, int VEC, int PSB>
__device__ __forceinline__ void row_load(const ST *p, Row<ST, VEC> &v)
{
    constexpr int NW = Row<ST, VEC>::NW;
    if constexpr (NW == 2) {
        const u32x2_t t = *reinterpret_cast<const u32x2_t *>(p);
        v.w[0] = t.x; v.w[1] = t.y;
    } else {
#pragma unroll
        for (int i = 0; i < NW / 4; ++i) {
            const u32x4_t t = *reinterpret_cast<const u32x4_t *>(
                reinterpret_cast<const char *>(p) + PSB * i);
            v.w[4 * i] = t.x; v.w[4 * i + 1] = t.y; v.w[4 * i + 2] = t.z; v.w[4 * i + 3] = t.w;
        }
    }
}

// channels (2i, 2i+1) of a row as two floats
template <typename ST, int VEC>
__device__ __forceinline__ f32x2 row_pair(const Row<ST, VEC> &v, int i)
{
    f32x2 r;
    if constexpr (sizeof(ST) == 4) {
        r.x = __uint_as_float(v.w[2 * i]); r.y = __uint_as_float(v.w[2 * i + 1]);
    } else {
        r.x = __uint_as_float(v.w[i] << 16); r.y = __uint_as_float(v.w[i] & 0xffff0000u);
    }
    return r;
}

// acc (VEC / 2 channel pairs) += w * row
#ifndef BOXATTN_TUNE_FWD_ABLATE
#define BOXATTN_TUNE_FWD_ABLATE 0   // timing experiments only (wrong results): 1 = the rows are
#endif                              // consumed by one XOR per word instead of unpack + FMA,
                                    // 2 = every row load hits the first 4 KiB of value (L1 hits),
                                    // 3 = no row loads
template <typename ST, int VEC>
__device__ __forceinline__ void row_axpy(f32x2 (&acc)[VEC / 2], float w, const Row<ST, VEC> &v)
{
    if constexpr (BOXATTN_TUNE_FWD_ABLATE == 1 && sizeof(ST) == 2) {
#pragma unroll
        for (int i = 0; i < VEC / 2; ++i)
            acc[i].x = __uint_as_float(__float_as_uint(acc[i].x) ^ v.w[i]);
        acc[0].y = __uint_as_float(__float_as_uint(acc[0].y) ^ __float_as_uint(w));
        return;
    }
    const f32x2 w2 = {w, w};
#pragma unroll
    for (int i = 0; i < VEC / 2; ++i)
        acc[i] = __builtin_elementwise_fma(w2, row_pair<ST, VEC>(v, i), acc[i]);
}

// sum_c g_c * v_c over the lane's VEC channels
template <typename ST, int VEC>
__device__ __forceinline__ float row_dot(const Row<ST, VEC> &g, const Row<ST, VEC> &v)
{
    if constexpr (sizeof(ST) == 4) {
        f32x2 a = {0.f, 0.f};
#pragma unroll
        for (int i = 0; i < VEC / 2; ++i)
            a = __builtin_elementwise_fma(row_pair<ST, VEC>(g, i), row_pair<ST, VEC>(v, i), a);
        return a.x + a.y;
    } else {
        float a = 0.f;                               // bf16 x bf16 products are exact in fp32
#pragma unroll
        for (int i = 0; i < VEC / 2; ++i)
            a = __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, g.w[i]),
                                                __builtin_bit_cast(bf16x2_t, v.w[i]), a, false);
        return a;
    }
}

template <typename ST, int VEC, int PSB>
__device__ __forceinline__ void row_store(ST *p, const f32x2 (&acc)[VEC / 2])
{
    if constexpr (sizeof(ST) == 4) {
#pragma unroll
        for (int i = 0; i < VEC / 4; ++i)
            *reinterpret_cast<float4 *>(reinterpret_cast<char *>(p) + PSB * i) =
                make_float4(acc[2 * i].x, acc[2 * i].y, acc[2 * i + 1].x, acc[2 * i + 1].y);
    } else if constexpr (VEC == 4) {
        u32x2_t t;
        t.x = pack_bf16x2(acc[0].x, acc[0].y); t.y = pack_bf16x2(acc[1].x, acc[1].y);
        *reinterpret_cast<u32x2_t *>(p) = t;
    } else {
        u32x4_t t;
        t.x = pack_bf16x2(acc[0].x, acc[0].y); t.y = pack_bf16x2(acc[1].x, acc[1].y);
        t.z = pack_bf16x2(acc[2].x, acc[2].y); t.w = pack_bf16x2(acc[3].x, acc[3].y);
        *reinterpret_cast<u32x4_t *>(p) = t;
    }
}

// G == 4: a pair is one DPP quad, so lane (j, t)'s step-A registers reach the other lanes of the
// pair with quad_perm broadcasts (VALU moves the compiler folds into the consuming add where it
// can) -- no LDS tile, no LDS round trip between the two steps.
#ifndef BOXATTN_TUNE_QUAD_DPP
#define BOXATTN_TUNE_QUAD_DPP 1
#endif
template <int G> struct UseQuadDpp { static constexpr bool value = G == 4 && BOXATTN_TUNE_QUAD_DPP; };
template <int CTRL> __device__ __forceinline__ unsigned dpp_mov(unsigned v)
{
    return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, true);
}
__device__ __forceinline__ unsigned quad_bcast(unsigned v, int t)   // t is a constant after unrolling
{
    switch (t & 3) {
    case 0: return dpp_mov<0x00>(v);
    case 1: return dpp_mov<0x55>(v);
    case 2: return dpp_mov<0xAA>(v);
    default: return dpp_mov<0xFF>(v);
    }
}
__device__ __forceinline__ u32x4_t quad_bcast(u32x4_t v, int t)
{
    u32x4_t r = {quad_bcast(v.x, t), quad_bcast(v.y, t), quad_bcast(v.z, t), quad_bcast(v.w, t)};
    return r;
}

// The wave-private geometry tile: per point NH 16-byte pieces {4 offsets}{4 weights}[{as, al}],
// one 16-byte pad per G-lane group so that the groups' broadcast reads start on different banks.
template <int G, int NH> struct GeoTile {
    static constexpr int kGroupStride = G * NH + 1;
    static constexpr int kSize = (kWave / G) * kGroupStride;
    static __device__ __forceinline__ int base(int lane) { return (lane / G) * kGroupStride; }
};

// Step A for one point: byte offsets of the four corners (row of head h, channel 0) or the
// out-of-map marker.
template <typename ST>
__device__ __forceinline__ u32x4_t corner_offsets(const Sample<float> &s, unsigned row0, int H, int h,
                                                  int C, bool have)
{
    unsigned off[4];
#pragma unroll
    for (int k = 0; k < 4; ++k)
        off[k] = (have && s.ok[k])
                     ? ((row0 + (unsigned)s.pix[k]) * H + h) * (unsigned)(C * sizeof(ST))
                     : kOobOffset;
    u32x4_t r = {off[0], off[1], off[2], off[3]};
    return r;
}

__device__ __forceinline__ u32x4_t as_u32x4(float a, float b, float c, float d)
{
    u32x4_t r = {__float_as_uint(a), __float_as_uint(b), __float_as_uint(c), __float_as_uint(d)};
    return r;
}

// Host-computed constants of the index arithmetic (no 64-bit divisions in the kernels: they were
// a fifth of the instructions of a wave).
struct GatherIdx {
    unsigned n_qh;        // B * Lq * H  (< 2^31)
    unsigned magic_h;     // floor(2^32 / H)   (divmod_magic)
    unsigned magic_lq;    // floor(2^32 / Lq)
    float rcp_p;          // 1.0f / P
    unsigned head_xcd;    // 1: workgroup b works on head b % 8 (H == 8), see pair_of_lane()
    // launch geometry (with_grid() on the host, right before the launch).  Explicit arguments,
    // not gridDim: the grid size lives in the hidden arguments (one more dependent scalar load
    // at the top of every wave) and `tiles per workgroup` was an emulated integer division there
    unsigned grid_x, grid_y;
    unsigned tps;         // point tiles per workgroup row: ceil(tiles / grid_y)
};

// Which (query, head) pair a lane group works on.
//   head_xcd = 0: consecutive pairs (a wave covers all heads of PAIRS / H queries), workgroups
//                 re-mapped so that every XCD gets one contiguous chunk of the query range;
//   head_xcd = 1 (H == 8): workgroup b -- which the hardware places on XCD b % 8 -- takes head
//                 b % 8 and PAIRS consecutive queries per wave: an XCD's L2 then only ever holds
//                 the rows of ONE head (1/8 of `value`), however far apart the queries sample.
template <int PAIRS>
__device__ __forceinline__ unsigned pair_of_lane(const GatherIdx &ix, unsigned blk, int H, int j, int wv,
                                                 bool &active)
{
    // blk: the kernel's own workgroup index (riders taken out, boxattn_ride.h; index mod 8 = XCD as before)
    unsigned qh;
    if (ix.head_xcd) {
        const unsigned tile = (blk / 8) * kGatherWaves + wv;
        const unsigned bq = tile * PAIRS + j;
        active = bq < ix.n_qh / (unsigned)H;
        qh = bq * (unsigned)H + blk % 8;
    } else {
        const unsigned bid = xcd_chunked_block(blk, ix.grid_x);
        qh = (bid * kGatherWaves + wv) * PAIRS + j;
        active = qh < ix.n_qh;
    }
    return active ? qh : ix.n_qh - 1;
}

// ---------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------
template <typename ST, int G, bool INST, int U, int VEC>
__global__ __launch_bounds__(256) void fwd2_kernel(
    const ST *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc,
    const float *__restrict__ w_sp, const float *__restrict__ w_lv, int S, int H, int L, int Lq,
    int P, ST *__restrict__ out, ST *__restrict__ mask, GatherIdx ix, unsigned value_bytes,
    BinRide ride)
{
    constexpr int C = VEC * G, PAIRS = kWave / G, NH = INST ? 3 : 2;
    // the training forward: the backward's count pass and the scans chained behind it ride in this launch
    // (boxattn_ride.h, bin_count_ride)
    __shared__ int ride_lds[kRideLdsInts];
    const RideRole role = ride_role(blockIdx.x, ride.grid);
    if (role.rider) {
        if (blockIdx.y == 0) bin_count_ride<256>(ride, role.id, ride_lds);
        return;
    }
    typedef GeoTile<G, NH> Tile;
    typedef Row<ST, VEC> RowT;
    constexpr int PSB = RowGeom<ST, VEC, G>::kPieceStride;    // bytes between a lane's pieces
    constexpr int LCH = RowT::kLaneBytes / (int)sizeof(ST);   // channels of one piece
    __shared__ LevelTable lv;
    __shared__ u32x4_t geo_all[4][Tile::kSize];
    const LevelRegs lv_regs = levels_request(shapes, lsi, L);   // published below, after the first loads

    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    u32x4_t *geo = geo_all[wv] + Tile::base(lane);   // this group's part of the tile
    bool active;
    const unsigned qh = pair_of_lane<PAIRS>(ix, role.id, H, lane / G, wv, active);
    const int slot = lane % G;                      // step A: point slot; step B: channel chunk
    unsigned bq, hu, b, qu;
    divmod_magic(qh, (unsigned)H, ix.magic_h, bq, hu);
    divmod_magic(bq, (unsigned)Lq, ix.magic_lq, b, qu);
    const int h = (int)hu;
    const size_t HC = (size_t)H * C;
    const int LP = L * P;
    const size_t pt0 = (size_t)qh * LP;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<ST *>(value), 0, value_bytes, 0x00020000);
    const unsigned lane_off = (unsigned)(slot * RowT::kLaneBytes);
    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);
    // (n + 0.5) * rcp_p truncates to n / P for every n < L * P <= 2^16: the product is at least
    // 0.5 / P away from an integer, far more than the rounding error of the two operations
    const float rcp_p = ix.rcp_p;

    f32x2 acc[VEC / 2];
#pragma unroll
    for (int i = 0; i < VEC / 2; ++i) acc[i] = f32x2{0.f, 0.f};

    if constexpr (!INST) {
        // one tile: G points of every pair of the wave.  xy / a: location and weight of THIS
        // lane's point of the tile (point t0 + slot; ignored past the end)
        auto tile = [&](int t0, float2 xy, float a_in) {
            u32x4_t my_off = {0u, 0u, 0u, 0u}, my_wt = {0u, 0u, 0u, 0u};
            {   // ---- step A
                const int lp = t0 + slot;
                const bool have = lp < LP;
                const int lq = have ? lp : LP - 1;
                const int l = (int)(((float)lq + 0.5f) * rcp_p);   // lq / P (exact, see rcp_p)
                const float a = have ? a_in : 0.f;
                const Sample<float> s = locate<float>(xy.x, xy.y, lv.h[l], lv.w[l]);
                const u32x4_t off =
                    corner_offsets<ST>(s, b * (unsigned)S + (unsigned)lv.start[l], H, h, C, have);
                const u32x4_t wt = as_u32x4(s.hh * s.hw * a, s.hh * s.lw * a, s.lh * s.hw * a,
                                            s.lh * s.lw * a);
                if constexpr (UseQuadDpp<G>::value) {
                    my_off = off;
                    my_wt = wt;
                } else {
                    wave_lds_sync();               // previous tile fully consumed
                    geo[slot * NH] = off;
                    geo[slot * NH + 1] = wt;
                    wave_lds_sync();
                }
            }
            // ---- step B (U points' loads in flight at a time)
#pragma unroll
            for (int tb = 0; tb < G; tb += U) {
                u32x4_t off[U], wt[U];
                RowT v[U][4];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if constexpr (UseQuadDpp<G>::value) {
                        off[u] = quad_bcast(my_off, tb + u);
                        wt[u] = quad_bcast(my_wt, tb + u);
                    } else {
                        off[u] = geo[(tb + u) * NH];
                        wt[u] = geo[(tb + u) * NH + 1];
                    }
                    if constexpr (BOXATTN_TUNE_FWD_ABLATE == 2) {
                        off[u].x &= 0xfc0u; off[u].y &= 0xfc0u; off[u].z &= 0xfc0u; off[u].w &= 0xfc0u;
                    }
                    if constexpr (BOXATTN_TUNE_FWD_ABLATE == 3) {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
#pragma unroll
                            for (int i = 0; i < RowT::NW; ++i) v[u][k].w[i] = off[u][k] + i;
                        continue;
                    }
                    row_load<ST, VEC, PSB>(rs, off[u].x + lane_off, v[u][0]);
                    row_load<ST, VEC, PSB>(rs, off[u].y + lane_off, v[u][1]);
                    row_load<ST, VEC, PSB>(rs, off[u].z + lane_off, v[u][2]);
                    row_load<ST, VEC, PSB>(rs, off[u].w + lane_off, v[u][3]);
                }
                loads_issued();
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    row_axpy<ST, VEC>(acc, __uint_as_float(wt[u].x), v[u][0]);
                    row_axpy<ST, VEC>(acc, __uint_as_float(wt[u].y), v[u][1]);
                    row_axpy<ST, VEC>(acc, __uint_as_float(wt[u].z), v[u][2]);
                    row_axpy<ST, VEC>(acc, __uint_as_float(wt[u].w), v[u][3]);
                }
            }
        };
        // The locations / weights of a wave are one contiguous run (64 / G pairs x L P points),
        // but a lane only ever has its next tile's 12 bytes in flight: with one tile of
        // look-ahead the kernel ran at the HBM LATENCY of that stream (4 dependent round trips
        // per wave, 1.6 TB/s on 54 MB).  BoxeR's shape (L P = 4 G: 4 levels x 2 x 2 points, 4
        // lanes per pair) requests all four tiles up front -- one round trip per wave.
        if (kGatherPreload && G == 4 && LP == kPreTiles * G) {
            float2 xy4[kPreTiles];
            float a4[kPreTiles];
#pragma unroll
            for (int k = 0; k < kPreTiles; ++k) {
                xy4[k] = loc2[pt0 + k * G + slot];
                a4[k] = w_sp[pt0 + k * G + slot];
            }
            levels_commit(lv, lv_regs, L);
            // a rolled loop over the tiles (the registers rotate): unrolled, the scheduler pulls
            // the row loads of all four tiles to the front (299 VGPRs)
#pragma unroll 1
            for (int t0 = 0; t0 < kPreTiles * G; t0 += G) {
                tile(t0, xy4[0], a4[0]);
#pragma unroll
                for (int k = 0; k + 1 < kPreTiles; ++k) {
                    xy4[k] = xy4[k + 1];
                    a4[k] = a4[k + 1];
                }
            }
        } else {
            float2 xy_n = loc2[pt0 + min(slot, LP - 1)];            // tile 0
            float a_n = w_sp[pt0 + min(slot, LP - 1)];
            levels_commit(lv, lv_regs, L);
            for (int t0 = 0; t0 < LP; t0 += G) {
                float2 xy = xy_n;
                float a = a_n;
                const int lp = t0 + slot;
                if constexpr (kGatherPrefetch) {
                    const int nq = min(lp + G, LP - 1);           // same point again past the end
                    xy_n = loc2[pt0 + nq];
                    a_n = w_sp[pt0 + nq];
                } else {
                    const int lq = min(lp, LP - 1);
                    xy = loc2[pt0 + lq];
                    a = w_sp[pt0 + lq];
                }
                tile(t0, xy, a);
            }
        }
    } else {
        levels_commit(lv, lv_regs, L);
        ST *mk = mask + (size_t)bq * P * HC + (size_t)h * C + slot * LCH;
        // few queries x many points (decoder, 14x14 grids): gridDim.y workgroups share the
        // point tiles of a (query, head) pair; out is then accumulated with atomics
        const int tps = (int)ix.tps;                   // of (P + G - 1) / G tiles
        const int p_begin = (int)blockIdx.y * tps * G, p_end = min(P, p_begin + tps * G);
        for (int p0 = p_begin; p0 < p_end; p0 += G) {
            f32x2 macc[G][VEC / 2];
#pragma unroll
            for (int t = 0; t < G; ++t)
#pragma unroll
                for (int i = 0; i < VEC / 2; ++i) macc[t][i] = f32x2{0.f, 0.f};
            for (int l = 0; l < L; ++l) {
                {   // ---- step A: points (l, p0 .. p0+G-1)
                    const int p = p0 + slot;
                    const bool have = p < P;
                    const size_t i = pt0 + (size_t)l * P + (have ? p : P - 1);
                    const float2 xy = loc2[i];
                    const float as = have ? w_sp[i] : 0.f;
                    const float al = have ? w_lv[i] : 0.f;
                    const Sample<float> s = locate<float>(xy.x, xy.y, lv.h[l], lv.w[l]);
                    const u32x4_t off = corner_offsets<ST>(
                        s, b * (unsigned)S + (unsigned)lv.start[l], H, h, C, have);
                    const u32x4_t wt =
                        as_u32x4(s.hh * s.hw, s.hh * s.lw, s.lh * s.hw, s.lh * s.lw);
                    wave_lds_sync();
                    geo[slot * NH] = off;
                    geo[slot * NH + 1] = wt;
                    geo[slot * NH + 2] = as_u32x4(as, al, 0.f, 0.f);
                    wave_lds_sync();
                }
#pragma unroll
                for (int tb = 0; tb < G; tb += U) {
                    u32x4_t off[U], wt[U], aa[U];
                    RowT v[U][4];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        off[u] = geo[(tb + u) * NH];
                        wt[u] = geo[(tb + u) * NH + 1];
                        aa[u] = geo[(tb + u) * NH + 2];
                        row_load<ST, VEC, PSB>(rs, off[u].x + lane_off, v[u][0]);
                        row_load<ST, VEC, PSB>(rs, off[u].y + lane_off, v[u][1]);
                        row_load<ST, VEC, PSB>(rs, off[u].z + lane_off, v[u][2]);
                        row_load<ST, VEC, PSB>(rs, off[u].w + lane_off, v[u][3]);
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        f32x2 val[VEC / 2];
#pragma unroll
                        for (int i = 0; i < VEC / 2; ++i) val[i] = f32x2{0.f, 0.f};
                        row_axpy<ST, VEC>(val, __uint_as_float(wt[u].x), v[u][0]);
                        row_axpy<ST, VEC>(val, __uint_as_float(wt[u].y), v[u][1]);
                        row_axpy<ST, VEC>(val, __uint_as_float(wt[u].z), v[u][2]);
                        row_axpy<ST, VEC>(val, __uint_as_float(wt[u].w), v[u][3]);
                        const float as = __uint_as_float(aa[u].x), al = __uint_as_float(aa[u].y);
                        const f32x2 as2 = {as, as}, al2 = {al, al};
#pragma unroll
                        for (int i = 0; i < VEC / 2; ++i) {
                            acc[i] = __builtin_elementwise_fma(val[i], as2, acc[i]);
                            macc[tb + u][i] = __builtin_elementwise_fma(val[i], al2, macc[tb + u][i]);
                        }
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < G; ++t)
                if (active && p0 + t < P) row_store<ST, VEC, PSB>(mk + (size_t)(p0 + t) * HC, macc[t]);
        }
    }
    if (active) {
        if constexpr (INST && std::is_same<ST, float>::value) {
            if (ix.grid_y > 1) {                   // split points: out was zero-filled by the host
#pragma unroll
                for (int i = 0; i < VEC / 2; ++i) {
                    // channel of pair i: piece (2 i / 4), position (2 i % 4) inside it
                    float *o = out + (size_t)qh * C + slot * LCH + (2 * i / 4) * (PSB / 4) + (2 * i) % 4;
                    atomic_add(o, acc[i].x);
                    atomic_add(o + 1, acc[i].y);
                }
                return;
            }
        }
        row_store<ST, VEC, PSB>(out + (size_t)qh * C + slot * LCH, acc);
    }
}

// ---------------------------------------------------------------------------------------
// forward, instance attention with FEW (query, head) pairs and many points (the mask decoder:
// 300 queries x 14x14 points): one WAVE per pair.  The 64 / G lane groups take different points
// of the pair; inside a group lane t does the geometry of level l0 + t of the group's point and
// the G lanes then walk these levels together (same step A / step B split as fwd2_kernel).  A
// group owns mask_out[b,q,p,m,:] of its points; `out` is summed over the groups with
// cross-lane adds at the end -- no atomics, no zero-fill, any storage type.
// ---------------------------------------------------------------------------------------
template <int G> __device__ __forceinline__ unsigned group_bcast(unsigned v, int t, int lane)
{
    if constexpr (G == 4) return quad_bcast(v, t);
    else return (unsigned)__shfl((int)v, (lane & ~(G - 1)) + t, kWave);
}

template <typename ST, int G, int VEC>
__global__ __launch_bounds__(256) void fwd_inst_wide_kernel(
    const ST *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc,
    const float *__restrict__ w_sp, const float *__restrict__ w_lv, int S, int H, int L, int Lq,
    int P, ST *__restrict__ out, ST *__restrict__ mask, GatherIdx ix, unsigned value_bytes,
    BinRide ride, unsigned ws)
{
    constexpr int C = VEC * G, NG = kWave / G;
    typedef Row<ST, VEC> RowT;
    constexpr int PSB = RowGeom<ST, VEC, G>::kPieceStride;
    constexpr int LCH = RowT::kLaneBytes / (int)sizeof(ST);
    // the training forward: the backward's count pass + scans ride in this launch (fwd2_kernel)
    __shared__ int ride_lds[kRideLdsInts];
    __shared__ float red[kGatherWaves][VEC * G];
    const RideRole role = ride_role(blockIdx.x, ride.grid);
    if (role.rider) {
        bin_count_ride<256>(ride, role.id, ride_lds);
        return;
    }
    const unsigned blk = role.id;
    __shared__ LevelTable lv;
    const LevelRegs lv_regs = levels_request(shapes, lsi, L);           // published below, behind the first loads

    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    // `ws` (1, 2, 4) waves share a pair: wave `sub` of them takes the point steps sub, sub + ws, ... (with many points a
    // pair -- 14 x 14: 25 steps of a float32 wave -- one wave per pair is 1 200 long-lived workgroups on 768 slots)
    const unsigned ppw = kGatherWaves / ws, sub = wv % ws;
    unsigned qh = blk * ppw + wv / ws;
    if (ix.head_xcd)                                                    // head = XCD (pair_of_lane)
        qh = ((blk / 8) * ppw + wv / ws) * (unsigned)H + blk % 8;
    const bool live = qh < ix.n_qh;                                     // wave-uniform
    qh = min(qh, ix.n_qh - 1u);
    const int slot = lane % G, grp = lane / G;
    unsigned bq, hu, b, qu;
    divmod_magic(qh, (unsigned)H, ix.magic_h, bq, hu);
    divmod_magic(bq, (unsigned)Lq, ix.magic_lq, b, qu);
    const int h = (int)hu;
    const size_t HC = (size_t)H * C;
    const size_t pt0 = (size_t)qh * L * P;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<ST *>(value), 0, value_bytes, 0x00020000);
    const unsigned lane_off = (unsigned)(slot * RowT::kLaneBytes);
    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);
    ST *mk = mask + (size_t)bq * P * HC + (size_t)h * C + slot * LCH;

    f32x2 acc[VEC / 2];
#pragma unroll
    for (int i = 0; i < VEC / 2; ++i) acc[i] = f32x2{0.f, 0.f};

    // The wave is a chain of round trips -- level table, locations, rows, per step of NG points -- and at 300
    // queries the whole launch is ONE wave's life (1 200 workgroups: a single round): the location / weight loads of
    // a step's first level group are requested a step ahead, the first step's in front of the level table's barrier.
    struct Ahead { float2 xy; float as, al; };
    auto request = [&](int p0) -> Ahead {
        const int p = min(p0 + grp, P - 1), lc = min(slot, L - 1);
        const size_t i = pt0 + (size_t)lc * P + p;
        return Ahead{loc2[i], w_sp[i], w_lv[i]};
    };
    const int pstep = (int)ws * NG;
    Ahead nxt = request((int)sub * NG);
    levels_commit(lv, lv_regs, L);
    if (ws == 1 && !live) return;

    for (int p0 = (int)sub * NG; live && p0 < P; p0 += pstep) {       // wave-uniform trip count
        const int p = p0 + grp;
        const bool have_p = p < P;
        const Ahead cur = nxt;
        if (p0 + pstep < P) nxt = request(p0 + pstep);
        f32x2 macc[VEC / 2];
#pragma unroll
        for (int i = 0; i < VEC / 2; ++i) macc[i] = f32x2{0.f, 0.f};
        for (int l0 = 0; l0 < L; l0 += G) {
            // ---- step A: lane t of the group -> level l0 + t of point p
            const int l = l0 + slot;
            const bool have = have_p && l < L;
            const int lc = min(l, L - 1);
            float2 xy = cur.xy;
            float as = cur.as, al = cur.al;
            if (l0 > 0) {                                              // (more levels than lanes in a group)
                const size_t i = pt0 + (size_t)lc * P + (have_p ? p : P - 1);
                xy = loc2[i];
                as = w_sp[i];
                al = w_lv[i];
            }
            as = have ? as : 0.f;
            al = have ? al : 0.f;
            const Sample<float> s = locate<float>(xy.x, xy.y, lv.h[lc], lv.w[lc]);
            const u32x4_t my_off =
                corner_offsets<ST>(s, b * (unsigned)S + (unsigned)lv.start[lc], H, h, C, have);
            const u32x4_t my_wt = as_u32x4(s.hh * s.hw, s.hh * s.lw, s.lh * s.hw, s.lh * s.lw);
            // ---- step B: the G lanes walk the group's levels together, four at a time (a
            //      group has G level slots; with fewer levels left the rest is skipped)
            constexpr int UL = 4;
#pragma unroll
            for (int t0 = 0; t0 < G; t0 += UL) {
                if (l0 + t0 >= L) break;                         // wave-uniform
                u32x4_t off[UL], wt[UL];
                unsigned aas[UL], aal[UL];
                RowT v[UL][4];
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    const int t = t0 + u;
                    off[u] = u32x4_t{group_bcast<G>(my_off.x, t, lane), group_bcast<G>(my_off.y, t, lane),
                                     group_bcast<G>(my_off.z, t, lane), group_bcast<G>(my_off.w, t, lane)};
                    wt[u] = u32x4_t{group_bcast<G>(my_wt.x, t, lane), group_bcast<G>(my_wt.y, t, lane),
                                    group_bcast<G>(my_wt.z, t, lane), group_bcast<G>(my_wt.w, t, lane)};
                    aas[u] = group_bcast<G>(__float_as_uint(as), t, lane);
                    aal[u] = group_bcast<G>(__float_as_uint(al), t, lane);
                    row_load<ST, VEC, PSB>(rs, off[u].x + lane_off, v[u][0]);
                    row_load<ST, VEC, PSB>(rs, off[u].y + lane_off, v[u][1]);
                    row_load<ST, VEC, PSB>(rs, off[u].z + lane_off, v[u][2]);
                    row_load<ST, VEC, PSB>(rs, off[u].w + lane_off, v[u][3]);
                }
                loads_issued();
#pragma unroll
                for (int u = 0; u < UL; ++u) {
                    f32x2 val[VEC / 2];
#pragma unroll
                    for (int k = 0; k < VEC / 2; ++k) val[k] = f32x2{0.f, 0.f};
                    row_axpy<ST, VEC>(val, __uint_as_float(wt[u].x), v[u][0]);
                    row_axpy<ST, VEC>(val, __uint_as_float(wt[u].y), v[u][1]);
                    row_axpy<ST, VEC>(val, __uint_as_float(wt[u].z), v[u][2]);
                    row_axpy<ST, VEC>(val, __uint_as_float(wt[u].w), v[u][3]);
                    const float a_s = __uint_as_float(aas[u]), a_l = __uint_as_float(aal[u]);
                    const f32x2 as2 = {a_s, a_s}, al2 = {a_l, a_l};
#pragma unroll
                    for (int k = 0; k < VEC / 2; ++k) {
                        acc[k] = __builtin_elementwise_fma(val[k], as2, acc[k]);
                        macc[k] = __builtin_elementwise_fma(val[k], al2, macc[k]);
                    }
                }
            }
        }
        if (have_p) row_store<ST, VEC, PSB>(mk + (size_t)p * HC, macc);
    }
    // out = sum over the lane groups (lanes with the same channel chunk: stride G)
#pragma unroll
    for (int o = G; o < kWave; o <<= 1)
#pragma unroll
        for (int k = 0; k < VEC / 2; ++k) {
            acc[k].x += __shfl_xor(acc[k].x, o, kWave);
            acc[k].y += __shfl_xor(acc[k].y, o, kWave);
        }
    if (ws > 1) {                                                       // ... and over the pair's waves
        if (grp == 0) {
#pragma unroll
            for (int k = 0; k < VEC / 2; ++k)
                *reinterpret_cast<f32x2 *>(&red[wv][slot * VEC + 2 * k]) = acc[k];
        }
        __syncthreads();
        if (sub != 0 || !live) return;
        if (grp == 0) {
            for (unsigned s = 1; s < ws; ++s)
#pragma unroll
                for (int k = 0; k < VEC / 2; ++k) {
                    const f32x2 o = *reinterpret_cast<const f32x2 *>(&red[wv + s][slot * VEC + 2 * k]);
                    acc[k].x += o.x;
                    acc[k].y += o.y;
                }
        }
    }
    if (grp == 0) row_store<ST, VEC, PSB>(out + (size_t)qh * C + slot * LCH, acc);
}

// ---------------------------------------------------------------------------------------
// backward, point gradients only (grad_loc, grad_weight[s]); grad_value is boxattn_binned.h
// ---------------------------------------------------------------------------------------
// WP ("wave per pair"; few pairs x many points, e.g. the 14x14 mask-decoder grid of instance
// attention): all lane groups of a wave work on ONE (query, head) pair, group g on the point
// tiles g, g + 16, ...: the wave's results are 64 consecutive points (whole lines instead of
// 16- and 32-byte pieces 3 KiB apart) and its location / weight reads are contiguous.
template <typename ST, int G, bool INST, int U, int VEC, bool WP = false>
__global__ __launch_bounds__(256) void pointgrad2_kernel(
    const ST *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const float *__restrict__ loc,
    const float *__restrict__ w_sp, const float *__restrict__ w_lv,
    const ST *__restrict__ grad_out, const ST *__restrict__ grad_mask, int S, int H, int L,
    int Lq, int P, float *__restrict__ grad_loc, float *__restrict__ grad_sp,
    float *__restrict__ grad_lv, GatherIdx ix, unsigned value_bytes, BinRide ride)
{
    constexpr int C = VEC * G, PAIRS = kWave / G;
    // the backward's fill pass rides in this launch (boxattn_ride.h, bin_fill_ride)
    __shared__ int ride_lds[kRideLdsInts];
    const RideRole role = ride_role(blockIdx.x, ride.grid);
    if (role.rider) {
        if (blockIdx.y == 0) bin_fill_ride<256>(ride, role.id, ride_lds);
        return;
    }
    const unsigned blk = role.id;
    typedef GeoTile<G, 1> Tile;                      // offsets only: the weights stay with lane t
    typedef Row<ST, VEC> RowT;
    constexpr int PSB = RowGeom<ST, VEC, G>::kPieceStride;    // bytes between a lane's pieces
    constexpr int LCH = RowT::kLaneBytes / (int)sizeof(ST);   // channels of one piece
    __shared__ LevelTable lv;
    __shared__ u32x4_t geo_all[4][Tile::kSize];
    // Box attention with L * P = 8 or 16 points per pair (BoxeR: 2x2 grids on 2 or 4 levels): the
    // per-point results are collected per pair in LDS and written once, 16 / 32 contiguous bytes
    // per lane.  (Written tile by tile -- 4-byte and 8-byte pieces per lane, 16 points of a pair
    // in four separate instructions -- the kernel's HBM write traffic was twice its output.)
#ifndef BOXATTN_TUNE_PG_TRACE
#define BOXATTN_TUNE_PG_TRACE 0     // experiments only (overwrites grad_weight): s_memtime stamps of wave 0 of
#endif                              // every workgroup: start, loop entry, each tile, end (tools/gpu_pg_trace.py)
    unsigned long long ts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int ts_n = 0;
    if constexpr (BOXATTN_TUNE_PG_TRACE) {
        ts[ts_n++] = __builtin_amdgcn_s_memtime();
        if (BOXATTN_TUNE_PG_TRACE == 2) {                     // finer stamps of the prologue instead of the tiles
            asm volatile("" ::"s"(S), "s"(Lq));               // explicit kernel arguments have arrived
            ts[2] = __builtin_amdgcn_s_memtime();
        }
    }
    constexpr int kBufLP = 16;
    constexpr bool kCanBuffer = !INST && (G == 4 || G == 8);
    __shared__ float res_all[kCanBuffer ? 4 * PAIRS * kBufLP * 3 : 1];
    const LevelRegs lv_regs = levels_request(shapes, lsi, L);   // published below, after the first loads

    const int lane = threadIdx.x & (kWave - 1), wv = threadIdx.x / kWave;
    u32x4_t *geo = geo_all[wv] + Tile::base(lane);
    bool active;
    unsigned qh;
    if constexpr (WP) {
        qh = blk * kGatherWaves + wv;                             // one pair per wave
        if (ix.head_xcd)                                                    // head = XCD (pair_of_lane)
            qh = ((blk / 8) * kGatherWaves + wv) * (unsigned)H + blk % 8;
        active = qh < ix.n_qh;
        qh = active ? qh : ix.n_qh - 1;
    } else {
        qh = pair_of_lane<PAIRS>(ix, blk, H, lane / G, wv, active);
    }
    const int slot = lane % G;
    if constexpr (BOXATTN_TUNE_PG_TRACE == 2) {
        asm volatile("" ::"v"(qh));                           // the pair index exists (grid size read)
        ts[3] = __builtin_amdgcn_s_memtime();
    }
    unsigned bq, hu, b, qu;
    divmod_magic(qh, (unsigned)H, ix.magic_h, bq, hu);
    divmod_magic(bq, (unsigned)Lq, ix.magic_lq, b, qu);
    const int h = (int)hu;
    const size_t HC = (size_t)H * C;
    const int LP = L * P;
    const size_t pt0 = (size_t)qh * LP;
    const __amdgpu_buffer_rsrc_t rs =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<ST *>(value), 0, value_bytes, 0x00020000);
    const unsigned lane_off = (unsigned)(slot * RowT::kLaneBytes);
    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);
    const float rcp_p = ix.rcp_p;                    // (n + 0.5) * rcp_p truncates to n / P

    RowT g;
    row_load<ST, VEC, PSB>(grad_out + (size_t)qh * C + slot * LCH, g);

    // gridDim.y workgroups share the point tiles of a pair (few queries x many points).
    // t_step: distance between two tiles of a lane group (WP: the groups interleave)
    const int tps = (int)ix.tps;                       // of (LP + G - 1) / G tiles
    const int t_first = (int)blockIdx.y * tps * G, t_end = min(LP, t_first + tps * G);
    const int t_begin = t_first + (WP ? (lane / G) * G : 0);
    constexpr int t_step = WP ? PAIRS * G : G;
    float2 xy_n = loc2[pt0 + min(t_begin + slot, LP - 1)];      // first tile
    float as_n = w_sp[pt0 + min(t_begin + slot, LP - 1)];
    float al_n = INST ? w_lv[pt0 + min(t_begin + slot, LP - 1)] : 0.f;
    // BoxeR's shape (L P = 4 G): all four tiles' locations / weights are requested up front,
    // one HBM round trip per wave instead of one per tile (see fwd2_kernel)
    constexpr bool kCanPre = kGatherPreload && !INST && !WP && G == 4;
    const bool pre = kCanPre && ix.grid_y == 1 && LP == kPreTiles * G;
    float2 xyq[kCanPre ? kPreTiles - 1 : 1];
    float asq[kCanPre ? kPreTiles - 1 : 1];
#pragma unroll
    for (int k = 0; k < (kCanPre ? kPreTiles - 1 : 1); ++k) {
        xyq[k] = make_float2(0.f, 0.f);
        asq[k] = 0.f;
        if constexpr (kCanPre) {
            // unconditional, from a clamped index (other shapes just do not use them): under
            // `if (pre)` the compiler merges the loaded registers with the zeros at the join and
            // WAITS for the load there -- the round trip this is meant to remove
            const int pq = min((k + 1) * G + slot, LP - 1);
            xyq[k] = loc2[pt0 + pq];
            asq[k] = w_sp[pt0 + pq];
        }
    }
    if constexpr (BOXATTN_TUNE_PG_TRACE) {
        __builtin_amdgcn_sched_barrier(0);
        ts[6] = __builtin_amdgcn_s_memtime();                 // first loads issued, before the barrier
        __builtin_amdgcn_sched_barrier(0);
    }
    levels_commit(lv, lv_regs, L);
    const bool buffered = kCanBuffer && !WP && ix.grid_y == 1 && (LP == 16 || LP == 8) &&
                          ((reinterpret_cast<uintptr_t>(grad_sp) | reinterpret_cast<uintptr_t>(grad_loc)) & 15) == 0;
    float *res = res_all + (kCanBuffer ? (wv * PAIRS + lane / G) * (kBufLP * 3) : 0);
    if constexpr (BOXATTN_TUNE_PG_TRACE) ts[ts_n++] = __builtin_amdgcn_s_memtime();
    // (wave-uniform trip count: with WP the groups whose tile lies past the end idle)
    for (int tu = t_first, t0 = t_begin; tu < t_end; tu += t_step, t0 += t_step) {
        // ---- step A (the lane keeps its point's geometry in registers for the finish)
        const int lp = t0 + slot;
        const bool have = lp < t_end;
        const int lq = have ? lp : LP - 1;
        const int l = (int)(((float)lq + 0.5f) * rcp_p);           // lq / P (exact, see rcp_p)
        float2 xy;
        float as, al;
        if constexpr (kGatherPrefetch) {
            xy = xy_n; as = as_n; al = al_n;
            if (kCanPre && pre) {
                xy_n = xyq[0];
                as_n = asq[0];
#pragma unroll
                for (int k = 0; k + 1 < (kCanPre ? kPreTiles - 1 : 1); ++k) {
                    xyq[k] = xyq[k + 1];
                    asq[k] = asq[k + 1];
                }
            } else {
                const int nq = min(lp + t_step, LP - 1);
                xy_n = loc2[pt0 + nq];
                as_n = w_sp[pt0 + nq];
                al_n = INST ? w_lv[pt0 + nq] : 0.f;
            }
        } else {
            xy = loc2[pt0 + lq];
            as = w_sp[pt0 + lq];
            al = INST ? w_lv[pt0 + lq] : 0.f;
        }
        const Sample<float> s = locate<float>(xy.x, xy.y, lv.h[l], lv.w[l]);
        const u32x4_t myoff =
            corner_offsets<ST>(s, b * (unsigned)S + (unsigned)lv.start[l], H, h, C, have);
        if constexpr (!UseQuadDpp<G>::value) {
            wave_lds_sync();
            geo[slot] = myoff;
            wave_lds_sync();
        }

        // ---- step B: corner sums of the G points of this lane's pair
        float s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f;         // S_k of "my" point (slot)
        float m1 = 0.f, m2 = 0.f, m3 = 0.f, m4 = 0.f;         // instance: sums with grad_mask
#pragma unroll
        for (int tb = 0; tb < G; tb += U) {
            RowT v[U][4], gm[INST ? U : 1];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                u32x4_t off;
                if constexpr (UseQuadDpp<G>::value) off = quad_bcast(myoff, tb + u);
                else off = geo[tb + u];
#ifndef BOXATTN_TUNE_PG_ABLATE
#define BOXATTN_TUNE_PG_ABLATE 0    // timing experiments only (wrong results): 2 = every row load hits
#endif                              // the first 4 KiB of value (L1 hits), 3 = no row loads
                if constexpr (BOXATTN_TUNE_PG_ABLATE == 2) {
                    off.x &= 0xfc0u; off.y &= 0xfc0u; off.z &= 0xfc0u; off.w &= 0xfc0u;
                }
                if constexpr (BOXATTN_TUNE_PG_ABLATE == 3 && !INST) {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int i = 0; i < RowT::NW; ++i) v[u][k].w[i] = off[k] + i;
                    continue;
                }
                row_load<ST, VEC, PSB>(rs, off.x + lane_off, v[u][0]);
                row_load<ST, VEC, PSB>(rs, off.y + lane_off, v[u][1]);
                row_load<ST, VEC, PSB>(rs, off.z + lane_off, v[u][2]);
                row_load<ST, VEC, PSB>(rs, off.w + lane_off, v[u][3]);
                if constexpr (INST) {
                    // grad_mask row of point (t0 + t): its p is uniform inside the group
                    const int lpt = min(t0 + tb + u, LP - 1);
                    const int pt = lpt - (int)(((float)lpt + 0.5f) * rcp_p) * P;   // lpt % P
                    row_load<ST, VEC, PSB>(grad_mask + ((size_t)bq * P + pt) * HC + (size_t)h * C + slot * LCH,
                                      gm[u]);
                }
            }
            loads_issued();
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float a1 = group_sum<G>(row_dot<ST, VEC>(g, v[u][0]));
                const float a2 = group_sum<G>(row_dot<ST, VEC>(g, v[u][1]));
                const float a3 = group_sum<G>(row_dot<ST, VEC>(g, v[u][2]));
                const float a4 = group_sum<G>(row_dot<ST, VEC>(g, v[u][3]));
                const bool mine = slot == tb + u;
                s1 = mine ? a1 : s1; s2 = mine ? a2 : s2; s3 = mine ? a3 : s3; s4 = mine ? a4 : s4;
                if constexpr (INST) {
                    const float b1 = group_sum<G>(row_dot<ST, VEC>(gm[u], v[u][0]));
                    const float b2 = group_sum<G>(row_dot<ST, VEC>(gm[u], v[u][1]));
                    const float b3 = group_sum<G>(row_dot<ST, VEC>(gm[u], v[u][2]));
                    const float b4 = group_sum<G>(row_dot<ST, VEC>(gm[u], v[u][3]));
                    m1 = mine ? b1 : m1; m2 = mine ? b2 : m2; m3 = mine ? b3 : m3; m4 = mine ? b4 : m4;
                }
            }
        }
        if constexpr (BOXATTN_TUNE_PG_TRACE) {
            asm volatile("" ::"v"(s1), "v"(s4));              // after the tile's sums exist
            if (BOXATTN_TUNE_PG_TRACE == 1 && ts_n < 7) ts[ts_n++] = __builtin_amdgcn_s_memtime();
        }
        // ---- finish: lane (pair, slot) owns point t0 + slot
        if (active && have) {
            const size_t i = pt0 + lp;
            const float Wl = (float)lv.w[l], Hl = (float)lv.h[l];
            const float w1 = s.hh * s.hw, w2 = s.hh * s.lw, w3 = s.lh * s.hw, w4 = s.lh * s.lw;
            float gs, gx, gy;
            gs = w1 * s1 + w2 * s2 + w3 * s3 + w4 * s4;
            if constexpr (!INST) {
                gx = Wl * as * (s.hh * (s2 - s1) + s.lh * (s4 - s3));
                gy = Hl * as * (s.hw * (s3 - s1) + s.lw * (s4 - s2));
            } else {
                const float t1 = as * s1 + al * m1, t2 = as * s2 + al * m2;
                const float t3 = as * s3 + al * m3, t4 = as * s4 + al * m4;
                gx = Wl * (s.hh * (t2 - t1) + s.lh * (t4 - t3));
                gy = Hl * (s.hw * (t3 - t1) + s.lw * (t4 - t2));
                grad_lv[i] = s.inside ? w1 * m1 + w2 * m2 + w3 * m3 + w4 * m4 : 0.f;
            }
            if constexpr (kCanBuffer) {
                if (buffered) {                               // wave-uniform
                    res[lp] = s.inside ? gs : 0.f;
                    res[kBufLP + 2 * lp] = s.inside ? gx : 0.f;
                    res[kBufLP + 2 * lp + 1] = s.inside ? gy : 0.f;
                    continue;
                }
            }
            grad_sp[i] = s.inside ? gs : 0.f;
            reinterpret_cast<float2 *>(grad_loc)[i] =
                s.inside ? make_float2(gx, gy) : make_float2(0.f, 0.f);
        }
    }
    if constexpr (kCanBuffer) {
        if (buffered && active) {
            wave_lds_sync();                                  // the pair's lanes wrote `res`
            const int n = LP / G, p0 = slot * n;              // this lane's points of the pair: 1, 2 or 4
            const float *r = res + kBufLP + 2 * p0;
            float *gsp = grad_sp + pt0 + p0, *gl = grad_loc + 2 * (pt0 + p0);
            if (n == 4) {
                *reinterpret_cast<float4 *>(gsp) = make_float4(res[p0], res[p0 + 1], res[p0 + 2], res[p0 + 3]);
                reinterpret_cast<float4 *>(gl)[0] = make_float4(r[0], r[1], r[2], r[3]);
                reinterpret_cast<float4 *>(gl)[1] = make_float4(r[4], r[5], r[6], r[7]);
            } else if (n == 2) {
                *reinterpret_cast<float2 *>(gsp) = make_float2(res[p0], res[p0 + 1]);
                *reinterpret_cast<float4 *>(gl) = make_float4(r[0], r[1], r[2], r[3]);
            } else {
                *gsp = res[p0];
                *reinterpret_cast<float2 *>(gl) = make_float2(r[0], r[1]);
            }
        }
    }
    if constexpr (BOXATTN_TUNE_PG_TRACE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the stores have left
        ts[7] = __builtin_amdgcn_s_memtime();
        if (lane == 0 && wv == 0 && active) {
#pragma unroll
            for (int k = 1; k < 8; ++k) grad_sp[pt0 + k] = (float)(unsigned)(ts[k] - ts[0]);
            grad_sp[pt0] = (float)(unsigned)(ts[0] & 0xffffffu);      // start time (low bits): launch order
        }
    }
}

}  // namespace boxattn
