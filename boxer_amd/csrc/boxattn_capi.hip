// C ABI of the box-attention operator (declared in include/boxattn.h): argument checks,
// kernel-variant choice and launches.  No torch / ATen types anywhere in this library.
//
// Replaces the reference host code e2edet/module/ops/src/box_attn/box_attn.cu:15-135 and
// e2edet/module/ops/src/instance_attn/instance_attn.cu:15-157 (checks, output zero-fill,
// launch) and the launchers box_attn_kernel.cuh:1078-1123, 1126-1505.
#include "../../include/boxattn.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <mutex>
#include <type_traits>
#include <vector>

#include "boxattn_binned.h"
#include "boxattn_binned_mfma.h"
#include "boxattn_fast.h"
#include "boxattn_gather2.h"
#include "boxattn_dense_plan.h"
#include "boxattn_generic.h"
#include "boxattn_grid.h"
#include "boxattn_qgrid.h"
#include "boxattn_tile.h"

using namespace boxattn;

namespace {

// 0 auto | 1 generic kernels only | 2 fast atomic kernels (error if the shape does not qualify,
// never the binned backward) | 3 binned backward required (error if not eligible)
// 7 = like 0 without the query-grid (LDS-tiled) kernels
std::atomic<int> g_variant{0};

// Tuning options (boxattn_set_option): process-wide knobs for A/B runs, relaxed atomics.
enum { kOptTileShape = 0, kOptTileRows = 1, kOptTileMarginCap = 2, kOptTileStatic = 3, kOptTileAblate = 4,
       kOptQgTarget = 5, kOptTileFwd = 6, kOptQgAblate = 7, kOptQgWaves = 8, kOptQgBwd = 9, kOptBinChunk = 10,
       kOptDense = 11,        // dense (matrix-core) encoder kernels: 0 default (BOXATTN_DENSE_DEFAULT), 1 off, 2 on
       kOptDenseJit = 12,     // window margin for the predicted box offset, tenths of a box quarter (0: 25)
       kOptDenseRef = 13,     // expected box size in pixels of the query's own level (0: 4, BoxeR's reference windows)
       kOptScanTail = 15,     // training forward: the block scans ride in the forward kernel's launch: 0 default (on), 1 off
       kOptDenseFill = 14,    // bin records counted / written by the window-staged kernels: 0 default (off), 1 off, 2 on
       kOptAccTr = 16,        // bf16 accumulate: 0 default (binned_accumulate_tr_kernel), 1 binned_accumulate_mfma_kernel
       kOptDenseFwd = 17,     // window-staged matrix-core forward for the encoder case: 0 default (BOXATTN_DENSE_FWD_DEFAULT), 1 off, 2 on
       kOptRec12 = 18,        // bf16 box attention: 12-byte bin records: 0 default (off), 1 off, 2 on
       kOptAccF32 = 19,       // float32 box attention, C = 32: accumulate on v_mfma_f32_32x32x2_f32: 0 / 1 off (default: VALU list walk), 2 on
       kNumOpts = 20 };
#ifndef BOXATTN_DENSE_FWD_DEFAULT
#define BOXATTN_DENSE_FWD_DEFAULT 2      // 2: on where eligible, 0: off
#endif
std::atomic<int> g_opt[kNumOpts];      // 0 = default
inline int opt(int k) { return g_opt[k].load(std::memory_order_relaxed); }

inline int ceil_div_sz(size_t a, size_t b) { return (int)((a + b - 1) / b); }

// zero-fill on the stream (zero_fill_kernel, see there why not hipMemsetAsync)
inline hipError_t zero_async(void *p, size_t bytes, hipStream_t st)
{
    if (!bytes) return hipSuccess;
    const int blocks = (int)std::min<size_t>((bytes / 16 + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(zero_fill_kernel, dim3(blocks), dim3(256), 0, st, (unsigned char *)p, bytes);
    return hipGetLastError();
}

inline bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }

struct Dims {
    int B, S, H, C, L, Lq, P;
    bool valid() const
    {
        return B >= 0 && S >= 0 && H > 0 && C > 0 && L > 0 && Lq >= 0 && P > 0;
    }
    size_t n_qh() const { return (size_t)B * Lq * H; }
    size_t n_value() const { return (size_t)B * S * H * C; }
    bool empty() const { return n_qh() == 0; }
};

// Which G (lanes per (query, head)) the fast kernels are instantiated for, VEC = 4.
inline int fast_group(const Dims &d)
{
    if (d.C % 4 != 0 || d.L > kMaxLevels) return 0;
    const int g = d.C / 4;
    return (g == 4 || g == 8 || g == 16) ? g : 0;
}

template <typename ST>
bool fast_ok(const Dims &d, const void *value, const void *loc, const void *a, const void *b,
             const void *c)
{
    if (g_variant == 1) return false;
    if (fast_group(d) == 0) return false;
    const size_t va = 4 * sizeof(ST);
    return aligned(value, va) && aligned(loc, 8) && aligned(a, va) && aligned(b, va) &&
           aligned(c, va);
}

// Lanes per (query, head) pair and channels per lane of the gather kernels (boxattn_gather2.h):
// 8 channels per lane when C allows and the rows can be fetched 16 bytes at a time.
#ifndef BOXATTN_TUNE_VEC_F32
#define BOXATTN_TUNE_VEC_F32 4      // measured: 128-byte fp32 rows gain nothing from 8 channels per lane
#endif
#ifndef BOXATTN_TUNE_VEC_BF16
#define BOXATTN_TUNE_VEC_BF16 8
#endif
#ifndef BOXATTN_TUNE_U8_F32
#define BOXATTN_TUNE_U8_F32 2
#endif
#ifndef BOXATTN_TUNE_U8_BF16
#define BOXATTN_TUNE_U8_BF16 4
#endif
#ifndef BOXATTN_TUNE_U4_F32
#define BOXATTN_TUNE_U4_F32 4      // 0: all G points of a tile
#endif
#ifndef BOXATTN_TUNE_U4_BF16
#define BOXATTN_TUNE_U4_BF16 4
#endif
struct GatherCfg { int G, VEC; };
template <typename ST> inline GatherCfg gather_cfg(const Dims &d, bool rows_16b_aligned)
{
    constexpr int want = sizeof(ST) == 2 ? BOXATTN_TUNE_VEC_BF16 : BOXATTN_TUNE_VEC_F32;
    if (want == 8 && rows_16b_aligned && (d.C == 32 || d.C == 64)) return {d.C / 8, 8};
    return {fast_group(d), 4};
}
// points of a pair in flight per lane (loads issued before the first use)
template <typename ST, int G, int VEC> struct GatherUnroll {
    static constexpr int u8 = sizeof(ST) == 2 ? BOXATTN_TUNE_U8_BF16 : BOXATTN_TUNE_U8_F32;
    static constexpr int u4 = sizeof(ST) == 2 ? BOXATTN_TUNE_U4_BF16 : BOXATTN_TUNE_U4_F32;
    static constexpr int want = VEC == 8 ? u8 : u4;
    static constexpr int value = (want <= 0 || want > G) ? G : want;
};
#ifndef BOXATTN_TUNE_HEAD_XCD
#define BOXATTN_TUNE_HEAD_XCD 1
#endif
inline unsigned div_magic(unsigned d) { return d <= 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / d); }
// index constants of the gather kernels; false if the problem is outside their 32-bit arithmetic
inline bool gather_idx(const Dims &d, GatherIdx &ix, size_t elem_bytes)
{
    const size_t n_qh = d.n_qh();
    if (n_qh >= (1ull << 31) || (size_t)d.B * d.S >= (1ull << 31)) return false;
    ix.n_qh = (unsigned)n_qh;
    ix.magic_h = div_magic((unsigned)d.H);
    ix.magic_lq = div_magic((unsigned)d.Lq);
    ix.rcp_p = 1.0f / (float)d.P;
    // Head-per-XCD placement (pair_of_lane): measured +3..20 % when the queries sample far apart
    // (decoder queries with big boxes, random locations) and -2..7 % for the encoder's local
    // windows, where the contiguous query chunks already keep an XCD's rows together.  So: only
    // for decoder-like shapes (few queries against the map) whose per-head rows fit an XCD's L2.
    ix.head_xcd = (BOXATTN_TUNE_HEAD_XCD && d.H == 8 && (long long)d.Lq * 4 <= d.S &&
                   (size_t)d.B * d.S * d.C * elem_bytes <= (3u << 20))
                      ? 1u : 0u;
    return true;
}
// workgroups (4 waves of `pairs` pairs) of a gather kernel
inline int gather_blocks(const Dims &d, const GatherIdx &ix, int pairs)
{
    if (ix.head_xcd) return 8 * ceil_div_sz((size_t)d.B * d.Lq, (size_t)pairs * 4);
    return ceil_div_sz(d.n_qh(), (size_t)pairs * 4);
}
// the launch geometry as explicit kernel arguments (GatherIdx: grid_x, grid_y, tps)
inline GatherIdx with_grid(GatherIdx ix, int grid_x, int grid_y, int tiles)
{
    ix.grid_x = (unsigned)grid_x;
    ix.grid_y = (unsigned)std::max(1, grid_y);
    ix.tps = (unsigned)((tiles + (int)ix.grid_y - 1) / (int)ix.grid_y);
    return ix;
}
// (the 8-channel kernels are only instantiated for the storage types that select them)
template <typename ST> struct GatherVec8 {
    static constexpr bool value = (sizeof(ST) == 2 ? BOXATTN_TUNE_VEC_BF16 : BOXATTN_TUNE_VEC_F32) == 8;
};
#define BOXATTN_GATHER_DISPATCH(cfg, X)                                                       \
    do {                                                                                      \
        if ((cfg).VEC == 8) {                                                                 \
            if constexpr (GatherVec8<ST>::value) {                                            \
                if ((cfg).G == 4) { X(4, 8) } else { X(8, 8) }                                \
            }                                                                                 \
        } else if ((cfg).G == 4) { X(4, 4) } else if ((cfg).G == 8) { X(8, 4) } else { X(16, 4) } \
    } while (0)

// How many workgroups should share the point tiles of a (query, head) pair: aim at ~4096
// workgroups when the query dimension alone gives fewer than 1024.
inline int point_split(int blocks, int tiles)
{
    if (blocks >= 1024 || tiles < 2) return 1;
    return std::max(1, std::min(tiles, (4096 + blocks - 1) / blocks));
}

inline int finish()
{
    return (int)hipGetLastError();
}
constexpr int kNotEligible = BOXATTN_NOT_ELIGIBLE;   // "use the unfused entry points" (no launch was made)

// ---- optional kernel timing (boxattn_profile_begin/_end) -------------------------------
struct EventPair { hipEvent_t a, b; };
enum { kSlotFwd = 0, kSlotBwdPoints = 1, kSlotBwdAccum = 2, kSlotBwdBin = 3, kSlotBwdCombine = 4,
       kSlotBwdPrep = 5, kNumSlots = BOXATTN_PROFILE_SLOTS };
struct Profile {
    std::atomic<bool> on{false};
    std::mutex mu;                          // forward and backward run on different host threads
    std::vector<EventPair> ev[kNumSlots];
} g_prof;
constexpr size_t kMaxProfiled = 4096;

struct ScopedKernelTimer {            // brackets one kernel launch when profiling is on
    hipStream_t st;
    EventPair ev{};
    std::vector<EventPair> *dst = nullptr;
    ScopedKernelTimer(std::vector<EventPair> &v, hipStream_t s) : st(s)
    {
        if (!g_prof.on.load(std::memory_order_relaxed)) return;
        if (hipEventCreate(&ev.a) != hipSuccess) return;
        if (hipEventCreate(&ev.b) != hipSuccess) { (void)hipEventDestroy(ev.a); return; }
        dst = &v;
        (void)hipEventRecord(ev.a, st);
    }
    ~ScopedKernelTimer()
    {
        if (!dst) return;
        (void)hipEventRecord(ev.b, st);
        std::lock_guard<std::mutex> g(g_prof.mu);
        if (dst->size() < kMaxProfiled) dst->push_back(ev);
        else { (void)hipEventDestroy(ev.a); (void)hipEventDestroy(ev.b); }
    }
};

inline void drain(std::vector<EventPair> &v, double *ms_sum, int *n)
{
    double sum = 0;
    int cnt = 0;
    for (auto &e : v) {
        float ms = 0;
        if (hipEventSynchronize(e.b) == hipSuccess &&
            hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) {
            sum += ms;
            ++cnt;
        }
        (void)hipEventDestroy(e.a);
        (void)hipEventDestroy(e.b);
    }
    v.clear();
    if (ms_sum) *ms_sum = sum;
    if (n) *n = cnt;
}

// ------------------------------------------------------ query-grid kernels (boxattn_tile.h)
struct TileShape { int tx, ty; };
template <typename ST> inline TileShape tile_shape()
{
    // option kOptTileShape: 1 = 16 x 8 queries (512 threads), 2 = 8 x 8 (256 threads)
    const int o = opt(kOptTileShape);
    if (o == 1) return {16, 8};
    if (o == 2) return {8, 8};
    return sizeof(ST) == 2 ? TileShape{16, 8} : TileShape{8, 8};
}
// LDS rows for the windows of one workgroup: two workgroups per CU (160 KiB) by default
template <typename ST> inline int tile_row_budget(const Dims &d)
{
    const int rowb = d.C * (int)sizeof(ST);
    const int o = opt(kOptTileRows);
    const int rows = o > 0 ? o : (78 * 1024) / rowb;
    return std::min(rows, (158 * 1024) / rowb);
}
// The encoder case: one query per pixel of the packed multi-level map, BoxeR's head geometry.
template <typename ST>
inline bool make_tile_plan(const Dims &d, const int64_t *sh, const int64_t *ls, TileShape ts,
                           TilePlan &p)
{
    if (!sh || !ls || g_variant == 1 || g_variant == 2 || g_variant == 7) return false;
    // (measured slower than the row gathers so far -- both are instruction-bound, DESIGN.md 4.1 --
    // so the kernel is opt-in: option kOptTileFwd = 1)
    if (opt(kOptTileFwd) != 1) return false;
    if (d.Lq != d.S || d.C != 32 || d.L > kTileMaxLevels || d.P % 4 != 0) return false;
    const int nr = d.L * d.P / 4;
    if (nr != 2 && nr != 4) return false;
    if ((size_t)d.B * d.S * d.H * d.C * sizeof(ST) >= kOobOffset ||
        (size_t)d.B * d.Lq * d.H >= (1ull << 31) / 64)
        return false;
    long long next = 0, tiles = 0;
    p.L = d.L;
    for (int l = 0; l < d.L; ++l) {
        const long long hl = sh[2 * l], wl = sh[2 * l + 1];
        if (hl <= 0 || wl <= 0 || hl > 32000 || wl > 32000 || ls[l] != next) return false;
        next += hl * wl;
        const long long ntx = (wl + ts.tx - 1) / ts.tx, nty = (hl + ts.ty - 1) / ts.ty;
        p.lv[l] = TileLevel{(int)hl, (int)wl, (int)ls[l], (int)ntx, (int)tiles, 1.0f / (float)wl,
                            1.0f / (float)hl};
        tiles += ntx * nty;
    }
    if (next != d.S || tiles * d.B * d.H >= (1ll << 31)) return false;
    p.n_tiles = (int)tiles;
    p.row_budget = tile_row_budget<ST>(d);
    p.margin_cap = opt(kOptTileMarginCap) > 0 ? opt(kOptTileMarginCap) : 24;
    p.static_q16 = opt(kOptTileStatic);
    p.ablate = opt(kOptTileAblate);
    return true;
}

template <typename K> inline hipError_t allow_dynamic_lds(K kernel, size_t bytes)
{
    return hipFuncSetAttribute(reinterpret_cast<const void *>(kernel),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <typename ST>
int launch_fwd_tile(const ST *value, const float *loc, const float *attn, const Dims &d,
                    const TilePlan &plan, TileShape ts, ST *out, hipStream_t st)
{
    const size_t lds = (size_t)plan.row_budget * d.C * sizeof(ST);
    const unsigned grid = (unsigned)((size_t)d.B * plan.n_tiles * d.H);
    const unsigned vbytes = (unsigned)(d.n_value() * sizeof(ST));
    const int nr = d.L * d.P / 4;
#define BOXATTN_FWD_TILE(NR_, TX_, TY_)                                                         \
    do {                                                                                        \
        auto k = fwd_tile_kernel<ST, NR_, TX_, TY_>;                                            \
        hipError_t e = allow_dynamic_lds(k, lds);                                               \
        if (e != hipSuccess) return (int)e;                                                     \
        hipLaunchKernelGGL(k, dim3(grid), dim3(TX_ * TY_ * 4), lds, st, value, loc, attn, plan, \
                           d.S, d.H, d.Lq, d.P, out, vbytes);                                   \
    } while (0)
    if (ts.tx == 16) {
        if (nr == 4) BOXATTN_FWD_TILE(4, 16, 8); else BOXATTN_FWD_TILE(2, 16, 8);
    } else {
        if (nr == 4) BOXATTN_FWD_TILE(4, 8, 8); else BOXATTN_FWD_TILE(2, 8, 8);
    }
#undef BOXATTN_FWD_TILE
    return finish();
}

// ------------------------------------------------------------------------------ forward
inline bool make_dense_plan(const Dims &d, const int64_t *sh, const int64_t *ls, DensePlan &p);

template <typename ST, bool INST>
int launch_fwd(const ST *value, const int64_t *shapes, const int64_t *lsi,
               const typename Storage<ST>::compute *loc,
               const typename Storage<ST>::compute *w_sp,
               const typename Storage<ST>::compute *w_lv, const Dims &d, ST *out, ST *mask,
               hipStream_t st, const int64_t *shapes_host = nullptr,
               const int64_t *lsi_host = nullptr, const ScanTail *scan_tail = nullptr,
               bool *scan_tail_taken = nullptr)
{
    // scan_tail: the backward's block scans to run as extra workgroups of the forward kernel (training
    // forward); *scan_tail_taken says whether the kernel that was launched carried them
    if (scan_tail_taken) *scan_tail_taken = false;
    if (!d.valid()) return (int)hipErrorInvalidValue;
    if (d.empty()) return 0;                                   // no queries: nothing to write
    if (!shapes || !lsi || !loc || !w_sp || !out || (INST && (!w_lv || !mask)))
        return (int)hipErrorInvalidValue;
    const size_t n_qh = d.n_qh();
    if (d.n_value() == 0) {                                    // no pixels: every sample is 0
        hipError_t e = zero_async(out, n_qh * d.C * sizeof(ST), st);
        if (e == hipSuccess && INST)
            e = zero_async(mask, n_qh * d.C * d.P * sizeof(ST), st);
        return (int)e;
    }
    if (!value) return (int)hipErrorInvalidValue;
    if constexpr (!std::is_same<ST, double>::value) {
        if (fast_ok<ST>(d, value, loc, out, INST ? (const void *)mask : (const void *)out,
                        out)) {
            if constexpr (!INST && std::is_same<ST, bf16_t>::value) {     // encoder case: window-staged matrix-core forward
                DensePlan dp;
                if (opt(kOptDenseFwd) != 1 && BOXATTN_DENSE_FWD_DEFAULT + opt(kOptDenseFwd) >= 2 && shapes_host &&
                    lsi_host && aligned(value, 16) && aligned(out, 16) && aligned(loc, 8) &&
                    make_dense_plan(d, shapes_host, lsi_host, dp)) {
                    const bool tail = scan_tail && scan_tail->n_wg > 0;
                    ScopedKernelTimer timer(g_prof.ev[kSlotFwd], st);
                    launch_fwd_dense(value, loc, w_sp, out, dp, (unsigned)(d.n_value() * sizeof(bf16_t)),
                                     tail ? scan_tail : nullptr, st);
                    if (tail && scan_tail_taken) *scan_tail_taken = true;
                    return finish();
                }
            }
            if constexpr (!INST) {          // encoder case: LDS-staged value windows
                TilePlan tp;
                const TileShape ts = tile_shape<ST>();
                if (aligned(value, 16) && aligned(out, 16) &&
                    make_tile_plan<ST>(d, shapes_host, lsi_host, ts, tp)) {
                    ScopedKernelTimer timer(g_prof.ev[kSlotFwd], st);
                    return launch_fwd_tile<ST>(value, loc, w_sp, d, tp, ts, out, st);
                }
            }
            const size_t vbytes = d.n_value() * sizeof(ST);
            GatherIdx ix{};
            const bool gen2 = g_variant != 2 && vbytes < kOobOffset && gather_idx(d, ix, sizeof(ST));   // buffer-load kernels
            const GatherCfg cfg =
                gen2 ? gather_cfg<ST>(d, aligned(value, 16) && aligned(out, 16) &&
                                         (!INST || aligned(mask, 16)))
                     : GatherCfg{fast_group(d), 4};
            const int G = cfg.G;
            const int pairs = kWave / G;
            const int blocks = gen2 ? gather_blocks(d, ix, pairs) : ceil_div_sz(n_qh, (size_t)pairs * 4);
            // instance attention with few queries and many points: split the points of a pair
            // over several workgroups (fp32 only: partial outs are combined with atomics)
            // (variant 5 = A/B switch: keep the split instead of the one-wave-per-pair kernel)
            const bool wide = INST && gen2 && g_variant != 5 && blocks < 1024 &&
                              d.P >= kWave / G;
            int fsplit = 1;
            if (INST && gen2 && !wide && std::is_same<ST, float>::value)
                fsplit = point_split(blocks, (d.P + G - 1) / G);
            if (fsplit > 1) {
                hipError_t e = zero_async(out, n_qh * d.C * sizeof(ST), st);
                if (e != hipSuccess) return (int)e;
            }
            ScopedKernelTimer timer(g_prof.ev[kSlotFwd], st);
            // instance attention with few pairs and many points: one wave per pair, the points
            // spread over the lane groups (any storage type; replaces the fp32-only atomic
            // split when it gives more waves)
            if constexpr (INST) {
                if (wide) {
                    const int wblocks = ix.head_xcd ? 8 * ceil_div_sz((size_t)d.B * d.Lq, 4)
                                                    : ceil_div_sz(n_qh, 4);
                    const bool wtail = scan_tail && scan_tail->n_wg > 0;
                    const ScanTail wsct = wtail ? *scan_tail : ScanTail{};
                    const int wlead = wtail ? wsct.plan.n_slices * kScanSub : 0;        // in FRONT of the grid
                    ix.lead = (unsigned)wlead;
#define BOXATTN_FWD_WIDE(GG, VV)                                                              \
    hipLaunchKernelGGL((fwd_inst_wide_kernel<ST, GG, VV>), dim3(wblocks + wlead), dim3(256), 0, st, \
                       value, shapes, lsi, loc, w_sp, w_lv, d.S, d.H, d.L, d.Lq, d.P, out,    \
                       mask, with_grid(ix, wblocks, 1, 1), (unsigned)vbytes, wsct);
                    BOXATTN_GATHER_DISPATCH(cfg, BOXATTN_FWD_WIDE);
#undef BOXATTN_FWD_WIDE
                    if (wtail && scan_tail_taken) *scan_tail_taken = true;
                    return finish();
                }
            }
            if (gen2) {
                const bool tail = scan_tail && scan_tail->n_wg > 0 && fsplit == 1;
                const ScanTail sct = tail ? *scan_tail : ScanTail{};
                const int tail_blocks = tail ? sct.plan.n_slices * kScanSub : 0;        // in FRONT of the grid (kScanSub = 8 each)
                ix.lead = (unsigned)tail_blocks;
#define BOXATTN_FWD2(GG, VV)                                                                  \
    hipLaunchKernelGGL((fwd2_kernel<ST, GG, INST, GatherUnroll<ST, GG, VV>::value, VV>),      \
                       dim3(blocks + tail_blocks, fsplit), dim3(256), 0, st, value, shapes, lsi, loc, w_sp, \
                       w_lv, d.S, d.H, d.L, d.Lq, d.P, out, mask,                             \
                       with_grid(ix, blocks, fsplit, (d.P + GG - 1) / GG), (unsigned)vbytes,  \
                       GridSrc{}, sct);
                BOXATTN_GATHER_DISPATCH(cfg, BOXATTN_FWD2);
#undef BOXATTN_FWD2
                if (tail && scan_tail_taken) *scan_tail_taken = true;
            } else {
#define BOXATTN_FWD_CASE(GG)                                                                  \
    case GG:                                                                                  \
        hipLaunchKernelGGL((fwd_fast_kernel<ST, 4, GG, INST>), dim3(blocks), dim3(256), 0, st, \
                           value, shapes, lsi, loc, w_sp, w_lv, d.S, d.H, d.L, d.Lq, d.P, out, \
                           mask, n_qh);                                                       \
        break;
                switch (G) {
                    BOXATTN_FWD_CASE(4)
                    BOXATTN_FWD_CASE(8)
                    BOXATTN_FWD_CASE(16)
                }
#undef BOXATTN_FWD_CASE
            }
            return finish();
        }
        if (g_variant == 2) return (int)hipErrorInvalidValue;
    }
    const size_t n = n_qh * d.C;
    const int blocks = (int)std::min<size_t>((n + 255) / 256, (size_t)1 << 20);
    ScopedKernelTimer timer(g_prof.ev[kSlotFwd], st);
    hipLaunchKernelGGL((fwd_generic_kernel<ST, INST>), dim3(blocks), dim3(256), 0, st, value,
                       shapes, lsi, loc, w_sp, w_lv, d.S, d.H, d.C, d.L, d.Lq, d.P, out, mask,
                       n);
    return finish();
}

// Forward from boxes (GRID flavour of fwd2_kernel): also writes the sampling grid.
template <typename ST>
int launch_fwd_grid(const ST *value, const int64_t *shapes, const int64_t *lsi, const float *attn,
                    const Dims &d, ST *out, const GridSrc &gs, hipStream_t st)
{
    if (!d.valid()) return (int)hipErrorInvalidValue;
    if (d.empty() || d.n_value() == 0) return kNotEligible;
    if (!value || !shapes || !lsi || !attn || !out || !gs.ref || !gs.offsets || !gs.kidx || !gs.grid_out)
        return (int)hipErrorInvalidValue;
    GatherIdx ix{};
    const size_t vbytes = d.n_value() * sizeof(ST);
    if (!fast_ok<ST>(d, value, gs.grid_out, out, out, out) || g_variant == 2 || vbytes >= kOobOffset ||
        !gather_idx(d, ix, sizeof(ST)))
        return kNotEligible;
    const GatherCfg cfg = gather_cfg<ST>(d, aligned(value, 16) && aligned(out, 16));
    const int blocks = gather_blocks(d, ix, kWave / cfg.G);
    ScopedKernelTimer timer(g_prof.ev[kSlotFwd], st);
#define BOXATTN_FWD2G(GG, VV)                                                                      \
    hipLaunchKernelGGL((fwd2_kernel<ST, GG, false, GatherUnroll<ST, GG, VV>::value, VV, true>),   \
                       dim3(blocks, 1), dim3(256), 0, st, value, shapes, lsi, (const float *)nullptr, \
                       attn, (const float *)nullptr, d.S, d.H, d.L, d.Lq, d.P, out, (ST *)nullptr, \
                       with_grid(ix, blocks, 1, 1), (unsigned)vbytes, gs);
    BOXATTN_GATHER_DISPATCH(cfg, BOXATTN_FWD2G);
#undef BOXATTN_FWD2G
    return finish();
}

// ----------------------------------------------------------------------------- backward
// GV = accumulation buffer for grad_value (the output itself for f32/f64, scratch for bf16)
template <typename ST, bool INST>
int launch_bwd(const ST *value, const int64_t *shapes, const int64_t *lsi,
               const typename Storage<ST>::compute *loc,
               const typename Storage<ST>::compute *w_sp,
               const typename Storage<ST>::compute *w_lv, const ST *grad_out,
               const ST *grad_mask, const Dims &d, ST *grad_value,
               typename Storage<ST>::compute *grad_loc, typename Storage<ST>::compute *grad_sp,
               typename Storage<ST>::compute *grad_lv,
               typename Storage<ST>::compute *grad_value_acc, hipStream_t st)
{
    typedef typename Storage<ST>::compute T;
    if (!d.valid()) return (int)hipErrorInvalidValue;
    const size_t nv = d.n_value();
    const size_t n_qh = d.n_qh();
    if (nv) {
        if (!grad_value || !grad_value_acc) return (int)hipErrorInvalidValue;
        hipError_t e = zero_async(grad_value_acc, nv * sizeof(T), st);
        if (e != hipSuccess) return (int)e;
    }
    if (n_qh) {
        if (!shapes || !lsi || !loc || !w_sp || !grad_out || !grad_loc || !grad_sp ||
            (INST && (!w_lv || !grad_mask || !grad_lv)))
            return (int)hipErrorInvalidValue;
        if (!nv) {                                             // no pixels: all gradients 0
            const size_t np = n_qh * d.L * d.P;
            hipError_t e = zero_async(grad_loc, 2 * np * sizeof(T), st);
            if (e == hipSuccess) e = zero_async(grad_sp, np * sizeof(T), st);
            if (e == hipSuccess && INST) e = zero_async(grad_lv, np * sizeof(T), st);
            return (int)e;
        }
        if (!value) return (int)hipErrorInvalidValue;
    }
    if (n_qh && nv) {
        ScopedKernelTimer timer(g_prof.ev[kSlotBwdPoints], st);
        bool done = false;
        if constexpr (!std::is_same<ST, double>::value) {
            if (fast_ok<ST>(d, value, loc, grad_out,
                            INST ? (const void *)grad_mask : (const void *)grad_out,
                            grad_loc)) {
                const int G = fast_group(d);
                const int pairs = kWave / G;
                const int blocks = ceil_div_sz(n_qh, (size_t)pairs * 4);
#define BOXATTN_BWD_CASE(GG)                                                                  \
    case GG:                                                                                  \
        hipLaunchKernelGGL((bwd_fast_kernel<ST, 4, GG, INST>), dim3(blocks), dim3(256), 0, st, \
                           value, shapes, lsi, loc, w_sp, w_lv, grad_out, grad_mask, d.S,    \
                           d.H, d.L, d.Lq, d.P, grad_value_acc, grad_loc, grad_sp, grad_lv,  \
                           n_qh);                                                             \
        break;
                switch (G) {
                    BOXATTN_BWD_CASE(4)
                    BOXATTN_BWD_CASE(8)
                    BOXATTN_BWD_CASE(16)
                }
#undef BOXATTN_BWD_CASE
                done = true;
            } else if (g_variant == 2) {
                return (int)hipErrorInvalidValue;
            }
        }
        if (!done) {
            const int blocks = (int)std::min<size_t>((n_qh + 3) / 4, (size_t)1 << 20);
            hipLaunchKernelGGL((bwd_generic_kernel<ST, INST>), dim3(blocks), dim3(256), 0, st,
                               value, shapes, lsi, loc, w_sp, w_lv, grad_out, grad_mask, d.S,
                               d.H, d.C, d.L, d.Lq, d.P, grad_value_acc, grad_loc, grad_sp,
                               grad_lv, n_qh);
        }
        int rc = finish();
        if (rc) return rc;
    }
    if constexpr (std::is_same<ST, bf16_t>::value) {
        if (nv) {
            const int blocks = (int)std::min<size_t>((nv / 4 + 255) / 256 + 1, 256 * 16);
            hipLaunchKernelGGL(cvt_f32_to_bf16_kernel, dim3(blocks), dim3(256), 0, st,
                               grad_value_acc, grad_value, nv);
            return finish();
        }
    }
    return 0;
}


// One helper stream + two events per device, created on first use and kept for the life of
// the process (no device memory).  fork: side waits for everything queued on `main` so far;
// join: `main` waits for everything queued on the side stream.
struct SideStream {
    static constexpr int kMaxDev = 16;
    struct PerDev { hipStream_t s = nullptr; hipEvent_t fork = nullptr, join = nullptr; };
    // The forward runs on the user's thread, the backward on an autograd worker: the lazy
    // creation and every record + wait pair on the shared events are done under one lock.
    static std::mutex &lock()
    {
        static std::mutex m;
        return m;
    }
    static PerDev &slot()
    {
        static PerDev devs[kMaxDev];
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::lock_guard<std::mutex> g(lock());
        PerDev &p = devs[dev % kMaxDev];
        if (!p.s) {
            // lowest priority: the side kernel fills what the main chain leaves idle
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            if (hipStreamCreateWithPriority(&p.s, hipStreamNonBlocking, least) != hipSuccess)
                p.s = nullptr;
            if (p.s && (hipEventCreateWithFlags(&p.fork, hipEventDisableTiming) != hipSuccess ||
                        hipEventCreateWithFlags(&p.join, hipEventDisableTiming) != hipSuccess))
                p.s = nullptr;
        }
        return p;
    }
    hipStream_t main_, side_;
    PerDev *p_;
    // `worth` = the caller's measured choice (problem big enough for the ~20 us of fork / join
    // latency, and a kernel mix that gains from running side by side); variant 4 forces the
    // one-stream schedule, variant 6 the two-stream one
    explicit SideStream(hipStream_t main, bool worth = true) : main_(main), side_(main), p_(&slot())
    {
        if (!p_->s || g_variant == 4 || (!worth && g_variant != 6)) return;   // stay on `main`
        std::lock_guard<std::mutex> g(lock());
        if (hipEventRecord(p_->fork, main_) == hipSuccess &&
            hipStreamWaitEvent(p_->s, p_->fork, 0) == hipSuccess)
            side_ = p_->s;
    }
    hipStream_t stream() const { return side_; }
    void join()
    {
        if (side_ == main_) return;
        std::lock_guard<std::mutex> g(lock());
        (void)hipEventRecord(p_->join, side_);
        (void)hipStreamWaitEvent(main_, p_->join, 0);
        side_ = main_;
    }
    ~SideStream() { join(); }
};

// ------------------------------------------------------- binned backward (boxattn_binned.h)
#ifndef BOXATTN_TUNE_CHUNK
#define BOXATTN_TUNE_CHUNK 1024
#endif
constexpr int kChunk = BOXATTN_TUNE_CHUNK;   // records per work item (upper bound)
// Records per work item.  One wavefront works an item off 64 records a round, and a round is a
// chain of dependent latencies (~3-5 us), so the kernel lasts at least rounds-per-item rounds:
// with few sample points (the decoders: 300 queries) 1 024-record items leave a handful of waves
// running 16 rounds while the rest of the chip idles.  Aim at ~256 items per (image, head) slice
// -- about the wave slots a slice gets -- between 128 and kChunk records.
inline int bin_chunk(const Dims &d)
{
    const int forced = opt(kOptBinChunk);
    if (forced > 0) return std::min(4096, std::max(64, (forced + 63) / 64 * 64));
    const long long rec_est = 3ll * d.Lq * d.L * d.P / 2;          // ~1.4 records per point
    const long long c = (rec_est / 256 + 63) / 64 * 64;
    return (int)std::min<long long>(kChunk, std::max<long long>(128, c));
}
// One stream by default.  Running the point-gradient kernel (and, in the training forward, the
// bin passes) on the library's helper stream once paid for fp32 storage (C2: 283 -> 270 us per
// step); with the faster bin passes it no longer does (C2 fp32 278 us on one stream, 283 on
// two; C5' 932 / 928; bf16 192 / 208): every cross-stream dependency costs ~20 us of event
// latency and the kernels mostly contend for the same CUs (rocprofv3 timelines).
// boxattn_set_variant(6) still forks.
template <typename ST> inline bool side_stream_worth(const Dims &) { return false; }
#ifndef BOXATTN_TUNE_ACC_MFMA
#define BOXATTN_TUNE_ACC_MFMA 1    // bf16 box attention: the round's scatter-add as a dense MFMA product
#endif
#ifndef BOXATTN_TUNE_WIDE_F32
#define BOXATTN_TUNE_WIDE_F32 0    // fp32 box attention (VALU accumulate kernel) from wide records: on uniformly random
#endif                             // locations accumulate 154 -> 103 us, fill +17; on model-like ones only the +17
// flavours whose accumulate step runs on the matrix cores (boxattn_binned_mfma.h): box attention
// in bf16 storage.  The float32 flavour of that kernel (two-term bf16 split of the upstream rows,
// 32 channels per head) exists and is parity-green, but measured no gain at BoxeR-R50 shapes --
// accumulate 103 -> 98 us, the wide records it needs +12 us in the fill pass -- so float32
// storage keeps the float32-exact VALU kernel unless boxattn_set_variant(11) asks for it.
// float32 storage with 32 channels per head: binned_accumulate_f32_kernel (float32 MFMAs: exact like the VALU kernel)
inline bool f32_mfma_ok(const Dims &d)
{
    // (opt-in: measured at C2 97 us against the VALU kernel's 102 on model-like inputs -- the matrix pipe is busy 63 us
    // of them -- and 100 against 133 on uniformly random ones; + 6 us for the wide records in the fill pass)
    return opt(kOptAccF32) == 2 && d.C == 32 && (size_t)d.B * d.Lq * d.H * 32 * 4 < kAccTrMaxBytes &&
           d.Lq < (1 << 24) && d.H * 128 < (1 << 24);
}
template <typename ST, bool INST> inline bool mfma_accumulate(const Dims &d)
{
    if (!BOXATTN_TUNE_ACC_MFMA || INST) return false;
    if (std::is_same<ST, bf16_t>::value) return true;
    return std::is_same<ST, float>::value && d.C == 32 && (g_variant == 11 || f32_mfma_ok(d));
}
// flavours whose bin passes write 16-byte records {id, x, y, weight}
template <typename ST, bool INST> inline bool wide_records(const Dims &d)
{
    return mfma_accumulate<ST, INST>(d) || (BOXATTN_TUNE_WIDE_F32 && !INST && sizeof(ST) == 4);
}
// the workspace query only knows the storage type and the dimensions: room for wide records
// whenever a flavour of that type may write them
inline bool wide_workspace(bool is_bf16, const Dims &d)
{
    // float32: only the opt-in matrix-core flavour (variant 11) writes wide records; a call whose
    // variant changed after the size query finds the workspace too small and takes the fallback
    return is_bf16 ? BOXATTN_TUNE_ACC_MFMA != 0
                   : (BOXATTN_TUNE_WIDE_F32 != 0 ||
                      (BOXATTN_TUNE_ACC_MFMA != 0 && d.C == 32 && (g_variant == 11 || f32_mfma_ok(d))));
}
constexpr int kMaxBlocks = 8192;      // per (image, head) slice: one LDS int each in bin_kernel

inline size_t align_up(size_t x, size_t a = 256) { return (x + a - 1) / a * a; }

struct WsLayout {
    size_t n_items;
    size_t part, tickets, subtot, offsets, items, combos, records, partials, scan_tmp, cursor, total;
    int q_per_wg, n_wg;                                     // launch geometry of the bin passes
};

inline bool make_plan_blocks(const Dims &d, const int64_t *sh, const int64_t *ls, BinPlan &p)
{
    constexpr int BW = 8, BH = 4;
    p.L = d.L;
    long long blk0 = 0, next_start = 0;
    for (int l = 0; l < d.L; ++l) {
        const long long hl = sh[2 * l], wl = sh[2 * l + 1], st = ls[l];
        // standard packed layout only: level l starts where level l-1 ends (every grad_value
        // row then has exactly one owner block)
        if (hl < 0 || wl < 0 || hl > INT32_MAX || wl > INT32_MAX || st != next_start ||
            st + hl * wl > d.S)
            return false;
        next_start = st + hl * wl;
        p.lv[l].H = (int)hl;
        p.lv[l].W = (int)wl;
        p.lv[l].start = (int)st;
        p.lv[l].nbx = (int)((wl + BW - 1) / BW);
        p.lv[l].nby = (int)((hl + BH - 1) / BH);
        // blk_of(): floor(x nb / size) by multiply-high is exact while x nb < 2^32 / size
        if ((unsigned long long)wl * wl * p.lv[l].nbx >= (1ull << 32) ||
            (unsigned long long)hl * hl * p.lv[l].nby >= (1ull << 32))
            return false;
        p.lv[l].mw = wl > 1 ? (unsigned)((1ull << 32) / (unsigned long long)wl + 1) : 0u;
        p.lv[l].mh = hl > 1 ? (unsigned)((1ull << 32) / (unsigned long long)hl + 1) : 0u;
        // blk_lo_magic(): first coordinate of a block by multiply-high, checked against the division
        const auto lo_magic = [](long long size, int nb) -> unsigned {
            const unsigned m = (unsigned)((1ull << 32) / (unsigned long long)nb + 1);
            if (nb == 1) return 0u;          // (one block: its origin is 0, which a magic of 0 delivers)
            for (int c = 0; c <= nb; ++c) {
                const unsigned long long n = (unsigned long long)c * size + nb - 1;
                if (n >= (1ull << 23) || (unsigned)((n * m) >> 32) != (unsigned)(n / nb)) return 0u;     // (n: a 24-bit multiply + nb)
            }
            return m;
        };
        p.lv[l].mnx = lo_magic(wl, p.lv[l].nbx);
        p.lv[l].mny = lo_magic(hl, p.lv[l].nby);
        p.lv[l].blk0 = (int)blk0;
        blk0 += (long long)p.lv[l].nbx * p.lv[l].nby;
    }
    // ... and the levels cover all of S: the binned kernels only store the rows a level block
    // owns, a padded tail (S > sum H_l W_l) would stay uninitialised (the atomic path zero-fills)
    if (next_start != d.S) return false;
    const long long rec_cap = 4ll * d.Lq * d.L * d.P;
    if (blk0 == 0 || blk0 > kMaxBlocks || rec_cap > INT32_MAX / 2 ||
        (long long)d.B * d.Lq * d.H > INT32_MAX || (long long)d.B * d.Lq * d.P * d.H > INT32_MAX)
        return false;
    int lp_bits = 0;
    while ((1ll << lp_bits) < (long long)d.L * d.P) ++lp_bits;
    if (((long long)d.Lq << lp_bits) > INT32_MAX) return false;
    if ((long long)d.L * d.P > (1 << 16)) return false;       // keeps per-workgroup point counts < 2^24
    p.lp_bits = lp_bits;
    p.n_slices = d.B * d.H;
    p.nblk = (int)blk0;
    p.rec_cap = (int)rec_cap;
    p.chunk = bin_chunk(d);
    p.item_cap = (int)(blk0 + rec_cap / p.chunk + 1);
    // blocks with more than one chunk: sum of their chunk counts <= 2 * records / chunk
    p.pslot_cap = (int)std::min<long long>(2 * (rec_cap / p.chunk) + 2, blk0 + rec_cap / p.chunk + 1);
    return true;
}

inline bool make_plan(const Dims &d, const int64_t *sh, const int64_t *ls, BinPlan &p)
{
    if (!sh || !ls || !d.valid() || fast_group(d) == 0 || d.L > kMaxBinLevels) return false;
    return make_plan_blocks(d, sh, ls, p);
}

// `wide`: 16-byte records {id, x, y, weight} (bf16 storage: the MFMA accumulate kernel) instead of
// 4-byte point ids
inline WsLayout ws_layout(const Dims &d, const BinPlan &p, bool wide)
{
    const size_t ns = (size_t)d.B * d.H;
    WsLayout w;
    // ~2048 workgroups for the two binning passes
    // ... and at most kScanSub * kScanWgPerSub workgroups per slice (bin_scan_a_kernel)
    const long long wg_target = 2048ll * 256 / kBinThreads;   // ~8 waves per SIMD over the chip
    w.q_per_wg = std::max(8, (int)(((long long)d.Lq * (long long)ns + wg_target - 1) / wg_target));
    w.q_per_wg = std::max(w.q_per_wg, (d.Lq + kScanSub * kScanWgPerSub - 1) /
                                          (kScanSub * kScanWgPerSub));
    // ... and every workgroup initialises, flushes and has scanned one counter per block: give it
    // at least 4 points per block (few queries on a big map: 1 000 queries on 468 x 468 = 6 903
    // blocks per slice now use 1 workgroup per slice instead of 67)
    const long long lp = (long long)d.L * d.P;
    w.q_per_wg = (int)std::max<long long>(w.q_per_wg, (4ll * p.nblk + lp - 1) / lp);
    w.n_wg = (d.Lq + w.q_per_wg - 1) / w.q_per_wg;
    size_t o = 0;
    w.n_items = o; o += align_up(ns * 2 * 4);
    w.part = o;    o += align_up(ns * std::max(w.n_wg, kDenseGroups) * (size_t)p.nblk * 4);
    w.tickets = o; o += align_up(ns * 4);        // the scans riding in the forward kernel's launch (ScanTail)
    w.subtot = o;  o += align_up(ns * kScanSub * (size_t)p.nblk * 4);
    w.offsets = o; o += align_up(ns * (p.nblk + 1) * 4);
    w.items = o;   o += align_up(ns * p.item_cap * 16);
    w.combos = o;  o += align_up(ns * (size_t)p.nblk * 16);
    w.records = o; o += align_up(ns * (size_t)p.rec_cap * (wide ? 16 : 4));
    w.partials = o; o += align_up(ns * (size_t)p.pslot_cap * 32 * d.C * 4);
    // multi-workgroup block scan (more than kScanThreads blocks per slice): per-block prefixes
    // inside a segment + the segments' totals
    w.scan_tmp = o;
    if (p.nblk > kScanThreads) o += align_up(ns * ((size_t)p.nblk + kMaxBlocks / kScanThreads) * 16);
    // window-staged kernels as the binning passes (bf16 box attention, encoder): their groups' counts use
    // the first kDenseGroups workgroup rows of `part`; one cursor per (slice, group, block) on top
    w.cursor = o;
    if (wide && p.nblk <= kDenseFillMaxBlocks) o += align_up(ns * kDenseGroups * (size_t)p.nblk * 4);
    w.total = o;
    return w;
}

// Binning passes (count, two scans, fill) of the binned backward into the workspace.  They only
// read the sampling locations, so the training forward can run them ahead of the backward.
template <bool WIDE, bool INTERLEAVE>
inline void launch_binning_t(const float *loc, const float *w_sp, const Dims &d, const BinPlan &plan,
                             const WsLayout &w, char *ws, hipStream_t st, int stages, bool rec12);
// wide: 16-byte records; interleave: queries interleaved over the bin workgroups (VALU accumulate)
// stages: kBinCount | kBinScan | kBinFill (all three in one call, or -- the training forward, which
// lets the scans ride in the forward kernel's launch -- one at a time)
enum { kBinCount = 1, kBinScan = 2, kBinFill = 4, kBinAll = 7,
       kBinTickets = 8 };    // with kBinCount: clear the tickets of the scans that will ride in the forward kernel
inline void launch_binning(bool wide, bool interleave, const float *loc, const float *w_sp,
                           const Dims &d, const BinPlan &plan, const WsLayout &w, char *ws,
                           hipStream_t st, int stages = kBinAll, bool rec12 = false)
{
    // rec12 (wide, not interleaved): the fill pass writes 12-byte records (touched_blocks12)
    if (wide && interleave) launch_binning_t<true, true>(loc, w_sp, d, plan, w, ws, st, stages, false);
    else if (wide) launch_binning_t<true, false>(loc, w_sp, d, plan, w, ws, st, stages, rec12);
    else if (interleave) launch_binning_t<false, true>(loc, w_sp, d, plan, w, ws, st, stages, false);
    else launch_binning_t<false, false>(loc, w_sp, d, plan, w, ws, st, stages, false);
}
// may the matrix-core accumulate of this call be binned_accumulate_tr_kernel (what 12-byte records need)?
template <typename ST, int C> inline bool accumulate_tr_ok(const Dims &d)
{
    if constexpr (std::is_same<ST, bf16_t>::value && (C == 16 || C == 32 || C == 64))
        return (size_t)d.B * d.Lq * d.H * C * sizeof(ST) < kAccTrMaxBytes && d.Lq < (1 << 24) && d.H * C * 2 < (1 << 24);
    return false;
}
// 12-byte bin records (bf16 box attention on the matrix-core accumulate): ids below 2^24 and exact block origins
template <typename ST, bool INST> inline bool rec12_ok(const Dims &d, const BinPlan &plan)
{
    if constexpr (std::is_same<ST, bf16_t>::value && !INST) {
        // (opt-in: the 16-bit fractions cost the "single terms correctly rounded" guarantee of the 16-byte records --
        // an ABSOLUTE 2^-17 on a bilinear fraction is a large relative error on a tiny weight -- for 2.6 us of 142)
        if (opt(kOptRec12) != 2 || opt(kOptAccTr) == 1 || g_variant == 11) return false;
        const bool tr = d.C == 16 ? accumulate_tr_ok<ST, 16>(d) : d.C == 32 ? accumulate_tr_ok<ST, 32>(d)
                                  : d.C == 64 ? accumulate_tr_ok<ST, 64>(d) : false;
        if (!tr || ((long long)d.Lq << plan.lp_bits) > (1ll << 24)) return false;
        for (int l = 0; l < plan.L; ++l)
            if ((plan.lv[l].nbx > 1 && !plan.lv[l].mnx) || (plan.lv[l].nby > 1 && !plan.lv[l].mny)) return false;
        return true;
    }
    return false;
}
// Can the two scan kernels of this plan run as bin_scan_tail_body workgroups?
inline bool scan_tail_ok(const BinPlan &plan, const WsLayout &w)
{
    return opt(kOptScanTail) != 1 && plan.nblk <= kScanThreads && w.n_wg <= kScanSub * kScanWgPerSub;
}
inline ScanTail scan_tail(const BinPlan &plan, const WsLayout &w, char *ws)
{
    ScanTail t{};
    t.subtot = (int *)(ws + w.subtot); t.offsets = (int *)(ws + w.offsets);
    t.items = (int4 *)(ws + w.items); t.combos = (int4 *)(ws + w.combos);
    t.n_items = (int *)(ws + w.n_items); t.part = (int *)(ws + w.part); t.tickets = (int *)(ws + w.tickets);
    t.plan = plan;
    t.n_wg = w.n_wg;
    return t;
}
template <bool WIDE, bool INTERLEAVE>
inline void launch_binning_t(const float *loc, const float *w_sp, const Dims &d, const BinPlan &plan,
                             const WsLayout &w, char *ws, hipStream_t st, int stages, bool rec12)
{
    constexpr int BW = 8, BH = 4;
    const int ns = d.B * d.H;
    int *part = (int *)(ws + w.part), *subtot = (int *)(ws + w.subtot);
    int *n_items = (int *)(ws + w.n_items);      // every scratch word is written before it is read
    int *offsets = (int *)(ws + w.offsets), *records = (int *)(ws + w.records);
    int4 *items = (int4 *)(ws + w.items), *combos = (int4 *)(ws + w.combos);
    const dim3 bgrid(w.n_wg, ns);
    const size_t bsh = ((size_t)plan.nblk + 1 + (BOXATTN_TUNE_COUNT_DUMP ? kBinThreads : 0)) * sizeof(int);
    ScopedKernelTimer timer(g_prof.ev[kSlotBwdBin], st);     // count + scan + fill
#ifndef BOXATTN_TUNE_BIN_PT
#define BOXATTN_TUNE_BIN_PT 4
#endif
    // four points per thread where the layout allows 16-byte loads of a (query, level)'s points
    const bool pt4 = BOXATTN_TUNE_BIN_PT == 4 && d.P % 4 == 0 && aligned(loc, 16) &&
                     (!WIDE || aligned(w_sp, 16));
    // (fill pass with 4-byte records: one point per thread -- neighbouring lanes then hold
    // neighbouring slots and their stores coalesce: 18.9 us against 23.0 with four)
#define BOXATTN_BIN(FILL_)                                                                         \
    do {                                                                                           \
        if (pt4 && (WIDE || !FILL_))                                                               \
            hipLaunchKernelGGL((bin_kernel<BW, BH, FILL_, WIDE, INTERLEAVE, 4>), bgrid,            \
                               dim3(kBinThreads), bsh, st, loc, w_sp, plan, d.H, d.Lq, d.P,        \
                               w.q_per_wg, w.n_wg, part, subtot, offsets, records, tickets);       \
        else                                                                                       \
            hipLaunchKernelGGL((bin_kernel<BW, BH, FILL_, WIDE, INTERLEAVE, 1>), bgrid,            \
                               dim3(kBinThreads), bsh, st, loc, w_sp, plan, d.H, d.Lq, d.P,        \
                               w.q_per_wg, w.n_wg, part, subtot, offsets, records, tickets);       \
    } while (0)
    int *tickets = (stages & kBinTickets) ? (int *)(ws + w.tickets) : nullptr;     // cleared by the count pass
    if (stages & kBinCount) BOXATTN_BIN(false);
    tickets = nullptr;
#ifndef BOXATTN_TUNE_SCAN_FUSE_WG
#define BOXATTN_TUNE_SCAN_FUSE_WG 48   // up to this many bin workgroups per slice the block scan does kernel A's work too
                                       // (38 workgroups, the 300-query decoders: C3'' fp32 68 -> 61 us; 64, C2: binning 49 -> 61 us)
#endif
    // (big maps, multi-workgroup block scan: fused up to 8 bin workgroups per slice -- the 1 000-query BEV decoder has 1)
    const bool fuse_a = plan.nblk <= kScanThreads ? w.n_wg <= BOXATTN_TUNE_SCAN_FUSE_WG : w.n_wg <= 8;
    if (!(stages & kBinScan)) {
    } else if (!fuse_a)
        hipLaunchKernelGGL(bin_scan_a_kernel,
                           dim3(kScanSub, ns, std::min(64, (plan.nblk + 255) / 256)), dim3(256), 0, st,
                           part, w.n_wg, subtot, plan);
    if (!(stages & kBinScan)) {
    } else if (plan.nblk > kScanThreads) {           // big maps: the block scan over several CUs
        const int nseg = (plan.nblk + kScanThreads - 1) / kScanThreads;
        int4 *tmp = (int4 *)(ws + w.scan_tmp), *segtot = tmp + (size_t)ns * plan.nblk;
        hipLaunchKernelGGL(bin_scan_seg_kernel, dim3(nseg, ns), dim3(kScanThreads), 0, st, subtot,
                           offsets, tmp, segtot, plan, part, fuse_a ? w.n_wg : 0);
        hipLaunchKernelGGL(bin_scan_emit_kernel, dim3(nseg, ns), dim3(kScanThreads), 0, st, offsets,
                           tmp, segtot, items, combos, n_items, plan);
    } else {
        hipLaunchKernelGGL(bin_scan_kernel, dim3(ns), dim3(kScanThreads), 0, st, subtot, offsets,
                           items, combos, n_items, plan, part, fuse_a ? w.n_wg : 0);
    }
    if ((stages & kBinFill) && rec12) {
        if constexpr (WIDE && !INTERLEAVE) {          // 12-byte records (bf16 box attention)
            if (pt4)
                hipLaunchKernelGGL((bin_kernel<BW, BH, true, true, false, 4, true>), bgrid, dim3(kBinThreads), bsh, st,
                                   loc, w_sp, plan, d.H, d.Lq, d.P, w.q_per_wg, w.n_wg, part, subtot, offsets,
                                   records, tickets);
            else
                hipLaunchKernelGGL((bin_kernel<BW, BH, true, true, false, 1, true>), bgrid, dim3(kBinThreads), bsh, st,
                                   loc, w_sp, plan, d.H, d.Lq, d.P, w.q_per_wg, w.n_wg, part, subtot, offsets,
                                   records, tickets);
        }
    } else if (stages & kBinFill) {
        BOXATTN_BIN(true);
    }
#undef BOXATTN_BIN
}

// Can the point-gradient kernel reduce the location gradients to box gradients itself (GRID
// flavour of pointgrad2_kernel: buffered epilogue, one launch row)?
// ------------------------------------------------- dense encoder kernels (boxattn_dense.h)
#ifndef BOXATTN_DENSE_DEFAULT
#define BOXATTN_DENSE_DEFAULT 1       // on unless boxattn_set_option(11, 1) switches them off
#endif
std::atomic<float *> g_dense_dbg{nullptr};      // debugging aid: per-point corner sums (boxattn_set_debug_buffer)
// Encoder case: one query per pixel of packed levels, bf16 storage, C = 32, 2x2 points, <= 4 levels.
inline bool make_dense_plan(const Dims &d, const int64_t *sh, const int64_t *ls, DensePlan &p)
{
    const bool want = opt(kOptDense) == 0 ? BOXATTN_DENSE_DEFAULT != 0 : opt(kOptDense) == 2;
    if (!sh || !ls || !d.valid() || !want || g_variant == 1 || g_variant == 2) return false;
    if (d.Lq != d.S || d.C != 32 || d.P != 4 || d.L > kDenseMaxLevels || d.B < 1 || d.S < 1) return false;
    if ((size_t)d.B * d.Lq * d.H * d.L * d.P >= (1ull << 31)) return false;       // 32-bit point ids
    if (d.n_value() * sizeof(bf16_t) >= kOobOffset) return false;
    p = DensePlan{};
    p.dbg = g_dense_dbg.load();
    p.L = d.L; p.B = d.B; p.Lq = d.Lq; p.S = d.S; p.H = d.H;
    const auto magic = [](int dd) { return dd == 1 ? 0xFFFFFFFFu : (unsigned)((1ull << 32) / (unsigned)dd); };
    p.mag_h = magic(d.H);
    long long next = 0;
    for (int l = 0; l < d.L; ++l) {
        const long long hl = sh[2 * l], wl = sh[2 * l + 1];
        if (hl <= 0 || wl <= 0 || hl > 32000 || wl > 32000 || ls[l] != next) return false;
        next += hl * wl;
        DenseLevel &v = p.lv[l];
        v.H = (int)hl; v.W = (int)wl; v.start = (int)ls[l];
        v.ntx = (int)((wl + kDenseTile - 1) / kDenseTile);
        v.ntiles = v.ntx * (int)((hl + kDenseTile - 1) / kDenseTile);
        if ((long long)v.ntiles * d.B * d.H >= (1 << 24)) return false;      // x * n and the block index < 2^31
        v.n_all = (unsigned)(v.ntiles * d.B);
        v.mag_ntx = magic(v.ntx);
        v.mag_ntiles = magic(v.ntiles);
    }
    if (next != d.S) return false;
    // windows: a tile's queries sit at pixel coordinate (qx + 0.5) r - 0.5 of the sampled level
    // (r = W_l / W_lq); their points lie a quarter box (ref / 4 pixels of the query's level, i.e.
    // ref / 4 * r here) to either side, the predicted offset may move them `jit` quarters further.
    // The levels are staged coarsest first into the workgroup's kDenseSlots pixel slots; what does
    // not fit (a coarse tile's window on a fine level) is not staged.
    const float ref4 = (opt(kOptDenseRef) > 0 ? (float)opt(kOptDenseRef) : 4.0f) / 4.0f;
    const float jit = (opt(kOptDenseJit) > 0 ? (float)opt(kOptDenseJit) : 25.0f) / 10.0f;
    for (int lq = 0; lq < d.L; ++lq) {
        int used = 0;
        for (int l = d.L - 1; l >= 0; --l) {
            DenseWin &w = p.win[lq][l];
            const float rx = (float)p.lv[l].W / (float)p.lv[lq].W, ry = (float)p.lv[l].H / (float)p.lv[lq].H;
            const float mx = rx * ref4 * (1.0f + jit), my = ry * ref4 * (1.0f + jit);
            const int cols = (int)std::ceil(rx * (kDenseTile - 1) + 2 * mx) + 2;
            const int rows = (int)std::ceil(ry * (kDenseTile - 1) + 2 * my) + 2;
            // 16.16 fixed point (a placement heuristic: any rounding will do, tile columns < 2^12)
            w.ax = (int)std::lround(kDenseTile * rx * 65536.0f); w.bx = (int)std::floor((0.5f * rx - 0.5f - mx) * 65536.0f);
            w.ay = (int)std::lround(kDenseTile * ry * 65536.0f); w.by = (int)std::floor((0.5f * ry - 0.5f - my) * 65536.0f);
            if (rx > 8.0f || ry > 8.0f) { w.ax = w.ay = 0; }      // (not staged anyway; keeps tx * ax inside 31 bits)
#ifndef BOXATTN_DENSE_PITCH_PAD
#define BOXATTN_DENSE_PITCH_PAD 2
#endif
            const int pitch = cols + BOXATTN_DENSE_PITCH_PAD;   // slots of neighbouring rows start 2 bank groups apart
            const bool fits = cols <= kDenseWinMax && rows <= kDenseWinMax && used + rows * pitch <= kDenseSlots - 1;   // (the last slot is the forward's zero row)
            w.geo = dense_win_pack(fits ? rows : 0, fits ? cols : 0, pitch, used);
            if (fits) used += rows * pitch;
        }
    }
    return true;
}

inline bool dense_pointgrad_ok(const DensePlan *dp, const void *value, const void *loc, const void *attn,
                               const void *grad_out, const void *grad_loc, const void *grad_attn)
{
    return dp && aligned(value, 16) && aligned(grad_out, 16) && aligned(loc, 8) &&
           aligned(attn, 4) && aligned(grad_loc, 16) && aligned(grad_attn, 16);
}

inline void run_pointgrad_dense(const bf16_t *value, const float *loc, const float *attn,
                                const bf16_t *grad_out, const Dims &d, const DensePlan &dp,
                                float *grad_loc, float *grad_attn, hipStream_t st, const CombineTail *ct,
                                const DenseBin *bin = nullptr)
{
    ScopedKernelTimer timer(g_prof.ev[kSlotBwdPoints], st);
    launch_pointgrad_dense(value, loc, attn, grad_out, dp, grad_loc, grad_attn,
                           (unsigned)(d.n_value() * sizeof(bf16_t)), st, ct ? *ct : CombineTail{},
                           bin ? *bin : DenseBin{});
}

static_assert(kDenseScanSub == kScanSub, "the window-staged fill reads the scan's sub-range tables");
// May the window-staged kernels do the binning (count + records) of this plan?
inline bool dense_fill_ok(const DensePlan *dp, const BinPlan &plan)
{
    // (opt-in: measured at C2 bf16 as fast as the two bin_kernel passes, not faster -- count 17 us + scan 10
    // + 26 us on top of the point-gradient kernel + 6 us for the combine step's own launch against
    // 12 + 12 + 23)
    return dp && opt(kOptDenseFill) == 2 && plan.nblk <= kDenseFillMaxBlocks && plan.nblk <= kScanThreads &&
           plan.L <= kDenseMaxLevels;
}

// Count + scan of the binned backward by the window-staged count kernel: counts[slice][group][block]
// through one global atomic per (workgroup, touched block) into the (zeroed) first kDenseGroups workgroup
// rows of `part`, then the block scan in its few-bin-workgroups form.  The records themselves are
// written by the point-gradient kernel of the backward (cursors: one per (slice, group, block), zeroed
// here).
inline void launch_dense_binning(const float *loc, const Dims &d, const BinPlan &plan, const WsLayout &w,
                                 char *ws, const DensePlan &dp, hipStream_t st)
{
    ScopedKernelTimer timer(g_prof.ev[kSlotBwdBin], st);
    const int ns = d.B * d.H;
    int *part = (int *)(ws + w.part), *subtot = (int *)(ws + w.subtot);
    int *n_items = (int *)(ws + w.n_items), *offsets = (int *)(ws + w.offsets);
    int4 *items = (int4 *)(ws + w.items), *combos = (int4 *)(ws + w.combos);
    const size_t gbytes = (size_t)ns * kDenseGroups * plan.nblk * sizeof(int);
    (void)zero_async(part, gbytes, st);
    (void)zero_async(ws + w.cursor, gbytes, st);
    launch_dense_count(loc, dp, dense_bin(plan, part, nullptr, nullptr, nullptr, nullptr), st);
    hipLaunchKernelGGL(bin_scan_kernel, dim3(ns), dim3(kScanThreads), 0, st, subtot, offsets, items, combos,
                       n_items, plan, part, kDenseGroups);
}

template <typename ST>
bool pointgrad_grid_ok(const Dims &d, const void *value, const void *grad_out, const void *grad_sp)
{
    GatherIdx ix{};
    if (d.P != 4 || (d.L * d.P != 16 && d.L * d.P != 8)) return false;
    if (d.n_value() * sizeof(ST) >= kOobOffset || !gather_idx(d, ix, sizeof(ST))) return false;
    const GatherCfg cfg = gather_cfg<ST>(d, aligned(value, 16) && aligned(grad_out, 16));
    if (cfg.G != 4 && cfg.G != 8) return false;
    const int blocks = gather_blocks(d, ix, kWave / cfg.G);
    return point_split(blocks, (d.L * d.P + cfg.G - 1) / cfg.G) == 1 && aligned(grad_sp, 16);
}

// Point gradients (grad_loc / grad_weight): query-major gathers, independent of how grad_value
// is accumulated.
template <typename ST, int G, bool INST>
void launch_pointgrad(const ST *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                      const float *w_sp, const float *w_lv, const ST *grad_out, const ST *grad_mask,
                      const Dims &d, float *grad_loc, float *grad_sp, float *grad_lv, hipStream_t st,
                      const GridSrc *gs = nullptr, const CombineTail *ct = nullptr,
                      const DensePlan *dp = nullptr)
{
    if constexpr (std::is_same<ST, bf16_t>::value && !INST) {
        if (!gs && dense_pointgrad_ok(dp, value, loc, w_sp, grad_out, grad_loc, grad_sp)) {
            run_pointgrad_dense(value, loc, w_sp, grad_out, d, *dp, grad_loc, grad_sp, st, ct);
            return;
        }
    }
    ScopedKernelTimer timer(g_prof.ev[kSlotBwdPoints], st);
    const size_t n_qh = d.n_qh();
    const size_t vbytes = d.n_value() * sizeof(ST);
    GatherIdx ix{};
    if (vbytes < kOobOffset && gather_idx(d, ix, sizeof(ST))) {
        const GatherCfg cfg = gather_cfg<ST>(d, aligned(value, 16) && aligned(grad_out, 16) &&
                                                (!INST || aligned(grad_mask, 16)));
        const int blocks = gather_blocks(d, ix, kWave / cfg.G);
        // few pairs x many points (instance attention on the mask-decoder grid): one wave
        // per pair, its lane groups over the point tiles; else one lane group per pair
#ifndef BOXATTN_TUNE_PG_WAVE_PER_PAIR
#define BOXATTN_TUNE_PG_WAVE_PER_PAIR 1
#endif
        const int tiles = (d.L * d.P + cfg.G - 1) / cfg.G;
        // the combine step's workers as extra workgroups of this launch (4 single-wave workers each)
        const CombineTail tail = ct ? *ct : CombineTail{};
        const int tail_blocks = tail.workers > 0 ? (tail.workers * tail.plan.n_slices + 3) / 4 : 0;
        const bool wpp = BOXATTN_TUNE_PG_WAVE_PER_PAIR && INST && g_variant != 5 &&
                         blocks < 1024 && tiles >= kWave / cfg.G;
        if (wpp) {
            const int wblocks = ix.head_xcd ? 8 * ceil_div_sz((size_t)d.B * d.Lq, 4)
                                            : ceil_div_sz(n_qh, 4);
            const int split = point_split(wblocks, tiles / (kWave / cfg.G));
#define BOXATTN_PG2W(GG, VV)                                                                         \
hipLaunchKernelGGL((pointgrad2_kernel<ST, GG, INST, GatherUnroll<ST, GG, VV>::value, VV, true>), \
                   dim3(wblocks + tail_blocks, split), dim3(256), 0, st, value, shapes, lsi, loc, w_sp, w_lv,  \
                   grad_out, grad_mask, d.S, d.H, d.L, d.Lq, d.P, grad_loc, grad_sp, grad_lv,    \
                   with_grid(ix, wblocks, split, tiles), (unsigned)vbytes, GridSrc{}, tail);
            BOXATTN_GATHER_DISPATCH(cfg, BOXATTN_PG2W);
#undef BOXATTN_PG2W
        } else {
            const int split = point_split(blocks, tiles);
            if constexpr (!INST) {
                if (gs) {        // boxes in, box gradients out (pointgrad_grid_ok() was checked)
#define BOXATTN_PG2G(GG, VV)                                                                        \
    if constexpr (GG == 4 || GG == 8)                                                               \
        hipLaunchKernelGGL((pointgrad2_kernel<ST, GG, false, GatherUnroll<ST, GG, VV>::value, VV,   \
                                              false, true>),                                        \
                           dim3(blocks + tail_blocks, 1), dim3(256), 0, st, value, shapes, lsi, loc, w_sp, w_lv,  \
                           grad_out, grad_mask, d.S, d.H, d.L, d.Lq, d.P, grad_loc, grad_sp,        \
                           grad_lv, with_grid(ix, blocks, 1, tiles), (unsigned)vbytes, *gs, tail);
                    BOXATTN_GATHER_DISPATCH(cfg, BOXATTN_PG2G);
#undef BOXATTN_PG2G
                    return;
                }
            }
#define BOXATTN_PG2(GG, VV)                                                                   \
hipLaunchKernelGGL((pointgrad2_kernel<ST, GG, INST, GatherUnroll<ST, GG, VV>::value, VV>), \
                   dim3(blocks + tail_blocks, split), dim3(256), 0, st, value, shapes, lsi, loc, w_sp,  \
                   w_lv, grad_out, grad_mask, d.S, d.H, d.L, d.Lq, d.P, grad_loc, grad_sp, \
                   grad_lv, with_grid(ix, blocks, split, tiles), (unsigned)vbytes, GridSrc{}, tail);
            BOXATTN_GATHER_DISPATCH(cfg, BOXATTN_PG2);
#undef BOXATTN_PG2
        }
    } else {
        const int blocks = ceil_div_sz(n_qh, (size_t)(kWave / G) * 4);
        hipLaunchKernelGGL((bwd_fast_kernel<ST, 4, G, INST, false>), dim3(blocks), dim3(256),
                           0, st, value, shapes, lsi, loc, w_sp, w_lv, grad_out, grad_mask,
                           d.S, d.H, d.L, d.Lq, d.P, (float *)nullptr, grad_loc, grad_sp, grad_lv,
                           n_qh);
        if (ct && ct->workers > 0)               // this kernel carries no tail: the combine step on its own
            hipLaunchKernelGGL((combine_partials_kernel<ST, 4 * G>), dim3(ct->workers, ct->plan.n_slices),
                               dim3(64), 0, st, ct->combos, ct->n_items, ct->partials, ct->plan, d.S,
                               d.H, static_cast<ST *>(ct->grad_value));
    }
}

// the matrix-core accumulate of bf16 box attention: binned_accumulate_tr_kernel (32-bit row offsets:
// grad_out below 2 GB) unless switched off, else binned_accumulate_mfma_kernel
template <typename ST, int C>
void launch_accumulate_mfma(const ST *grad_out, const Dims &d, const BinPlan &plan, int wg_per_slice, int ns8,
                            const int4 *items, const int *n_items, const int *records, ST *grad_value,
                            float *partials, hipStream_t st, bool rec12 = false)
{
    if constexpr (std::is_same<ST, bf16_t>::value && (C == 16 || C == 32 || C == 64)) {
        const size_t go_bytes = (size_t)d.B * d.Lq * d.H * C * sizeof(ST);
        if ((rec12 || opt(kOptAccTr) != 1) && accumulate_tr_ok<ST, C>(d)) {      // (12-byte records: this kernel only)
            launch_accumulate_tr(C, grad_out, go_bytes, plan, d.S, d.H, d.Lq, items, n_items, records, grad_value,
                                 partials, wg_per_slice, ns8, rec12, st);
            return;
        }
    }
    if constexpr (std::is_same<ST, float>::value && C == 32) {
        if (g_variant != 11 && f32_mfma_ok(d)) {
            launch_accumulate_f32(grad_out, (size_t)d.B * d.Lq * d.H * C * sizeof(float), plan, d.S, d.H, d.Lq, items,
                                  n_items, records, grad_value, partials, wg_per_slice, ns8, st);
            return;
        }
    }
    hipLaunchKernelGGL((binned_accumulate_mfma_kernel<ST, C>), dim3(wg_per_slice, ns8), dim3(64), 0, st, grad_out,
                       plan, d.S, d.H, d.Lq, items, n_items, records, grad_value, partials);
}

#define BOXATTN_TUNE_ACC_WG_CAP_DEFAULT 1024     // accumulate workgroups per slice at most (see BOXATTN_TUNE_ACC_WG_CAP)
template <typename ST, int G, bool INST>
int run_binned(const ST *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
               const float *w_sp, const float *w_lv, const ST *grad_out, const ST *grad_mask,
               const Dims &d, const BinPlan &plan, const WsLayout &w, char *ws, ST *grad_value,
               float *grad_loc, float *grad_sp, float *grad_lv, bool plan_ready, hipStream_t st,
               const GridSrc *gs = nullptr, const DensePlan *dp = nullptr, int dense_fill = 0, bool rec12 = false)
{
    // rec12: 12-byte bin records (this call's fill pass or the training forward's whose plan this is)
    // dense_fill: 0 no; 1 the window-staged kernels bin (count + scan here, records by the point-gradient
    // kernel); 2 the same, counted and scanned already (by the training forward)
    const int ns = d.B * d.H;
    int *n_items = (int *)(ws + w.n_items);
    int *offsets = (int *)(ws + w.offsets), *records = (int *)(ws + w.records);
    int4 *items = (int4 *)(ws + w.items), *combos = (int4 *)(ws + w.combos);
    float *partials = (float *)(ws + w.partials);
    if constexpr (std::is_same<ST, bf16_t>::value && !INST) {
        if (dense_fill && dp) {
            // [count, scan] -> point gradients + records -> accumulate -> combine
            if (dense_fill == 1) launch_dense_binning(loc, d, plan, w, ws, *dp, st);
            const DenseBin bin = dense_bin(plan, (int *)(ws + w.part), (const int *)(ws + w.subtot), offsets,
                                           (int *)(ws + w.cursor), records);
            run_pointgrad_dense(value, loc, w_sp, grad_out, d, *dp, grad_loc, grad_sp, st, nullptr, &bin);
            const int ns8 = (ns + 7) / 8 * 8;
            const int wg_per_slice = std::min(BOXATTN_TUNE_ACC_WG_CAP_DEFAULT, std::max(1, plan.item_cap));
            {
                ScopedKernelTimer timer(g_prof.ev[kSlotBwdAccum], st);
                launch_accumulate_mfma<ST, 4 * G>(grad_out, d, plan, wg_per_slice, ns8, items, n_items, records,
                                                  grad_value, partials, st);
            }
            {
                ScopedKernelTimer timer(g_prof.ev[kSlotBwdCombine], st);
                hipLaunchKernelGGL((combine_partials_kernel<ST, 4 * G>), dim3(64, ns), dim3(64), 0, st, combos,
                                   n_items, partials, combine_plan(plan), d.S, d.H, grad_value);
            }
            return finish();
        }
    }
    // grad_loc / grad_weight (query-major gathers) do not depend on the binning: where it pays
    // (side_stream_worth) they are launched on the library's helper stream, next to the bin
    // passes and the accumulate kernel.  Fork/join with events, so the caller still sees one
    // in-order stream (also valid under stream capture).
    SideStream side(st, side_stream_worth<ST>(d));
    const bool use_mfma = mfma_accumulate<ST, INST>(d), wide = wide_records<ST, INST>(d);
    if (!plan_ready) launch_binning(wide, !use_mfma, loc, w_sp, d, plan, w, ws, st, kBinAll, rec12);
    // On one stream the point gradients go LAST and carry the combine step's workers as extra
    // workgroups (CombineTail): one launch less, 5-7 us of every step.  With the helper stream
    // (variant 6) they run next to the binning / accumulate kernels and the combine step keeps
    // its own launch.
#ifndef BOXATTN_TUNE_COMBINE_TAIL
#define BOXATTN_TUNE_COMBINE_TAIL 1
#endif
    const bool tail = BOXATTN_TUNE_COMBINE_TAIL && side.stream() == st;
    if (!tail)
        launch_pointgrad<ST, G, INST>(value, shapes, lsi, loc, w_sp, w_lv, grad_out, grad_mask, d,
                                      grad_loc, grad_sp, grad_lv, side.stream(), gs, nullptr, dp);
    // One single-wave workgroup per potential work item (item_cap is the host-side bound; the
    // real count lives on the device, surplus workgroups exit at once); the hardware dispatcher
    // hands them out as waves retire -- dynamic load balancing without a work-queue atomic (a
    // persistent-waves version with a software queue was 15 % slower).  The
    // kernel maps workgroups to (slice, worker) itself (XCD affinity), hence the 8-aligned grid.
    const int ns8 = (ns + 7) / 8 * 8;
#ifndef BOXATTN_TUNE_ACC_WG_CAP
#define BOXATTN_TUNE_ACC_WG_CAP 1024   // per slice; beyond that a workgroup takes several items (its next one in flight);
                                       // 256 / 512 / 1024 / 3072 / 6144: C5 99 / 85 / 83 / 93 / 111 us, C2 67 / 54 / 54 / 55 / 54
#endif
    const int wg_per_slice = BOXATTN_TUNE_ACC_WG_CAP
                                 ? std::min(BOXATTN_TUNE_ACC_WG_CAP, std::max(1, plan.item_cap))
                                 : std::max(1, plan.item_cap);
    {
        ScopedKernelTimer timer(g_prof.ev[kSlotBwdAccum], st);
        // records per lane and round: one is best for every flavour now that the bin passes
        // interleave the queries (two were better for bf16 box attention before that)
#ifndef BOXATTN_TUNE_RPL_BF16
#define BOXATTN_TUNE_RPL_BF16 1
#endif
#ifndef BOXATTN_TUNE_RPL_F32
#define BOXATTN_TUNE_RPL_F32 1
#endif
        constexpr int kRpl = INST ? 1 : (sizeof(ST) == 2 ? BOXATTN_TUNE_RPL_BF16 : BOXATTN_TUNE_RPL_F32);
        constexpr bool kMfmaBuilt = !INST && (std::is_same<ST, bf16_t>::value ||
                                              (std::is_same<ST, float>::value && 4 * G == 32));
        bool done = false;
        if constexpr (kMfmaBuilt) {
            if (use_mfma) {
                launch_accumulate_mfma<ST, 4 * G>(grad_out, d, plan, wg_per_slice, ns8, items, n_items, records,
                                                  grad_value, partials, st, rec12);
                done = true;
            }
        }
        if (!done) {
            if constexpr (!INST && sizeof(ST) == 4 && BOXATTN_TUNE_WIDE_F32) {
                hipLaunchKernelGGL((binned_accumulate_kernel<ST, 4 * G, INST, kRpl, true>),
                                   dim3(wg_per_slice, ns8), dim3(64), 0, st, grad_out, grad_mask, loc,
                                   w_sp, w_lv, plan, d.S, d.H, d.Lq, d.P, offsets, items, n_items,
                                   records, grad_value, partials);
            } else if constexpr (!std::is_same<ST, bf16_t>::value || INST) {
                hipLaunchKernelGGL((binned_accumulate_kernel<ST, 4 * G, INST, kRpl, false>),
                                   dim3(wg_per_slice, ns8), dim3(64), 0, st, grad_out, grad_mask, loc,
                                   w_sp, w_lv, plan, d.S, d.H, d.Lq, d.P, offsets, items, n_items,
                                   records, grad_value, partials);
            } else {
                return (int)hipErrorInvalidValue;    // bf16 box attention always has the MFMA kernel
            }
        }
    }
    if (tail) {
        CombineTail ct{combos, n_items, partials, grad_value, combine_plan(plan), 64};
        launch_pointgrad<ST, G, INST>(value, shapes, lsi, loc, w_sp, w_lv, grad_out, grad_mask, d,
                                      grad_loc, grad_sp, grad_lv, st, gs, &ct, dp);
    } else {
        ScopedKernelTimer timer(g_prof.ev[kSlotBwdCombine], st);
        hipLaunchKernelGGL((combine_partials_kernel<ST, 4 * G>), dim3(64, ns), dim3(64), 0, st, combos,
                           n_items, partials, combine_plan(plan), d.S, d.H, grad_value);
    }
    side.join();
    return finish();
}

// ------------------------------------------------- query-grid backward (boxattn_qgrid.h)
struct QgLayout { size_t bbox, cand, cand_w, partials, total; };

// Encoder case (Lq == S, packed levels), BoxeR's head geometry, bf16 storage, 2x2 grids.
inline bool make_qg_plan(const Dims &d, const int64_t *sh, const int64_t *ls, QgPlan &p)
{
    if (!sh || !ls || !d.valid() || g_variant == 1 || g_variant == 2 || g_variant == 3 || g_variant == 8)
        return false;
    // (measured slower than the binned backward so far -- 21 + 118 + 10 us against 50 + 57 + 8 at
    // BoxeR-R50 shapes, DESIGN.md 4.2 -- so the path is opt-in: option kOptQgBwd = 1)
    if (opt(kOptQgBwd) != 1) return false;
    if (d.Lq != d.S || d.C != 32 || d.P != 4 || d.L > kQgMaxLevels) return false;
    if ((size_t)d.B * d.Lq * d.H * d.C >= (1ull << 31)) return false;       // 32-bit row ids
    long long next = 0, tiles = 0, items = 0, parts = 0;
    const int target = opt(kOptQgTarget) > 0 ? opt(kOptQgTarget) : 512;     // records per item
    p.L = d.L;
    for (int l = 0; l < d.L; ++l) {
        const long long hl = sh[2 * l], wl = sh[2 * l + 1];
        if (hl <= 0 || wl <= 0 || hl > 32000 || wl > 32000 || ls[l] != next) return false;
        next += hl * wl;
        QgLevel &v = p.lv[l];
        v.H = (int)hl; v.W = (int)wl; v.start = (int)ls[l];
        v.ntx4 = (int)((wl + 3) / 4);
        v.tile0 = (int)tiles;
        tiles += (long long)v.ntx4 * ((hl + 3) / 4);
        const long long nbx = (wl + 7) / 8, nby = (hl + 3) / 4;
        // expected records per block: every query puts P points on every level, a record per
        // block its footprint meets (~1.3)
        const double per_block = (double)d.Lq * d.P * 1.3 / (double)(nbx * nby);
        v.gx = v.gy = 1;
        v.cpb = 1;
        if (per_block * 4 <= target * 1.5) v.gx = v.gy = 2;
        else if (per_block * 2 <= target * 1.25) v.gx = 2;
        else v.cpb = (int)std::max(1.0, std::min(4096.0, per_block / target + 0.5));
        v.ngx = (int)((nbx + v.gx - 1) / v.gx);
        v.ngy = (int)((nby + v.gy - 1) / v.gy);
        v.item0 = (int)items;
        items += (long long)v.ngx * v.ngy * v.cpb;
        v.part0 = v.cpb > 1 ? (int)parts : -1;
        if (v.cpb > 1) parts += (long long)v.ngx * v.ngy * v.cpb;
    }
    if (next != d.S || tiles > 65535 || items > (1 << 20) || parts > (1 << 20)) return false;
    for (int l = 0; l < d.L; ++l)
        p.lv[l].cpb = std::min<long long>(p.lv[l].cpb, tiles);
    p.n_tiles4 = (int)tiles;
    p.n_items = (int)items;
    p.n_parts = (int)parts;
    p.ablate = opt(kOptQgAblate);
    return true;
}

inline QgLayout qg_layout(const Dims &d, const QgPlan &p)
{
    QgLayout w;
    const size_t ns = (size_t)d.B * d.H;
    size_t o = 0;
    w.bbox = o;     o += align_up(ns * d.L * (size_t)p.n_tiles4 * sizeof(uint2));
    w.cand = o;     o += align_up(ns * d.L * (size_t)p.n_tiles4 * 64 * sizeof(int4));
    w.cand_w = o;   o += align_up(ns * d.L * (size_t)p.n_tiles4 * 64 * sizeof(float));
    w.partials = o; o += align_up(ns * (size_t)p.n_parts * 32 * d.C * sizeof(float));
    w.total = o;
    return w;
}

inline int launch_qg_prep(const float *loc, const float *attn, const Dims &d, const QgPlan &p,
                          const QgLayout &w, char *ws, hipStream_t st)
{
    ScopedKernelTimer timer(g_prof.ev[kSlotBwdPrep], st);
    const unsigned waves = (unsigned)((size_t)d.B * d.H * p.n_tiles4);
    uint2 *bbox = (uint2 *)(ws + w.bbox);
    int4 *cand = (int4 *)(ws + w.cand);
    float *cand_w = (float *)(ws + w.cand_w);
#define BOXATTN_QG_BBOX(LV_)                                                                     \
    case LV_:                                                                                    \
        hipLaunchKernelGGL((qg_prep_kernel<LV_>), dim3((waves + 3) / 4), dim3(256), 0, st, loc,  \
                           attn, p, d.B, d.H, d.Lq, bbox, cand, cand_w);                         \
        break;
    switch (d.L) {
        BOXATTN_QG_BBOX(1) BOXATTN_QG_BBOX(2) BOXATTN_QG_BBOX(3) BOXATTN_QG_BBOX(4)
        BOXATTN_QG_BBOX(5) BOXATTN_QG_BBOX(6) BOXATTN_QG_BBOX(7) BOXATTN_QG_BBOX(8)
    }
#undef BOXATTN_QG_BBOX
    return finish();
}

// grad_value of bf16 box attention on a query grid: [tile boxes] -> accumulate -> combine
inline int run_qgrid(const bf16_t *grad_out, const float *loc, const float *attn, const Dims &d,
                     const QgPlan &p, const QgLayout &w, char *ws, bf16_t *grad_value,
                     bool boxes_ready, hipStream_t st)
{
    int rc = 0;
    if (!boxes_ready && (rc = launch_qg_prep(loc, attn, d, p, w, ws, st))) return rc;
    const int ns = d.B * d.H;
    uint2 *bbox = (uint2 *)(ws + w.bbox);
    float *partials = (float *)(ws + w.partials);
    {
        ScopedKernelTimer timer(g_prof.ev[kSlotBwdAccum], st);
        // persistent waves: 12 per CU (registers / LDS of the kernel), 32 CUs per XCD
        const int per_xcd_items = ((ns + 7) / 8) * p.n_items;
        const int waves = std::max(1, std::min(opt(kOptQgWaves) > 0 ? opt(kOptQgWaves) : 12 * 32, per_xcd_items));
        hipLaunchKernelGGL((qg_accumulate_kernel<32>), dim3(8 * waves), dim3(64), 0, st, grad_out,
                           bbox, (const int4 *)(ws + w.cand), (const float *)(ws + w.cand_w), p, ns,
                           d.S, d.H, grad_value, partials);
    }
    int split_groups = 0;
    for (int l = 0; l < p.L; ++l)
        if (p.lv[l].cpb > 1) split_groups += p.lv[l].ngx * p.lv[l].ngy;
    if (split_groups) {
        ScopedKernelTimer timer(g_prof.ev[kSlotBwdCombine], st);
        hipLaunchKernelGGL((qg_combine_kernel<32>), dim3(split_groups, ns), dim3(128), 0, st,
                           partials, p, d.S, d.H, grad_value);
    }
    return finish();
}

// Backward with a caller-provided workspace; falls back to the atomic kernels when the
// binned algorithm does not apply.
template <typename ST, bool INST>
int launch_bwd_ws(const ST *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                  const float *w_sp, const float *w_lv, const ST *grad_out, const ST *grad_mask,
                  const Dims &d, ST *grad_value, float *grad_loc, float *grad_sp, float *grad_lv,
                  const int64_t *shapes_host, const int64_t *lsi_host, void *workspace,
                  size_t workspace_bytes, int plan_kind, hipStream_t st, const GridSrc *gs = nullptr)
{
    // plan_kind: what the training forward left in the workspace -- 0 nothing, 1 the binning
    // plan, 2 the query-grid tile boxes (the value *_fwd_train_* returned in *plan_built)
    constexpr bool kBf16 = std::is_same<ST, bf16_t>::value;
    if (!d.valid()) return (int)hipErrorInvalidValue;
    BinPlan plan;
    const size_t nv = d.n_value();
    bool binned = (g_variant == 0 || g_variant >= 3) && workspace && nv && d.n_qh() &&
                  make_plan(d, shapes_host, lsi_host, plan) &&
                  fast_ok<ST>(d, value, loc, grad_out,
                              INST ? (const void *)grad_mask : (const void *)grad_out, grad_loc) &&
                  aligned(workspace, 256) && aligned(grad_value, 16);
    if (gs && (!binned || INST || !pointgrad_grid_ok<ST>(d, value, grad_out, grad_sp)))
        return kNotEligible;                 // the caller falls back to the grid tensor's own kernels
    if constexpr (kBf16 && !INST) {      // encoder case: no global binning (boxattn_qgrid.h)
        QgPlan qp;
        if (binned && plan_kind != 1 && plan_kind != 4 && aligned(grad_out, 16) && aligned(loc, 16) &&
            make_qg_plan(d, shapes_host, lsi_host, qp)) {
            const QgLayout qw = qg_layout(d, qp);
            if (workspace_bytes >= qw.total) {
                if (!shapes || !lsi || !loc || !w_sp || !grad_out || !grad_loc || !grad_sp ||
                    !grad_value || !value)
                    return (int)hipErrorInvalidValue;
                launch_pointgrad<ST, 8, false>(value, shapes, lsi, loc, w_sp, w_lv, grad_out,
                                               grad_mask, d, grad_loc, grad_sp, grad_lv, st, gs);
                return run_qgrid(grad_out, loc, w_sp, d, qp, qw, (char *)workspace, grad_value,
                                 plan_kind == 2, st);
            }
        }
        if (plan_kind == 2) plan_kind = 0;     // boxes in the workspace, but the binned path runs
    }
    const bool plan_ready = plan_kind == 1 || plan_kind == 4;   // 4: with 12-byte records (kind 3 -- counted by the window-staged kernels -- see below)
    WsLayout w{};
    if (binned) {
        w = ws_layout(d, plan, wide_workspace(kBf16, d));
        binned = workspace_bytes >= w.total;
    }
    // the boxes-in entry points never take the atomic fallback (its grad_loc slot is aliased to
    // grad_offsets there, a smaller buffer): an undersized workspace is "not eligible"
    if (gs && !binned) return kNotEligible;
    if (!binned) {
        // (a plan the forward built is simply not used when the backward's own checks -- e.g. an
        // unaligned grad_out view -- rule the binned path out: the atomic path needs no plan)
        if (g_variant == 3) return (int)hipErrorInvalidValue;
        float *acc = nullptr;
        if constexpr (kBf16) {
            if (!workspace || workspace_bytes < nv * sizeof(float)) return (int)hipErrorInvalidValue;
            acc = (float *)workspace;
        } else {
            acc = grad_value;
        }
        return launch_bwd<ST, INST>(value, shapes, lsi, loc, w_sp, w_lv, grad_out, grad_mask, d,
                                    grad_value, grad_loc, grad_sp, grad_lv, acc, st);
    }
    if (!shapes || !lsi || !loc || !w_sp || !grad_out || !grad_loc || !grad_sp || !grad_value ||
        !value || (INST && (!w_lv || !grad_mask || !grad_lv)))
        return (int)hipErrorInvalidValue;
    char *ws = (char *)workspace;
    int rc = 0;
    DensePlan dense;
    const DensePlan *dp = kBf16 && !INST && make_dense_plan(d, shapes_host, lsi_host, dense) ? &dense : nullptr;
    // the window-staged kernels as the binning passes: whenever they compute the point gradients and the
    // matrix-core accumulate reads wide records (a forward-built classic plan, kind 1, is used as it is)
    int dense_fill = 0;
    if (!gs && !plan_ready && dense_fill_ok(dp, plan) && mfma_accumulate<ST, INST>(d) &&
        dense_pointgrad_ok(dp, value, loc, w_sp, grad_out, grad_loc, grad_sp))
        dense_fill = plan_kind == 3 ? 2 : 1;
    // 12-byte records: as the plan's forward wrote them, or this call's own fill pass
    const bool rec12 = plan_kind == 4 || (!plan_ready && !dense_fill && rec12_ok<ST, INST>(d, plan));
    switch (fast_group(d)) {
#define BOXATTN_BINNED_CASE(GG)                                                                 \
    case GG:                                                                                    \
        rc = run_binned<ST, GG, INST>(value, shapes, lsi, loc, w_sp, w_lv, grad_out, grad_mask, \
                                      d, plan, w, ws, grad_value, grad_loc, grad_sp, grad_lv,   \
                                      plan_ready, st, gs, dp, dense_fill, rec12);               \
        break;
        BOXATTN_BINNED_CASE(4)
        BOXATTN_BINNED_CASE(8)
        BOXATTN_BINNED_CASE(16)
#undef BOXATTN_BINNED_CASE
    }
    return rc;
}

// Training forward: the forward kernel on `stream`, and on the helper stream (concurrently) the
// binning passes of the backward, which only depend on the sampling locations.  The workspace
// then carries the plan to the *_bwd_ws_* call (plan_ready = 1).
template <typename ST, bool INST>
int launch_fwd_train(const ST *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                     const float *w_sp, const float *w_lv, const Dims &d, ST *out, ST *mask,
                     const int64_t *shapes_host, const int64_t *lsi_host, void *workspace,
                     size_t workspace_bytes, int *plan_built, hipStream_t st)
{
    if (plan_built) *plan_built = 0;
    BinPlan plan;
    bool ok = (g_variant == 0 || g_variant >= 3) && workspace && d.valid() &&
              d.n_value() && d.n_qh() && make_plan(d, shapes_host, lsi_host, plan) &&
              aligned(workspace, 256) &&
              // what the backward will check and the forward can already see (its other operands,
              // grad_out / grad_value, are re-checked there; an ineligible backward ignores the plan)
              fast_ok<ST>(d, value, loc, out, INST ? (const void *)mask : (const void *)out, out);
    WsLayout w{};
    if (ok) {
        w = ws_layout(d, plan, wide_workspace(std::is_same<ST, bf16_t>::value, d));
        ok = workspace_bytes >= w.total;
    }
    if (!ok)
        return launch_fwd<ST, INST>(value, shapes, lsi, loc, w_sp, w_lv, d, out, mask, st,
                                    shapes_host, lsi_host);
    if constexpr (std::is_same<ST, bf16_t>::value && !INST) {      // query grid: tile boxes only
        QgPlan qp;
        if (aligned(loc, 16) && make_qg_plan(d, shapes_host, lsi_host, qp)) {
            const QgLayout qw = qg_layout(d, qp);
            if (workspace_bytes >= qw.total) {
                int rc = launch_fwd<ST, INST>(value, shapes, lsi, loc, w_sp, w_lv, d, out, mask, st,
                                              shapes_host, lsi_host);
                if (rc == 0) rc = launch_qg_prep(loc, w_sp, d, qp, qw, (char *)workspace, st);
                if (rc == 0 && plan_built) *plan_built = 2;
                return rc;
            }
        }
    }
    if constexpr (std::is_same<ST, bf16_t>::value && !INST) {      // window-staged kernels: count + scan only
        DensePlan dense;
        if (make_dense_plan(d, shapes_host, lsi_host, dense) && dense_fill_ok(&dense, plan) &&
            mfma_accumulate<ST, INST>(d) && aligned(loc, 8)) {
            launch_dense_binning(loc, d, plan, w, (char *)workspace, dense, st);
            const int rc = launch_fwd<ST, INST>(value, shapes, lsi, loc, w_sp, w_lv, d, out, mask, st,
                                                shapes_host, lsi_host);
            if (rc == 0 && plan_built) *plan_built = 3;
            return rc;
        }
    }
    SideStream side(st, side_stream_worth<ST>(d));
    const bool wide = wide_records<ST, INST>(d), inter = !mfma_accumulate<ST, INST>(d);
    const bool rec12 = wide && !inter && rec12_ok<ST, INST>(d, plan);
    if (side.stream() == st && scan_tail_ok(plan, w)) {
        // count -> forward kernel + the scans as extra workgroups of its launch -> fill
        launch_binning(wide, inter, loc, w_sp, d, plan, w, (char *)workspace, st, kBinCount | kBinTickets);
        const ScanTail tail = scan_tail(plan, w, (char *)workspace);
        bool taken = false;
        const int rc = launch_fwd<ST, INST>(value, shapes, lsi, loc, w_sp, w_lv, d, out, mask, st,
                                            shapes_host, lsi_host, &tail, &taken);
        launch_binning(wide, inter, loc, w_sp, d, plan, w, (char *)workspace, st,
                       taken ? kBinFill : (kBinScan | kBinFill), rec12);
        if (rc == 0 && plan_built) *plan_built = rec12 ? 4 : 1;
        return rc;
    }
    launch_binning(wide, inter, loc, w_sp, d, plan, w, (char *)workspace, side.stream(), kBinAll, rec12);
    const int rc = launch_fwd<ST, INST>(value, shapes, lsi, loc, w_sp, w_lv, d, out, mask, st,
                                        shapes_host, lsi_host);
    side.join();
    if (rc == 0 && plan_built) *plan_built = rec12 ? 4 : 1;
    return rc;
}

}  // namespace

extern "C" {

int boxattn_fwd_train_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                          const float *loc, const float *attn, int B, int S, int H, int C, int L,
                          int Lq, int P, float *out, const int64_t *shapes_host,
                          const int64_t *lsi_host, void *workspace, size_t workspace_bytes,
                          int *plan_built, void *stream)
{
    return launch_fwd_train<float, false>(value, shapes, lsi, loc, attn, nullptr,
                                          Dims{B, S, H, C, L, Lq, P}, out, nullptr, shapes_host,
                                          lsi_host, workspace, workspace_bytes, plan_built,
                                          (hipStream_t)stream);
}
int boxattn_fwd_train_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                           const float *loc, const float *attn, int B, int S, int H, int C, int L,
                           int Lq, int P, uint16_t *out, const int64_t *shapes_host,
                           const int64_t *lsi_host, void *workspace, size_t workspace_bytes,
                           int *plan_built, void *stream)
{
    return launch_fwd_train<bf16_t, false>(value, shapes, lsi, loc, attn, nullptr,
                                           Dims{B, S, H, C, L, Lq, P}, out, nullptr, shapes_host,
                                           lsi_host, workspace, workspace_bytes, plan_built,
                                           (hipStream_t)stream);
}
int instattn_fwd_train_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                           const float *loc, const float *spatial_w, const float *level_w, int B,
                           int S, int H, int C, int L, int Lq, int P, float *out, float *mask_out,
                           const int64_t *shapes_host, const int64_t *lsi_host, void *workspace,
                           size_t workspace_bytes, int *plan_built, void *stream)
{
    return launch_fwd_train<float, true>(value, shapes, lsi, loc, spatial_w, level_w,
                                         Dims{B, S, H, C, L, Lq, P}, out, mask_out, shapes_host,
                                         lsi_host, workspace, workspace_bytes, plan_built,
                                         (hipStream_t)stream);
}
int instattn_fwd_train_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                            const float *loc, const float *spatial_w, const float *level_w, int B,
                            int S, int H, int C, int L, int Lq, int P, uint16_t *out,
                            uint16_t *mask_out, const int64_t *shapes_host,
                            const int64_t *lsi_host, void *workspace, size_t workspace_bytes,
                            int *plan_built, void *stream)
{
    return launch_fwd_train<bf16_t, true>(value, shapes, lsi, loc, spatial_w, level_w,
                                          Dims{B, S, H, C, L, Lq, P}, out, mask_out, shapes_host,
                                          lsi_host, workspace, workspace_bytes, plan_built,
                                          (hipStream_t)stream);
}

size_t boxattn_bwd_workspace_bytes(int is_bf16, int B, int S, int H, int C, int L, int Lq, int P,
                                   const int64_t *shapes_host, const int64_t *lsi_host)
{
    const Dims d{B, S, H, C, L, Lq, P};
    if (!d.valid()) return 0;
    const size_t fallback = is_bf16 ? align_up(d.n_value() * sizeof(float)) : 0;
    BinPlan plan;
    if (!make_plan(d, shapes_host, lsi_host, plan)) return fallback;
    size_t need = std::max(fallback, ws_layout(d, plan, wide_workspace(is_bf16 != 0, d)).total);
    QgPlan qp;
    if (is_bf16 && make_qg_plan(d, shapes_host, lsi_host, qp))
        need = std::max(need, qg_layout(d, qp).total);
    return need;
}

int boxattn_bwd_ws_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                       const float *loc, const float *attn, const float *grad_out, int B, int S,
                       int H, int C, int L, int Lq, int P, float *grad_value, float *grad_loc,
                       float *grad_attn, const int64_t *shapes_host, const int64_t *lsi_host,
                       void *workspace, size_t workspace_bytes, int plan_ready, void *stream)
{
    return launch_bwd_ws<float, false>(value, shapes, lsi, loc, attn, nullptr, grad_out, nullptr,
                                       Dims{B, S, H, C, L, Lq, P}, grad_value, grad_loc,
                                       grad_attn, nullptr, shapes_host, lsi_host, workspace,
                                       workspace_bytes, plan_ready, (hipStream_t)stream);
}
int boxattn_bwd_ws_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                        const float *loc, const float *attn, const uint16_t *grad_out, int B,
                        int S, int H, int C, int L, int Lq, int P, uint16_t *grad_value,
                        float *grad_loc, float *grad_attn, const int64_t *shapes_host,
                        const int64_t *lsi_host, void *workspace, size_t workspace_bytes,
                        int plan_ready, void *stream)
{
    return launch_bwd_ws<bf16_t, false>(value, shapes, lsi, loc, attn, nullptr, grad_out, nullptr,
                                        Dims{B, S, H, C, L, Lq, P}, grad_value, grad_loc,
                                        grad_attn, nullptr, shapes_host, lsi_host, workspace,
                                        workspace_bytes, plan_ready, (hipStream_t)stream);
}
int instattn_bwd_ws_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                        const float *loc, const float *spatial_w, const float *level_w,
                        const float *grad_out, const float *grad_mask, int B, int S, int H, int C,
                        int L, int Lq, int P, float *grad_value, float *grad_loc,
                        float *grad_spatial_w, float *grad_level_w, const int64_t *shapes_host,
                        const int64_t *lsi_host, void *workspace, size_t workspace_bytes,
                        int plan_ready, void *stream)
{
    return launch_bwd_ws<float, true>(value, shapes, lsi, loc, spatial_w, level_w, grad_out,
                                      grad_mask, Dims{B, S, H, C, L, Lq, P}, grad_value, grad_loc,
                                      grad_spatial_w, grad_level_w, shapes_host, lsi_host,
                                      workspace, workspace_bytes, plan_ready, (hipStream_t)stream);
}
int instattn_bwd_ws_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                         const float *loc, const float *spatial_w, const float *level_w,
                         const uint16_t *grad_out, const uint16_t *grad_mask, int B, int S, int H,
                         int C, int L, int Lq, int P, uint16_t *grad_value, float *grad_loc,
                         float *grad_spatial_w, float *grad_level_w, const int64_t *shapes_host,
                         const int64_t *lsi_host, void *workspace, size_t workspace_bytes,
                         int plan_ready, void *stream)
{
    return launch_bwd_ws<bf16_t, true>(value, shapes, lsi, loc, spatial_w, level_w, grad_out,
                                       grad_mask, Dims{B, S, H, C, L, Lq, P}, grad_value, grad_loc,
                                       grad_spatial_w, grad_level_w, shapes_host, lsi_host,
                                       workspace, workspace_bytes, plan_ready, (hipStream_t)stream);
}


int boxattn_abi_version(void) { return BOXATTN_ABI_VERSION; }

const char *boxattn_build_info(void)
{
    return "boxattn gfx950 (CDNA4, wave64) | hipcc " __VERSION__
           " | kernels: generic{f32,f64,bf16}, gather{f32 4ch/lane, bf16 8ch/lane} C={16,32,64}, "
           "binned-bwd{f32,bf16; bf16 accumulate on MFMA}, window-staged encoder point gradients{bf16}, "
           "box-grid{f32} | abi 6";
}

int boxattn_profile_begin(void)
{
    std::lock_guard<std::mutex> g(g_prof.mu);
    for (auto &v : g_prof.ev) drain(v, nullptr, nullptr);
    g_prof.on = true;
    return 0;
}

int boxattn_profile_end(double *ms_sum, int *launches)
{
    g_prof.on = false;
    std::lock_guard<std::mutex> g(g_prof.mu);
    for (int i = 0; i < kNumSlots; ++i)
        drain(g_prof.ev[i], ms_sum ? ms_sum + i : nullptr, launches ? launches + i : nullptr);
    return 0;
}

int boxattn_set_variant(int variant) { return g_variant.exchange(variant); }

// debugging aid, not part of the documented ABI: a device buffer of 8 floats per sample point that the
// dense point-gradient kernel fills with its corner sums (nullptr: off)
void boxattn_set_debug_buffer(float *p) { g_dense_dbg.store(p); }

int boxattn_set_option(int key, int value)
{
    if (key < 0 || key >= kNumOpts) return -1;
    return g_opt[key].exchange(value);
}

#define DIMS Dims{B, S, H, C, L, Lq, P}
#define ST_ (hipStream_t) stream

int boxattn_fwd_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                    const float *loc, const float *attn, int B, int S, int H, int C, int L,
                    int Lq, int P, float *out, void *stream)
{
    return launch_fwd<float, false>(value, shapes, lsi, loc, attn, nullptr, DIMS, out, nullptr,
                                    ST_);
}
int boxattn_fwd_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                    const double *loc, const double *attn, int B, int S, int H, int C, int L,
                    int Lq, int P, double *out, void *stream)
{
    return launch_fwd<double, false>(value, shapes, lsi, loc, attn, nullptr, DIMS, out,
                                     nullptr, ST_);
}
int boxattn_fwd_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                     const float *loc, const float *attn, int B, int S, int H, int C, int L,
                     int Lq, int P, uint16_t *out, void *stream)
{
    return launch_fwd<bf16_t, false>(value, shapes, lsi, loc, attn, nullptr, DIMS, out,
                                     nullptr, ST_);
}

int boxattn_fwd_hl_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                       const float *loc, const float *attn, int B, int S, int H, int C, int L,
                       int Lq, int P, float *out, const int64_t *shapes_host,
                       const int64_t *lsi_host, void *stream)
{
    return launch_fwd<float, false>(value, shapes, lsi, loc, attn, nullptr, DIMS, out, nullptr,
                                    ST_, shapes_host, lsi_host);
}
int boxattn_fwd_hl_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                        const float *loc, const float *attn, int B, int S, int H, int C, int L,
                        int Lq, int P, uint16_t *out, const int64_t *shapes_host,
                        const int64_t *lsi_host, void *stream)
{
    return launch_fwd<bf16_t, false>(value, shapes, lsi, loc, attn, nullptr, DIMS, out, nullptr,
                                     ST_, shapes_host, lsi_host);
}

int boxattn_bwd_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                    const float *loc, const float *attn, const float *grad_out, int B, int S,
                    int H, int C, int L, int Lq, int P, float *grad_value, float *grad_loc,
                    float *grad_attn, void *stream)
{
    return launch_bwd<float, false>(value, shapes, lsi, loc, attn, nullptr, grad_out, nullptr,
                                    DIMS, grad_value, grad_loc, grad_attn, nullptr, grad_value,
                                    ST_);
}
int boxattn_bwd_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                    const double *loc, const double *attn, const double *grad_out, int B,
                    int S, int H, int C, int L, int Lq, int P, double *grad_value,
                    double *grad_loc, double *grad_attn, void *stream)
{
    return launch_bwd<double, false>(value, shapes, lsi, loc, attn, nullptr, grad_out, nullptr,
                                     DIMS, grad_value, grad_loc, grad_attn, nullptr,
                                     grad_value, ST_);
}
int boxattn_bwd_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                     const float *loc, const float *attn, const uint16_t *grad_out, int B,
                     int S, int H, int C, int L, int Lq, int P, uint16_t *grad_value,
                     float *grad_loc, float *grad_attn, float *grad_value_ws, void *stream)
{
    return launch_bwd<bf16_t, false>(value, shapes, lsi, loc, attn, nullptr, grad_out, nullptr,
                                     DIMS, grad_value, grad_loc, grad_attn, nullptr,
                                     grad_value_ws, ST_);
}

int instattn_fwd_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                     const float *loc, const float *spatial_w, const float *level_w, int B,
                     int S, int H, int C, int L, int Lq, int P, float *out, float *mask_out,
                     void *stream)
{
    return launch_fwd<float, true>(value, shapes, lsi, loc, spatial_w, level_w, DIMS, out,
                                   mask_out, ST_);
}
int instattn_fwd_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                     const double *loc, const double *spatial_w, const double *level_w, int B,
                     int S, int H, int C, int L, int Lq, int P, double *out, double *mask_out,
                     void *stream)
{
    return launch_fwd<double, true>(value, shapes, lsi, loc, spatial_w, level_w, DIMS, out,
                                    mask_out, ST_);
}
int instattn_fwd_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                      const float *loc, const float *spatial_w, const float *level_w, int B,
                      int S, int H, int C, int L, int Lq, int P, uint16_t *out,
                      uint16_t *mask_out, void *stream)
{
    return launch_fwd<bf16_t, true>(value, shapes, lsi, loc, spatial_w, level_w, DIMS, out,
                                    mask_out, ST_);
}

int instattn_bwd_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                     const float *loc, const float *spatial_w, const float *level_w,
                     const float *grad_out, const float *grad_mask, int B, int S, int H, int C,
                     int L, int Lq, int P, float *grad_value, float *grad_loc,
                     float *grad_spatial_w, float *grad_level_w, void *stream)
{
    return launch_bwd<float, true>(value, shapes, lsi, loc, spatial_w, level_w, grad_out,
                                   grad_mask, DIMS, grad_value, grad_loc, grad_spatial_w,
                                   grad_level_w, grad_value, ST_);
}
int instattn_bwd_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                     const double *loc, const double *spatial_w, const double *level_w,
                     const double *grad_out, const double *grad_mask, int B, int S, int H,
                     int C, int L, int Lq, int P, double *grad_value, double *grad_loc,
                     double *grad_spatial_w, double *grad_level_w, void *stream)
{
    return launch_bwd<double, true>(value, shapes, lsi, loc, spatial_w, level_w, grad_out,
                                    grad_mask, DIMS, grad_value, grad_loc, grad_spatial_w,
                                    grad_level_w, grad_value, ST_);
}
int instattn_bwd_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                      const float *loc, const float *spatial_w, const float *level_w,
                      const uint16_t *grad_out, const uint16_t *grad_mask, int B, int S, int H,
                      int C, int L, int Lq, int P, uint16_t *grad_value, float *grad_loc,
                      float *grad_spatial_w, float *grad_level_w, float *grad_value_ws,
                      void *stream)
{
    return launch_bwd<bf16_t, true>(value, shapes, lsi, loc, spatial_w, level_w, grad_out,
                                    grad_mask, DIMS, grad_value, grad_loc, grad_spatial_w,
                                    grad_level_w, grad_value_ws, ST_);
}


// ---- reference windows + offsets -> sampling grid ------------------------------------------
static int grid_dims(int ref_dim, int ref_per_head, int V, int angle_mode, int B, int Lq, int H,
                     int L, int P, GridDims &d)
{
    if (B < 0 || Lq < 0 || H <= 0 || L <= 0 || P <= 0 || angle_mode < 0 || angle_mode > 2)
        return 0;
    if (V != (angle_mode == 1 ? 5 : 4) || ref_dim < (angle_mode ? 5 : 4)) return 0;
    d = GridDims{Lq, H, L, P, V, ref_dim, ref_per_head ? 1 : 0, angle_mode};
    return ((size_t)B * Lq == 0) ? 2 : 1;                       // 2: nothing to do
}

int boxattn_grid_fwd_f32(const float *ref, int ref_dim, int ref_per_head, const float *offsets,
                         int V, int angle_mode, const float *kernel_idx,
                         const float *valid_ratios, int B, int Lq, int H, int L, int P,
                         float *grid, void *stream)
{
    GridDims d{};
    const int ok = grid_dims(ref_dim, ref_per_head, V, angle_mode, B, Lq, H, L, P, d);
    if (ok == 2) return 0;
    if (ok != 1 || !ref || !offsets || !kernel_idx || !grid) return (int)hipErrorInvalidValue;
    const size_t n_pts = (size_t)B * Lq * H * L * P;
    const size_t blocks = (n_pts + 255) / 256;
    if (blocks > 0x7fffffffu) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(grid_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       ref, offsets, kernel_idx, valid_ratios, d, n_pts, grid);
    return finish();
}

int boxattn_grid_bwd_f32(const float *ref, int ref_dim, int ref_per_head, const float *offsets,
                         int V, int angle_mode, const float *kernel_idx,
                         const float *valid_ratios, const float *grad_grid, int B, int Lq, int H,
                         int L, int P, float *grad_offsets, float *grad_ref_rows, void *stream)
{
    GridDims d{};
    const int ok = grid_dims(ref_dim, ref_per_head, V, angle_mode, B, Lq, H, L, P, d);
    if (ok == 2) return 0;
    if (ok != 1 || !ref || !offsets || !kernel_idx || !grad_grid || !grad_offsets)
        return (int)hipErrorInvalidValue;
    const size_t n_rows = (size_t)B * Lq * H * L;
    const size_t blocks = (n_rows * 4 + 255) / 256;
    if (blocks > 0x7fffffffu) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(grid_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       ref, offsets, kernel_idx, valid_ratios, grad_grid, d, n_rows, grad_offsets,
                       grad_ref_rows);
    return finish();
}



// ---- box attention straight from boxes (second step) ----------------------
static int grid_src(const float *ref, int ref_dim, int ref_per_head, const float *offsets, int V,
                    int angle_mode, const float *kernel_idx, const float *valid_ratios, int B, int Lq,
                    int H, int L, int P, GridSrc &gs)
{
    GridDims gd{};
    const int ok = grid_dims(ref_dim, ref_per_head, V, angle_mode, B, Lq, H, L, P, gd);
    if (ok != 1 || !ref || !offsets || !kernel_idx) return ok == 2 ? BOXATTN_NOT_ELIGIBLE : (int)hipErrorInvalidValue;
    gs = GridSrc{ref, offsets, kernel_idx, valid_ratios, gd, nullptr, nullptr, nullptr};
    return 0;
}

int boxattn_fwd_grid_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                         const float *ref, int ref_dim, int ref_per_head, const float *offsets, int V,
                         int angle_mode, const float *kernel_idx, const float *valid_ratios,
                         const float *attn, int B, int S, int H, int C, int L, int Lq, int P,
                         float *out, float *grid, void *stream)
{
    GridSrc gs;
    if (int rc = grid_src(ref, ref_dim, ref_per_head, offsets, V, angle_mode, kernel_idx, valid_ratios,
                          B, Lq, H, L, P, gs))
        return rc;
    gs.grid_out = grid;
    return launch_fwd_grid<float>(value, shapes, lsi, attn, DIMS, out, gs, ST_);
}
int boxattn_fwd_grid_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                          const float *ref, int ref_dim, int ref_per_head, const float *offsets, int V,
                          int angle_mode, const float *kernel_idx, const float *valid_ratios,
                          const float *attn, int B, int S, int H, int C, int L, int Lq, int P,
                          uint16_t *out, float *grid, void *stream)
{
    GridSrc gs;
    if (int rc = grid_src(ref, ref_dim, ref_per_head, offsets, V, angle_mode, kernel_idx, valid_ratios,
                          B, Lq, H, L, P, gs))
        return rc;
    gs.grid_out = grid;
    return launch_fwd_grid<bf16_t>(value, shapes, lsi, attn, DIMS, out, gs, ST_);
}
int boxattn_bwd_ws_grid_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                            const float *grid, const float *attn, const float *grad_out,
                            const float *ref, int ref_dim, int ref_per_head, const float *offsets, int V,
                            int angle_mode, const float *kernel_idx, const float *valid_ratios, int B,
                            int S, int H, int C, int L, int Lq, int P, float *grad_value,
                            float *grad_offsets, float *grad_ref_rows, float *grad_attn,
                            const int64_t *shapes_host, const int64_t *lsi_host, void *workspace,
                            size_t workspace_bytes, void *stream)
{
    GridSrc gs;
    if (int rc = grid_src(ref, ref_dim, ref_per_head, offsets, V, angle_mode, kernel_idx, valid_ratios,
                          B, Lq, H, L, P, gs))
        return rc;
    if (!grad_offsets) return (int)hipErrorInvalidValue;
    gs.grad_offsets = grad_offsets;
    gs.grad_ref_rows = grad_ref_rows;
    return launch_bwd_ws<float, false>(value, shapes, lsi, grid, attn, nullptr, grad_out, nullptr,
                                       DIMS, grad_value, grad_offsets /* unused grad_loc slot */,
                                       grad_attn, nullptr, shapes_host, lsi_host, workspace,
                                       workspace_bytes, 0, ST_, &gs);
}
int boxattn_bwd_ws_grid_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                             const float *grid, const float *attn, const uint16_t *grad_out,
                             const float *ref, int ref_dim, int ref_per_head, const float *offsets,
                             int V, int angle_mode, const float *kernel_idx, const float *valid_ratios,
                             int B, int S, int H, int C, int L, int Lq, int P, uint16_t *grad_value,
                             float *grad_offsets, float *grad_ref_rows, float *grad_attn,
                             const int64_t *shapes_host, const int64_t *lsi_host, void *workspace,
                             size_t workspace_bytes, void *stream)
{
    GridSrc gs;
    if (int rc = grid_src(ref, ref_dim, ref_per_head, offsets, V, angle_mode, kernel_idx, valid_ratios,
                          B, Lq, H, L, P, gs))
        return rc;
    if (!grad_offsets) return (int)hipErrorInvalidValue;
    gs.grad_offsets = grad_offsets;
    gs.grad_ref_rows = grad_ref_rows;
    return launch_bwd_ws<bf16_t, false>(value, shapes, lsi, grid, attn, nullptr, grad_out, nullptr,
                                        DIMS, grad_value, grad_offsets /* unused grad_loc slot */,
                                        grad_attn, nullptr, shapes_host, lsi_host, workspace,
                                        workspace_bytes, 0, ST_, &gs);
}

}  // extern "C"

// ---- pointwise work around the operator ---------------------------------
// lanes per row of the vector kernels (rows of 4, 8, 16, 32 or 64 values, 16-byte aligned
// tensors), 0: the one-thread-per-row kernels
template <typename T>
static int softmax_group(int n, const T *typed, const float *f32)
{
    if (n % 4 != 0 || (n / 4 & (n / 4 - 1)) != 0 || n > 64) return 0;
    return aligned(typed, 16) && aligned(f32, 16) ? n / 4 : 0;
}

template <typename T>
static int softmax_fwd(const T *logits, long long rows, int n, float *attn, hipStream_t st)
{
    if (rows < 0 || n <= 0 || n > 64) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    if (!logits || !attn) return (int)hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((rows + 255) / 256);
    const int g = softmax_group(n, logits, attn);
    const size_t total = (size_t)rows * n;
    const unsigned vblocks = (unsigned)((total / 4 + 255) / 256);
#define BOXATTN_SOFTMAX_VEC(G) \
    hipLaunchKernelGGL((softmax_vec_fwd_kernel<T, G>), dim3(vblocks), dim3(256), 0, st, logits, total, attn)
    if (g == 1) BOXATTN_SOFTMAX_VEC(1);
    else if (g == 2) BOXATTN_SOFTMAX_VEC(2);
    else if (g == 4) BOXATTN_SOFTMAX_VEC(4);
    else if (g == 8) BOXATTN_SOFTMAX_VEC(8);
    else if (g == 16) BOXATTN_SOFTMAX_VEC(16);
#undef BOXATTN_SOFTMAX_VEC
    else if (n <= 16)
        hipLaunchKernelGGL((softmax_rows_fwd_kernel<T, 16>), dim3(blocks), dim3(256), 0, st, logits,
                           (size_t)rows, n, attn);
    else
        hipLaunchKernelGGL((softmax_rows_fwd_kernel<T, 64>), dim3(blocks), dim3(256), 0, st, logits,
                           (size_t)rows, n, attn);
    return finish();
}
template <typename T>
static int softmax_bwd(const float *attn, const float *grad_attn, long long rows, int n,
                       T *grad_logits, hipStream_t st)
{
    if (rows < 0 || n <= 0 || n > 64) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    if (!attn || !grad_attn || !grad_logits) return (int)hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((rows + 255) / 256);
    const int g = aligned(grad_attn, 16) ? softmax_group(n, grad_logits, attn) : 0;
    const size_t total = (size_t)rows * n;
    const unsigned vblocks = (unsigned)((total / 4 + 255) / 256);
#define BOXATTN_SOFTMAX_VEC(G) \
    hipLaunchKernelGGL((softmax_vec_bwd_kernel<T, G>), dim3(vblocks), dim3(256), 0, st, attn, \
                       grad_attn, total, grad_logits)
    if (g == 1) BOXATTN_SOFTMAX_VEC(1);
    else if (g == 2) BOXATTN_SOFTMAX_VEC(2);
    else if (g == 4) BOXATTN_SOFTMAX_VEC(4);
    else if (g == 8) BOXATTN_SOFTMAX_VEC(8);
    else if (g == 16) BOXATTN_SOFTMAX_VEC(16);
#undef BOXATTN_SOFTMAX_VEC
    else if (n <= 16)
        hipLaunchKernelGGL((softmax_rows_bwd_kernel<T, 16>), dim3(blocks), dim3(256), 0, st, attn,
                           grad_attn, (size_t)rows, n, grad_logits);
    else
        hipLaunchKernelGGL((softmax_rows_bwd_kernel<T, 64>), dim3(blocks), dim3(256), 0, st, attn,
                           grad_attn, (size_t)rows, n, grad_logits);
    return finish();
}

template <typename T>
static int value_prep(const T *value, const unsigned char *mask, long long rows, int d,
                      uint16_t *out, hipStream_t st)
{
    if (rows < 0 || d <= 0 || d % 8 != 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    if (!value || !out || !aligned(value, 16) || !aligned(out, 16)) return (int)hipErrorInvalidValue;
    const size_t n8 = (size_t)rows * d / 8;
    hipLaunchKernelGGL((value_mask_cast_kernel<T>), dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0,
                       st, value, mask, (size_t)rows, d, out);
    return finish();
}

extern "C" {

int boxattn_softmax_fwd_f32(const float *logits, long long rows, int n, float *attn, void *stream)
{
    return softmax_fwd<float>(logits, rows, n, attn, (hipStream_t)stream);
}
int boxattn_softmax_fwd_bf16(const uint16_t *logits, long long rows, int n, float *attn, void *stream)
{
    return softmax_fwd<bf16_t>(logits, rows, n, attn, (hipStream_t)stream);
}
int boxattn_softmax_bwd_f32(const float *attn, const float *grad_attn, long long rows, int n,
                            float *grad_logits, void *stream)
{
    return softmax_bwd<float>(attn, grad_attn, rows, n, grad_logits, (hipStream_t)stream);
}
int boxattn_softmax_bwd_bf16(const float *attn, const float *grad_attn, long long rows, int n,
                             uint16_t *grad_logits, void *stream)
{
    return softmax_bwd<bf16_t>(attn, grad_attn, rows, n, grad_logits, (hipStream_t)stream);
}

int boxattn_value_prep_f32(const float *value, const unsigned char *mask, long long rows, int d,
                           uint16_t *out, void *stream)
{
    return value_prep<float>(value, mask, rows, d, out, (hipStream_t)stream);
}
int boxattn_value_prep_bf16(const uint16_t *value, const unsigned char *mask, long long rows, int d,
                            uint16_t *out, void *stream)
{
    return value_prep<bf16_t>(value, mask, rows, d, out, (hipStream_t)stream);
}

}  // extern "C"
