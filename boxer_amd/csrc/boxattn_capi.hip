// C ABI of the box-attention operator (declared in include/boxattn.h): argument checks,
// kernel choice and launches.  No torch / ATen types anywhere in this library.
//
// Replaces the reference host code e2edet/module/ops/src/box_attn/box_attn.cu:15-135 and
// e2edet/module/ops/src/instance_attn/instance_attn.cu:15-157 (checks, output zero-fill,
// launch) and the launchers box_attn_kernel.cuh:1078-1123, 1126-1505.
#include "../../include/boxattn.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <map>
#include <mutex>
#include <type_traits>
#include <vector>

#include "boxattn_binned.h"
#include "boxattn_fast.h"
#include "boxattn_gather2.h"
#include "boxattn_dense_plan.h"
#include "boxattn_generic.h"

using namespace boxattn;

namespace {

// 0 auto | 1 generic kernels only | 2 fast atomic kernels (error if the shape does not qualify,
// never the binned backward) | 3 binned backward required (error if not eligible)
std::atomic<int> g_variant{0};

// Tuning options (boxattn_set_option): process-wide knobs for A/B runs, relaxed atomics.  (The key numbers
// of rounds 1-3 are kept; the keys of the kernels that lost their A/B and were removed are gone.)
enum { kOptBinChunk = 10,     // records per work item of the accumulate kernels (0: from the number of sample points)
       kOptDense = 11,        // window-staged encoder kernels (forward + point gradients, bf16 and float32): 0 default (on),
                              // 1 off (row-gather kernels: the parity cross-check of the two kernel families), 2 on
       kOptRiders = 15,       // binning passes inside the point-gradient / accumulate (/ forward) launches: 0 default (on; one-pass
                              // fill where the map allows it, boxattn_spec.h; the combine inside the accumulate launch only
                              // for small problems), 1 off (launches of their own), 2 on except the combine (its own
                              // launch), 3 on, combine inside, 4 on with the two-pass binning of round 4 (count riders in the
                              // training forward, fill riders against the exact scan) instead of the one-pass fill
       kOptAccF32 = 19,       // float32 box attention, C = 32, accumulate: 0 default (bf16 matrix cores on exact three-term
                              // splits), 1 VALU list walk (4-byte records), 2 v_mfma_f32_32x32x2_f32
       kOptRideShift = 20,    // where the riders sit: (s_count + 1) | (s_fill + 1) << 4, a rider group every 2^s groups
                              // of 8 workgroups (s = 0: all in front); | v << 8: 64 v bin workgroups (riders) in all;
                              // 0: defaults
       kNumOpts = 21 };
// (round 6 removed the keys whose non-default values had lost their A/B: 12 / 13 window margins, 17 staged forward off,
// 21 staged float32 kernels off -- 11 = 1 switches every window-staged kernel off)
std::atomic<int> g_opt[kNumOpts];      // 0 = default
inline int opt(int k) { return g_opt[k].load(std::memory_order_relaxed); }
inline bool opt_live(int k)
{
    return k == kOptBinChunk || k == kOptDense || k == kOptRiders || k == kOptAccF32 || k == kOptRideShift;
}
#ifndef BOXATTN_RIDE_SHIFT_COUNT
#define BOXATTN_RIDE_SHIFT_COUNT 0     // count riders: all in front of the forward kernel's grid (measured: interleaving
                                       // them with the tiles only stretches the count pass -- riders and tiles want the same
                                       // wave slots, profiles/r04_rider_sweep.log)
#endif
#ifndef BOXATTN_RIDE_SHIFT_FILL
#define BOXATTN_RIDE_SHIFT_FILL 0      // fill riders: in front of the point-gradient kernel's grid
#endif
#ifndef BOXATTN_BIN_WG_TARGET
#define BOXATTN_BIN_WG_TARGET 256      // bin workgroups (= riders) in all, of 256 threads' worth: one per CU.  FEW, FAT riders
                                       // with the next step's locations always in flight cost a quarter of the chip's
                                       // wave slots; 1024 thin ones cost all of them for as long (C2 bf16 step
                                       // 146 / 139 / 135 / 154 us with 1024 / 512 / 256 / 128 riders)
#endif
#ifndef BOXATTN_SPEC_Q_PER_RIDER
#define BOXATTN_SPEC_Q_PER_RIDER 600   // ... of the one-pass fill (boxattn_spec.h) at encoder-sized problems: one rider per ~600
                                       // (query, slice) pairs, 256 ... 768 of them.  Its riders are bound by latency (two
                                       // barriers and an atomic round trip a step), not by instruction issue, so somewhat more,
                                       // shorter riders than the two-pass fill wants.  C2 bf16 (213 k pairs), one box, resident /
                                       // cache-cold inputs, us a step: 256 riders 126.3 / 140.2, 384: 122.5 / 132.9, 512: 119.8 /
                                       // 135.0, 768: 119.3 / 135.0, 1024: 122.8 / 136.2 (two-pass, 256: 124.0 / 133.2); C2' (354 k):
                                       // 512: 209.8 / 219.4, 1024: 218.5 / 225.1 (two-pass: 211.3 / 221.5) -- profiles/r06_onepass_ab.log
#endif
// one_pass: the shape's box-attention backward fills its bins in one pass (a property of the map, plan_layout)
inline long long bin_wg_target(bool one_pass, long long queries)
{
    const int v = (opt(kOptRideShift) >> 8) & 31;       // 64 v riders
    if (v > 0) return 64ll * v;
    if (!one_pass || queries < 65536) return BOXATTN_BIN_WG_TARGET;
    const long long r = (queries / BOXATTN_SPEC_Q_PER_RIDER + 32) / 64 * 64;
    return std::min(768ll, std::max(256ll, r));
}
inline unsigned ride_shift(bool fill)
{
    const int o = opt(kOptRideShift), v = fill ? (o >> 4) & 15 : o & 15;
    return v > 0 ? (unsigned)(v - 1) : (unsigned)(fill ? BOXATTN_RIDE_SHIFT_FILL : BOXATTN_RIDE_SHIFT_COUNT);
}

inline int ceil_div_sz(size_t a, size_t b) { return (int)((a + b - 1) / b); }

// zero-fill on the stream (zero_fill_kernel, see there why not hipMemsetAsync)
inline hipError_t zero_async(void *p, size_t bytes, hipStream_t st)
{
    if (!bytes) return hipSuccess;
    const int blocks = (int)std::min<size_t>((bytes / 16 + 255) / 256 + 1, 2048);
    hipLaunchKernelGGL(zero_fill_kernel, dim3(blocks), dim3(256), 0, st, (unsigned char *)p, bytes,
                       (int)(bytes >= ((size_t)8 << 20)));
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
    ix.head_xcd = (d.H == 8 && (long long)d.Lq * 4 <= d.S &&
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

// Over how many workgroup rows (grid.y) the point tiles of a (query, head) pair are spread in the
// point-gradient kernel: aim at ~4096 workgroups when the query dimension alone gives fewer than 1024.
inline int point_split(int blocks, int tiles)
{
    if (blocks >= 1024 || tiles < 2) return 1;
    return std::max(1, std::min(tiles, (4096 + blocks - 1) / blocks));
}

inline int finish()
{
    return (int)hipGetLastError();
}

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


// rider workgroups of a launch: the caller sets ride.grid.n_riders / .shift, the rest follows from the grid
inline BinRide place_riders(const BinRide *ride_in, unsigned own_blocks, unsigned *total)
{
    BinRide ride = ride_in ? *ride_in : BinRide{};
    ride.grid = ride_grid(ride.grid.n_riders, own_blocks, ride.grid.shift, total);
    return ride;
}

// ------------------------------------------------------------------------------ forward
inline bool make_dense_plan(const Dims &d, const int64_t *sh, const int64_t *ls, DensePlan &p, int elem);

// count_ride: the backward's count pass + scans to run as rider workgroups of the forward kernel (training
// forward); *ride_taken says whether the kernel that was launched carried them
template <typename ST, bool INST>
int launch_fwd(const ST *value, const int64_t *shapes, const int64_t *lsi,
               const typename Storage<ST>::compute *loc,
               const typename Storage<ST>::compute *w_sp,
               const typename Storage<ST>::compute *w_lv, const Dims &d, ST *out, ST *mask,
               hipStream_t st, const int64_t *shapes_host = nullptr,
               const int64_t *lsi_host = nullptr, const BinRide *count_ride = nullptr,
               bool *ride_taken = nullptr, bool allow_dense = true, unsigned long long *stats = nullptr)
{
    if (ride_taken) *ride_taken = false;
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
                if (allow_dense && shapes_host && lsi_host && aligned(value, 16) && aligned(out, 16) &&
                    aligned(loc, 8) && make_dense_plan(d, shapes_host, lsi_host, dp, 2)) {
                    ScopedKernelTimer timer(g_prof.ev[kSlotFwd], st);
                    launch_fwd_dense(value, loc, w_sp, out, dp, (unsigned)(d.n_value() * sizeof(bf16_t)),
                                     count_ride ? *count_ride : BinRide{}, stats, st);
                    if (count_ride && ride_taken) *ride_taken = true;
                    return finish();
                }
            }
            if constexpr (!INST && std::is_same<ST, float>::value) {      // encoder case, float32: window-staged VALU forward
                DensePlan dp;
                if (allow_dense && shapes_host && lsi_host && aligned(value, 16) && aligned(out, 16) &&
                    aligned(loc, 8) && make_dense_plan(d, shapes_host, lsi_host, dp, 4)) {
                    ScopedKernelTimer timer(g_prof.ev[kSlotFwd], st);
                    launch_fwd_dense_f32(value, loc, w_sp, out, dp, (unsigned)(d.n_value() * sizeof(float)),
                                         count_ride ? *count_ride : BinRide{}, stats, st);
                    if (count_ride && ride_taken) *ride_taken = true;
                    return finish();
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
            ScopedKernelTimer timer(g_prof.ev[kSlotFwd], st);
            // instance attention with few (query, head) pairs and many points (the mask decoder): one wave per
            // pair, the points spread over the lane groups (any storage type, no atomics)
            if constexpr (INST) {
                if (gen2 && blocks < 1024 && d.P >= kWave / G) {
                    // 8 or more point steps a pair: its points over the four waves of a workgroup
                    const int steps = ceil_div_sz((size_t)d.P, (size_t)(kWave / G));
                    const unsigned ws = steps >= 8 ? 4 : steps >= 4 ? 2 : 1;
                    const size_t ppw = 4 / ws;
                    const int wblocks = ix.head_xcd ? 8 * ceil_div_sz((size_t)d.B * d.Lq, ppw)
                                                    : ceil_div_sz(n_qh, ppw);
                    unsigned total = 0;
                    const BinRide ride = place_riders(count_ride, (unsigned)wblocks, &total);
#define BOXATTN_FWD_WIDE(GG, VV)                                                              \
    hipLaunchKernelGGL((fwd_inst_wide_kernel<ST, GG, VV>), dim3(total), dim3(256), 0, st,     \
                       value, shapes, lsi, loc, w_sp, w_lv, d.S, d.H, d.L, d.Lq, d.P, out,    \
                       mask, with_grid(ix, wblocks, 1, 1), (unsigned)vbytes, ride, ws);
                    BOXATTN_GATHER_DISPATCH(cfg, BOXATTN_FWD_WIDE);
#undef BOXATTN_FWD_WIDE
                    if (count_ride && ride_taken) *ride_taken = true;
                    return finish();
                }
            }
            if (gen2) {
                unsigned total = 0;
                const BinRide ride = place_riders(count_ride, (unsigned)blocks, &total);
#define BOXATTN_FWD2(GG, VV)                                                                  \
    hipLaunchKernelGGL((fwd2_kernel<ST, GG, INST, GatherUnroll<ST, GG, VV>::value, VV>),      \
                       dim3(total, 1), dim3(256), 0, st, value, shapes, lsi, loc, w_sp,       \
                       w_lv, d.S, d.H, d.L, d.Lq, d.P, out, mask,                             \
                       with_grid(ix, blocks, 1, (d.P + GG - 1) / GG), (unsigned)vbytes, ride);
                BOXATTN_GATHER_DISPATCH(cfg, BOXATTN_FWD2);
#undef BOXATTN_FWD2
                if (count_ride && ride_taken) *ride_taken = true;
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



#include "boxattn_host_plan.h"       // make_plan / make_dense_plan, plan_layout / scratch_layout, acc_kind, riders_ok

// record format / query order / points per thread of the bin passes of a call
template <typename ST, bool INST>
inline int bin_flavour(const Dims &d, const float *loc, const float *w_sp)
{
    const bool wide = acc_kind<ST, INST>(d) != kAccValu;
    // four points per thread where the layout allows 16-byte loads of a (query, level)'s points
    const bool pt4 = d.P % 4 == 0 && aligned(loc, 16) && (!wide || aligned(w_sp, 16));
    return (wide ? kRideWide : kRideInterleave) | (pt4 ? kRidePt4 : 0);
}
inline BinRide make_ride(const float *loc, const float *w_sp, const Dims &d, const BinPlan &plan,
                         const PlanLayout &pl, char *pbuf, const ScratchLayout *sl, char *sbuf, int flavour,
                         bool fill)
{
    BinRide r{};
    r.loc = loc; r.w_sp = w_sp;
    r.part = (int *)(pbuf + pl.part); r.subtot = (int *)(pbuf + pl.subtot); r.offsets = (int *)(pbuf + pl.offsets);
    r.items = (int4 *)(pbuf + pl.items); r.combos = (int4 *)(pbuf + pl.combos);
    r.n_items = (int *)(pbuf + pl.n_items); r.tickets = (int *)(pbuf + pl.tickets);
    r.records = sl ? (int *)(sbuf + sl->records) : nullptr;
    r.ctickets = fill && sl ? (int *)(sbuf + sl->ctickets) : nullptr;
    r.plan = plan;
    r.H = d.H; r.Lq = d.Lq; r.P = d.P; r.q_per_wg = pl.q_per_wg; r.n_wg = pl.n_wg;
    r.nwg_magic = div_magic((unsigned)pl.n_wg);
    r.flavour = flavour;
    r.grid.n_riders = (unsigned)(pl.n_wg * d.B * d.H);
    r.grid.shift = ride_shift(fill);
    r.trace = (unsigned long long *)g_dense_dbg.load();      // (read only by builds with BOXATTN_RIDE_TRACE)
    return r;
}

// The bin passes as launches of their own (a backward that plans for itself; maps too big for the riders;
// host kernels that carry none): stages kBinCount | kBinScan | kBinFill.
enum { kBinCount = 1, kBinScan = 2, kBinFill = 4 };
inline void launch_binning(int flavour, const float *loc, const float *w_sp, const Dims &d, const BinPlan &plan,
                           const PlanLayout &w, char *pbuf, int *records, hipStream_t st, int stages,
                           int *ctickets = nullptr)
{
    constexpr int BW = 8, BH = 4;
    const int ns = d.B * d.H;
    int *part = (int *)(pbuf + w.part), *subtot = (int *)(pbuf + w.subtot);
    int *n_items = (int *)(pbuf + w.n_items), *offsets = (int *)(pbuf + w.offsets);
    int4 *items = (int4 *)(pbuf + w.items), *combos = (int4 *)(pbuf + w.combos);
    const dim3 bgrid(w.n_wg, ns);
    const size_t bsh = ((size_t)plan.nblk + 1) * sizeof(int);
    const int inter = (flavour & kRideInterleave) ? 1 : 0;
    const bool wide = (flavour & kRideWide) != 0, pt4 = (flavour & kRidePt4) != 0;
    ScopedKernelTimer timer(g_prof.ev[kSlotBwdBin], st);
#define BOXATTN_BIN(FILL_, WIDE_, PT_)                                                              \
    hipLaunchKernelGGL((bin_kernel<BW, BH, FILL_, WIDE_, PT_>), bgrid, dim3(kBinThreads), bsh, st, loc, w_sp, \
                       plan, d.H, d.Lq, d.P, w.q_per_wg, w.n_wg, inter, part, subtot, offsets, records, ctickets)
    if (stages & kBinCount) {
        if (pt4) BOXATTN_BIN(false, false, 4); else BOXATTN_BIN(false, false, 1);
        if (plan.zero_workers > 0)         // sparse map: the zero workers' geometry table is part of the plan
            hipLaunchKernelGGL(zero_geo_kernel, dim3((plan.nblk + 255) / 256), dim3(256), 0, st, plan,
                               (int2 *)(pbuf + w.zgeo));
    }
#ifndef BOXATTN_TUNE_SCAN_FUSE_WG
#define BOXATTN_TUNE_SCAN_FUSE_WG 48   // up to this many bin workgroups per slice the block scan does kernel A's work too
                                       // (38 workgroups, the 300-query decoders: C3'' fp32 68 -> 61 us; 64, C2: binning 49 -> 61 us)
#endif
    // (big maps, multi-workgroup block scan: fused up to 8 bin workgroups per slice -- the 1 000-query BEV decoder has 1)
    const bool fuse_a = plan.nblk <= kScanThreads ? w.n_wg <= BOXATTN_TUNE_SCAN_FUSE_WG : w.n_wg <= 8;
    if (stages & kBinScan) {
        if (!fuse_a)
            hipLaunchKernelGGL(bin_scan_a_kernel,
                               dim3(kScanSub, ns, std::min(64, (plan.nblk + 255) / 256)), dim3(256), 0, st,
                               part, w.n_wg, subtot, plan);
        if (plan.nblk > kScanThreads) {           // big maps: the block scan over several CUs
            const int nseg = (plan.nblk + kScanThreads - 1) / kScanThreads;
            int4 *tmp = (int4 *)(pbuf + w.scan_tmp), *segtot = tmp + (size_t)ns * plan.nblk;
            hipLaunchKernelGGL(bin_scan_seg_kernel, dim3(nseg, ns), dim3(kScanThreads), 0, st, subtot,
                               offsets, tmp, segtot, plan, part, fuse_a ? w.n_wg : 0);
            hipLaunchKernelGGL(bin_scan_emit_kernel, dim3(nseg, ns), dim3(kScanThreads), 0, st, offsets,
                               tmp, segtot, items, combos, n_items, plan);
        } else {
            hipLaunchKernelGGL(bin_scan_kernel, dim3(ns), dim3(kScanThreads), 0, st, subtot, offsets,
                               items, combos, n_items, plan, part, fuse_a ? w.n_wg : 0);
        }
    }
    if (stages & kBinFill) {
        // (4-byte records: one point per thread -- neighbouring lanes then hold neighbouring slots and their
        // stores coalesce: 18.9 us against 23.0 with four)
        if (wide && pt4) BOXATTN_BIN(true, true, 4);
        else if (wide) BOXATTN_BIN(true, true, 1);
        else BOXATTN_BIN(true, false, 1);
    }
#undef BOXATTN_BIN
}

inline bool dense_pointgrad_ok(const DensePlan *dp, const void *value, const void *loc, const void *attn,
                               const void *grad_out, const void *grad_loc, const void *grad_attn)
{
    return dp && aligned(value, 16) && aligned(grad_out, 16) && aligned(loc, 8) &&
           aligned(attn, 4) && aligned(grad_loc, 16) && aligned(grad_attn, 16);
}

// Point gradients (grad_loc / grad_weight): query-major, independent of how grad_value is accumulated;
// `fill_ride`: the backward's fill pass as rider workgroups of this launch (*ride_taken: carried).
template <typename ST, int G, bool INST>
void launch_pointgrad(const ST *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                      const float *w_sp, const float *w_lv, const ST *grad_out, const ST *grad_mask,
                      const Dims &d, float *grad_loc, float *grad_sp, float *grad_lv, hipStream_t st,
                      const BinRide *fill_ride, bool *ride_taken, const DensePlan *dp)
{
    if (ride_taken) *ride_taken = false;
    ScopedKernelTimer timer(g_prof.ev[kSlotBwdPoints], st);
    if constexpr (std::is_same<ST, bf16_t>::value && !INST) {
        if (dense_pointgrad_ok(dp, value, loc, w_sp, grad_out, grad_loc, grad_sp)) {
            launch_pointgrad_dense(value, loc, w_sp, grad_out, *dp, grad_loc, grad_sp,
                                   (unsigned)(d.n_value() * sizeof(bf16_t)), st, fill_ride ? *fill_ride : BinRide{});
            if (fill_ride && ride_taken) *ride_taken = true;
            return;
        }
    }
    if constexpr (std::is_same<ST, float>::value && !INST) {
        if (dense_pointgrad_ok(dp, value, loc, w_sp, grad_out, grad_loc, grad_sp)) {
            launch_pointgrad_dense_f32(value, loc, w_sp, grad_out, *dp, grad_loc, grad_sp,
                                       (unsigned)(d.n_value() * sizeof(float)), st, fill_ride ? *fill_ride : BinRide{});
            if (fill_ride && ride_taken) *ride_taken = true;
            return;
        }
    }
    const size_t n_qh = d.n_qh();
    const size_t vbytes = d.n_value() * sizeof(ST);
    GatherIdx ix{};
    if (vbytes < kOobOffset && gather_idx(d, ix, sizeof(ST))) {
        const GatherCfg cfg = gather_cfg<ST>(d, aligned(value, 16) && aligned(grad_out, 16) &&
                                                (!INST || aligned(grad_mask, 16)));
        const int blocks = gather_blocks(d, ix, kWave / cfg.G);
        // few pairs x many points (instance attention on the mask-decoder grid): one wave
        // per pair, its lane groups over the point tiles; else one lane group per pair
        const int tiles = (d.L * d.P + cfg.G - 1) / cfg.G;
        const bool wpp = INST && blocks < 1024 && tiles >= kWave / cfg.G;
        unsigned total = 0;
        if (wpp) {
            const int wblocks = ix.head_xcd ? 8 * ceil_div_sz((size_t)d.B * d.Lq, 4)
                                            : ceil_div_sz(n_qh, 4);
            const int split = point_split(wblocks, tiles / (kWave / cfg.G));
            const BinRide ride = place_riders(fill_ride, (unsigned)wblocks, &total);
#define BOXATTN_PG2W(GG, VV)                                                                         \
hipLaunchKernelGGL((pointgrad2_kernel<ST, GG, INST, GatherUnroll<ST, GG, VV>::value, VV, true>), \
                   dim3(total, split), dim3(256), 0, st, value, shapes, lsi, loc, w_sp, w_lv,    \
                   grad_out, grad_mask, d.S, d.H, d.L, d.Lq, d.P, grad_loc, grad_sp, grad_lv,    \
                   with_grid(ix, wblocks, split, tiles), (unsigned)vbytes, ride);
            BOXATTN_GATHER_DISPATCH(cfg, BOXATTN_PG2W);
#undef BOXATTN_PG2W
        } else {
            const int split = point_split(blocks, tiles);
            const BinRide ride = place_riders(fill_ride, (unsigned)blocks, &total);
            {
#define BOXATTN_PG2(GG, VV)                                                                   \
hipLaunchKernelGGL((pointgrad2_kernel<ST, GG, INST, GatherUnroll<ST, GG, VV>::value, VV>), \
                   dim3(total, split), dim3(256), 0, st, value, shapes, lsi, loc, w_sp,    \
                   w_lv, grad_out, grad_mask, d.S, d.H, d.L, d.Lq, d.P, grad_loc, grad_sp, \
                   grad_lv, with_grid(ix, blocks, split, tiles), (unsigned)vbytes, ride);
                BOXATTN_GATHER_DISPATCH(cfg, BOXATTN_PG2);
#undef BOXATTN_PG2
            }
        }
        if (fill_ride && ride_taken) *ride_taken = true;
    } else {
        const int blocks = ceil_div_sz(n_qh, (size_t)(kWave / G) * 4);
        hipLaunchKernelGGL((bwd_fast_kernel<ST, 4, G, INST, false>), dim3(blocks), dim3(256),
                           0, st, value, shapes, lsi, loc, w_sp, w_lv, grad_out, grad_mask,
                           d.S, d.H, d.L, d.Lq, d.P, (float *)nullptr, grad_loc, grad_sp, grad_lv,
                           n_qh);
    }
}

// The accumulate launch of a binned backward: one single-wave workgroup per potential work item (item_cap is
// the host-side bound; the real count lives on the device, surplus workgroups exit at once); the hardware
// dispatcher hands them out as waves retire -- dynamic load balancing without a work-queue atomic (a
// persistent-waves version with a software queue was 15 % slower).  The kernels map workgroups to (slice,
// worker) themselves (XCD affinity), hence the 8-aligned grid.
constexpr int kAccWgCap = 1024;   // accumulate workgroups per slice; beyond that a workgroup takes several items (its next one in
                                  // flight); 256 / 512 / 1024 / 3072 / 6144: C5 99 / 85 / 83 / 93 / 111 us, C2 67 / 54 / 54 / 55 / 54
template <typename ST, int G, bool INST>
int launch_accumulate(AccKind acc, const ST *grad_out, const ST *grad_mask, const float *loc, const float *w_sp,
                      const float *w_lv, const Dims &d, const BinPlan &plan, const int *offsets, const int4 *items,
                      const int *n_items, const int *records, ST *grad_value, float *partials,
                      const ChunkCombine &cc, const ZeroRole &zr, hipStream_t st)
{
    constexpr int C = 4 * G;
    const int ns = d.B * d.H, ns8 = (ns + 7) / 8 * 8;
    const int wg_per_slice = std::min(kAccWgCap, std::max(1, plan.item_cap));
    ScopedKernelTimer timer(g_prof.ev[kSlotBwdAccum], st);
    if constexpr (std::is_same<ST, bf16_t>::value && !INST) {
        if (acc == kAccTr) {
            launch_accumulate_tr(C, grad_out, (size_t)d.B * d.Lq * d.H * C * sizeof(ST), plan, d.S, d.H, d.Lq, items,
                                 n_items, records, grad_value, partials, wg_per_slice, ns8, cc, zr, st);
            return finish();
        }
    }
    if constexpr (std::is_same<ST, float>::value && !INST && C == 32) {
        if (acc == kAccSplit) {
            launch_accumulate_split(grad_out, (size_t)d.B * d.Lq * d.H * C * sizeof(float), plan, d.S, d.H, d.Lq, items,
                                    n_items, records, grad_value, partials, wg_per_slice, ns8, cc, zr, st);
            return finish();
        }
        if (acc == kAccF32) {
            launch_accumulate_f32(grad_out, (size_t)d.B * d.Lq * d.H * C * sizeof(float), plan, d.S, d.H, d.Lq, items,
                                  n_items, records, grad_value, partials, wg_per_slice, ns8, cc, zr, st);
            return finish();
        }
    }
    if constexpr (std::is_same<ST, float>::value && INST && C == 32) {
        if (acc == kAccSplit) {
            launch_accumulate_split(grad_out, (size_t)d.B * d.Lq * d.H * C * sizeof(float), plan, d.S, d.H, d.Lq, items,
                                    n_items, records, grad_value, partials, wg_per_slice, ns8, cc, zr, st, grad_mask,
                                    (size_t)d.B * d.Lq * d.P * d.H * C * sizeof(float), w_lv, d.P);
            return finish();
        }
    }
    if (acc != kAccValu) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL((binned_accumulate_kernel<ST, C, INST, 1, false>), dim3(wg_per_slice + plan.zero_workers, ns8), dim3(64), 0, st,
                       grad_out, grad_mask, loc, w_sp, w_lv, plan, d.S, d.H, d.Lq, d.P, offsets, items, n_items,
                       records, grad_value, partials, cc, zr);
    return finish();
}

// A binned backward: [count -> scans, unless the training forward left a plan] -> point gradients with the
// fill pass riding in their launch -> accumulate (chunked blocks summed by their last chunk).
// spec: the caller's state of the one-pass fill (boxattn_spec.h) -- nullptr: two-pass binning; spec_warm: its ranges were
// planned by an earlier call (else this call runs the two-pass passes and spec_layout_kernel plans them from its scan)
template <typename ST, int G, bool INST>
int run_binned(const ST *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
               const float *w_sp, const float *w_lv, const ST *grad_out, const ST *grad_mask,
               const Dims &d, const BinPlan &plan, const PlanLayout &pl, char *pbuf, const ScratchLayout &sl,
               char *sbuf, ST *grad_value, float *grad_loc, float *grad_sp, float *grad_lv, bool plan_ready,
               hipStream_t st, const DensePlan *dp, const SpecRide *spec = nullptr,
               bool spec_warm = false, int *spec_tickets = nullptr)
{
    const int ns = d.B * d.H;
    const AccKind acc = acc_kind<ST, INST>(d);
    const int flavour = bin_flavour<ST, INST>(d, loc, w_sp);
    int *records = (int *)(sbuf + sl.records);
    float *partials = (float *)(sbuf + sl.partials);
    const int *n_items = (const int *)(pbuf + pl.n_items), *offsets = (const int *)(pbuf + pl.offsets);
    const int4 *items = (const int4 *)(pbuf + pl.items), *combos = (const int4 *)(pbuf + pl.combos);
    const bool one_pass = spec && spec_warm;
    if (!plan_ready && !one_pass) launch_binning(flavour, loc, w_sp, d, plan, pl, pbuf, records, st, kBinCount | kBinScan);
    bool filled = false;
    if (one_pass) {
        // point gradients + the fill riders' ONE pass over the locations into the ranges the state holds; the slice's
        // last rider writes the work items (and the next call's ranges)
        BinRide ride = make_ride(loc, w_sp, d, plan, pl, pbuf, &sl, sbuf, flavour | kRideSpec, true);
        ride.spec = *spec;
        ride.tickets = spec_tickets;
        launch_pointgrad<ST, G, INST>(value, shapes, lsi, loc, w_sp, w_lv, grad_out, grad_mask, d, grad_loc,
                                      grad_sp, grad_lv, st, &ride, &filled, dp);
        if (!filled)        // (the launched kernel carries no riders: the two-pass passes as launches, below)
            launch_binning(flavour, loc, w_sp, d, plan, pl, pbuf, records, st, kBinCount | kBinScan);
    } else if (riders_ok(plan, pl)) {
        const BinRide ride = make_ride(loc, w_sp, d, plan, pl, pbuf, &sl, sbuf, flavour, true);
        launch_pointgrad<ST, G, INST>(value, shapes, lsi, loc, w_sp, w_lv, grad_out, grad_mask, d, grad_loc,
                                      grad_sp, grad_lv, st, &ride, &filled, dp);
    } else {
        launch_pointgrad<ST, G, INST>(value, shapes, lsi, loc, w_sp, w_lv, grad_out, grad_mask, d, grad_loc,
                                      grad_sp, grad_lv, st, nullptr, nullptr, dp);
    }
    // Chunked blocks: summed by their last chunk inside the accumulate launch (chunk_finish; the fill pass -- riding
    // or not -- clears the blocks' tickets) wherever the riders run: one launch less, C2 bf16 125.7 -> 122.9 us, C2 fp32
    // 245 -> 240, C2' fp32 420 -> 411 (profiles/r04_combine_sweep.log; with 1 024-record chunks and a spilling store
    // path it had been the other way round at encoder sizes) and for the short launch chains of few sample points.
    // Encoder-sized maps too big for the riders (the BEV encoder) keep combine_partials_kernel behind the
    // accumulate launch (C5' bf16 464 against 478 us).
    const bool small = (long long)d.Lq * d.L * d.P < 65536;
    const bool own_combine = opt(kOptRiders) == 1 || opt(kOptRiders) == 2 ||
                             (opt(kOptRiders) == 0 && !riders_ok(plan, pl) && !small);
    if (!filled)
        launch_binning(flavour, loc, w_sp, d, plan, pl, pbuf, records, st, kBinFill,
                       own_combine ? nullptr : (int *)(sbuf + sl.ctickets));
    ChunkCombine cc{};
    if (!own_combine) cc = ChunkCombine{(int *)(sbuf + sl.ctickets), combos, plan.nblk, plan.pslot_cap};
    // (sparse maps: zero workers in front of the accumulate grid store the zeros of the blocks without records)
    ZeroRole zr{offsets, (const int2 *)(pbuf + pl.zgeo)};
    BinPlan plan_acc = plan;
    if (one_pass && filled) {       // the front rows of the grid: redo workers for the blocks that outgrew their range
        zr = ZeroRole{nullptr, nullptr, spec->redo, loc, w_sp, d.P};
        plan_acc.zero_workers = kSpecRedoWorkers;
    }
    int rc = launch_accumulate<ST, G, INST>(acc, grad_out, grad_mask, loc, w_sp, w_lv, d, plan_acc, offsets, items,
                                            n_items, records, grad_value, partials, cc, zr, st);
    if (rc) return rc;
    if (spec && !(one_pass && filled))      // a cold state: the next call's ranges from this call's exact scan
        hipLaunchKernelGGL(spec_layout_kernel<256>, dim3(ns), dim3(256), 0, st, plan, offsets, spec->cursor, spec->cbase,
                           spec->redo);
    if (own_combine) {
        ScopedKernelTimer timer(g_prof.ev[kSlotBwdCombine], st);
        hipLaunchKernelGGL((combine_partials_kernel<ST, 4 * G>), dim3(64, ns), dim3(64), 0, st, combos,
                           n_items, partials, combine_plan(plan), d.S, d.H, grad_value);
    }
    return finish();
}

// Backward with a caller-provided workspace (and, optionally, the plan a training forward built); falls back
// to the atomic kernels when the binned algorithm does not apply.
template <typename ST, bool INST>
int launch_bwd_ws(const ST *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                  const float *w_sp, const float *w_lv, const ST *grad_out, const ST *grad_mask,
                  const Dims &d, ST *grad_value, float *grad_loc, float *grad_sp, float *grad_lv,
                  const int64_t *shapes_host, const int64_t *lsi_host, void *workspace,
                  size_t workspace_bytes, const void *plan_buf, size_t plan_bytes, int hints, hipStream_t st,
                  void *state = nullptr, size_t state_bytes = 0)
{
    constexpr bool kBf16 = std::is_same<ST, bf16_t>::value;
    if (!d.valid()) return (int)hipErrorInvalidValue;
    BinPlan plan;
    const size_t nv = d.n_value();
    bool binned = (g_variant == 0 || g_variant == 3) && workspace && nv && d.n_qh() &&
                  make_plan(d, shapes_host, lsi_host, plan) &&
                  fast_ok<ST>(d, value, loc, grad_out,
                              INST ? (const void *)grad_mask : (const void *)grad_out, grad_loc) &&
                  aligned(workspace, 256) && aligned(grad_value, 16);
    PlanLayout pl{};
    ScratchLayout sl{};
    bool plan_ready = false;
    if (binned) {
        pl = plan_layout(d, plan);
        sl = scratch_layout(d, plan, wide_workspace(kBf16, d));
        plan_ready = plan_buf && plan_bytes >= pl.total && aligned(plan_buf, 256);
        binned = workspace_bytes >= (plan_ready ? sl.total : pl.total + sl.total);
    }
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
    char *pbuf = plan_ready ? const_cast<char *>((const char *)plan_buf) : ws;
    char *sbuf = plan_ready ? ws : ws + pl.total;
    int rc = 0;
    DensePlan dense;
    const DensePlan *dp = !std::is_same<ST, double>::value && !INST && !(hints & BOXATTN_HINT_NOT_LOCAL) &&
                                  make_dense_plan(d, shapes_host, lsi_host, dense, (int)sizeof(ST)) ? &dense : nullptr;
    // the one-pass fill (boxattn_spec.h): the caller's state holds the bins' ranges
    SpecRide spec{};
    const SpecRide *sp = nullptr;
    bool spec_warm = false;
    int *spec_tickets = nullptr;
    if (!plan_ready && spec_ok<ST, INST>(d, plan, pl)) {
        const StateLayout sy = state_layout(d, &plan);
        const int chk = state_check(state, state_bytes, sy, d, shapes_host, st, (hints & BOXATTN_HINT_FRESH_STATE) != 0);
        if (chk < 0) return (int)hipErrorInvalidValue;
        if (chk > 0) {
            char *sb = (char *)state;
            spec = SpecRide{(int *)(sb + sy.cursor), (int *)(sb + sy.cbase), (int2 *)(sb + sy.redo),
                            (unsigned long long *)(sb + sy.spec_stats)};
            sp = &spec;
            spec_warm = chk == 2;
            spec_tickets = (int *)(sb + sy.tickets);
            state_learned(state);        // (whichever way this call goes, it leaves the next call's ranges behind)
        }
    }
    switch (fast_group(d)) {
#define BOXATTN_BINNED_CASE(GG)                                                                 \
    case GG:                                                                                    \
        rc = run_binned<ST, GG, INST>(value, shapes, lsi, loc, w_sp, w_lv, grad_out, grad_mask, \
                                      d, plan, pl, pbuf, sl, sbuf, grad_value, grad_loc, grad_sp, grad_lv, \
                                      plan_ready, st, dp, sp, spec_warm, spec_tickets);     \
        break;
        BOXATTN_BINNED_CASE(4)
        BOXATTN_BINNED_CASE(8)
        BOXATTN_BINNED_CASE(16)
#undef BOXATTN_BINNED_CASE
    }
    return rc;
}

// Training forward: the forward kernel with the backward's count pass and scans riding in its launch (they only
// depend on the sampling locations); `plan_buf` then carries the plan to the *_bwd_ws_* call.
template <typename ST, bool INST>
int launch_fwd_train(const ST *value, const int64_t *shapes, const int64_t *lsi, const float *loc,
                     const float *w_sp, const float *w_lv, const Dims &d, ST *out, ST *mask,
                     const int64_t *shapes_host, const int64_t *lsi_host, void *plan_buf,
                     size_t plan_bytes, void *state, size_t state_bytes, int hints, int *plan_built,
                     hipStream_t st)
{
    if (plan_built) *plan_built = 0;
    const bool allow_dense = !(hints & BOXATTN_HINT_NOT_LOCAL);
    // the locality counters of the window-staged forward, then the riders' tickets, then the one-pass fill's ranges
    const size_t tbytes = (size_t)std::max(0, d.B) * (size_t)std::max(0, d.H) * kRideTickets * sizeof(int);
    BinPlan plan;
    const bool planned = d.valid() && make_plan(d, shapes_host, lsi_host, plan);
    const StateLayout sy = state_layout(d, planned ? &plan : nullptr);
    // (a state buffer that is too small or misaligned is an error, not "no state": the caller would silently lose the
    // locality counters and pay a zero-fill launch in front of every forward)
    if (state_check(state, state_bytes, sy, d, shapes_host, st, (hints & BOXATTN_HINT_FRESH_STATE) != 0) < 0)
        return (int)hipErrorInvalidValue;
    const bool have_state = state != nullptr;
    unsigned long long *stats = have_state ? (unsigned long long *)state : nullptr;
    // The backward of this shape fills its bins in one pass, from ranges it keeps in the state buffer: nothing of the
    // backward rides in the forward, and there is no plan to hand over (*plan_built stays 0)
    if (have_state && planned && d.n_value() && d.n_qh() && (g_variant == 0 || g_variant == 3) &&
        fast_ok<ST>(d, value, loc, out, INST ? (const void *)mask : (const void *)out, out) &&
        spec_ok<ST, INST>(d, plan, plan_layout(d, plan)))
        return launch_fwd<ST, INST>(value, shapes, lsi, loc, w_sp, w_lv, d, out, mask, st,
                                    shapes_host, lsi_host, nullptr, nullptr, allow_dense, stats);
    bool ok = (g_variant == 0 || g_variant == 3) && plan_buf && planned &&
              d.n_value() && d.n_qh() &&
              aligned(plan_buf, 256) &&
              // what the backward will check and the forward can already see (its other operands,
              // grad_out / grad_value, are re-checked there; an ineligible backward ignores the plan)
              fast_ok<ST>(d, value, loc, out, INST ? (const void *)mask : (const void *)out, out);
    PlanLayout pl{};
    if (ok) {
        pl = plan_layout(d, plan);
        ok = plan_bytes >= pl.total;
    }
    if (!ok)
        return launch_fwd<ST, INST>(value, shapes, lsi, loc, w_sp, w_lv, d, out, mask, st,
                                    shapes_host, lsi_host, nullptr, nullptr, allow_dense, stats);
    char *pbuf = (char *)plan_buf;
    const int flavour = bin_flavour<ST, INST>(d, loc, w_sp);
    bool taken = false;
    int rc = 0;
    if (riders_ok(plan, pl)) {
        // The riders' tickets start at zero and are left zero by the call.  `state`: the caller's zeroed,
        // persistent ticket buffer (one per stream: calls on a stream never overlap); without it the tickets live
        // in the -- possibly fresh -- plan buffer and are cleared by a launch of their own (~5 us in front of
        // the forward kernel).
        BinRide ride = make_ride(loc, w_sp, d, plan, pl, pbuf, nullptr, nullptr, flavour, false);
        if (have_state) {
            ride.tickets = (int *)((char *)state + sy.tickets);
        } else {
            hipError_t e = zero_async(pbuf + pl.tickets, tbytes, st);
            if (e != hipSuccess) return (int)e;
        }
        rc = launch_fwd<ST, INST>(value, shapes, lsi, loc, w_sp, w_lv, d, out, mask, st, shapes_host, lsi_host,
                                  &ride, &taken, allow_dense, stats);
    } else {
        rc = launch_fwd<ST, INST>(value, shapes, lsi, loc, w_sp, w_lv, d, out, mask, st, shapes_host, lsi_host,
                                  nullptr, nullptr, allow_dense, stats);
    }
    if (rc) return rc;
    if (!taken)
        launch_binning(flavour, loc, w_sp, d, plan, pl, pbuf, nullptr, st, kBinCount | kBinScan);
    rc = finish();
    if (rc == 0 && plan_built) *plan_built = 1;
    return rc;
}

}  // namespace


extern "C" {

#define DIMS Dims{B, S, H, C, L, Lq, P}
#define ST_ (hipStream_t) stream

int boxattn_fwd_train_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                          const float *loc, const float *attn, int B, int S, int H, int C, int L,
                          int Lq, int P, float *out, const int64_t *shapes_host,
                          const int64_t *lsi_host, void *plan, size_t plan_bytes, void *state,
                          size_t state_bytes, int hints, int *plan_built, void *stream)
{
    return launch_fwd_train<float, false>(value, shapes, lsi, loc, attn, nullptr, DIMS, out, nullptr,
                                          shapes_host, lsi_host, plan, plan_bytes, state, state_bytes, hints, plan_built, ST_);
}
int boxattn_fwd_train_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                           const float *loc, const float *attn, int B, int S, int H, int C, int L,
                           int Lq, int P, uint16_t *out, const int64_t *shapes_host,
                           const int64_t *lsi_host, void *plan, size_t plan_bytes, void *state,
                           size_t state_bytes, int hints, int *plan_built, void *stream)
{
    return launch_fwd_train<bf16_t, false>(value, shapes, lsi, loc, attn, nullptr, DIMS, out, nullptr,
                                           shapes_host, lsi_host, plan, plan_bytes, state, state_bytes, hints, plan_built, ST_);
}
int instattn_fwd_train_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                           const float *loc, const float *spatial_w, const float *level_w, int B,
                           int S, int H, int C, int L, int Lq, int P, float *out, float *mask_out,
                           const int64_t *shapes_host, const int64_t *lsi_host, void *plan,
                           size_t plan_bytes, void *state, size_t state_bytes, int hints, int *plan_built, void *stream)
{
    return launch_fwd_train<float, true>(value, shapes, lsi, loc, spatial_w, level_w, DIMS, out, mask_out,
                                         shapes_host, lsi_host, plan, plan_bytes, state, state_bytes, hints, plan_built, ST_);
}
int instattn_fwd_train_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                            const float *loc, const float *spatial_w, const float *level_w, int B,
                            int S, int H, int C, int L, int Lq, int P, uint16_t *out,
                            uint16_t *mask_out, const int64_t *shapes_host,
                            const int64_t *lsi_host, void *plan, size_t plan_bytes, void *state,
                            size_t state_bytes, int hints, int *plan_built, void *stream)
{
    return launch_fwd_train<bf16_t, true>(value, shapes, lsi, loc, spatial_w, level_w, DIMS, out, mask_out,
                                          shapes_host, lsi_host, plan, plan_bytes, state, state_bytes, hints, plan_built, ST_);
}

size_t boxattn_plan_bytes(int is_bf16, int B, int S, int H, int C, int L, int Lq, int P,
                          const int64_t *shapes_host, const int64_t *lsi_host)
{
    (void)is_bf16;
    const Dims d = DIMS;
    BinPlan plan;
    if (!d.valid() || !make_plan(d, shapes_host, lsi_host, plan)) return 0;
    return plan_layout(d, plan).total;
}

size_t boxattn_state_bytes(int B, int S, int H, int C, int L, int Lq, int P, const int64_t *shapes_host,
                           const int64_t *lsi_host)
{
    const Dims d = DIMS;
    BinPlan plan;
    const bool planned = d.valid() && make_plan(d, shapes_host, lsi_host, plan);
    return state_layout(d, planned ? &plan : nullptr).total;
}

size_t boxattn_bwd_workspace_bytes(int is_bf16, int B, int S, int H, int C, int L, int Lq, int P,
                                   const int64_t *shapes_host, const int64_t *lsi_host)
{
    const Dims d = DIMS;
    if (!d.valid()) return 0;
    const size_t fallback = is_bf16 ? align_up(d.n_value() * sizeof(float)) : 0;
    BinPlan plan;
    if (!make_plan(d, shapes_host, lsi_host, plan)) return fallback;
    return std::max(fallback, plan_layout(d, plan).total +
                                  scratch_layout(d, plan, wide_workspace(is_bf16 != 0, d)).total);
}

int boxattn_bwd_ws_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                       const float *loc, const float *attn, const float *grad_out, int B, int S,
                       int H, int C, int L, int Lq, int P, float *grad_value, float *grad_loc,
                       float *grad_attn, const int64_t *shapes_host, const int64_t *lsi_host,
                       void *workspace, size_t workspace_bytes, const void *plan, size_t plan_bytes, void *state,
        size_t state_bytes, int hints, void *stream)
{
    return launch_bwd_ws<float, false>(value, shapes, lsi, loc, attn, nullptr, grad_out, nullptr, DIMS,
                                       grad_value, grad_loc, grad_attn, nullptr, shapes_host, lsi_host,
                                       workspace, workspace_bytes, plan, plan_bytes, hints, ST_, state, state_bytes);
}
int boxattn_bwd_ws_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                        const float *loc, const float *attn, const uint16_t *grad_out, int B,
                        int S, int H, int C, int L, int Lq, int P, uint16_t *grad_value,
                        float *grad_loc, float *grad_attn, const int64_t *shapes_host,
                        const int64_t *lsi_host, void *workspace, size_t workspace_bytes,
                        const void *plan, size_t plan_bytes, void *state,
        size_t state_bytes, int hints, void *stream)
{
    return launch_bwd_ws<bf16_t, false>(value, shapes, lsi, loc, attn, nullptr, grad_out, nullptr, DIMS,
                                        grad_value, grad_loc, grad_attn, nullptr, shapes_host, lsi_host,
                                        workspace, workspace_bytes, plan, plan_bytes, hints, ST_, state, state_bytes);
}
int instattn_bwd_ws_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                        const float *loc, const float *spatial_w, const float *level_w,
                        const float *grad_out, const float *grad_mask, int B, int S, int H, int C,
                        int L, int Lq, int P, float *grad_value, float *grad_loc,
                        float *grad_spatial_w, float *grad_level_w, const int64_t *shapes_host,
                        const int64_t *lsi_host, void *workspace, size_t workspace_bytes,
                        const void *plan, size_t plan_bytes, void *state,
        size_t state_bytes, int hints, void *stream)
{
    return launch_bwd_ws<float, true>(value, shapes, lsi, loc, spatial_w, level_w, grad_out, grad_mask, DIMS,
                                      grad_value, grad_loc, grad_spatial_w, grad_level_w, shapes_host,
                                      lsi_host, workspace, workspace_bytes, plan, plan_bytes, hints, ST_, state, state_bytes);
}
int instattn_bwd_ws_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                         const float *loc, const float *spatial_w, const float *level_w,
                         const uint16_t *grad_out, const uint16_t *grad_mask, int B, int S, int H,
                         int C, int L, int Lq, int P, uint16_t *grad_value, float *grad_loc,
                         float *grad_spatial_w, float *grad_level_w, const int64_t *shapes_host,
                         const int64_t *lsi_host, void *workspace, size_t workspace_bytes,
                         const void *plan, size_t plan_bytes, void *state,
        size_t state_bytes, int hints, void *stream)
{
    return launch_bwd_ws<bf16_t, true>(value, shapes, lsi, loc, spatial_w, level_w, grad_out, grad_mask, DIMS,
                                       grad_value, grad_loc, grad_spatial_w, grad_level_w, shapes_host,
                                       lsi_host, workspace, workspace_bytes, plan, plan_bytes, hints, ST_, state, state_bytes);
}


int boxattn_abi_version(void) { return BOXATTN_ABI_VERSION; }

const char *boxattn_build_info(void)
{
    return "boxattn gfx950 (CDNA4, wave64) | hipcc " __VERSION__
           " | kernels: generic{f32,f64,bf16}, gather{f32 4ch/lane, bf16 8ch/lane} C={16,32,64}, "
           "window-staged encoder forward + point gradients{bf16 on MFMA 4x4x4, f32 on VALU}, "
           "binned-bwd{bf16 on MFMA 32x32x16, f32 on the same MFMA over exact three-term bf16 splits} "
           "with fill / combine riding in the point-gradient and accumulate launches, "
           "one-pass fill into guessed bin ranges (caller-owned state), box-grid{f32} | abi 8";
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

std::atomic<int> g_epoch{0};
int boxattn_options_epoch(void) { return g_epoch.load(std::memory_order_relaxed); }

int boxattn_set_variant(int variant)
{
    g_epoch.fetch_add(1, std::memory_order_relaxed);
    return g_variant.exchange(variant);
}

// debugging aid, not part of the documented ABI: a device buffer the window-staged point-gradient kernel fills
// with per-wave time stamps in builds with BOXATTN_DENSE_DEBUG (nullptr: off)
void boxattn_set_debug_buffer(float *p) { g_dense_dbg.store(p); }

int boxattn_set_option(int key, int value)
{
    if (key < 0 || key >= kNumOpts || !opt_live(key)) return -1;
    g_epoch.fetch_add(1, std::memory_order_relaxed);
    return g_opt[key].exchange(value);
}


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


}  // extern "C"
