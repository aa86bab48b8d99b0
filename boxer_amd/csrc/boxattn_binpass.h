// The two passes over the sampling locations of the destination-binned backward (boxattn_binned.h steps
// 1 and 3: count, fill) as device functions: the stand-alone bin_kernel calls them with 512 threads, the
// riders (boxattn_ride.h) with the 256 threads of the kernel they ride in.  A header of its own: the
// window-staged encoder kernels (boxattn_dense.hip, a separate translation unit) carry riders too.
#pragma once
#include "boxattn_binplan.h"
#include "boxattn_ride.h"
#include "boxattn_scan_tail.h"

namespace boxattn {

// One pass of workgroup `wg` of slice `s` over its queries' sample points: every point is assigned to the
// <= 4 blocks its 2x2 footprint touches (touched_blocks).
//   count (!FILL): per-block counts in the LDS histogram `hist` (left there; the caller writes them out);
//   fill  (FILL) : hist starts at the workgroup's first slot in every bin (part + subtot + offsets), each
//                  point's record goes to the slot an LDS atomic hands out.
// WIDE records (bf16 box attention, matrix-core accumulate): {point id, x, y, attention weight} instead of
// the id alone, so that the accumulate kernel reads everything but the upstream row from its (coalesced)
// record stream instead of gathering two more 128-byte lines per record.
// PT = 4 (P % 4 == 0, 16-byte aligned tensors): a thread takes four consecutive points of one (query,
// level) with two 16-byte loads and one level lookup (count pass 13.5 -> 11.7 us); PT = 1: any P.
// interleave (VALU accumulate kernel): workgroup w takes the queries w, w + n_wg, w + 2 n_wg, ...: every
// workgroup's records are then a uniform sample of the map, and so is any run of consecutive records of a
// bin.  That kernel works through a bin 64 records at a time with one lane per destination pixel; with
// contiguous query ranges a round's records came from neighbouring queries and piled up on a few pixels
// (longest per-pixel list 3.3x the mean; interleaved 2.1x; accumulate kernel 133 -> 103 us).
// Matrix-core accumulate (a dense product, indifferent to the order): contiguous query ranges, whose
// records land in few bins, in runs -- 27 -> 24 us for the fill pass.
// No global atomics anywhere in the binning (they cost ~20 us per pass: ~200 k single-lane atomics on 226
// cache lines), and the record order is deterministic.
template <int THREADS, int BW, int BH, bool FILL, bool WIDE, int PT>
__device__ __forceinline__ void bin_pass_body(int *hist, BinLevel *s_lv, const float *__restrict__ loc,
                                              const float *__restrict__ w_sp, const BinPlan &plan, int H,
                                              int Lq, int P, int q_per_wg, int n_wg, bool interleave,
                                              const int *__restrict__ part, const int *__restrict__ subtot,
                                              const int *__restrict__ offsets, int *__restrict__ records,
                                              int s, int wg)
{
    const int b = s / H, h = s % H;
    const int LP = plan.L * P;
    const int q0 = interleave ? wg : wg * q_per_wg, qstep = interleave ? n_wg : 1;
    const int n_q = interleave ? (q0 < Lq ? (Lq - q0 + qstep - 1) / qstep : 0)
                               : max(0, min(q0 + q_per_wg, Lq) - q0);
    const int *mypart = part + ((size_t)s * n_wg + wg) * plan.nblk;

    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);
    static_assert(PT == 1 || PT == 4, "points per thread and step");
#ifndef BOXATTN_TUNE_BIN_U
#define BOXATTN_TUNE_BIN_U 2
#endif
    constexpr int U = PT == 4 ? BOXATTN_TUNE_BIN_U : 4;     // groups of PT points per thread per step (loads in flight)
    const size_t pid0 = (((size_t)b * Lq + q0) * H + h) * LP;     // first point of query q0
    const size_t qstride = (size_t)H * LP * qstep;           // points between this WG's queries
    const int LPG = LP / PT, n_grp = n_q * LPG;               // groups per query, groups of this WG
    const float rcp_lpg = 1.0f / (float)LPG, rcp_p = 1.0f / (float)P;
    int *rec = records + (size_t)s * plan.rec_cap * (WIDE ? 4 : 1);
    // one step's points of this thread (clamped: every thread loads from valid addresses)
    struct Step {
        float2 xy[U][PT];
        float wv[U][PT];
        int lp0[U], ql[U];
    };
    auto load_step = [&](Step &t, int g0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int g = min(g0 + u * THREADS, n_grp - 1);
            int lg;
            divmod_small(g, LPG, rcp_lpg, t.ql[u], lg);
            t.lp0[u] = lg * PT;
            const size_t base = pid0 + t.ql[u] * qstride + t.lp0[u];
            if constexpr (PT == 4) {
                const float4 *p4 = reinterpret_cast<const float4 *>(loc2 + base);
                const float4 a = p4[0], c = p4[1];
                t.xy[u][0] = make_float2(a.x, a.y); t.xy[u][1] = make_float2(a.z, a.w);
                t.xy[u][2] = make_float2(c.x, c.y); t.xy[u][3] = make_float2(c.z, c.w);
                if constexpr (FILL && WIDE) {
                    const float4 w4 = *reinterpret_cast<const float4 *>(w_sp + base);
                    t.wv[u][0] = w4.x; t.wv[u][1] = w4.y; t.wv[u][2] = w4.z; t.wv[u][3] = w4.w;
                }
            } else {
                t.xy[u][0] = loc2[base];
                if constexpr (FILL && WIDE) t.wv[u][0] = w_sp[base];
            }
            if constexpr (!(FILL && WIDE)) {
#pragma unroll
                for (int k = 0; k < PT; ++k) t.wv[u][k] = 0.f;
            }
        }
    };
    // one record of point (u, k) into slot `slot` of its bin
    auto put_record = [&](const Step &t, int u, int k, int slot) {
        const int id = ((q0 + t.ql[u] * qstep) << plan.lp_bits) | (t.lp0[u] + k);
        if constexpr (WIDE)
            reinterpret_cast<int4 *>(rec)[slot] = make_int4(id, __float_as_int(t.xy[u][k].x),
                                                            __float_as_int(t.xy[u][k].y), __float_as_int(t.wv[u][k]));
        else
            rec[slot] = id;
    };
    auto work_step = [&](const Step &t, int g0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (g0 + u * THREADS >= n_grp) break;
            const BinLevel lv = s_lv[(int)(((float)t.lp0[u] + 0.5f) * rcp_p)];   // level = lp / P
            // (One LDS atomic per touched block and (query, level) GROUP -- adding how many of its four points have a
            // record there -- instead of one per point and block was built and measured: the 2 x 2 block bookkeeping
            // costs more vector instructions than the atomics it saves; a fat rider's count pass 21 -> 26 us.)
#pragma unroll
            for (int k = 0; k < PT; ++k) {
                int blk[4];
                touched_blocks(t.xy[u][k].x, t.xy[u][k].y, lv, blk);
                // predicated, not redirected to a dump slot: same-address LDS atomics serialise
                // per lane, a shared dump slot made this kernel 1.6x slower
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (blk[j] >= 0) {
                        const int slot = atomicAdd(&hist[blk[j]], 1);            // LDS
                        if constexpr (FILL) put_record(t, u, k, slot);
                    }
                }
            }
        }
    };
    // The first step's loads go out BEFORE the level table / histogram set-up and its barrier: the
    // set-up's own round trip (the fill pass reads three tables per block) then runs under theirs
    // instead of in front of it.
    constexpr int STRIDE = THREADS * U;
    Step sa, sb;                                       // two static buffers: a rotating one would MOVE registers
    if (n_grp > 0) load_step(sa, (int)threadIdx.x);    // with loads in flight, i.e. wait for them
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < kMaxBinLevels; ++k) s_lv[k] = plan.lv[k];
    }
    const int wps = scan_wps(n_wg);                // as in the scan
    const int *mysub = subtot + ((size_t)s * kScanSub + wg / wps) * plan.nblk;
    for (int k = threadIdx.x; k < plan.nblk; k += THREADS)
        hist[k] = FILL ? mypart[k] + mysub[k] + offsets[(size_t)s * (plan.nblk + 1) + k] : 0;
    __syncthreads();
    // A workgroup with several steps (few, fat bin workgroups: the riders) keeps the NEXT step's
    // locations in flight while it ranks the current one's: a serial chain of round trips otherwise.
    for (int base = 0; base < n_grp; base += 2 * STRIDE) {          // workgroup-uniform trip count
        const int g0 = base + (int)threadIdx.x;
        if (base + STRIDE < n_grp) load_step(sb, g0 + STRIDE);
        work_step(sa, g0);
        if (base + STRIDE >= n_grp) break;
        if (base + 2 * STRIDE < n_grp) load_step(sa, g0 + 2 * STRIDE);
        work_step(sb, g0 + STRIDE);
    }
    if constexpr (!FILL) __syncthreads();            // the histogram is complete
}

// ---------------------------------------------------------------------------------------
// riders
// ---------------------------------------------------------------------------------------
enum { kRideWide = 1, kRideInterleave = 2, kRidePt4 = 4,
       kRideSpec = 8 };     // one-pass fill into guessed bin ranges (boxattn_spec.h); wide records, contiguous queries
constexpr int kRideMaxBlocks = 3072;      // blocks per slice the riders' LDS histogram is built for (12 KB; beyond kScanThreads the
                                          // slice's last arriver scans in two passes over it: scan_blocks_big_body)
constexpr int kRideLdsInts = kRideMaxBlocks + 8 * (int)(sizeof(BinLevel) / sizeof(int)) + 4 * 4 + 2;

// What a rider needs: the arguments of bin_kernel + the scan's outputs.  Passed BY VALUE all the way (a
// kernel-argument struct handed on by reference was copied to scratch at the top of every wave of the host
// kernel).
// The caller's state of the one-pass fill (boxattn_spec.h): per slice the bins' ranges, their cursors, the redo list
struct SpecRide {
    int *cursor;                 // [slice][nblk]      next free record slot of every block (== cbase between calls)
    int *cbase;                  // [slice][nblk + 1]  first slot of every block's range
    int2 *redo;                  // [slice][1 + nblk]  {blocks that outgrew their range, 0}, then {block, geometry} each
    unsigned long long *stats;   // [2] calls, blocks redone (only ever added to)
};
struct BinRide {
    const float *loc, *w_sp;
    int *part, *subtot, *offsets, *records;
    int4 *items, *combos;
    int *n_items;
    int *tickets;            // count: [slice][kRideTickets], zero on entry, left zero
    int *ctickets;           // fill: [slice][nblk] the accumulate launch's combine tickets, zeroed here (or null)
    BinPlan plan;
    int H, Lq, P, q_per_wg, n_wg;
    unsigned nwg_magic;      // floor(2^32 / n_wg) (0xFFFFFFFF for 1): rider id -> (slice, bin workgroup) without a division
    int flavour;             // kRideWide | kRideInterleave | kRidePt4
    RideGrid grid;           // grid.n_riders == 0: no riders in this launch
    SpecRide spec;           // flavour & kRideSpec
    unsigned long long *trace;   // builds with BOXATTN_RIDE_TRACE: 8 s_memrealtime stamps per rider (tools/gpu_ride_trace.py)
};
#ifndef BOXATTN_RIDE_TRACE
#define BOXATTN_RIDE_TRACE 0
#endif
#define RIDE_STAMP(k_)                                                                              \
    do {                                                                                            \
        if (BOXATTN_RIDE_TRACE && r.trace && threadIdx.x == 0) {                                    \
            __builtin_amdgcn_sched_barrier(0);                                                      \
            r.trace[(size_t)id * 8 + (k_)] = __builtin_amdgcn_s_memrealtime();                      \
        }                                                                                           \
    } while (0)

// LDS of a rider: [histogram: kRideMaxBlocks ints][level table][4 x 4 wave sums][flag]
struct RideLds {
    int *hist;
    BinLevel *lv;
    int *wsum;
    int *flag;
    __device__ __forceinline__ explicit RideLds(int *base)
        : hist(base), lv(reinterpret_cast<BinLevel *>(base + kRideMaxBlocks)),
          wsum(base + kRideMaxBlocks + 8 * (int)(sizeof(BinLevel) / sizeof(int))),
          flag(base + kRideMaxBlocks + 8 * (int)(sizeof(BinLevel) / sizeof(int)) + 16) {}
};
static_assert(kMaxBinLevels == 8, "RideLds reserves 8 levels");

// Count rider `id` = (slice, bin workgroup) and whatever scan stages it turns out to be the last arriver of.
template <int THREADS>
__device__ __forceinline__ void bin_count_ride(const BinRide r, unsigned id, int *lds)
{
    static_assert(THREADS == 256, "4 wave sums per scan quantity");
    if (id >= r.grid.n_riders) return;
#ifndef BOXATTN_TUNE_RIDE_PRIO
#define BOXATTN_TUNE_RIDE_PRIO 0
#endif
    if (BOXATTN_TUNE_RIDE_PRIO) __builtin_amdgcn_s_setprio(BOXATTN_TUNE_RIDE_PRIO);
    RIDE_STAMP(0);
    const RideLds m(lds);
    const BinPlan plan = r.plan;
    unsigned s_u, wg_u;
    divmod_magic(id, (unsigned)r.n_wg, r.nwg_magic, s_u, wg_u);
    const int s = (int)s_u, wg = (int)wg_u;
    const bool inter = (r.flavour & kRideInterleave) != 0;
    if (r.flavour & kRidePt4)
        bin_pass_body<THREADS, 8, 4, false, false, 4>(m.hist, m.lv, r.loc, r.w_sp, plan, r.H, r.Lq, r.P, r.q_per_wg,
                                                     r.n_wg, inter, r.part, r.subtot, r.offsets, r.records, s, wg);
    else
        bin_pass_body<THREADS, 8, 4, false, false, 1>(m.hist, m.lv, r.loc, r.w_sp, plan, r.H, r.Lq, r.P, r.q_per_wg,
                                                     r.n_wg, inter, r.part, r.subtot, r.offsets, r.records, s, wg);
    RIDE_STAMP(1);
    // publish my row of counts, then the chain of last arrivers
    int *mypart = r.part + ((size_t)s * r.n_wg + wg) * plan.nblk;
    for (int k = threadIdx.x; k < plan.nblk; k += THREADS) agent_store(mypart + k, m.hist[k]);
    stores_left();
    const int wps = scan_wps(r.n_wg), u = wg / wps;
    const int n_in_sub = min(r.n_wg, (u + 1) * wps) - u * wps, n_sub = (r.n_wg + wps - 1) / wps;
    int *tk = r.tickets + (size_t)s * kRideTickets;
    const bool last = last_arriver<THREADS>(tk + u, n_in_sub, m.flag);
    RIDE_STAMP(2);
    if (!last) return;       // workgroup-uniform
    const ScanOut o{r.subtot, r.offsets, r.items, r.combos, r.n_items};
    if (n_sub == 1) {        // one sub-range: its last arriver is the slice's, and does both scan stages in one pass
        RIDE_STAMP(3);
        if (plan.nblk <= kScanThreads) scan_blocks_body<THREADS>(o, plan, m.lv, 1, s, m.wsum, r.part, r.n_wg);
        else scan_blocks_big_body<THREADS>(o, plan, m.lv, 1, s, m.wsum, m.hist, r.part, r.n_wg);
    } else {
        scan_sub_body<THREADS>(r.part, r.subtot, plan, r.n_wg, s, u);
        stores_left();
        RIDE_STAMP(3);
        if (!last_arriver<THREADS>(tk + kScanSub, n_sub, m.flag)) return;
        if (plan.nblk <= kScanThreads) scan_blocks_body<THREADS>(o, plan, m.lv, n_sub, s, m.wsum);
        else scan_blocks_big_body<THREADS>(o, plan, m.lv, n_sub, s, m.wsum, m.hist);
    }
    RIDE_STAMP(4);
}

template <int THREADS> __device__ __forceinline__ void bin_fill_spec_ride(const BinRide r, unsigned id, int *lds);   // boxattn_spec.h

// Fill rider `id` = (slice, bin workgroup); bin workgroup 0 of a slice also clears the slice's combine tickets.
template <int THREADS>
__device__ __forceinline__ void bin_fill_ride(const BinRide r, unsigned id, int *lds)
{
    if (id >= r.grid.n_riders) return;
    if (BOXATTN_TUNE_RIDE_PRIO) __builtin_amdgcn_s_setprio(BOXATTN_TUNE_RIDE_PRIO);
    if (r.flavour & kRideSpec) {        // one pass into guessed ranges, then the chain of the slice's last rider
        bin_fill_spec_ride<THREADS>(r, id, lds);
        return;
    }
    RIDE_STAMP(5);
    const RideLds m(lds);
    const BinPlan plan = r.plan;
    unsigned s_u, wg_u;
    divmod_magic(id, (unsigned)r.n_wg, r.nwg_magic, s_u, wg_u);
    const int s = (int)s_u, wg = (int)wg_u;
    const bool inter = (r.flavour & kRideInterleave) != 0;
    if (wg == 0 && r.ctickets)
        for (int k = threadIdx.x; k < plan.nblk; k += THREADS) r.ctickets[(size_t)s * plan.nblk + k] = 0;
    if ((r.flavour & kRideWide) && (r.flavour & kRidePt4))
        bin_pass_body<THREADS, 8, 4, true, true, 4>(m.hist, m.lv, r.loc, r.w_sp, plan, r.H, r.Lq, r.P, r.q_per_wg,
                                                   r.n_wg, inter, r.part, r.subtot, r.offsets, r.records, s, wg);
    else if (r.flavour & kRideWide)
        bin_pass_body<THREADS, 8, 4, true, true, 1>(m.hist, m.lv, r.loc, r.w_sp, plan, r.H, r.Lq, r.P, r.q_per_wg,
                                                   r.n_wg, inter, r.part, r.subtot, r.offsets, r.records, s, wg);
    else      // 4-byte records: one point per thread keeps neighbouring lanes on neighbouring slots (18.9 against 23.0 us)
        bin_pass_body<THREADS, 8, 4, true, false, 1>(m.hist, m.lv, r.loc, r.w_sp, plan, r.H, r.Lq, r.P, r.q_per_wg,
                                                    r.n_wg, inter, r.part, r.subtot, r.offsets, r.records, s, wg);
    if (BOXATTN_RIDE_TRACE) { stores_left(); __syncthreads(); }
    RIDE_STAMP(6);
}

}  // namespace boxattn

#include "boxattn_spec.h"
