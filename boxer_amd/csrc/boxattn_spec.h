// One-pass, speculative fill of the destination-binned backward (round 6).
//
// The two-pass binning (count -> scan -> fill, boxattn_binpass.h) locates every sample point twice because the
// fill pass needs every bin's first record slot before it writes the first record.  The reference has no such
// passes at all -- its backward is one launch of float atomics (box_attn_kernel.cuh:352-472).  Here the fill riders
// of the point-gradient launch write the records in ONE pass over the sampling locations, into bin ranges whose
// CAPACITIES were guessed from the previous call on the same caller-owned state buffer:
//
//   state (per stream and shape, zeroed once by the caller, kept by the caller between calls):
//     cbase[slice][nblk + 1]   first record slot of every block's range; the range of block k is
//                              [cbase[k], cbase[k + 1]), all ranges inside the slice's rec_cap slots;
//     cursor[slice][nblk]      the next free slot of every block; == cbase[k] between calls;
//     redo[slice][1 + nblk]    written by every call: how many blocks outgrew their range, and which.
//
//   fill rider, per step of THREADS x U x PT sample points (spec_fill_body):
//     rank    every (point, touched block) takes its rank inside the step from an LDS histogram (as the two-pass fill);
//     claim   every block the step touched gets its `count` slots with ONE returned global atomic on cursor[block]
//             (thread t speaks for the blocks t, t + THREADS, ...); a claim that would pass the end of the range is
//             remembered as "dropped";
//     store   record -> claimed base + rank.
//   the slice's LAST rider to finish (ticket, nobody waits) then knows every block's true record count:
//     spec_chain_body  writes the work-item list of the accumulate launch exactly as the scan of the two-pass
//             algorithm does (equal shares of the true counts), lists the blocks that outgrew their range in `redo`,
//             and re-plans the ranges for the NEXT call from this call's counts (twice the count + 64, decaying
//             slowly) -- cbase is rewritten in place, cursor reset.
//   accumulate launch: unchanged kernels.  REDO workers -- extra single-wave workgroups in front of its grid, which
//             exit at once when `redo` is empty -- recompute a block that outgrew its range from the sampling
//             locations themselves (redo_blocks: every point of the block's level is tested; slow, and exact).
//
// Results never depend on the guess: a range that is too small costs time (its block is recomputed by a redo worker
// and the next call's ranges follow the new counts), never accuracy; a zeroed state is simply a state whose every
// range is empty.  The host library sends the FIRST call on a zeroed state through the two-pass passes instead
// (cheaper than recomputing every block) and lets spec_layout_kernel derive the ranges from its exact scan.
//
// What it buys (C2 bf16): the count riders (4.9 M wave instructions) and the count -> scan chain leave
// the training forward's launch, the forward needs no plan, and the reference's four-function API needs no parked plan.
#pragma once
#include "boxattn_binpass.h"

namespace boxattn {

constexpr int kSpecDump = 256;                              // one dump counter per thread of a rider (see spec_fill_body)
constexpr int kSpecMaxBlocks = kRideMaxBlocks / 2 - kSpecDump - 1;   // two LDS arrays (ranks + dump counters / claimed bases) in the riders' histogram
constexpr int kSpecRedoWorkers = 64;                        // per slice, in front of the accumulate grid
constexpr int kSpecRankBits = 16;                           // a rank inside a step (< points of a step x 4) fits 16 bits
static_assert(kSpecMaxBlocks < (1 << 12), "block index bits of a packed candidate");

// LDS barrier of the spec riders: waits for this wave's LDS traffic only -- the next step's location loads (and the
// tiles' workgroups' traffic on the same CU) stay in flight across it, which __syncthreads()'s vmcnt(0) would drain.
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// new capacity of a block with c records whose range was cap_old slots: room for twice the count, never less than
// three quarters of what it had (layers of one shape alternate on one state), a multiple of 4 records = 64 bytes
__device__ __forceinline__ int spec_want(int c, int cap_old)
{
    return (max(2 * c + 64, cap_old - (cap_old >> 2)) + 3) & ~3;
}

// ---------------------------------------------------------------------------------------------------------------
// fill: one pass of bin workgroup `wg` of slice `s` over its (contiguous) queries; wide records only
// ---------------------------------------------------------------------------------------------------------------
#ifndef BOXATTN_TUNE_SPEC_U
#define BOXATTN_TUNE_SPEC_U 1
#endif
template <int THREADS, int PT>
__device__ __forceinline__ void spec_fill_body(int *hist, int *base, BinLevel *s_lv, const float *__restrict__ loc,
                                               const float *__restrict__ w_sp, const BinPlan &plan, int H, int Lq,
                                               int P, int q_per_wg, int *__restrict__ cursor_s,
                                               const int *__restrict__ cbase_s, int *__restrict__ records, int s, int wg,
                                               unsigned long long *trace = nullptr)
{
    static_assert(PT == 1 || PT == 4, "points per thread and group");
    constexpr int U = PT == 4 ? BOXATTN_TUNE_SPEC_U : 4;          // groups of PT points per thread and step
    constexpr int STRIDE = THREADS * U;
    static_assert(STRIDE * PT * 4 < (1 << kSpecRankBits), "ranks of a step");
    const int b = s / H, h = s % H;
    const int LP = plan.L * P;
    const int q0 = wg * q_per_wg;
    const int n_q = max(0, min(q0 + q_per_wg, Lq) - q0);
    // (32-bit point indices from uniform bases: the call has < 2^29 sample points, spec_ok)
    const unsigned pid0 = (unsigned)(((b * Lq + q0) * H + h) * LP);
    const unsigned qstride = (unsigned)(H * LP);
    const int LPG = LP / PT, n_grp = n_q * LPG;
    const float rcp_lpg = 1.0f / (float)LPG, rcp_p = 1.0f / (float)P;
    int4 *rec = reinterpret_cast<int4 *>(records) + (size_t)s * plan.rec_cap;
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
            const unsigned pbase = pid0 + (unsigned)t.ql[u] * qstride + (unsigned)t.lp0[u];
            const char *lp8 = reinterpret_cast<const char *>(loc) + (pbase << 3);
            const char *wp4 = reinterpret_cast<const char *>(w_sp) + (pbase << 2);
            if constexpr (PT == 4) {
                const float4 a = *reinterpret_cast<const float4 *>(lp8), c = *reinterpret_cast<const float4 *>(lp8 + 16);
                t.xy[u][0] = make_float2(a.x, a.y); t.xy[u][1] = make_float2(a.z, a.w);
                t.xy[u][2] = make_float2(c.x, c.y); t.xy[u][3] = make_float2(c.z, c.w);
                const float4 w4 = *reinterpret_cast<const float4 *>(wp4);
                t.wv[u][0] = w4.x; t.wv[u][1] = w4.y; t.wv[u][2] = w4.z; t.wv[u][3] = w4.w;
            } else {
                t.xy[u][0] = *reinterpret_cast<const float2 *>(lp8);
                t.wv[u][0] = *reinterpret_cast<const float *>(wp4);
            }
        }
    };
    // What a point keeps from the rank phase to the store phase: meta = first block | inside << 12 | crosses a block
    // column << 13 | crosses a block row << 14, and its <= 4 ranks, 16 bits each.  The footprint's blocks are b, b + 1
    // (next block column), b + nbx (next block row), b + nbx + 1: block indices step by at most one per pixel.
    constexpr unsigned kIn = 1u << 12, kCc = 1u << 13, kCr = 1u << 14;
    unsigned long long ph[4] = {0, 0, 0, 0};      // trace builds: time in rank / wait + claim / wait / store (thread 0)
#define SPEC_PHASE(i_, t0_)                                                                            \
    do {                                                                                             \
        if (BOXATTN_RIDE_TRACE) {                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                       \
            const unsigned long long now_ = __builtin_amdgcn_s_memrealtime();                        \
            ph[i_] += now_ - t0_;                                                                    \
            t0_ = now_;                                                                              \
        }                                                                                            \
    } while (0)
    auto work_step = [&](const Step &t, int g0) {
        unsigned meta[U][PT], rk01[U][PT], rk23[U][PT];
        int nbx_u[U];
        unsigned long long tph = BOXATTN_RIDE_TRACE ? __builtin_amdgcn_s_memrealtime() : 0ull;
        // ---- rank.  Three loops, so that the <= 4 PT LDS atomics of a group are all IN FLIGHT before the first rank is
        // used: one returned LDS atomic is ~100 cycles, and waited for one at a time (16 of them a step) they were a
        // third of the phase (rider trace: 1.75 us a step for ~300 instructions).
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool have = g0 + u * THREADS < n_grp;
            const BinLevel lv = s_lv[(int)(((float)t.lp0[u] + 0.5f) * rcp_p)];
            nbx_u[u] = lv.nbx;
            const float Hf = (float)lv.H, Wf = (float)lv.W;
            int b0[PT];
            bool in[PT], cc[PT], cr[PT];
#pragma unroll
            for (int k = 0; k < PT; ++k) {
                float h_im, w_im;
                {
#pragma clang fp contract(off)                   // two roundings, as in locate()
                    h_im = t.xy[u][k].y * Hf - 0.5f;
                    w_im = t.xy[u][k].x * Wf - 0.5f;
                }
                in[k] = have & (h_im > -1.f) & (w_im > -1.f) & (h_im < Hf) & (w_im < Wf) & (lv.H > 0) & (lv.W > 0);
                const int y0 = (int)floorf(in[k] ? h_im : 0.f), x0 = (int)floorf(in[k] ? w_im : 0.f);
                // block row / column of the first corner: floor(y nby / H) by multiply-high (blk_of); the second corner
                // y + 1 lies in the next block row iff the remainder y nby - row H reaches H - nby -- no second division
                const int ty = __mul24(max(y0, 0), lv.nby), tx = __mul24(max(x0, 0), lv.nbx);
                const int ra = (int)__umulhi((unsigned)ty, lv.mh), ca = (int)__umulhi((unsigned)tx, lv.mw);
                b0[k] = lv.blk0 + ra * lv.nbx + ca;
                // (bitwise, not short-circuit: no exec-masked regions around three instructions)
                cr[k] = in[k] & ((unsigned)y0 < (unsigned)(lv.H - 1)) & (ty - __mul24(ra, lv.H) + lv.nby >= lv.H);
                cc[k] = in[k] & ((unsigned)x0 < (unsigned)(lv.W - 1)) & (tx - __mul24(ca, lv.W) + lv.nbx >= lv.W);
            }
            // UNCONDITIONAL atomics: a candidate that does not exist counts in the thread's own dump counter (one per
            // thread, behind the histogram: no two lanes share one) -- a select instead of an exec-masked region, whose end
            // the compiler waits at for the returned rank
            unsigned r[PT][4];
            const int dump = kSpecMaxBlocks + 1 + (int)threadIdx.x;
#pragma unroll
            for (int k = 0; k < PT; ++k) {
                r[k][0] = (unsigned)atomicAdd(&hist[in[k] ? b0[k] : dump], 1);                      // LDS
                r[k][1] = (unsigned)atomicAdd(&hist[cc[k] ? b0[k] + 1 : dump], 1);
                r[k][2] = (unsigned)atomicAdd(&hist[cr[k] ? b0[k] + lv.nbx : dump], 1);
                r[k][3] = (unsigned)atomicAdd(&hist[cc[k] & cr[k] ? b0[k] + lv.nbx + 1 : dump], 1);
            }
#pragma unroll
            for (int k = 0; k < PT; ++k) {
                meta[u][k] = (in[k] ? (unsigned)b0[k] | kIn : 0u) | (cc[k] ? kCc : 0u) | (cr[k] ? kCr : 0u);
                rk01[u][k] = (r[k][0] & 0xFFFFu) | (r[k][1] << 16);      // (a dump counter only ever grows: keep 16 bits)
                rk23[u][k] = (r[k][2] & 0xFFFFu) | (r[k][3] << 16);
            }
        }
        SPEC_PHASE(0, tph);
        lds_barrier();
        // ---- claim: thread t speaks for the blocks t, t + THREADS, ...: ONE returned global atomic per touched block and
        // step, all of a thread's claims of a batch in flight at once (the rank-0 thread of a block claiming it saved this
        // scan of the histogram and serialised up to sixteen atomic round trips a wave: 117 us instead of 53)
        constexpr int KB = 2;
        for (int k0 = (int)threadIdx.x; k0 < plan.nblk; k0 += KB * THREADS) {
            int c[KB], old[KB], end[KB];
#pragma unroll
            for (int i = 0; i < KB; ++i) {
                const int bk = k0 + i * THREADS;
                c[i] = bk < plan.nblk ? hist[bk] : 0;
                old[i] = end[i] = 0;
            }
#pragma unroll
            for (int i = 0; i < KB; ++i)
                if (c[i] > 0) {
                    // (uniform base + 32-bit byte offset: address = a scalar pair and ONE vector register)
                    const unsigned bo = (unsigned)(k0 + i * THREADS) << 2;
                    old[i] = __hip_atomic_fetch_add(reinterpret_cast<int *>(reinterpret_cast<char *>(cursor_s) + bo), c[i],
                                                    __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    end[i] = *reinterpret_cast<const int *>(reinterpret_cast<const char *>(cbase_s) + bo + 4u);
                }
#pragma unroll
            for (int i = 0; i < KB; ++i)
                if (c[i] > 0) {
                    const int bk = k0 + i * THREADS;
                    hist[bk] = 0;                                             // ready for the next step
                    base[bk] = old[i] + c[i] <= end[i] ? old[i] : -1;         // -1: the range is full, the block will be redone
                }
        }
        SPEC_PHASE(1, tph);
        lds_barrier();
        SPEC_PHASE(2, tph);
        // ---- store (the slice's records are < 4 GB, spec_ok: 32-bit byte offsets from a uniform base).  The bases of a
        // group's points are all read before the first is used (as the ranks above).
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idq = (q0 + t.ql[u]) << plan.lp_bits;
            int sb[PT][4];
#pragma unroll
            for (int k = 0; k < PT; ++k) {
                // (the neighbours' bases are read whether the point crosses or not: two paired LDS reads, no masks; past the
                // last block -- and for points outside, block 0 -- they read something harmless inside the LDS array)
                const int b0 = (int)(meta[u][k] & 0xFFFu);
                sb[k][0] = base[b0]; sb[k][1] = base[b0 + 1];
                sb[k][2] = base[b0 + nbx_u[u]]; sb[k][3] = base[b0 + nbx_u[u] + 1];
            }
            char *rb8 = reinterpret_cast<char *>(rec);
#pragma unroll
            for (int k = 0; k < PT; ++k) {
                const unsigned m = meta[u][k];
                const int4 r = make_int4(idq | (t.lp0[u] + k), __float_as_int(t.xy[u][k].x),
                                         __float_as_int(t.xy[u][k].y), __float_as_int(t.wv[u][k]));
                if ((m & kIn) && sb[k][0] >= 0)
                    *reinterpret_cast<int4 *>(rb8 + ((unsigned)(sb[k][0] + (int)(rk01[u][k] & 0xFFFFu)) << 4)) = r;
                if ((m & kCc) && sb[k][1] >= 0)
                    *reinterpret_cast<int4 *>(rb8 + ((unsigned)(sb[k][1] + (int)(rk01[u][k] >> 16)) << 4)) = r;
                if ((m & kCr) && sb[k][2] >= 0)
                    *reinterpret_cast<int4 *>(rb8 + ((unsigned)(sb[k][2] + (int)(rk23[u][k] & 0xFFFFu)) << 4)) = r;
                if ((m & kCc) && (m & kCr) && sb[k][3] >= 0)
                    *reinterpret_cast<int4 *>(rb8 + ((unsigned)(sb[k][3] + (int)(rk23[u][k] >> 16)) << 4)) = r;
            }
        }
        SPEC_PHASE(3, tph);
    };
#ifndef BOXATTN_TUNE_SPEC_PREFETCH
#define BOXATTN_TUNE_SPEC_PREFETCH 1      // the next step's locations in flight while a step is worked (a second Step of registers)
#endif
    Step sa;
    if (n_grp > 0) load_step(sa, (int)threadIdx.x);
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < kMaxBinLevels; ++k) s_lv[k] = plan.lv[k];
    }
    for (int k = threadIdx.x; k < plan.nblk; k += THREADS) hist[k] = 0;
    __syncthreads();
    if constexpr (BOXATTN_TUNE_SPEC_PREFETCH != 0) {
        Step sb;
        for (int start = 0; start < n_grp; start += 2 * STRIDE) {           // workgroup-uniform trip count
            const int g0 = start + (int)threadIdx.x;
            if (start + STRIDE < n_grp) load_step(sb, g0 + STRIDE);
            work_step(sa, g0);
            if (start + STRIDE >= n_grp) break;
            if (start + 2 * STRIDE < n_grp) load_step(sa, g0 + 2 * STRIDE);
            work_step(sb, g0 + STRIDE);
        }
    } else {
        for (int start = 0; start < n_grp; start += STRIDE) {
            const int g0 = start + (int)threadIdx.x;
            work_step(sa, g0);
            if (start + STRIDE < n_grp) load_step(sa, g0 + STRIDE);
        }
    }
#undef SPEC_PHASE
    if (BOXATTN_RIDE_TRACE && trace && threadIdx.x == 0)
        for (int i = 0; i < 4; ++i) trace[i] = ph[i];
}

// ---------------------------------------------------------------------------------------------------------------
// chain: the slice's last rider -- true counts -> work items, redo list, the next call's ranges
// ---------------------------------------------------------------------------------------------------------------
// cnt / cbl: nblk (+ 1) ints of LDS each (the fill's two arrays); wsum: 4 x THREADS / 64 ints.
template <int THREADS>
__device__ __forceinline__ void spec_chain_body(const ScanOut o, const BinPlan &plan, const BinLevel *lv_lds, int s,
                                                int *wsum, int *cnt, int *cbl, int *__restrict__ cursor_s,
                                                int *__restrict__ cbase_s, int2 *__restrict__ redo_s,
                                                unsigned long long *__restrict__ stats)
{
    constexpr int NW = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nblk = plan.nblk;
    __syncthreads();
    for (int k = tid; k <= nblk; k += THREADS) {
        const int cb = cbase_s[k];
        cbl[k] = cb;
        if (k < nblk) cnt[k] = agent_load(cursor_s + k) - cb;      // (claimed by riders all over the chip: past the L1)
    }
    __syncthreads();
    const int per = (nblk + THREADS - 1) / THREADS;
    const int k_lo = min(tid * per, nblk), k_hi = min(k_lo + per, nblk);
    // two rounds of four running sums (the riders' LDS has 4 x NW ints for them):
    //   A  {wanted capacity, items, partial slots, chunked blocks}   B  {blocks to redo, records rounded up to 4}
    int sumA[4] = {0, 0, 0, 0}, sumB[4] = {0, 0, 0, 0};
    for (int k = k_lo; k < k_hi; ++k) {
        const int c = cnt[k], cap = cbl[k + 1] - cbl[k];
        const bool over = c > cap;
        const int nch = over ? 0 : max(1, (c + plan.chunk - 1) / plan.chunk);
        sumA[0] += spec_want(c, cap); sumA[1] += nch; sumA[2] += nch > 1 ? nch : 0; sumA[3] += nch > 1 ? 1 : 0;
        sumB[0] += over ? 1 : 0; sumB[1] += (c + 3) & ~3;
    }
    auto scan4 = [&](const int (&sum)[4], int (&run)[4], int (&tot)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int x = sum[i];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const int y = __shfl_up(x, d, 64);
                if (lane >= d) x += y;
            }
            run[i] = x - sum[i];
            if (lane == 63) wsum[i * NW + wv] = x;
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int pre = 0, all = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                pre += w < wv ? wsum[i * NW + w] : 0;
                all += wsum[i * NW + w];
            }
            run[i] += pre;
            tot[i] = all;
        }
        __syncthreads();
    };
    int runA[4], totA[4], runB[4], totB[4];
    scan4(sumA, runA, totA);
    scan4(sumB, runB, totB);
    // the next call's ranges: what every block wants, if the slice has room for it; else the records themselves and
    // an equal share of what is left (ranges never reach past rec_cap: a block that gets no room is redone)
    const bool roomy = totA[0] <= plan.rec_cap;
    // (float quotient, rounded down twice: a uniform integer division is ~40 scalar instructions and their registers)
    const int share = roomy ? 0 : max(0, (int)((float)max(0, plan.rec_cap - totB[1]) / (float)max(nblk, 1)) - 4) & ~3;
    int next = roomy ? runA[0] : runB[1] + k_lo * share;
    int n_over = runB[0];
    for (int k = k_lo; k < k_hi; ++k) {
        const int c = cnt[k], cap = cbl[k + 1] - cbl[k];
        const bool over = c > cap;
        const int nch = over ? 0 : max(1, (c + plan.chunk - 1) / plan.chunk);
        int level = 0;
        for (int l = 1; l < plan.L; ++l)
            if (k >= lv_lds[l].blk0) level = l;
        const BinLevel lv = lv_lds[level];
        const int geo = (int)pack_block_geo(lv, level, k);
        const int csz = chunk_records(c, nch);
        for (int jj = 0; jj < nch; ++jj)               // heaviest first, as the two-pass scan lists them
            o.items[(size_t)s * plan.item_cap + (totA[1] - 1 - (runA[1] + jj))] =
                make_int4(geo, cbl[k] + jj * csz, cbl[k] + min(c, (jj + 1) * csz),
                          nch > 1 ? (runA[2] + jj) | (runA[3] << kItemSlotBits) : -1);
        if (nch > 1) o.combos[(size_t)s * nblk + runA[3]] = make_int4(geo, runA[2], nch, 0);
        if (over) redo_s[1 + n_over++] = make_int2(k, geo);
        runA[1] += nch; runA[2] += nch > 1 ? nch : 0; runA[3] += nch > 1 ? 1 : 0;
        const int first = min(next, plan.rec_cap);
        cbase_s[k] = first;
        cursor_s[k] = first;
        next += roomy ? spec_want(c, cap) : ((c + 3) & ~3) + share;
    }
    if (k_hi == nblk && k_lo < nblk) cbase_s[nblk] = min(next, plan.rec_cap);      // (the thread of the last block)
    if (tid == 0) {
        o.n_items[2 * s] = totA[1];
        o.n_items[2 * s + 1] = totA[3];
        redo_s[0] = make_int2(totB[0], 0);
        if (stats) {
            atomicAdd(stats, 1ull);
            if (totB[0]) atomicAdd(stats + 1, (unsigned long long)totB[0]);
        }
    }
}

// A cold state (the host library's decision: zeroed by the caller, or re-initialised for another shape): the call ran
// the two-pass passes, whose exact scan left every block's first record in `offsets`; the ranges of the NEXT call
// follow from the counts.  grid = slices, block 256.  (A template: this header is part of two translation units.)
template <int THREADS>
__global__ __launch_bounds__(THREADS) void spec_layout_kernel(BinPlan plan, const int *__restrict__ offsets,
                                                          int *__restrict__ cursor, int *__restrict__ cbase,
                                                          int2 *__restrict__ redo)
{
    static_assert(THREADS == 256, "4 wave sums per quantity");
    __shared__ int wsum[2 * 4];
    const int s = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nblk = plan.nblk;
    const int *off = offsets + (size_t)s * (nblk + 1);
    int *cursor_s = cursor + (size_t)s * nblk, *cbase_s = cbase + (size_t)s * (nblk + 1);
    const int per = (nblk + 255) / 256;
    const int k_lo = min(tid * per, nblk), k_hi = min(k_lo + per, nblk);
    int sum[2] = {0, 0};
    for (int k = k_lo; k < k_hi; ++k) {
        const int c = off[k + 1] - off[k];
        sum[0] += spec_want(c, 0);
        sum[1] += (c + 3) & ~3;
    }
    int run[2], tot[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int x = sum[i];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        run[i] = x - sum[i];
        if (lane == 63) wsum[i * 4 + wv] = x;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int pre = 0, all = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            pre += w < wv ? wsum[i * 4 + w] : 0;
            all += wsum[i * 4 + w];
        }
        run[i] += pre;
        tot[i] = all;
    }
    const bool roomy = tot[0] <= plan.rec_cap;
    const int share = roomy ? 0 : max(0, (int)((float)max(0, plan.rec_cap - tot[1]) / (float)max(nblk, 1)) - 4) & ~3;
    int next = roomy ? run[0] : run[1] + k_lo * share;
    for (int k = k_lo; k < k_hi; ++k) {
        const int c = off[k + 1] - off[k];
        const int first = min(next, plan.rec_cap);
        cbase_s[k] = first;
        cursor_s[k] = first;
        next += roomy ? spec_want(c, 0) : ((c + 3) & ~3) + share;
    }
    if (k_hi == nblk && k_lo < nblk) cbase_s[nblk] = min(next, plan.rec_cap);
    if (tid == 0) redo[(size_t)s * (nblk + 1)] = make_int2(0, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// fill rider (the counterpart of bin_fill_ride for BinRide::flavour & kRideSpec)
// ---------------------------------------------------------------------------------------------------------------
template <int THREADS>
__device__ __forceinline__ void bin_fill_spec_ride(const BinRide r, unsigned id, int *lds)
{
    static_assert(THREADS == 256, "4 wave sums per scan quantity");
    const RideLds m(lds);
    int *hist = m.hist, *base = m.hist + kRideMaxBlocks / 2;      // hist: nblk counters + kSpecDump dump counters
    const BinPlan plan = r.plan;
    unsigned s_u, wg_u;
    divmod_magic(id, (unsigned)r.n_wg, r.nwg_magic, s_u, wg_u);
    const int s = (int)s_u, wg = (int)wg_u;
    if (wg == 0 && r.ctickets)
        for (int k = threadIdx.x; k < plan.nblk; k += THREADS) r.ctickets[(size_t)s * plan.nblk + k] = 0;
    int *cursor_s = r.spec.cursor + (size_t)s * plan.nblk;
    int *cbase_s = r.spec.cbase + (size_t)s * (plan.nblk + 1);
    RIDE_STAMP(5);
    unsigned long long *tr = BOXATTN_RIDE_TRACE && r.trace ? r.trace + (size_t)id * 8 : nullptr;     // [0..3]: phase times
    if (r.flavour & kRidePt4)
        spec_fill_body<THREADS, 4>(hist, base, m.lv, r.loc, r.w_sp, plan, r.H, r.Lq, r.P, r.q_per_wg, cursor_s, cbase_s,
                                   r.records, s, wg, tr);
    else
        spec_fill_body<THREADS, 1>(hist, base, m.lv, r.loc, r.w_sp, plan, r.H, r.Lq, r.P, r.q_per_wg, cursor_s, cbase_s,
                                   r.records, s, wg, tr);
    RIDE_STAMP(6);
    // every claim of this workgroup has returned (its value was used); the ticket orders the riders of the slice
    if (!last_arriver<THREADS>(r.tickets + (size_t)s * kRideTickets + kScanSub, r.n_wg, m.flag)) return;
    const ScanOut o{r.subtot, r.offsets, r.items, r.combos, r.n_items};
    spec_chain_body<THREADS>(o, plan, m.lv, s, m.wsum, hist, base, cursor_s, cbase_s,
                             r.spec.redo + (size_t)s * (plan.nblk + 1), r.spec.stats);
    RIDE_STAMP(7);
}

// ---------------------------------------------------------------------------------------------------------------
// redo: one wave recomputes the grad_value rows of a block from the sampling locations (accumulate launch)
// ---------------------------------------------------------------------------------------------------------------
// Lane = (pixel of the block, channel half).  Every sample point of the block's level is tested (64 a step); a point
// whose 2x2 footprint meets the block is broadcast and every lane whose pixel is one of its corners adds weight x
// upstream row -- the reference's sum (box_attn_kernel.cuh:100-184) gathered by the destination, float32 order.
template <typename ST, int C>
__device__ __forceinline__ void redo_blocks(const ZeroRole zr, const BinPlan &plan, int s, int zw, int n_zw, int S, int H,
                                            int Lq, const ST *__restrict__ grad_out, ST *__restrict__ grad_value, int lane)
{
    constexpr int BW = 8, CH = C / 2, EPL = 16 / (int)sizeof(ST);
    const int2 *redo_s = zr.redo + (size_t)s * (plan.nblk + 1);
    const int n = redo_s[0].x;
    if (zw >= n) return;
    const int b = s / H, h = s % H, P = zr.P, LP = plan.L * P;
    const float2 *loc2 = reinterpret_cast<const float2 *>(zr.loc);
    const int mypix = lane >> 1, half = lane & 1;
    const int np = Lq * P;
    const float rcp_p = 1.0f / (float)P;
    for (int i = zw; i < n; i += n_zw) {
        const BlockGeo bg = unpack_block_geo((unsigned)redo_s[1 + i].y);
        int lvH = plan.lv[0].H, lvW = plan.lv[0].W, lv_start = plan.lv[0].start;
#pragma unroll
        for (int k = 1; k < kMaxBinLevels; ++k)
            if (k == bg.level) { lvH = plan.lv[k].H; lvW = plan.lv[k].W; lv_start = plan.lv[k].start; }
        const float Hf = (float)lvH, Wf = (float)lvW;
        const int Y = bg.oy + mypix / BW, X = bg.ox + mypix % BW;
        const bool live = mypix / BW < bg.bh && mypix % BW < bg.bw;
        float acc[CH];
#pragma unroll
        for (int c = 0; c < CH; ++c) acc[c] = 0.f;
        for (int i0 = 0; i0 < np; i0 += 64) {
            const bool have = i0 + lane < np;
            int q, p;
            divmod_small(have ? i0 + lane : 0, P, rcp_p, q, p);
            const size_t pid = (((size_t)b * Lq + q) * H + h) * LP + (size_t)bg.level * P + p;
            const float2 xy = loc2[pid];
            float h_im, w_im;
            {
#pragma clang fp contract(off)                   // two roundings, as in locate()
                h_im = xy.y * Hf - 0.5f;
                w_im = xy.x * Wf - 0.5f;
            }
            const bool inside = h_im > -1.f && w_im > -1.f && h_im < Hf && w_im < Wf;
            const int y0 = (int)floorf(inside ? h_im : 0.f), x0 = (int)floorf(inside ? w_im : 0.f);
            // (corners outside the map belong to no block, so the interval test needs no clamping)
            const bool hit = have && inside && y0 + 1 >= bg.oy && y0 < bg.oy + bg.bh && x0 + 1 >= bg.ox && x0 < bg.ox + bg.bw;
            const float a = hit ? zr.w_sp[pid] : 0.f;
            unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
            while (mask) {
                const int j = (int)__builtin_ctzll(mask);            // wave-uniform
                mask &= mask - 1;
                const float hj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, h_im), j));
                const float wj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, w_im), j));
                const float aj = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, a), j));
                const int qj = __builtin_amdgcn_readlane(q, j);
                const float yf = floorf(hj), xf = floorf(wj);
                const float lh = hj - yf, lw = wj - xf, hh = 1.f - lh, hw = 1.f - lw;
                const int yj = (int)yf, xj = (int)xf;
                const bool corner = (Y == yj || Y == yj + 1) && (X == xj || X == xj + 1);
                const float w = (Y == yj ? hh : lh) * (X == xj ? hw : lw) * aj;
                const ST *g = grad_out + (((size_t)b * Lq + qj) * H + h) * C + half * CH;
#pragma unroll
                for (int c = 0; c < CH; c += EPL) {
                    float t[EPL];
                    VecIO<ST, EPL>::ld(g + c, t);
                    if (corner) {
#pragma unroll
                        for (int e = 0; e < EPL; ++e) acc[c + e] = fmaf(w, t[e], acc[c + e]);
                    }
                }
            }
        }
        if (live) {
            ST *dst = grad_value + (((size_t)b * S + lv_start + (size_t)Y * lvW + X) * H + h) * C + half * CH;
#pragma unroll
            for (int c = 0; c < CH; c += EPL) {
                float t[EPL];
#pragma unroll
                for (int e = 0; e < EPL; ++e) t[e] = acc[c + e];
                VecIO<ST, EPL>::st(dst + c, t);
            }
        }
    }
}

}  // namespace boxattn
