// The block scans of the destination-binned backward (boxattn_binned.h step 2) as a device function that rides
// in ANOTHER kernel's launch -- the training forward's (fwd2_kernel, fwd_dense_kernel).  A header of its own:
// the window-staged forward lives in a separate translation unit (boxattn_dense.hip).
#pragma once
#include "boxattn_device.h"
#include "boxattn_combine.h"
#include "boxattn_binplan.h"

namespace boxattn {

// Origin, extent and level of a block in one word (computed once per block by the scan kernel:
// five integer divisions that every work item used to repeat -- 5 % of the accumulate kernel).
//   bits 0-11 oy, 12-23 ox, 24-25 bh - 1, 26-28 bw - 1, 29-31 level   (maps < 4096 x 4096)
__device__ __forceinline__ unsigned pack_block_geo(const BinLevel &lv, int level, int blk)
{
    const int by = (blk - lv.blk0) / lv.nbx, bx = (blk - lv.blk0) % lv.nbx;
    const int oy = blk_lo(by, lv.H, lv.nby), ox = blk_lo(bx, lv.W, lv.nbx);
    const int bh = blk_lo(by + 1, lv.H, lv.nby) - oy, bw = blk_lo(bx + 1, lv.W, lv.nbx) - ox;
    return (unsigned)oy | ((unsigned)ox << 12) | ((unsigned)(bh - 1) << 24) |
           ((unsigned)(bw - 1) << 26) | ((unsigned)level << 29);
}
constexpr int kScanSub = 8, kScanWgPerSub = 16;   // bin_scan_a_kernel: sub-ranges of workgroups
constexpr int kScanThreads = 1024;     // one workgroup per slice walks the blocks 1024 at a time

// The two scan kernels inside ANOTHER kernel's launch (the training forward puts kScanSub such
// workgroups per slice in front of the forward kernel's grid, ScanTail): between the count pass and the
// fill pass the stream otherwise runs two launches of 16-128 small workgroups, 12 us of a 170 us step
// during which the chip idles; next to the forward kernel's thousands of workgroups they cost nothing.
//   workgroup (slice, u): bin_scan_a_kernel's work for sub-range u of the bin workgroups -- every
//       workgroup's first slot inside the sub-range, the sub-range's total per block -- then a ticket;
//   the slice's LAST workgroup to arrive: bin_scan_kernel's work (nblk <= kScanThreads), a thread taking
//       kScanThreads / THREADS CONSECUTIVE blocks so that the one block scan runs over the threads' sums.
// (One workgroup per slice doing all of it took 60 us -- 8 x 16 dependent-free loads per block and
// thread, behind three forward waves on its SIMD -- and held the forward kernel's launch open.)
// Results as the two kernels'.  Inter-workgroup hand-off (MI355X_MICROARCH.md, inter-workgroup
// visibility): agent-scope (write-through) stores of the totals -> s_waitcnt vmcnt(0) -> __syncthreads
// -> relaxed agent atomic (the ticket); the last arriver reads them with agent-scope loads.
struct ScanTail {
    int *subtot, *offsets;
    int4 *items, *combos;
    int *n_items;
    int *part;                // [slice][n_wg][nblk] the count pass's counts -> first slots (in place)
    int *tickets;             // [slice] zeros on entry (the count pass clears them), left zero again
    BinPlan plan;
    int n_wg;                 // bin workgroups per slice (0: no scan work in this launch)
};
template <int THREADS>
__device__ __forceinline__ void bin_scan_tail_body(const ScanTail t, int s, int u)
{
    // (t by VALUE: through a reference the compiler kept the whole struct in private memory and copied
    // it there at the top of the kernel -- 40 scratch stores in front of EVERY wave of the forward kernel,
    // which then took 56 us instead of 39)
    constexpr int PER = kScanThreads / THREADS;
    __shared__ int wsum[4][THREADS / 64];
    __shared__ int ticket;
    const BinPlan plan = t.plan;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // ---- sub-range u (bin_scan_a_kernel)
    {
        const int wps = (t.n_wg + kScanSub - 1) / kScanSub;          // <= kScanWgPerSub (host)
        const int w_lo = u * wps, w_hi = min(t.n_wg, w_lo + wps);
        int *sp = t.part + (size_t)s * t.n_wg * plan.nblk;
        for (int k = tid; k < plan.nblk; k += THREADS) {
            int tv[kScanWgPerSub], sum = 0;
#pragma unroll
            for (int i = 0; i < kScanWgPerSub; ++i)
                tv[i] = w_lo + i < w_hi ? sp[(size_t)(w_lo + i) * plan.nblk + k] : 0;
#pragma unroll
            for (int i = 0; i < kScanWgPerSub; ++i) {
                if (w_lo + i < w_hi) sp[(size_t)(w_lo + i) * plan.nblk + k] = sum;
                sum += tv[i];
            }
            // (the totals are what the slice's last workgroup reads: written through to memory -- an
            // agent-scope store -- and read with agent-scope loads below; no release / acquire fence: a
            // release is a write-back of the XCD's whole L2, which the forward kernel next door keeps
            // full of dirty `out` lines.  The first slots in `part` are for the NEXT launch.)
            __hip_atomic_store(t.subtot + ((size_t)s * kScanSub + u) * plan.nblk + k, sum, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // my stores have left
    __syncthreads();
    if (tid == 0) {
        ticket = __hip_atomic_fetch_add(t.tickets + s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (ticket == kScanSub - 1)
            __hip_atomic_store(t.tickets + s, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // for the next call
    }
    __syncthreads();
    if (ticket != kScanSub - 1) return;                               // workgroup-uniform
    // ---- the slice's block scan (bin_scan_kernel)
    int c[PER], nch[PER], sum[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int k = tid * PER + j;
        c[j] = 0;
        if (k < plan.nblk) {                       // sub-range totals -> sub-range first slots
            int tv[kScanSub];
#pragma unroll
            for (int uu = 0; uu < kScanSub; ++uu) {
                // (read past the L1: the other workgroups' stores are in the L2 / memory)
                tv[uu] = __hip_atomic_load(t.subtot + ((size_t)s * kScanSub + uu) * plan.nblk + k, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int uu = 0; uu < kScanSub; ++uu) {
                t.subtot[((size_t)s * kScanSub + uu) * plan.nblk + k] = c[j];
                c[j] += tv[uu];
            }
        }
        // every block gets at least one item (an empty block still has to be zero-filled)
        nch[j] = k < plan.nblk ? max(1, (c[j] + plan.chunk - 1) / plan.chunk) : 0;
        sum[0] += c[j]; sum[1] += nch[j]; sum[2] += nch[j] > 1 ? nch[j] : 0; sum[3] += nch[j] > 1 ? 1 : 0;
    }
    int run[4], tot[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int x = sum[i];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int y = __shfl_up(x, o, 64);
            if (lane >= o) x += y;
        }
        run[i] = x - sum[i];                       // exclusive inside the wave
        if (lane == 63) wsum[i][wv] = x;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int pre = 0, all = 0;
#pragma unroll
        for (int w = 0; w < THREADS / 64; ++w) {
            pre += w < wv ? wsum[i][w] : 0;
            all += wsum[i][w];
        }
        run[i] += pre;
        tot[i] = all;
    }
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int k = tid * PER + j;
        if (k < plan.nblk) {
            t.offsets[(size_t)s * (plan.nblk + 1) + k] = run[0];
            int level = 0;
#pragma unroll
            for (int l = 1; l < kMaxBinLevels; ++l)
                if (l < plan.L && k >= plan.lv[l].blk0) level = l;
            BinLevel lv = plan.lv[0];
#pragma unroll
            for (int l = 1; l < kMaxBinLevels; ++l)
                if (l == level) lv = plan.lv[l];
            const int geo = (int)pack_block_geo(lv, level, k);
            for (int jj = 0; jj < nch[j]; ++jj)           // heaviest first, as bin_scan_kernel lists them
                t.items[(size_t)s * plan.item_cap + (tot[1] - 1 - (run[1] + jj))] =
                    make_int4(geo, run[0] + jj * plan.chunk, run[0] + min(c[j], (jj + 1) * plan.chunk),
                              nch[j] > 1 ? run[2] + jj : -1);
            if (nch[j] > 1) t.combos[(size_t)s * plan.nblk + run[3]] = make_int4(geo, run[2], nch[j], 0);
            run[0] += c[j]; run[1] += nch[j]; run[2] += nch[j] > 1 ? nch[j] : 0; run[3] += nch[j] > 1 ? 1 : 0;
        }
    }
    if (tid == 0) {
        t.offsets[(size_t)s * (plan.nblk + 1) + plan.nblk] = tot[0];
        t.n_items[2 * s] = tot[1];
        t.n_items[2 * s + 1] = tot[3];             // chunked blocks to combine
    }
}

}  // namespace boxattn
