// The two scan stages of the destination-binned backward (boxattn_binned.h step 2) as device functions
// that run INSIDE the training forward's launch, chained behind the count riders by tickets
// (boxattn_ride.h; the stand-alone kernels bin_scan_a_kernel / bin_scan_kernel of boxattn_binned.h do the
// same work for a backward that plans for itself):
//
//   count rider (slice s, bin workgroup w)   counts its points per block, publishes its row of `part`,
//                                            takes a ticket on sub-range u = w / wps of the slice;
//   the sub-range's LAST arriver             scan_sub_body: every workgroup's first slot inside the
//                                            sub-range, the sub-range's total per block -> `subtot`;
//                                            ticket on the slice;
//   the slice's LAST arriver                 scan_blocks_body: prefix over the sub-ranges, exclusive scan
//                                            over the blocks -> `offsets`, the work-item list, the list of
//                                            chunked blocks.
// Nobody waits: a workgroup that is not the last arriver simply ends.  Hand-offs: agent-scope
// (write-through) stores, s_waitcnt vmcnt(0) in every storing wave, barrier, relaxed agent-scope ticket;
// the last arriver reads with agent-scope (L1-bypassing) loads.  Tickets are zero on entry and left zero.
#pragma once
#include "boxattn_device.h"
#include "boxattn_combine.h"
#include "boxattn_binplan.h"
#include "boxattn_ride.h"

namespace boxattn {

// Origin, extent and level of a block in one word (computed once per block by the scan:
// five integer divisions that every work item used to repeat -- 5 % of the accumulate kernel).
//   bits 0-11 oy, 12-23 ox, 24-25 bh - 1, 26-28 bw - 1, 29-31 level   (maps < 4096 x 4096)
__device__ __forceinline__ unsigned pack_block_geo(const BinLevel &lv, int level, int blk)
{
    // (five integer divisions per block were ~200 instructions each way: a third of the block scan, which sits
    // on the critical path of the training forward's launch; multiply-high with host-checked magic numbers)
    const int rel = blk - lv.blk0;
    const int by = div_nb(rel, lv.nbx, lv.mnx), bx = rel - by * lv.nbx;
    const int oy = div_nb(by * lv.H + lv.nby - 1, lv.nby, lv.mny), ox = div_nb(bx * lv.W + lv.nbx - 1, lv.nbx, lv.mnx);
    const int bh = div_nb((by + 1) * lv.H + lv.nby - 1, lv.nby, lv.mny) - oy;
    const int bw = div_nb((bx + 1) * lv.W + lv.nbx - 1, lv.nbx, lv.mnx) - ox;
    return (unsigned)oy | ((unsigned)ox << 12) | ((unsigned)(bh - 1) << 24) |
           ((unsigned)(bw - 1) << 26) | ((unsigned)level << 29);
}
// Records per item of a block with c records cut into nch chunks: EQUAL shares (a block of 9 300 records in chunks of
// 1 536 is seven items of 1 329, not six of 1 536 and one of 84: the longest item sets the tail of the accumulate
// launch, and a partial tile + ticket costs the same whatever it sums).  (nch - 1) * share < c for every nch <= c.
__device__ __forceinline__ int chunk_records(int c, int nch) { return nch > 1 ? (c + nch - 1) / nch : c; }
constexpr int kScanSub = 8, kScanWgPerSub = 16;   // sub-ranges of bin workgroups per slice, workgroups per sub-range at most
constexpr int kScanThreads = 1024;     // bin_scan_kernel: one workgroup per slice walks the blocks 1024 at a time
constexpr int kRideTickets = kScanSub + 1;        // per slice: one ticket per sub-range + the slice's
// Bin workgroups per sub-range.  Up to kScanWgPerSub bin workgroups per slice (the fat riders: 16) form ONE
// sub-range: its last arriver reads all their rows itself and goes straight on to the block scan -- one
// hand-off in the chain instead of two (each is a write-through store + a ticket round trip, ~3 us, on the
// critical path of the training forward's launch).
__host__ __device__ inline int scan_wps(int n_wg)
{
    return n_wg <= kScanWgPerSub ? (n_wg > 0 ? n_wg : 1) : (n_wg + kScanSub - 1) / kScanSub;
}

struct ScanOut {
    int *subtot, *offsets;
    int4 *items, *combos;
    int *n_items;
};

// Sub-range u of slice s: first slots of its bin workgroups (in place in `part`), totals -> subtot (published)
template <int THREADS>
__device__ __forceinline__ void scan_sub_body(int *part, int *subtot, const BinPlan &plan, int n_wg, int s, int u)
{
    const int wps = scan_wps(n_wg);          // <= kScanWgPerSub (host)
    const int w_lo = u * wps, w_hi = min(n_wg, w_lo + wps);
    int *sp = part + (size_t)s * n_wg * plan.nblk;
    for (int k = threadIdx.x; k < plan.nblk; k += THREADS) {
        int tv[kScanWgPerSub], sum = 0;
#pragma unroll
        for (int i = 0; i < kScanWgPerSub; ++i)                // (published by the count riders: read past the L1)
            tv[i] = w_lo + i < w_hi ? agent_load(sp + (size_t)(w_lo + i) * plan.nblk + k) : 0;
#pragma unroll
        for (int i = 0; i < kScanWgPerSub; ++i) {
            if (w_lo + i < w_hi) sp[(size_t)(w_lo + i) * plan.nblk + k] = sum;     // for the NEXT launch (fill)
            sum += tv[i];
        }
        agent_store(subtot + ((size_t)s * kScanSub + u) * plan.nblk + k, sum);
    }
}

// The slice's block scan (bin_scan_kernel's work, nblk <= kScanThreads): a thread takes
// kScanThreads / THREADS CONSECUTIVE blocks, so the one block scan runs over the threads' sums.
// fuse_part != nullptr (one sub-range, fuse_wg <= kScanWgPerSub bin workgroups): the sub-range stage is done here as
// well -- every thread reads the fuse_wg published counts of its blocks itself -- instead of in a pass of its own
// with a hand-off in between: a round trip and a barrier less on the chain.
template <int THREADS>
__device__ __forceinline__ void scan_blocks_body(const ScanOut o, const BinPlan &plan, const BinLevel *lv_lds,
                                                 int n_sub, int s, int *wsum /* 4 x THREADS / 64 ints of LDS */,
                                                 int *fuse_part = nullptr, int fuse_wg = 0)
{
    constexpr int PER = kScanThreads / THREADS, NW = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    // consecutive blocks per thread: as few as the slice's blocks need (452 blocks at BoxeR-R50 shapes: two per
    // thread, 226 threads at work instead of 113 -- this scan is a serial stretch on the forward launch's critical path)
    const int per = (plan.nblk + THREADS - 1) / THREADS;
    int c[PER], nch[PER], sum[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int k = tid * per + j;
        c[j] = 0;
        nch[j] = 0;
        if (j >= per) continue;
        if (k < plan.nblk && fuse_part) {          // the bin workgroups' counts -> their first slots (in place)
            int *sp = fuse_part + (size_t)s * fuse_wg * plan.nblk + k;
            int tv[kScanWgPerSub];
#pragma unroll
            for (int w = 0; w < kScanWgPerSub; ++w)            // (published by the count riders: read past the L1)
                tv[w] = w < fuse_wg ? agent_load(sp + (size_t)w * plan.nblk) : 0;
#pragma unroll
            for (int w = 0; w < kScanWgPerSub; ++w) {
                if (w < fuse_wg) sp[(size_t)w * plan.nblk] = c[j];                 // for the NEXT launch (fill)
                c[j] += tv[w];
            }
            o.subtot[(size_t)s * kScanSub * plan.nblk + k] = 0;                    // the one sub-range starts the bin
        } else if (k < plan.nblk) {                // sub-range totals -> sub-range first slots
            int tv[kScanSub];
#pragma unroll
            for (int uu = 0; uu < kScanSub; ++uu)
                tv[uu] = uu < n_sub ? agent_load(o.subtot + ((size_t)s * kScanSub + uu) * plan.nblk + k) : 0;
#pragma unroll
            for (int uu = 0; uu < kScanSub; ++uu) {
                if (uu < n_sub) o.subtot[((size_t)s * kScanSub + uu) * plan.nblk + k] = c[j];
                c[j] += tv[uu];
            }
        }
        // every block gets at least one item (an empty block still has to be zero-filled) unless the map is sparse
        nch[j] = k < plan.nblk ? max(plan.min_items, (c[j] + plan.chunk - 1) / plan.chunk) : 0;
        sum[0] += c[j]; sum[1] += nch[j]; sum[2] += nch[j] > 1 ? nch[j] : 0; sum[3] += nch[j] > 1 ? 1 : 0;
    }
    int run[4], tot[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int x = sum[i];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int y = __shfl_up(x, d, 64);
            if (lane >= d) x += y;
        }
        run[i] = x - sum[i];                       // exclusive inside the wave
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
#pragma unroll
    for (int j = 0; j < PER; ++j) {
        const int k = tid * per + j;
        if (j < per && k < plan.nblk) {
            o.offsets[(size_t)s * (plan.nblk + 1) + k] = run[0];
            // (the level table in LDS -- the count pass left it there: a per-lane index into the kernel
            // argument would put a copy of the plan into scratch)
            int level = 0;
            for (int l = 1; l < plan.L; ++l)
                if (k >= lv_lds[l].blk0) level = l;
            const BinLevel lv = lv_lds[level];
            const int geo = (int)pack_block_geo(lv, level, k);
            const int csz = chunk_records(c[j], nch[j]);
            for (int jj = 0; jj < nch[j]; ++jj)           // heaviest first, as bin_scan_kernel lists them
                o.items[(size_t)s * plan.item_cap + (tot[1] - 1 - (run[1] + jj))] =
                    make_int4(geo, run[0] + jj * csz, run[0] + min(c[j], (jj + 1) * csz),
                              nch[j] > 1 ? (run[2] + jj) | (run[3] << kItemSlotBits) : -1);
            if (nch[j] > 1) o.combos[(size_t)s * plan.nblk + run[3]] = make_int4(geo, run[2], nch[j], 0);
            run[0] += c[j]; run[1] += nch[j]; run[2] += nch[j] > 1 ? nch[j] : 0; run[3] += nch[j] > 1 ? 1 : 0;
        }
    }
    if (tid == 0) {
        o.offsets[(size_t)s * (plan.nblk + 1) + plan.nblk] = tot[0];
        o.n_items[2 * s] = tot[1];
        o.n_items[2 * s + 1] = tot[3];             // chunked blocks
    }
}

// The same scan for slices with more blocks than scan_blocks_body's threads hold in registers (kScanThreads < nblk <=
// kRideMaxBlocks: the BEV encoder's 234 x 234 + 117 x 117 = 2 220 blocks): two passes over `cnt`, nblk ints of LDS -- the
// count rider's histogram, free once its row of counts has left.  Pass 1 (block = thread + i THREADS: coalesced rows)
// brings every block's total into LDS and turns the bin workgroups' / sub-ranges' counts into first slots in place;
// pass 2 (`per` consecutive blocks per thread) is the scan and the emit of scan_blocks_body with the totals read from LDS.
template <int THREADS>
__device__ __forceinline__ void scan_blocks_big_body(const ScanOut o, const BinPlan &plan, const BinLevel *lv_lds,
                                                     int n_sub, int s, int *wsum, int *cnt,
                                                     int *fuse_part = nullptr, int fuse_wg = 0)
{
    constexpr int NW = THREADS / 64;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    __syncthreads();                               // every wave has published its part of the histogram
    for (int k = tid; k < plan.nblk; k += THREADS) {
        int c = 0;
        if (fuse_part) {
            int *sp = fuse_part + (size_t)s * fuse_wg * plan.nblk + k;
            int tv[kScanWgPerSub];
#pragma unroll
            for (int w = 0; w < kScanWgPerSub; ++w) tv[w] = w < fuse_wg ? agent_load(sp + (size_t)w * plan.nblk) : 0;
#pragma unroll
            for (int w = 0; w < kScanWgPerSub; ++w) {
                if (w < fuse_wg) sp[(size_t)w * plan.nblk] = c;
                c += tv[w];
            }
            o.subtot[(size_t)s * kScanSub * plan.nblk + k] = 0;
        } else {
            int tv[kScanSub];
#pragma unroll
            for (int uu = 0; uu < kScanSub; ++uu)
                tv[uu] = uu < n_sub ? agent_load(o.subtot + ((size_t)s * kScanSub + uu) * plan.nblk + k) : 0;
#pragma unroll
            for (int uu = 0; uu < kScanSub; ++uu) {
                if (uu < n_sub) o.subtot[((size_t)s * kScanSub + uu) * plan.nblk + k] = c;
                c += tv[uu];
            }
        }
        cnt[k] = c;
    }
    __syncthreads();
    const int per = (plan.nblk + THREADS - 1) / THREADS;
    const int k_lo = min(tid * per, plan.nblk), k_hi = min(k_lo + per, plan.nblk);
    int sum[4] = {0, 0, 0, 0};
    for (int k = k_lo; k < k_hi; ++k) {
        const int c = cnt[k], nch = max(plan.min_items, (c + plan.chunk - 1) / plan.chunk);
        sum[0] += c; sum[1] += nch; sum[2] += nch > 1 ? nch : 0; sum[3] += nch > 1 ? 1 : 0;
    }
    int run[4], tot[4];
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
    for (int k = k_lo; k < k_hi; ++k) {
        const int c = cnt[k], nch = max(plan.min_items, (c + plan.chunk - 1) / plan.chunk);
        o.offsets[(size_t)s * (plan.nblk + 1) + k] = run[0];
        int level = 0;
        for (int l = 1; l < plan.L; ++l)
            if (k >= lv_lds[l].blk0) level = l;
        const BinLevel lv = lv_lds[level];
        const int geo = (int)pack_block_geo(lv, level, k);
        const int csz = chunk_records(c, nch);
        for (int jj = 0; jj < nch; ++jj)
            o.items[(size_t)s * plan.item_cap + (tot[1] - 1 - (run[1] + jj))] =
                make_int4(geo, run[0] + jj * csz, run[0] + min(c, (jj + 1) * csz),
                          nch > 1 ? (run[2] + jj) | (run[3] << kItemSlotBits) : -1);
        if (nch > 1) o.combos[(size_t)s * plan.nblk + run[3]] = make_int4(geo, run[2], nch, 0);
        run[0] += c; run[1] += nch; run[2] += nch > 1 ? nch : 0; run[3] += nch > 1 ? 1 : 0;
    }
    if (tid == 0) {
        o.offsets[(size_t)s * (plan.nblk + 1) + plan.nblk] = tot[0];
        o.n_items[2 * s] = tot[1];
        o.n_items[2 * s + 1] = tot[3];
    }
}

// Sparse maps (BinPlan::min_items == 0: far more blocks than records -- 1 000 queries against a 468 x 468 BEV
// map): the blocks without records get no work item; their rows of grad_value are zeroed by ZERO WORKERS, extra
// single-wave workgroups in front of the accumulate launch's grid.  A zero worker takes `per` consecutive blocks
// of its slice, finds the empty ones in the block offsets (one coalesced load per 64 blocks) and streams their
// zeros -- no per-block latency chain (as one-wave items the empty blocks ran at 2.9 TB/s: an item is a load
// -> store round trip however little it stores).  Disjoint from the rows the items write: no ordering needed.
constexpr int kZeroPer = 32;     // blocks per zero worker (BinPlan::zero_workers of them per slice)
struct ZeroRole {
    const int *offsets;          // [slice][nblk + 1], written by the scan
    const int2 *geo;             // [nblk] {first pixel of the block (index into S), W | bh - 1 << 16 | bw - 1 << 18}: zero_geo_kernel (boxattn_binned.h)
    // the same front rows of the accumulate grid as REDO workers of the one-pass fill (boxattn_spec.h; redo != nullptr, offsets == nullptr):
    const int2 *redo;            // [slice][1 + nblk] {count, 0}, {block, geometry}...: blocks that outgrew their guessed range
    const float *loc, *w_sp;     // the call's sampling locations / attention weights
    int P;
};
// (The block geometry comes from a table, not from the plan in the kernel arguments: selecting a level's
// entry there costs the accumulate kernels ~40 scalar registers at their top, which they do not have --
// scalar spills into vector lanes, vector spills to scratch.)
template <typename ST, int C>
__device__ __forceinline__ void zero_empty_blocks(const ZeroRole zr, int nblk, int s, int zw, int S, int H,
                                                  ST *__restrict__ grad_value, int lane)
{
    constexpr int PPR = C * (int)sizeof(ST) / 16;            // 16-byte pieces of a (pixel, head) row
    constexpr int PIX = 64 / PPR;                            // pixels a wave covers per store
    typedef unsigned int zr_u32x4 __attribute__((ext_vector_type(4)));
    const int b = s / H, h = s % H;
    const int *off = zr.offsets + (size_t)s * (nblk + 1);
    const int k_lo = zw * kZeroPer, k_hi = min(nblk, k_lo + kZeroPer);
    for (int k0 = k_lo; k0 < k_hi; k0 += 64) {
        const int k = min(k0 + lane, k_hi - 1);
        const bool empty = k0 + lane < k_hi && off[k + 1] == off[k];
        const int2 g = zr.geo[k];
        unsigned long long m = __builtin_amdgcn_ballot_w64(empty);
        while (m) {
            const int j = (int)__builtin_ctzll(m);            // wave-uniform
            m &= m - 1;
            const int pix0 = __builtin_amdgcn_readlane(g.x, j), gy = __builtin_amdgcn_readlane(g.y, j);
            const int W = gy & 0xffff, bh = ((gy >> 16) & 3) + 1, bw = ((gy >> 18) & 7) + 1;
#pragma unroll
            for (int p0 = 0; p0 < 32; p0 += PIX) {
                const int pix = p0 + lane / PPR, py = pix >> 3, px = pix & 7;
                if (pix < 32 && py < bh && px < bw) {
                    ST *row = grad_value + (((size_t)b * S + pix0 + py * W + px) * H + h) * C;
                    // (plain stores: the two 64-byte halves of a bf16 line belong to neighbouring heads = slices on
                    // the same XCD, whose L2 merges them -- non-temporal they leave as half lines, C5 bf16 accumulate
                    // 56 -> 67 us; float32 rows, whole lines, measured the same either way)
                    reinterpret_cast<zr_u32x4 *>(row)[lane % PPR] = zr_u32x4{0u, 0u, 0u, 0u};
                }
            }
        }
    }
}

}  // namespace boxattn
