// Destination-binned backward for grad_value: no floating-point atomics on the hot path.
//
// Why: the reference scatters 4*C atomicAdds per sample point into grad_value
// (box_attn_kernel.cuh:100-184; 436 M atomics at BoxeR-R50 COCO shapes).  On MI355X global
// fp32 atomics retire ~10 G 128-byte row-ops/s whatever their scope or locality, and LDS
// ds_add_f32 ~3 clk per LANE (profiles/r01_microbench_*.log), so any "atomic per
// contribution" design is 1.3-5 ms.  Here every grad_value row is instead produced by ONE
// owner that gathers its contributions:
//
//   1. bin_kernel<count>   one pass over the sampling locations: every sample point is
//                          assigned to the blocks (up to 8x4 pixels, a balanced partition of
//                          every level, per image and head) its 2x2 footprint touches;
//                          per-workgroup LDS histograms, written out densely per workgroup
//                          (no global atomics); queries interleaved over the workgroups.
//   2. bin_scan_a_kernel, bin_scan_kernel   per (image, head) slice: prefix over the
//                          workgroups and exclusive scan over the blocks -> every workgroup's
//                          first slot in every bin, and the work-item list (big bins are cut
//                          into chunks).
//   3. bin_kernel<fill>    same pass again, now writing the point ids into their bins.
//   4. pointgrad2_kernel (boxattn_gather2.h): grad_loc / grad_weight, query-major.
//   5. accumulate, one wavefront per work item, two formulations:
//      binned_accumulate_tr_kernel (boxattn_binned_tr.h; bf16 box attention): a round of 64
//                          records x 32 pixels as a dense product on the matrix cores, from
//                          wide records {id, x, y, weight};
//      binned_accumulate_kernel (here; fp32 storage and instance attention): records ->
//                          geometry -> per-pixel entry lists in LDS (integer LDS atomics for
//                          the ranks); the records' upstream-gradient rows are staged once in
//                          LDS; each lane pair owns one destination pixel and sums (w*a) * row
//                          over its list in registers.
//                          Either way: one plain coalesced row store per pixel in the storage
//                          type; blocks cut into chunks (the coarse levels) write fp32 partial
//                          tiles instead.
//   6. combine: the partial tiles of a chunked block are summed by the block's last chunk item to finish
//                          (chunk_finish, boxattn_combine.h); combine_partials_kernel for plans too big for that.
// In the training step the passes 1-3 are not launches of their own: count + scans ride in the forward
// kernel's launch, fill in the point-gradient kernel's (boxattn_ride.h, boxattn_binpass.h).
// No zero-fill, no conversion pass, no float atomics (run-to-run differences are limited to
// the fp32 summation order inside a bin, which follows integer LDS atomics).
//
// The result is the same sum as the reference's, in a different (still unspecified) order.
#pragma once
#include <type_traits>

#include "boxattn_device.h"
#include "boxattn_combine.h"
#include "boxattn_binplan.h"
#include "boxattn_scan_tail.h"
#include "boxattn_binpass.h"

namespace boxattn {

// ---------------------------------------------------------------------------------------
// 1 + 3: count / fill as launches of their own (a backward that plans for itself, maps too big for the
//        riders).  grid = (workgroups, slices), block kBinThreads, dynamic LDS nblk ints.
// ---------------------------------------------------------------------------------------
#ifndef BOXATTN_TUNE_BIN_THREADS
#define BOXATTN_TUNE_BIN_THREADS 512
#endif
constexpr int kBinThreads = BOXATTN_TUNE_BIN_THREADS;
template <int BW, int BH, bool FILL, bool WIDE, int PT>
__global__ __launch_bounds__(kBinThreads) void bin_kernel(const float *__restrict__ loc,
                                                  const float *__restrict__ w_sp, BinPlan plan,
                                                  int H, int Lq, int P, int q_per_wg, int n_wg, int interleave,
                                                  int *__restrict__ part,
                                                  const int *__restrict__ subtot,
                                                  const int *__restrict__ offsets,
                                                  int *__restrict__ records, int *__restrict__ ctickets)
{
    // fill pass: bin workgroup 0 of a slice clears the slice's combine tickets of the accumulate launch (chunk_finish)
    if (FILL && ctickets && blockIdx.x == 0)
        for (int k = threadIdx.x; k < plan.nblk; k += kBinThreads) ctickets[(size_t)blockIdx.y * plan.nblk + k] = 0;
    // part[slice][workgroup][block]: after the count pass the number of records this workgroup
    // has for the block; the scan turns it into the workgroup's first slot inside the block's bin.
    // grid = (n_wg, slices).  (Placing all workgroups of a slice on one XCD, so that the record
    // writes of a bin merge in one L2, changed nothing: the fill pass is bound by the bytes.)
    extern __shared__ int sh_bins[];
    __shared__ BinLevel s_lv[kMaxBinLevels];       // indexed per lane (no select chains)
    const int s = blockIdx.y, wg = blockIdx.x;
    bin_pass_body<kBinThreads, BW, BH, FILL, WIDE, PT>(sh_bins, s_lv, loc, w_sp, plan, H, Lq, P, q_per_wg, n_wg,
                                                       interleave != 0, part, subtot, offsets, records, s, wg);
    if constexpr (!FILL) {
        int *mypart = part + ((size_t)s * n_wg + wg) * plan.nblk;
        for (int k = threadIdx.x; k < plan.nblk; k += kBinThreads) mypart[k] = sh_bins[k];
    }
}

// ---------------------------------------------------------------------------------------
// 2: per-slice scan.  grid = slices, block 256.
// ---------------------------------------------------------------------------------------
// Two small kernels so that no thread walks more than 16 dependent-free loads:
//   A  grid (kScanSub, slices): prefix of the per-workgroup counts inside each of the
//      kScanSub sub-ranges of workgroups (<= kScanWgPerSub workgroups each), sub-range totals;
//   B  grid (slices): prefix over the sub-ranges, then the exclusive scan over the blocks.
__global__ __launch_bounds__(256) void bin_scan_a_kernel(int *__restrict__ part, int n_wg,
                                                         int *__restrict__ subtot, BinPlan plan)
{
    const int sub = blockIdx.x, s = blockIdx.y;
    const int wps = scan_wps(n_wg);
    const int w_lo = sub * wps, w_hi = min(n_wg, w_lo + wps);
    int *sp = part + (size_t)s * n_wg * plan.nblk;
    // grid.z covers the blocks 256 at a time (big maps: thousands of blocks per slice)
    for (int k = blockIdx.z * 256 + threadIdx.x; k < plan.nblk; k += 256 * gridDim.z) {
        int t[kScanWgPerSub], sum = 0;
#pragma unroll
        for (int u = 0; u < kScanWgPerSub; ++u)
            t[u] = w_lo + u < w_hi ? sp[(size_t)(w_lo + u) * plan.nblk + k] : 0;
#pragma unroll
        for (int u = 0; u < kScanWgPerSub; ++u) {
            if (w_lo + u < w_hi) sp[(size_t)(w_lo + u) * plan.nblk + k] = sum;
            sum += t[u];
        }
        subtot[((size_t)s * kScanSub + sub) * plan.nblk + k] = sum;
    }
}

// fuse_wg > 0: kernel A's work is done here as well (every thread walks the fuse_wg workgroup
// counts of its block itself): one launch less -- what the decoder shapes, whose step is a chain
// of short kernels, are made of -- at the price of fuse_wg loads per thread instead of 16 + 8.
__global__ __launch_bounds__(kScanThreads) void bin_scan_kernel(int *__restrict__ subtot,
                                                       int *__restrict__ offsets,
                                                       int4 *__restrict__ items,
                                                       int4 *__restrict__ combos,
                                                       int *__restrict__ n_items, BinPlan plan,
                                                       int *__restrict__ part, int fuse_wg)
{
    // four running sums over the blocks: records, items, partial slots, chunked blocks
    __shared__ int wsum[4][kScanThreads / 64];
    __shared__ int carry[4];
    const int s = blockIdx.x, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x < 4) carry[threadIdx.x] = 0;
    __syncthreads();
    for (int k0 = 0; k0 < plan.nblk; k0 += kScanThreads) {
        const int k = k0 + threadIdx.x;
        const bool live = k < plan.nblk;
        int c = 0;
        if (live && fuse_wg > 0) {                 // workgroup counts -> first slots, both levels
            const int wps = scan_wps(fuse_wg);
            int *sp = part + (size_t)s * fuse_wg * plan.nblk + k;
            for (int u = 0; u < kScanSub; ++u) {
                subtot[((size_t)s * kScanSub + u) * plan.nblk + k] = c;
                int in_sub = 0;
                const int w_hi = min(fuse_wg, (u + 1) * wps);
                for (int w = u * wps; w < w_hi; ++w) {
                    const int t = sp[(size_t)w * plan.nblk];
                    sp[(size_t)w * plan.nblk] = in_sub;
                    in_sub += t;
                }
                c += in_sub;
            }
        } else if (live) {                         // sub-range totals -> sub-range first slots
            int t[kScanSub];
#pragma unroll
            for (int u = 0; u < kScanSub; ++u)
                t[u] = subtot[((size_t)s * kScanSub + u) * plan.nblk + k];
#pragma unroll
            for (int u = 0; u < kScanSub; ++u) {
                subtot[((size_t)s * kScanSub + u) * plan.nblk + k] = c;
                c += t[u];
            }
        }
        // every block gets at least one item (an empty block still has to be zero-filled) unless the map is sparse
        const int nch = live ? max(plan.min_items, (c + plan.chunk - 1) / plan.chunk) : 0;
        const int v[4] = {c, nch, nch > 1 ? nch : 0, nch > 1 ? 1 : 0};
        int inc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int x = v[i];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int t = __shfl_up(x, o, 64);
                if (lane >= o) x += t;
            }
            inc[i] = x;
            if (lane == 63) wsum[i][wv] = x;
        }
        __syncthreads();
        int ex[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int pre = carry[i];
            for (int w = 0; w < wv; ++w) pre += wsum[i][w];
            ex[i] = pre + inc[i] - v[i];
        }
        // Items are listed HEAVIEST FIRST (the coarse levels' chunked blocks are the last ones
        // in block order): item i of the list is the (total - 1 - i)-th in block order.  Stored that
        // way round, an accumulate worker's first item is items[worker] -- a load it can issue at
        // once, next to the one for the item count, instead of after it.
        int tot_items = carry[1];                  // (one pass: the host sends bigger maps to the segment kernels)
        for (int w = 0; w < kScanThreads / 64; ++w) tot_items += wsum[1][w];
        if (live) {
            offsets[(size_t)s * (plan.nblk + 1) + k] = ex[0];
            int level = 0;
#pragma unroll
            for (int l = 1; l < kMaxBinLevels; ++l)
                if (l < plan.L && k >= plan.lv[l].blk0) level = l;
            BinLevel lv = plan.lv[0];
#pragma unroll
            for (int l = 1; l < kMaxBinLevels; ++l)
                if (l == level) lv = plan.lv[l];
            const int geo = (int)pack_block_geo(lv, level, k);
            const int csz = chunk_records(c, nch);
            for (int j = 0; j < nch; ++j)
                items[(size_t)s * plan.item_cap + (tot_items - 1 - (ex[1] + j))] =   // record range inside the slice
                    make_int4(geo, ex[0] + j * csz, ex[0] + min(c, (j + 1) * csz),
                              nch > 1 ? (ex[2] + j) | (ex[3] << kItemSlotBits) : -1);   // .w: see kItemSlotBits
            if (nch > 1)
                combos[(size_t)s * plan.nblk + ex[3]] = make_int4(geo, ex[2], nch, 0);
        }
        __syncthreads();
        if (threadIdx.x == kScanThreads - 1) {
#pragma unroll
            for (int i = 0; i < 4; ++i) carry[i] = ex[i] + v[i];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        offsets[(size_t)s * (plan.nblk + 1) + plan.nblk] = carry[0];
        n_items[2 * s] = carry[1];
        n_items[2 * s + 1] = carry[3];             // chunked blocks to combine
    }
}

// Maps with more than kScanThreads blocks per slice (a 468 x 468 BEV level: 6 903): the same
// scan over SEVERAL workgroups per slice.  One workgroup walking the blocks 1 024 at a time is
// ~100 k uncoalesced 4-byte accesses through a single CU's memory pipeline (35 us at 6 903
// blocks, the longest kernel of the BEV decoder's backward but one); here every segment of 1 024
// blocks has its own workgroup, in two launches:
//   bin_scan_seg_kernel   grid (segments, slices): per block the sub-range prefix (as above), the
//                         block's record count -> offsets[] (scratch use), its exclusive prefixes
//                         INSIDE the segment -> tmp[], the segment's four totals -> segtot[];
//   bin_scan_emit_kernel  same grid: segment base = sum of the preceding segments' totals, then
//                         offsets / items / combos exactly as bin_scan_kernel writes them.
__global__ __launch_bounds__(kScanThreads) void bin_scan_seg_kernel(int *__restrict__ subtot,
                                                                    int *__restrict__ offsets,
                                                                    int4 *__restrict__ tmp,
                                                                    int4 *__restrict__ segtot,
                                                                    BinPlan plan, int *__restrict__ part = nullptr,
                                                                    int fuse_wg = 0)
{
    // fuse_wg > 0 (few bin workgroups per slice -- the big maps with few queries have ONE): bin_scan_a_kernel's
    // work is done here as well, as in bin_scan_kernel: one launch less in a chain of short kernels
    __shared__ int wsum[4][kScanThreads / 64];
    const int seg = blockIdx.x, s = blockIdx.y, lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int k = seg * kScanThreads + (int)threadIdx.x;
    const bool live = k < plan.nblk;
    int c = 0;
    if (live && fuse_wg > 0) {
        const int wps = scan_wps(fuse_wg);
        int *sp = part + (size_t)s * fuse_wg * plan.nblk + k;
        for (int u = 0; u < kScanSub; ++u) {
            subtot[((size_t)s * kScanSub + u) * plan.nblk + k] = c;
            int in_sub = 0;
            const int w_hi = min(fuse_wg, (u + 1) * wps);
            for (int w = u * wps; w < w_hi; ++w) {
                const int t = sp[(size_t)w * plan.nblk];
                sp[(size_t)w * plan.nblk] = in_sub;
                in_sub += t;
            }
            c += in_sub;
        }
    } else if (live) {
        int t[kScanSub];
#pragma unroll
        for (int u = 0; u < kScanSub; ++u) t[u] = subtot[((size_t)s * kScanSub + u) * plan.nblk + k];
#pragma unroll
        for (int u = 0; u < kScanSub; ++u) {
            subtot[((size_t)s * kScanSub + u) * plan.nblk + k] = c;
            c += t[u];
        }
    }
    const int nch = live ? max(plan.min_items, (c + plan.chunk - 1) / plan.chunk) : 0;
    const int v[4] = {c, nch, nch > 1 ? nch : 0, nch > 1 ? 1 : 0};
    int inc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int x = v[i];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(x, o, 64);
            if (lane >= o) x += t;
        }
        inc[i] = x;
        if (lane == 63) wsum[i][wv] = x;
    }
    __syncthreads();
    int ex[4], tot[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int pre = 0, all = 0;
#pragma unroll
        for (int w = 0; w < kScanThreads / 64; ++w) {
            pre += w < wv ? wsum[i][w] : 0;
            all += wsum[i][w];
        }
        ex[i] = pre + inc[i] - v[i];
        tot[i] = all;
    }
    if (live) {
        offsets[(size_t)s * (plan.nblk + 1) + k] = c;                 // scratch: the emit kernel reads it
        tmp[(size_t)s * plan.nblk + k] = make_int4(ex[0], ex[1], ex[2], ex[3]);
    }
    if (threadIdx.x == 0)
        segtot[(size_t)s * gridDim.x + seg] = make_int4(tot[0], tot[1], tot[2], tot[3]);
}

__global__ __launch_bounds__(kScanThreads) void bin_scan_emit_kernel(int *__restrict__ offsets,
                                                                     const int4 *__restrict__ tmp,
                                                                     const int4 *__restrict__ segtot,
                                                                     int4 *__restrict__ items,
                                                                     int4 *__restrict__ combos,
                                                                     int *__restrict__ n_items,
                                                                     BinPlan plan)
{
    const int seg = blockIdx.x, s = blockIdx.y;
    int base[4] = {0, 0, 0, 0}, all[4] = {0, 0, 0, 0};
    for (int g = 0; g < (int)gridDim.x; ++g) {                         // <= kMaxBlocks / kScanThreads
        const int4 t = segtot[(size_t)s * gridDim.x + g];
        const int tv[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            base[i] += g < seg ? tv[i] : 0;
            all[i] += tv[i];
        }
    }
    const int k = seg * kScanThreads + (int)threadIdx.x;
    if (k < plan.nblk) {
        const int c = offsets[(size_t)s * (plan.nblk + 1) + k];
        const int4 e = tmp[(size_t)s * plan.nblk + k];
        const int ex[4] = {base[0] + e.x, base[1] + e.y, base[2] + e.z, base[3] + e.w};
        const int nch = max(plan.min_items, (c + plan.chunk - 1) / plan.chunk);
        offsets[(size_t)s * (plan.nblk + 1) + k] = ex[0];
        int level = 0;
#pragma unroll
        for (int l = 1; l < kMaxBinLevels; ++l)
            if (l < plan.L && k >= plan.lv[l].blk0) level = l;
        BinLevel lv = plan.lv[0];
#pragma unroll
        for (int l = 1; l < kMaxBinLevels; ++l)
            if (l == level) lv = plan.lv[l];
        const int geo = (int)pack_block_geo(lv, level, k);
        const int csz = chunk_records(c, nch);
        for (int j = 0; j < nch; ++j)
            items[(size_t)s * plan.item_cap + (all[1] - 1 - (ex[1] + j))] =      // heaviest first
                make_int4(geo, ex[0] + j * csz, ex[0] + min(c, (j + 1) * csz),
                          nch > 1 ? (ex[2] + j) | (ex[3] << kItemSlotBits) : -1);
        if (nch > 1) combos[(size_t)s * plan.nblk + ex[3]] = make_int4(geo, ex[2], nch, 0);
    }
    if (seg == 0 && threadIdx.x == 0) {
        offsets[(size_t)s * (plan.nblk + 1) + plan.nblk] = all[0];
        n_items[2 * s] = all[1];
        n_items[2 * s + 1] = all[3];
    }
}

// ---------------------------------------------------------------------------------------
// 5: accumulate.  One WAVEFRONT per work item (64-thread workgroups, no cross-wave barriers):
//    every wave has private LDS and runs independently, so the many dependent latencies of an
//    item (record id -> location / weight -> upstream-gradient row) are hidden by the other
//    ~12 waves of the CU instead of stalling a 256-thread workgroup at barriers.
//    grid = (waves per slice, slices); waves are persistent and pull items from the slice's
//    queue.  A block is 8x4 pixels; in the summation phase lane = (pixel, channel half).
// ---------------------------------------------------------------------------------------
#ifndef BOXATTN_TUNE_ABLATE
#define BOXATTN_TUNE_ABLATE 0      // timing experiments only (wrong results): 1 no list walk,
#endif                             // 2 no ranks / entries, 3 no row fetch + stage, 4 walk without row reads
#ifndef BOXATTN_TUNE_ACC_WPE
#define BOXATTN_TUNE_ACC_WPE 1
#endif
template <typename ST, int C, bool INST, int RPL = 1, bool WIDE = false>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(BOXATTN_TUNE_ACC_WPE)))
void binned_accumulate_kernel(
    const ST *__restrict__ grad_out, const ST *__restrict__ grad_mask,
    const float *__restrict__ loc, const float *__restrict__ w_sp,
    const float *__restrict__ w_lv, BinPlan plan, int S, int H, int Lq, int P,
    const int *__restrict__ offsets, const int4 *__restrict__ items,
    const int *__restrict__ n_items, const int *__restrict__ records,
    ST *__restrict__ grad_value, float *__restrict__ partials, ChunkCombine cc, ZeroRole zr)
{
    constexpr int BW = 8, PB = 32;                      // blocks: up to 8 x 4 pixels
    constexpr int R = 64 * RPL;                        // records per round, RPL per lane
    constexpr int CH = C / 2;                          // channels per lane while summing
    constexpr int ROWB = C * (int)sizeof(ST);          // bytes of one upstream-gradient row
    // Box attention: rows are staged in the storage type.  (Staging bf16 rows as fp32 saves the
    // unpack in the summation loop but doubles the LDS bytes: slower.)  Instance attention: the
    // two upstream rows of a record are combined while staging, t = a_s * g + a_l * g_mask
    // (reference instance_attn_kernel.cuh:139), as fp32 -- one staged row and one weight per
    // entry, exactly the box flavour's walk, instead of two of each.
#ifndef BOXATTN_TUNE_STAGE_F32
#define BOXATTN_TUNE_STAGE_F32 0
#endif
    constexpr bool kCvt = INST || (BOXATTN_TUNE_STAGE_F32 && sizeof(ST) == 2);   // staged as fp32
    constexpr int SB = kCvt ? 4 : (int)sizeof(ST);     // bytes per staged element
    constexpr int RS = C * SB + 16;                    // LDS row stride: row + pad (banks)
    constexpr int LPR = ROWB / 16;                     // lanes that fetch one row, 16 B each
    constexpr int RPP = 64 / LPR;                      // rows staged per pass
    constexpr int NPASS = R / RPP;                     // staging passes per round
    constexpr int EPL = 16 / (int)sizeof(ST);          // elements per fetched 16-byte piece
#ifndef BOXATTN_TUNE_UNR
#define BOXATTN_TUNE_UNR 2
#endif
    constexpr int UNR = BOXATTN_TUNE_UNR;              // list entries handled per step
    typedef float2 Entry;                              // {weight, LDS offset of the staged row}
    typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));  // plain vector: stays in VGPRs
    constexpr int NQ = CH * SB / 16;                   // 16-byte pieces of a lane's half row
    // channels (2i, 2i+1) of a staged half row as two floats
    auto staged_pair = [](const u32x4 (&rw)[NQ], int i) -> f32x2 {
        f32x2 r;
        if constexpr (SB == 4) {
            r.x = __uint_as_float(rw[(2 * i) / 4][(2 * i) % 4]);
            r.y = __uint_as_float(rw[(2 * i + 1) / 4][(2 * i + 1) % 4]);
        } else {
            const unsigned wd = rw[i / 4][i % 4];
            r.x = __uint_as_float(wd << 16); r.y = __uint_as_float(wd & 0xffff0000u);
        }
        return r;
    };
    // row R of the stages is all zeros: the target of padded (unused) list entries
    __shared__ __attribute__((aligned(16))) unsigned char gstage[(R + 1) * RS];
    // 4 entries per record + the lists' padding + the prefetch overrun of the last step
    __shared__ __attribute__((aligned(16))) Entry ent[4 * R + PB * (UNR - 1) + UNR];
    __shared__ int pcnt[PB + 1], poff[PB + 1];                             // pcnt[PB] = dump slot
    __shared__ int last_flag;

    // Workgroup -> (slice, worker) so that all workers of a slice sit on ONE XCD (workgroup b
    // runs on XCD b % 8): a slice only reads the upstream-gradient / location / weight rows of
    // its own head, so each XCD's L2 then holds 1/8 of those tensors instead of all of them.
    const int n_slices = plan.n_slices, workers = (int)gridDim.x - plan.zero_workers;
    const int bid = blockIdx.y * gridDim.x + blockIdx.x;
    const int xcd = bid % 8, k = bid / 8;
    const int per_xcd = (n_slices + 7) / 8;                 // slices handled by one XCD
    const int s = slice_on_xcd(xcd, k % per_xcd, per_xcd);
    const int worker = k / per_xcd - plan.zero_workers;                  // 0 .. workers-1 (grid is 8-aligned); < 0: a sparse map's zero workers
    if (s >= n_slices || worker >= workers) return;
    const int b = s / H, h = s % H;
    const int LP = plan.L * P;
    const int lane = threadIdx.x;
    if (worker < 0) {
        zero_empty_blocks<ST, C>(zr, plan.nblk, s, worker + plan.zero_workers, S, H, grad_value, lane);
        return;
    }
    const int mypix = lane >> 1, half = lane & 1;
    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);
    const int n_it = n_items[2 * s];
    const int lp_mask = (1 << plan.lp_bits) - 1;

    if (lane <= PB) pcnt[lane] = 0;
    for (int i = lane; i < RS / 4; i += 64) {          // the zero rows
        reinterpret_cast<int *>(&gstage[R * RS])[i] = 0;
    }

    // Items are taken heaviest first.  The launch provides one workgroup per potential item, so
    // this loop normally runs once and the hardware workgroup dispatcher does the load
    // balancing.  (A software queue was slower both ways it was tried: with the 16 queue heads
    // in one cache line every dequeue of the chip serialised on that line -- 13 k atomics =
    // 160 us --, and with padded heads the dequeue round trip still cost 10 %.)
    const int4 *my_items = items + (size_t)s * plan.item_cap;
    int4 item_n = my_items[min(worker, plan.item_cap - 1)];      // (list is heaviest first)
    for (int it = worker; it < n_it; it += workers) {
        // coarse levels sit at the end of the list and carry the long chunked items: take
        // them first so the tail of the kernel is made of short items (the next item of this
        // workgroup is requested while it works on the current one)
        const int4 item = item_n;
        item_n = my_items[min(it + workers, plan.item_cap - 1)];
        const BlockGeo bg = unpack_block_geo((unsigned)item.x);
        BinLevel lv = plan.lv[0];                    // select, no dynamic indexing of kernel args
#pragma unroll
        for (int k = 1; k < kMaxBinLevels; ++k)
            if (k == bg.level) lv = plan.lv[k];
        const int oy = bg.oy, ox = bg.ox, bh = bg.bh, bw = bg.bw;
        const int *rec = records + (size_t)s * plan.rec_cap * (WIDE ? 4 : 1);   // item.y / .z index the slice
        (void)offsets;
        f32x2 acc[CH / 2];                           // channel pairs (2i, 2i+1) of this half
#pragma unroll
        for (int i = 0; i < CH / 2; ++i) acc[i] = f32x2{0.f, 0.f};

        // Software pipeline over rounds of R records: everything global that round r+1 needs
        // (record ids, locations, weights, upstream-gradient rows) is issued at the top of
        // round r and consumed one iteration later.  Inactive lanes use record 0 (valid).
        // WIDE records {id, x, y, weight} (box attention): location and weight come with the
        // coalesced record stream instead of two more gathers per record.
        static_assert(!(WIDE && INST), "wide records carry one weight");
        struct Ids { int v[RPL]; float2 xy[RPL]; float a[RPL]; };
        auto fetch_ids = [&](int rr) -> Ids {
            Ids r;
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const bool have = rr + i * 64 + lane < item.z;
                if constexpr (WIDE) {
                    const int4 w = have ? reinterpret_cast<const int4 *>(rec)[rr + i * 64 + lane]
                                        : make_int4(0, 0, 0, 0);
                    r.v[i] = w.x;
                    r.xy[i] = make_float2(__int_as_float(w.y), __int_as_float(w.z));
                    r.a[i] = __int_as_float(w.w);
                } else {
                    r.v[i] = have ? rec[rr + i * 64 + lane] : 0;
                }
            }
            return r;
        };
        int row_n[RPL], mr_n[RPL];                     // rows of the round being fetched
        auto fetch_point = [&](const Ids &r, float2 (&xy)[RPL], float (&as)[RPL],
                               float (&al)[RPL]) {
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const int q = r.v[i] >> plan.lp_bits, lp = r.v[i] & lp_mask;
                row_n[i] = (int)(((size_t)b * Lq + q) * H + h);
                const size_t pid = (size_t)row_n[i] * LP + lp;
                if constexpr (WIDE) {
                    xy[i] = r.xy[i];
                    as[i] = r.a[i];
                } else {
                    xy[i] = loc2[pid];
                    as[i] = w_sp[pid];
                }
                al[i] = INST ? w_lv[pid] : 0.f;
                mr_n[i] = INST ? (int)((((size_t)b * Lq + q) * P + lp % P) * H + h) : 0;
            }
        };
        u32x4 grow[NPASS], mrow[INST ? NPASS : 1];     // rows in flight (registers)
        auto stage_piece = [&](unsigned char *dst, u32x4 v) {
            *reinterpret_cast<u32x4 *>(dst) = v;
        };
        auto fetched_elem = [](const u32x4 &v, int e) -> float {   // element e of a 16-byte piece
            if constexpr (sizeof(ST) == 4) return __uint_as_float(v[e]);
            else return (e & 1) ? __uint_as_float(v[e / 2] & 0xffff0000u)
                                : __uint_as_float(v[e / 2] << 16);
        };
#define BOXATTN_FETCH_ROWS()                                                                    \
    _Pragma("unroll") for (int ps = 0; ps < NPASS; ++ps) {                                      \
        const int j_ = ps * RPP + lane / LPR, piece_ = lane % LPR;                              \
        const int rj_ = __shfl(row_n[(ps * RPP) / 64], j_ % 64, 64);                            \
        grow[ps] = *reinterpret_cast<const u32x4 *>(grad_out + (size_t)rj_ * C + piece_ * EPL); \
        if constexpr (INST) {                                                                   \
            const int mj_ = __shfl(mr_n[(ps * RPP) / 64], j_ % 64, 64);                         \
            mrow[ps] =                                                                          \
                *reinterpret_cast<const u32x4 *>(grad_mask + (size_t)mj_ * C + piece_ * EPL);   \
        }                                                                                       \
    }
#define BOXATTN_STAGE_ROWS(AS, AL)                                                              \
    _Pragma("unroll") for (int ps = 0; ps < NPASS; ++ps) {                                      \
        const int j_ = ps * RPP + lane / LPR, piece_ = lane % LPR;                              \
        if constexpr (!kCvt) {                                                                  \
            stage_piece(&gstage[j_ * RS + piece_ * 16], grow[ps]);                              \
        } else {                                                                                \
            const float as_ = INST ? __shfl(AS[(ps * RPP) / 64], j_ % 64, 64) : 1.f;            \
            const float al_ = INST ? __shfl(AL[(ps * RPP) / 64], j_ % 64, 64) : 0.f;            \
            float t_[EPL];                                                                      \
            _Pragma("unroll") for (int e_ = 0; e_ < EPL; ++e_)                                  \
                t_[e_] = INST ? as_ * fetched_elem(grow[ps], e_) +                              \
                                    al_ * fetched_elem(mrow[INST ? ps : 0], e_)                 \
                              : fetched_elem(grow[ps], e_);                                     \
            _Pragma("unroll") for (int e_ = 0; e_ < EPL; e_ += 4)                               \
                stage_piece(&gstage[j_ * RS + piece_ * (EPL * 4) + e_ * 4],                     \
                            u32x4{__float_as_uint(t_[e_]), __float_as_uint(t_[e_ + 1]),         \
                                  __float_as_uint(t_[e_ + 2]), __float_as_uint(t_[e_ + 3])});   \
        }                                                                                       \
    }
        float2 xy_c[RPL], xy_n[RPL];
        float as_c[RPL], al_c[RPL], as_n[RPL], al_n[RPL];
#pragma unroll
        for (int i = 0; i < RPL; ++i) { xy_n[i] = make_float2(0.f, 0.f); as_n[i] = al_n[i] = 0.f; }
        fetch_point(fetch_ids(item.y), xy_c, as_c, al_c);
        if (BOXATTN_TUNE_ABLATE != 3) { BOXATTN_FETCH_ROWS() }
        Ids rec_n = fetch_ids(item.y + R);
        if (BOXATTN_TUNE_ABLATE != 3) { BOXATTN_STAGE_ROWS(as_c, al_c) }   // round 0 staged directly
        for (int rr = item.y; rr < item.z; rr += R) {
            const int n = min(R, item.z - rr);
            const bool more = rr + R < item.z;         // wave-uniform
            if (more) {                                // issue everything round r+1 needs
                fetch_point(rec_n, xy_n, as_n, al_n);
                if (BOXATTN_TUNE_ABLATE != 3) { BOXATTN_FETCH_ROWS() }
                rec_n = fetch_ids(rr + 2 * R);
            }
            // ---- phase 1: lane = RPL records: geometry, rank inside the destination pixel
            //      lists (pixk == PB: corner not in this block / idle lane).
            float wk[RPL][4];
            int pixk[RPL][4], rank[RPL][4];
#pragma unroll
            for (int i = 0; i < RPL; ++i) {
                const Sample<float> sm = locate<float>(xy_c[i].x, xy_c[i].y, lv.H, lv.W);
                wk[i][0] = sm.hh * sm.hw; wk[i][1] = sm.hh * sm.lw;
                wk[i][2] = sm.lh * sm.hw; wk[i][3] = sm.lh * sm.lw;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int yy = sm.y0 + (k >> 1), xx = sm.x0 + (k & 1);
                    const bool use = i * 64 + lane < n && sm.ok[k] &&
                                     (unsigned)(yy - oy) < (unsigned)bh &&
                                     (unsigned)(xx - ox) < (unsigned)bw;
                    pixk[i][k] = use ? (yy - oy) * BW + (xx - ox) : PB;
                }
            }
            // predicated, not redirected to a dump counter: same-address LDS atomics serialise per
            // lane, and a third of the corners (other blocks, idle lanes) would all hit the dump
#pragma unroll
            for (int i = 0; i < RPL; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    rank[i][k] = 0;
                    if (BOXATTN_TUNE_ABLATE != 2 && pixk[i][k] < PB)
                        rank[i][k] = atomicAdd(&pcnt[pixk[i][k]], 1);
                }
            wave_lds_sync();
            {   // inclusive scan of the 32 pixel counts with DPP row shifts (no LDS round trips).
                // Every list is padded to a multiple of UNR entries with {weight 0, zero row}, so
                // the walk below needs no per-entry "still inside my list" masking.
                const int cnt = lane < PB ? pcnt[lane] : 0;
                if (lane < PB) pcnt[lane] = 0;                  // ready for the next round
                const int padded = (cnt + UNR - 1) / UNR * UNR;
                int ic = padded;
                ic += __builtin_amdgcn_update_dpp(0, ic, 0x111, 0xF, 0xF, true);   // row_shr:1
                ic += __builtin_amdgcn_update_dpp(0, ic, 0x112, 0xF, 0xF, true);   // row_shr:2
                ic += __builtin_amdgcn_update_dpp(0, ic, 0x114, 0xF, 0xF, true);   // row_shr:4
                ic += __builtin_amdgcn_update_dpp(0, ic, 0x118, 0xF, 0xF, true);   // row_shr:8
                ic += __builtin_amdgcn_update_dpp(0, ic, 0x142, 0xA, 0xF, true);   // row_bcast:15
                if (lane < PB) poff[lane + 1] = ic;
                if (lane == 0) poff[0] = 0;
                if (lane < PB) {
                    const float zrow = __int_as_float(R * RS);
                    for (int e = ic - padded + cnt; e < ic; ++e) ent[e] = make_float2(0.f, zrow);
                }
            }
            wave_lds_sync();
            // all list offsets first (one LDS round trip), then the predicated entry writes
            int epos[RPL][4];
#pragma unroll
            for (int i = 0; i < RPL; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) epos[i][k] = poff[pixk[i][k]] + rank[i][k];
#pragma unroll
            for (int i = 0; i < RPL; ++i)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (BOXATTN_TUNE_ABLATE == 2 || pixk[i][k] >= PB) continue;
                    const float slot = __int_as_float((i * 64 + lane) * RS);   // row's LDS offset
                    // instance attention: both weights are already in the staged row
                    ent[epos[i][k]] = make_float2(INST ? wk[i][k] : wk[i][k] * as_c[i], slot);
                }
            wave_lds_sync();
            // ---- phase 2: lane = (destination pixel, channel half): sum w * row over the
            //      pixel's (padded) list, UNR entries per step (independent LDS reads in flight)
            const int e0 = poff[mypix], e1 = poff[mypix + 1];
            Entry en_n[UNR];                           // entries of the next step (prefetched)
#pragma unroll
            for (int u = 0; u < UNR; ++u) en_n[u] = ent[e0 + u];
            for (int e = e0; e < (BOXATTN_TUNE_ABLATE == 1 ? e0 : e1); e += UNR) {
                Entry en[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    en[u] = en_n[u];
                    en_n[u] = ent[e + UNR + u];        // may run into the next list: not used then
                }
                float wa[UNR];
                int jj[UNR];                           // LDS byte offset of the entry's row
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    wa[u] = en[u].x;
                    jj[u] = __float_as_int(en[u].y);
                }
                // rows as raw words, then packed math: one v_pk_fma_f32 per channel pair
                u32x4 rw[UNR][NQ];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const u32x4 *gp = reinterpret_cast<const u32x4 *>(
                        &gstage[jj[u] + half * (CH * SB)]);
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
                        rw[u][q] = BOXATTN_TUNE_ABLATE == 4 ? u32x4{(unsigned)jj[u], 1u, 2u, 3u} : gp[q];
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    const f32x2 w2 = {wa[u], wa[u]};
#pragma unroll
                    for (int i = 0; i < CH / 2; ++i)
                        acc[i] = __builtin_elementwise_fma(w2, staged_pair(rw[u], i), acc[i]);
                }
            }
            wave_lds_sync();
            if (more) {                                // stage round r+1 (rows have arrived)
                if (BOXATTN_TUNE_ABLATE != 3) { BOXATTN_STAGE_ROWS(as_n, al_n) }
#pragma unroll
                for (int i = 0; i < RPL; ++i) { xy_c[i] = xy_n[i]; as_c[i] = as_n[i]; al_c[i] = al_n[i]; }
            }
        }
#undef BOXATTN_FETCH_ROWS
#undef BOXATTN_STAGE_ROWS
        // ---- store.  A block handled by one item writes its rows once, in the storage type (no
        //      zero-fill before, no conversion pass after).  A block cut into chunks writes one
        //      fp32 partial tile per chunk; combine_partials_kernel sums them.  No float atomics
        //      anywhere.
        if (item.w < 0) {
            const int yy = oy + mypix / BW, xx = ox + mypix % BW;
            if (mypix / BW < bh && mypix % BW < bw) {
                ST *dst = grad_value +
                          (((size_t)b * S + lv.start + (size_t)yy * lv.W + xx) * H + h) * C +
                          half * CH;
#pragma unroll
                for (int c = 0; c < CH; c += EPL) {
                    float t[EPL];
#pragma unroll
                    for (int i = 0; i < EPL; i += 2) {
                        t[i] = acc[(c + i) / 2].x; t[i + 1] = acc[(c + i) / 2].y;
                    }
                    VecIO<ST, EPL>::st(dst + c, t);
                }
            }
        } else {
            // a chunk: fp32 partial tile; the block's last chunk to finish sums them (chunk_finish)
            const bool publish = cc.tickets != nullptr;
            const __amdgpu_buffer_rsrc_t tile =
                partial_tile(partials, s, plan.pslot_cap, item.w & ((1 << kItemSlotBits) - 1), C);
#pragma unroll
            for (int c = 0; c < CH; c += 4)
                partial_store(tile, (unsigned)((mypix * C + half * CH + c) * 4),
                              make_float4(acc[c / 2].x, acc[c / 2].y, acc[c / 2 + 1].x, acc[c / 2 + 1].y), publish);
            if (publish) chunk_finish<ST, C>(cc, partials, lv.start, lv.W, S, H, grad_value, s, item.w, lane, &last_flag);
        }
    }
}

// ---------------------------------------------------------------------------------------
// 6: sum the partial tiles of the blocks that were cut into chunks.  grid = (workers, slices),
//    one wavefront each, looping over the slice's chunked blocks (only the coarse levels).
// ---------------------------------------------------------------------------------------
inline CombinePlan combine_plan(const BinPlan &p)
{
    CombinePlan c;
    c.nblk = p.nblk; c.pslot_cap = p.pslot_cap; c.n_slices = p.n_slices;
    for (int k = 0; k < kMaxBinLevels; ++k) { c.start[k] = p.lv[k].start; c.W[k] = p.lv[k].W; }
    return c;
}
// The zero workers' geometry table of a sparse map (boxattn_scan_tail.h: ZeroRole), once per plan.
__global__ __launch_bounds__(256) void zero_geo_kernel(BinPlan plan, int2 *__restrict__ geo)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= plan.nblk) return;
    int level = 0;
#pragma unroll
    for (int l = 1; l < kMaxBinLevels; ++l)
        if (l < plan.L && k >= plan.lv[l].blk0) level = l;
    BinLevel lv = plan.lv[0];
#pragma unroll
    for (int l = 1; l < kMaxBinLevels; ++l)
        if (l == level) lv = plan.lv[l];
    const BlockGeo bg = unpack_block_geo(pack_block_geo(lv, level, k));
    geo[k] = make_int2(lv.start + bg.oy * lv.W + bg.ox, lv.W | ((bg.bh - 1) << 16) | ((bg.bw - 1) << 18));
}

template <typename ST, int C>
__global__ __launch_bounds__(64) void combine_partials_kernel(const int4 *__restrict__ combos,
                                                              const int *__restrict__ n_items,
                                                              const float *__restrict__ partials,
                                                              CombinePlan plan, int S, int H,
                                                              ST *__restrict__ grad_value)
{
    combine_partials_body<ST, C>(combos, n_items, partials, plan, S, H, grad_value, (int)blockIdx.y,
                                 (int)blockIdx.x, (int)gridDim.x, (int)threadIdx.x);
}

}  // namespace boxattn
