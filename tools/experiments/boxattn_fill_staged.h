// RETIRED EXPERIMENT (round 2) -- not compiled into the library.
//
// Fill pass of the binned backward with a workgroup's records sorted by bin in LDS and streamed
// out in runs, instead of every lane storing its own records (bin_kernel<FILL> in
// boxer_amd/csrc/boxattn_binned.h, which it was included into, after bin_kernel).
// Built to parity (402 GPU tests green) and measured at C2: binning passes 47-52 -> 64 us (bf16,
// 16-byte records), 40 -> 57 us (fp32, 4-byte records); C3' step 396 -> 427 us.  The PMC picture
// that motivated it (54 % of the direct fill's wave cycles wait for an issue slot) is the LDS
// atomics and the geometry, not the scattered stores: the direct kernel without its stores
// still takes 14 of its 24-29 us, and three more barriers, a scan and the staging traffic per
// 1 024-point chunk cost more than the coalescing returns.
// ---------------------------------------------------------------------------------------
// 3, staged: the fill pass with its records sorted by bin in LDS before they leave the workgroup.
// The direct fill (bin_kernel<FILL>) stores every record from the lane that produced it: a wave's
// store instruction is 64 separate 16-byte (4-byte) writes to as many bins, and the pass is bound
// by issuing them (PMC: 54 % of the wave cycles wait for an issue slot; 14 us without the stores,
// 24-29 with).  Here a workgroup takes its points in chunks of kStagePts: (a) geometry, rank of
// every record inside its bin (LDS integer atomics, as before), (b) exclusive scan of the chunk's
// per-bin counts, (c) every record is written to its sorted position in an LDS staging buffer
// together with its global slot, (d) the staging buffer is streamed out by consecutive lanes --
// consecutive records of a bin are consecutive in memory, so a wave's store covers a few whole
// runs instead of 64 pieces.  A chunk with more records than the buffer holds (more than 1.5 per
// point: possible, not seen) stores directly as before.  grid = (workgroups, slices), block
// kBinThreads, dynamic LDS stage_lds_bytes().
// ---------------------------------------------------------------------------------------
#ifndef BOXATTN_TUNE_STAGE_U
#define BOXATTN_TUNE_STAGE_U 2
#endif
constexpr int kStageU = BOXATTN_TUNE_STAGE_U;            // points per thread and chunk
constexpr int kStagePts = kBinThreads * kStageU;
constexpr int kStageCap = kStagePts * 3 / 2;             // records staged per chunk
constexpr size_t stage_lds_bytes(int nblk, bool wide)
{
    return (size_t)3 * ((nblk + 3) / 4 * 4) * 4 + (size_t)kStageCap * (wide ? 20 : 8);
}

template <int BW, int BH, bool WIDE, bool INTERLEAVE>
__global__ __launch_bounds__(kBinThreads) void bin_fill_staged_kernel(
    const float *__restrict__ loc, const float *__restrict__ w_sp, BinPlan plan, int H, int Lq, int P,
    int q_per_wg, int n_wg, const int *__restrict__ part, const int *__restrict__ subtot,
    const int *__restrict__ offsets, int *__restrict__ records)
{
    extern __shared__ int sh_bins[];
    const int nb4 = (plan.nblk + 3) / 4 * 4;
    int *gbase = sh_bins, *cnt = sh_bins + nb4, *lstart = sh_bins + 2 * nb4;
    typedef typename std::conditional<WIDE, int4, int>::type Rec;
    Rec *stage_rec = reinterpret_cast<Rec *>(sh_bins + 3 * nb4);        // 16-byte aligned (nb4 % 4 == 0)
    int *stage_slot = reinterpret_cast<int *>(stage_rec + kStageCap);
    __shared__ int wave_tot[kBinThreads / 64];
    __shared__ BinLevel s_lv[kMaxBinLevels];

    const int s = blockIdx.y, wg = blockIdx.x;
    const int b = s / H, h = s % H;
    const int LP = plan.L * P;
    constexpr bool kInterleave = BOXATTN_TUNE_INTERLEAVE && INTERLEAVE;
    const int q0 = kInterleave ? wg : wg * q_per_wg, qstep = kInterleave ? n_wg : 1;
    const int n_q = kInterleave ? (q0 < Lq ? (Lq - q0 + qstep - 1) / qstep : 0)
                                : max(0, min(q0 + q_per_wg, Lq) - q0);
    const int n_pts = n_q * LP;
    const int *mypart = part + ((size_t)s * n_wg + wg) * plan.nblk;
    const int wps = (n_wg + kScanSub - 1) / kScanSub;
    const int *mysub = subtot + ((size_t)s * kScanSub + wg / wps) * plan.nblk;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int k = 0; k < kMaxBinLevels; ++k) s_lv[k] = plan.lv[k];
    }
    for (int k = threadIdx.x; k < plan.nblk; k += blockDim.x) {
        gbase[k] = mypart[k] + mysub[k] + offsets[(size_t)s * (plan.nblk + 1) + k];
        cnt[k] = 0;
    }
    __syncthreads();

    const float2 *loc2 = reinterpret_cast<const float2 *>(loc);
    const size_t pid0 = (((size_t)b * Lq + q0) * H + h) * LP;
    const size_t qstride = (size_t)H * LP * qstep;
    const float rcp_lp = 1.0f / (float)LP, rcp_p = 1.0f / (float)P;
    Rec *rec = reinterpret_cast<Rec *>(records + (size_t)s * plan.rec_cap * (WIDE ? 4 : 1));
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ipt = (plan.nblk + (int)blockDim.x - 1) / (int)blockDim.x;   // scan strip per thread

    for (int c0 = 0; c0 < n_pts; c0 += kStagePts) {
        // ---- (a) geometry + ranks
        float2 xy[kStageU];
        float wv_[kStageU];
        int ids[kStageU];
        int blk[kStageU][4], rank[kStageU][4];
#pragma unroll
        for (int u = 0; u < kStageU; ++u) {
            const int i = min(c0 + u * (int)blockDim.x + (int)threadIdx.x, n_pts - 1);
            int ql, lp;
            divmod_small(i, LP, rcp_lp, ql, lp);
            const size_t at = pid0 + ql * qstride + lp;
            xy[u] = loc2[at];
            wv_[u] = WIDE ? w_sp[at] : 0.f;
            ids[u] = ((q0 + ql * qstep) << plan.lp_bits) | lp;
        }
#pragma unroll
        for (int u = 0; u < kStageU; ++u) {
            const bool live = c0 + u * (int)blockDim.x + (int)threadIdx.x < n_pts;
            const int lp = ids[u] & ((1 << plan.lp_bits) - 1);
            const BinLevel lv = s_lv[(int)(((float)lp + 0.5f) * rcp_p)];       // level = lp / P
            touched_blocks(xy[u].x, xy[u].y, lv, blk[u]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (!live) blk[u][j] = -1;
                rank[u][j] = 0;
                if (blk[u][j] >= 0) rank[u][j] = atomicAdd(&cnt[blk[u][j]], 1);      // LDS, predicated
            }
        }
        __syncthreads();
        // ---- (b) exclusive scan of the chunk's counts -> first staging position of every bin
        int strip = 0;
        for (int k = 0; k < ipt; ++k) {
            const int idx = (int)threadIdx.x * ipt + k;
            strip += idx < plan.nblk ? cnt[idx] : 0;
        }
        int inc = strip;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(inc, o, 64);
            if (lane >= o) inc += t;
        }
        if (lane == 63) wave_tot[wv] = inc;
        __syncthreads();
        int pre = inc - strip, total = 0;
#pragma unroll
        for (int w = 0; w < kBinThreads / 64; ++w) {
            pre += w < wv ? wave_tot[w] : 0;
            total += wave_tot[w];
        }
        for (int k = 0; k < ipt; ++k) {
            const int idx = (int)threadIdx.x * ipt + k;
            if (idx < plan.nblk) {
                lstart[idx] = pre;
                pre += cnt[idx];
            }
        }
        __syncthreads();
        // ---- (c) records to their sorted staging position (or straight to memory)
        const bool staged = total <= kStageCap;
#pragma unroll
        for (int u = 0; u < kStageU; ++u) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (blk[u][j] >= 0) {
                    const int k = blk[u][j];
                    const int gslot = gbase[k] + rank[u][j];
                    Rec r;
                    if constexpr (WIDE)
                        r = make_int4(ids[u], __float_as_int(xy[u].x), __float_as_int(xy[u].y),
                                      __float_as_int(wv_[u]));
                    else
                        r = ids[u];
                    if (staged) {
                        const int pos = lstart[k] + rank[u][j];
                        stage_rec[pos] = r;
                        stage_slot[pos] = gslot;
                    } else {
                        rec[gslot] = r;
                    }
                }
            }
        }
        __syncthreads();
        // ---- (d) stream the staging buffer out; advance the bins' first slots
        if (staged) {
            for (int i = threadIdx.x; i < total; i += blockDim.x) rec[stage_slot[i]] = stage_rec[i];
        }
        for (int k = threadIdx.x; k < plan.nblk; k += blockDim.x) {
            gbase[k] += cnt[k];
            cnt[k] = 0;
        }
        __syncthreads();
    }
}

