// Translation unit of the kernels that run float32 VALU arithmetic next to MFMAs: the window-staged
// encoder kernels (boxattn_dense.h) and the matrix-core accumulate (boxattn_binned_tr.h).  Built with
// -fno-slp-vectorize (boxer_amd/_lib.py SOURCES): a packed float32 instruction
// (v_pk_mul_f32 / v_pk_fma_f32, which the SLP vectoriser makes of neighbouring scalar operations) issued
// while an MFMA of the same wave is completing was seen to return wrong values on MI355X.
#include "boxattn_dense.h"
#include "boxattn_binned_tr.h"
#include "boxattn_dense_fwd.h"

namespace boxattn {

void launch_dense_count(const float *loc, const DensePlan &dp, const DenseBin &bin, hipStream_t st)
{
    const unsigned blocks = dense_blocks(dp);
    const size_t lds = (size_t)bin.nblk * sizeof(int);
#define BOXATTN_DENSE_CNT(LV_)                                                                      \
    case LV_:                                                                                       \
        hipLaunchKernelGGL((dense_count_kernel<LV_>), dim3(blocks), dim3(256), lds, st, loc, dp, bin); \
        break;
    switch (dp.L) {
        BOXATTN_DENSE_CNT(1) BOXATTN_DENSE_CNT(2) BOXATTN_DENSE_CNT(3) BOXATTN_DENSE_CNT(4)
    }
#undef BOXATTN_DENSE_CNT
}

void launch_pointgrad_dense(const uint16_t *value, const float *loc, const float *attn,
                            const uint16_t *grad_out, const DensePlan &dp, float *grad_loc,
                            float *grad_attn, unsigned value_bytes, hipStream_t st,
                            const CombineTail &tail, const DenseBin &bin)
{
    const unsigned tail_blocks = tail.workers > 0 ? (unsigned)(tail.workers * tail.plan.n_slices + 3) / 4 : 0;
    const unsigned blocks = dense_blocks(dp);
    const size_t lds = bin.on ? (size_t)bin.nblk * sizeof(int) : 0;
#define BOXATTN_DENSE_PG(LV_)                                                                           \
    case LV_:                                                                                           \
        if (bin.on)                                                                                     \
            hipLaunchKernelGGL((pointgrad_dense_kernel<LV_, true>), dim3(blocks + tail_blocks), dim3(256), lds, \
                               st, value, loc, attn, grad_out, grad_loc, grad_attn, dp, value_bytes,   \
                               blocks, tail, bin);                                                      \
        else                                                                                            \
            hipLaunchKernelGGL((pointgrad_dense_kernel<LV_, false>), dim3(blocks + tail_blocks), dim3(256), 0, \
                               st, value, loc, attn, grad_out, grad_loc, grad_attn, dp, value_bytes,   \
                               blocks, tail, bin);                                                      \
        break;
    switch (dp.L) {
        BOXATTN_DENSE_PG(1) BOXATTN_DENSE_PG(2) BOXATTN_DENSE_PG(3) BOXATTN_DENSE_PG(4)
    }
#undef BOXATTN_DENSE_PG
}

void launch_fwd_dense(const uint16_t *value, const float *loc, const float *attn, uint16_t *out,
                      const DensePlan &dp, unsigned value_bytes, const ScanTail *scan_tail, hipStream_t st)
{
    const unsigned blocks = dense_blocks(dp);
    const ScanTail sct = scan_tail ? *scan_tail : ScanTail{};
    const unsigned lead = scan_tail ? (unsigned)(sct.plan.n_slices * kScanSub) : 0u;    // in FRONT of the grid
#define BOXATTN_DENSE_FWD(LV_)                                                                       \
    case LV_:                                                                                        \
        hipLaunchKernelGGL((fwd_dense_kernel<LV_>), dim3(blocks + lead), dim3(256), 0, st, value, loc, attn, out, \
                           dp, value_bytes, lead, sct);                                              \
        break;
    switch (dp.L) {
        BOXATTN_DENSE_FWD(1) BOXATTN_DENSE_FWD(2) BOXATTN_DENSE_FWD(3) BOXATTN_DENSE_FWD(4)
    }
#undef BOXATTN_DENSE_FWD
}

void launch_accumulate_tr(int C, const uint16_t *grad_out, size_t grad_out_bytes, const BinPlan &plan, int S,
                          int H, int Lq, const int4 *items, const int *n_items, const int *records,
                          uint16_t *grad_value, float *partials, int wg_per_slice, int ns8, bool rec12, hipStream_t st)
{
#define BOXATTN_ACC_TR(C_, R_)                                                                          \
    hipLaunchKernelGGL((binned_accumulate_tr_kernel<uint16_t, C_, R_>), dim3(wg_per_slice, ns8), dim3(64), 0, st, \
                       grad_out, (unsigned)grad_out_bytes, plan, S, H, Lq, items, n_items, records,     \
                       grad_value, partials)
    switch (C) {
    case 16: if (rec12) BOXATTN_ACC_TR(16, true); else BOXATTN_ACC_TR(16, false); break;
    case 32: if (rec12) BOXATTN_ACC_TR(32, true); else BOXATTN_ACC_TR(32, false); break;
    case 64: if (rec12) BOXATTN_ACC_TR(64, true); else BOXATTN_ACC_TR(64, false); break;
    }
#undef BOXATTN_ACC_TR
}

void launch_accumulate_f32(const float *grad_out, size_t grad_out_bytes, const BinPlan &plan, int S, int H, int Lq,
                           const int4 *items, const int *n_items, const int *records, float *grad_value,
                           float *partials, int wg_per_slice, int ns8, hipStream_t st)
{
    hipLaunchKernelGGL((binned_accumulate_f32_kernel<32>), dim3(wg_per_slice, ns8), dim3(64), 0, st, grad_out,
                       (unsigned)grad_out_bytes, plan, S, H, Lq, items, n_items, records, grad_value, partials);
}

}  // namespace boxattn
