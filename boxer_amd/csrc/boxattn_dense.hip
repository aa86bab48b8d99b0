// Translation unit of the kernels that run float32 VALU arithmetic next to MFMAs: the window-staged
// encoder kernels (boxattn_dense.h) and the matrix-core accumulate (boxattn_binned_tr.h).  Built with
// -fno-slp-vectorize (boxer_amd/_lib.py SOURCES): a packed float32 instruction
// (v_pk_mul_f32 / v_pk_fma_f32, which the SLP vectoriser makes of neighbouring scalar operations) issued
// while an MFMA of the same wave is completing was seen to return wrong values on MI355X.
#include "boxattn_dense.h"
#include "boxattn_binned_tr.h"
#include "boxattn_dense_fwd.h"
#include "boxattn_dense_f32.h"

namespace boxattn {

// rider workgroups of a launch: the caller sets ride.grid.n_riders / .shift, the rest follows from the grid
static BinRide place_riders(BinRide ride, unsigned own_blocks, unsigned *total)
{
    ride.grid = ride_grid(ride.grid.n_riders, own_blocks, ride.grid.shift, total);
    return ride;
}

void launch_pointgrad_dense(const uint16_t *value, const float *loc, const float *attn,
                            const uint16_t *grad_out, const DensePlan &dp, float *grad_loc,
                            float *grad_attn, unsigned value_bytes, hipStream_t st, const BinRide &ride_in)
{
    unsigned total = 0;
    const BinRide ride = place_riders(ride_in, dense_blocks(dp), &total);
#define BOXATTN_DENSE_PG(LV_)                                                                           \
    case LV_:                                                                                           \
        hipLaunchKernelGGL((pointgrad_dense_kernel<LV_>), dim3(total), dim3(256), 0, st, value, loc, attn, \
                           grad_out, grad_loc, grad_attn, dp, value_bytes, ride);                      \
        break;
    switch (dp.L) {
        BOXATTN_DENSE_PG(1) BOXATTN_DENSE_PG(2) BOXATTN_DENSE_PG(3) BOXATTN_DENSE_PG(4)
    }
#undef BOXATTN_DENSE_PG
}

void launch_fwd_dense(const uint16_t *value, const float *loc, const float *attn, uint16_t *out,
                      const DensePlan &dp, unsigned value_bytes, const BinRide &ride_in,
                      unsigned long long *stats, hipStream_t st)
{
    unsigned total = 0;
    const BinRide ride = place_riders(ride_in, dense_blocks(dp), &total);
#define BOXATTN_DENSE_FWD(LV_)                                                                       \
    case LV_:                                                                                        \
        hipLaunchKernelGGL((fwd_dense_kernel<LV_>), dim3(total), dim3(256), 0, st, value, loc, attn, out, \
                           dp, value_bytes, ride, stats);                                            \
        break;
    switch (dp.L) {
        BOXATTN_DENSE_FWD(1) BOXATTN_DENSE_FWD(2) BOXATTN_DENSE_FWD(3) BOXATTN_DENSE_FWD(4)
    }
#undef BOXATTN_DENSE_FWD
}

void launch_pointgrad_dense_f32(const float *value, const float *loc, const float *attn, const float *grad_out,
                                const DensePlan &dp, float *grad_loc, float *grad_attn, unsigned value_bytes,
                                hipStream_t st, const BinRide &ride_in)
{
    unsigned total = 0;
    const BinRide ride = place_riders(ride_in, dense_blocks(dp), &total);
#define BOXATTN_DENSE_PG32(LV_)                                                                         \
    case LV_:                                                                                           \
        hipLaunchKernelGGL((pointgrad_dense_f32_kernel<LV_>), dim3(total), dim3(256), 0, st, value, loc, attn, \
                           grad_out, grad_loc, grad_attn, dp, value_bytes, ride);                      \
        break;
    switch (dp.L) {
        BOXATTN_DENSE_PG32(1) BOXATTN_DENSE_PG32(2) BOXATTN_DENSE_PG32(3) BOXATTN_DENSE_PG32(4)
    }
#undef BOXATTN_DENSE_PG32
}

void launch_fwd_dense_f32(const float *value, const float *loc, const float *attn, float *out, const DensePlan &dp,
                          unsigned value_bytes, const BinRide &ride_in, unsigned long long *stats, hipStream_t st)
{
    unsigned total = 0;
    const BinRide ride = place_riders(ride_in, dense_blocks(dp), &total);
#define BOXATTN_DENSE_FWD32(LV_)                                                                     \
    case LV_:                                                                                        \
        hipLaunchKernelGGL((fwd_dense_f32_kernel<LV_>), dim3(total), dim3(256), 0, st, value, loc, attn, out, \
                           dp, value_bytes, ride, stats);                                            \
        break;
    switch (dp.L) {
        BOXATTN_DENSE_FWD32(1) BOXATTN_DENSE_FWD32(2) BOXATTN_DENSE_FWD32(3) BOXATTN_DENSE_FWD32(4)
    }
#undef BOXATTN_DENSE_FWD32
}

void launch_accumulate_tr(int C, const uint16_t *grad_out, size_t grad_out_bytes, const BinPlan &plan, int S,
                          int H, int Lq, const int4 *items, const int *n_items, const int *records,
                          uint16_t *grad_value, float *partials, int wg_per_slice, int ns8, const ChunkCombine &cc,
                          const ZeroRole &zr, hipStream_t st)
{
#define BOXATTN_ACC_TR(C_)                                                                              \
    hipLaunchKernelGGL((binned_accumulate_tr_kernel<uint16_t, C_>), dim3(wg_per_slice + plan.zero_workers, ns8), dim3(64), 0, st, \
                       grad_out, (unsigned)grad_out_bytes, plan, S, H, Lq, items, n_items, records,     \
                       grad_value, partials, cc, zr)
    switch (C) {
    case 16: BOXATTN_ACC_TR(16); break;
    case 32: BOXATTN_ACC_TR(32); break;
    case 64: BOXATTN_ACC_TR(64); break;
    }
#undef BOXATTN_ACC_TR
}

void launch_accumulate_split(const float *grad_out, size_t grad_out_bytes, const BinPlan &plan, int S, int H, int Lq,
                             const int4 *items, const int *n_items, const int *records, float *grad_value,
                             float *partials, int wg_per_slice, int ns8, const ChunkCombine &cc, const ZeroRole &zr,
                             hipStream_t st, const float *grad_mask, size_t grad_mask_bytes, const float *w_lv, int P)
{
    const InstRows inst{grad_mask, (unsigned)grad_mask_bytes, w_lv, P};
    if (grad_mask)      // instance attention: two upstream rows per record
        hipLaunchKernelGGL((binned_accumulate_split_kernel<32, true>), dim3(wg_per_slice + plan.zero_workers, ns8), dim3(64), 0, st,
                           grad_out, (unsigned)grad_out_bytes, plan, S, H, Lq, items, n_items, records, grad_value, partials, cc,
                           zr, inst);
    else
        hipLaunchKernelGGL((binned_accumulate_split_kernel<32, false>), dim3(wg_per_slice + plan.zero_workers, ns8), dim3(64), 0, st,
                           grad_out, (unsigned)grad_out_bytes, plan, S, H, Lq, items, n_items, records, grad_value, partials, cc,
                           zr, inst);
}

void launch_accumulate_f32(const float *grad_out, size_t grad_out_bytes, const BinPlan &plan, int S, int H, int Lq,
                           const int4 *items, const int *n_items, const int *records, float *grad_value,
                           float *partials, int wg_per_slice, int ns8, const ChunkCombine &cc, const ZeroRole &zr,
                           hipStream_t st)
{
    hipLaunchKernelGGL((binned_accumulate_f32_kernel<32>), dim3(wg_per_slice + plan.zero_workers, ns8), dim3(64), 0, st, grad_out,
                       (unsigned)grad_out_bytes, plan, S, H, Lq, items, n_items, records, grad_value, partials, cc, zr);
}

}  // namespace boxattn
