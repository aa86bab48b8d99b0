/*
 * boxattn.h -- C ABI of the MI355X (gfx950) box-attention / instance-attention operator.
 *
 * This is the drop-in boundary for the one native component of kienduynguyen/BoxeR: the
 * `e2edet.ops` extension with its four entry points (reference:
 * e2edet/module/ops/src/vision.cpp:7-12).  Each group of functions below replaces one of
 * them; the reference-side binding a maintainer would add is shown in INTEGRATION.md.
 *
 * Conventions (all functions):
 *   - Every pointer is a DEVICE pointer on the current HIP device, including the int64
 *     shape tables (the reference passes them as CUDA tensors, box_attn.cu:25-26).
 *   - Tensors are dense, row-major, contiguous (reference CHECK_CONTIGUOUS, box_attn.cu:9-11):
 *       value        (B, S, H, C)          S = sum_l H_l*W_l
 *       shapes       (L, 2) int64          (H_l, W_l)
 *       lsi          (L,)   int64          first row of level l inside S
 *       loc          (B, Lq, H, L, P, 2)   normalised [0,1]; [..,0]=x (width), [..,1]=y
 *       attn / spatial_w / level_w         (B, Lq, H, L, P)
 *       out, grad_out                      (B, Lq, H, C)
 *       mask_out, grad_mask                (B, Lq, P, H, C)
 *   - Inputs are borrowed and never written.  Outputs are FULLY DEFINED by the call: the
 *     functions zero-fill whatever they accumulate into (the reference relies on
 *     at::zeros / zeros_like in its host code, box_attn.cu:44,105-107), so callers may pass
 *     uninitialised memory.  Nothing is allocated or freed behind the ABI.
 *   - `stream` is a hipStream_t (NULL = the default stream).  Calls are asynchronous.
 *   - Return value: 0 on success, otherwise a hipError_t value (1 = hipErrorInvalidValue
 *     for bad arguments).  Launch failures are returned, never just printed (the reference
 *     only printf's them, box_attn_kernel.cuh:1118-1122).
 *   - There is no `im2col_step`: in the reference it is pure batch chunking of the launch
 *     (box_attn.cu:40-66) with no numerical effect.  One call covers the whole batch.
 *
 * Numeric types (suffix):
 *   _f32   everything float32                      (the reference's training path)
 *   _f64   everything float64                      (the reference's gradcheck path)
 *   _bf16  value / out / mask_out / grad_out / grad_mask / grad_value are bfloat16;
 *          loc, weights and their gradients stay float32; accumulation is float32.
 *          (New: the reference raises on BFloat16, box_attn.cu:54.)  The backward needs a
 *          float32 scratch of B*S*H*C elements for the grad_value accumulation.
 */
#ifndef BOXATTN_H_
#define BOXATTN_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version of this header; bumped on any signature change. */
#define BOXATTN_ABI_VERSION 8
int boxattn_abi_version(void);

/* Static description of the build ("gfx950", compiler, kernel variants); never NULL. */
const char *boxattn_build_info(void);

/* ---- replaces box_attn_forward (box_attn.h:29-54, box_attn.cu:15-71) ------------------ */
int boxattn_fwd_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                    const float *loc, const float *attn, int B, int S, int H, int C, int L,
                    int Lq, int P, float *out, void *stream);
int boxattn_fwd_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                    const double *loc, const double *attn, int B, int S, int H, int C, int L,
                    int Lq, int P, double *out, void *stream);
int boxattn_fwd_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                     const float *loc, const float *attn, int B, int S, int H, int C, int L,
                     int Lq, int P, uint16_t *out, void *stream);

/* Forward with HOST copies of the two level tables next to the device ones (either may be
 * NULL): lets the library recognise the encoder case -- one query per pixel of the packed
 * multi-level map (Lq == S; box_transformer.py:346-354) -- and run the window-staged kernels
 * (a query tile's value windows in LDS, arithmetic on the matrix cores) instead of the row
 * gathers.  Same results as boxattn_fwd_*. */
int boxattn_fwd_hl_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                       const float *loc, const float *attn, int B, int S, int H, int C, int L,
                       int Lq, int P, float *out, const int64_t *shapes_host,
                       const int64_t *lsi_host, void *stream);
int boxattn_fwd_hl_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                        const float *loc, const float *attn, int B, int S, int H, int C, int L,
                        int Lq, int P, uint16_t *out, const int64_t *shapes_host,
                        const int64_t *lsi_host, void *stream);

/* ---- replaces box_attn_backward (box_attn.h:56-83, box_attn.cu:74-135) ---------------- */
int boxattn_bwd_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                    const float *loc, const float *attn, const float *grad_out, int B, int S,
                    int H, int C, int L, int Lq, int P, float *grad_value, float *grad_loc,
                    float *grad_attn, void *stream);
int boxattn_bwd_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                    const double *loc, const double *attn, const double *grad_out, int B,
                    int S, int H, int C, int L, int Lq, int P, double *grad_value,
                    double *grad_loc, double *grad_attn, void *stream);
/* grad_value_ws: float32 scratch, B*S*H*C elements, contents ignored and clobbered. */
int boxattn_bwd_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                     const float *loc, const float *attn, const uint16_t *grad_out, int B,
                     int S, int H, int C, int L, int Lq, int P, uint16_t *grad_value,
                     float *grad_loc, float *grad_attn, float *grad_value_ws, void *stream);

/* ---- replaces instance_attn_forward (instance_attn.h:32-59, instance_attn.cu:15-82) --- */
int instattn_fwd_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                     const float *loc, const float *spatial_w, const float *level_w, int B,
                     int S, int H, int C, int L, int Lq, int P, float *out, float *mask_out,
                     void *stream);
int instattn_fwd_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                     const double *loc, const double *spatial_w, const double *level_w, int B,
                     int S, int H, int C, int L, int Lq, int P, double *out, double *mask_out,
                     void *stream);
int instattn_fwd_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                      const float *loc, const float *spatial_w, const float *level_w, int B,
                      int S, int H, int C, int L, int Lq, int P, uint16_t *out,
                      uint16_t *mask_out, void *stream);

/* ---- replaces instance_attn_backward (instance_attn.h:61-92, instance_attn.cu:85-157) - */
int instattn_bwd_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                     const float *loc, const float *spatial_w, const float *level_w,
                     const float *grad_out, const float *grad_mask, int B, int S, int H, int C,
                     int L, int Lq, int P, float *grad_value, float *grad_loc,
                     float *grad_spatial_w, float *grad_level_w, void *stream);
int instattn_bwd_f64(const double *value, const int64_t *shapes, const int64_t *lsi,
                     const double *loc, const double *spatial_w, const double *level_w,
                     const double *grad_out, const double *grad_mask, int B, int S, int H,
                     int C, int L, int Lq, int P, double *grad_value, double *grad_loc,
                     double *grad_spatial_w, double *grad_level_w, void *stream);
int instattn_bwd_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                      const float *loc, const float *spatial_w, const float *level_w,
                      const uint16_t *grad_out, const uint16_t *grad_mask, int B, int S, int H,
                      int C, int L, int Lq, int P, uint16_t *grad_value, float *grad_loc,
                      float *grad_spatial_w, float *grad_level_w, float *grad_value_ws,
                      void *stream);

/*
 * ---- backward with a caller-provided workspace (the fast path) ---------------------------
 * Same contract as the plain backward entry points above, plus:
 *   shapes_host / lsi_host : HOST copies of the two int64 level tables (the reference keeps
 *       them only on the device; the binding caches a host copy per tensor).  They let the
 *       library plan the destination-binned algorithm without a
 *       device->host sync.  May be NULL: the call then behaves like the plain backward.
 *   workspace / workspace_bytes : device scratch, 256-byte aligned, contents ignored and
 *       clobbered; size from boxattn_bwd_workspace_bytes() (covers the bf16 fp32 scratch too).
 *       It only lives inside the call (bin records, partial tiles).
 *   plan / plan_bytes : NULL / 0 -- the call plans for itself (count + scan passes first) -- or the
 *       buffer a *_fwd_train_* call filled (*plan_built == 1) for the SAME sampling locations,
 *       dimensions and option settings; it is only read.
 *   hints : 0, or BOXATTN_HINT_* bits (see *_fwd_train_*); speed only, never results.
 *   state / state_bytes : NULL / 0, or the caller's state buffer of this (stream, dimensions, level shapes) -- see
 *       *_fwd_train_* below.  With it, box attention whose accumulate runs on the matrix cores (bf16 storage at 16 /
 *       32 / 64 channels per head, float32 at 32) on maps of up to 1 535 blocks of 8x4 pixels per (image, head) fills
 *       its bins in ONE pass: the state keeps, per block, the record range the previous call planned
 *       from its own counts; the fill riders claim slots in those ranges with one returned atomic per block and step;
 *       a block that outgrows its range is recomputed from the sampling locations by a redo worker of the accumulate
 *       launch; every call re-plans the ranges for the next one.  Results never depend on what the state holds; the
 *       first call on a zeroed (or foreign-shaped) state runs the two-pass passes.  Such a call takes no plan.
 * If the shape is not eligible or the workspace is too small, the call falls back to the
 * atomic kernels of the plain entry points (for _bf16 the workspace must then still hold
 * B*S*H*C floats); a plan the backward cannot use (operands the fast paths reject) is ignored.
 * Only float32 and bfloat16 exist here; float64 uses the plain backward.
 * The binned path uses no float atomics and no zero-fill.  Everything runs on `stream`:
 *   [count + scans, unless a plan is given] -> point-gradient kernel with the fill pass riding in its
 *   launch -> accumulate kernel (the partial tiles of chunked blocks summed by their last chunk).
 */
size_t boxattn_bwd_workspace_bytes(int is_bf16, int B, int S, int H, int C, int L, int Lq, int P,
                                   const int64_t *shapes_host, const int64_t *lsi_host);
int boxattn_bwd_ws_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                       const float *loc, const float *attn, const float *grad_out, int B, int S,
                       int H, int C, int L, int Lq, int P, float *grad_value, float *grad_loc,
                       float *grad_attn, const int64_t *shapes_host, const int64_t *lsi_host,
                       void *workspace, size_t workspace_bytes, const void *plan, size_t plan_bytes,
                       void *state, size_t state_bytes, int hints, void *stream);
int boxattn_bwd_ws_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                        const float *loc, const float *attn, const uint16_t *grad_out, int B,
                        int S, int H, int C, int L, int Lq, int P, uint16_t *grad_value,
                        float *grad_loc, float *grad_attn, const int64_t *shapes_host,
                        const int64_t *lsi_host, void *workspace, size_t workspace_bytes,
                        const void *plan, size_t plan_bytes, void *state, size_t state_bytes, int hints,
                        void *stream);
int instattn_bwd_ws_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                        const float *loc, const float *spatial_w, const float *level_w,
                        const float *grad_out, const float *grad_mask, int B, int S, int H, int C,
                        int L, int Lq, int P, float *grad_value, float *grad_loc,
                        float *grad_spatial_w, float *grad_level_w, const int64_t *shapes_host,
                        const int64_t *lsi_host, void *workspace, size_t workspace_bytes,
                        const void *plan, size_t plan_bytes, void *state, size_t state_bytes, int hints,
                        void *stream);
int instattn_bwd_ws_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                         const float *loc, const float *spatial_w, const float *level_w,
                         const uint16_t *grad_out, const uint16_t *grad_mask, int B, int S, int H,
                         int C, int L, int Lq, int P, uint16_t *grad_value, float *grad_loc,
                         float *grad_spatial_w, float *grad_level_w, const int64_t *shapes_host,
                         const int64_t *lsi_host, void *workspace, size_t workspace_bytes,
                         const void *plan, size_t plan_bytes, void *state, size_t state_bytes, int hints,
                        void *stream);

/*
 * ---- reference windows + box offsets -> sampling grid (opt-in) ---------------
 * Everything of the modules' `_where_to_attend` after the box-offset projection
 * (e2edet/module/box_attention.py:63-81 BoxAttention / InstanceAttention, :304-338
 * Box3dAttention) as one kernel each way:
 *     box   = ref[:4] + offsets[:4] / 8 * (w, h, w, h)_ref
 *     theta = none (angle_mode 0) | (ref[4] + offsets[4] / 16) * 2 pi (1) | ref[4] (2)
 *     grid[b,q,h,l,p,:] = (c + R(theta) (kernel_idx[p] * relu(size))) * valid_ratios[b,l,:]
 *   ref          (B, Lq, ref_dim) or, ref_per_head != 0, (B, Lq, H, ref_dim); float32,
 *                ref_dim >= 4 (>= 5 with an angle); columns beyond the angle are ignored
 *   offsets      (B, Lq, H, L, V) float32, V = 4, or 5 for angle_mode 1
 *   kernel_idx   (P, 2) float32, the module's `kernel_indices` buffer
 *   valid_ratios (B, L, 2) float32 or NULL (`v_valid_ratios`, (B,1,1,L,1,2) in the reference)
 *   grid / grad_grid (B, Lq, H, L, P, 2) float32 -- the `sampling_locations` of the operator
 *   grad_offsets (B, Lq, H, L, V) float32, fully written
 *   grad_ref_rows (B, Lq, H, L, 5) float32 or NULL: per-(head, level) gradient of
 *                (cx, cy, w, h, angle)_ref; the caller sums it over L (and H) if it needs
 *                the gradient of the reference windows
 */
int boxattn_grid_fwd_f32(const float *ref, int ref_dim, int ref_per_head, const float *offsets,
                         int V, int angle_mode, const float *kernel_idx,
                         const float *valid_ratios, int B, int Lq, int H, int L, int P,
                         float *grid, void *stream);
int boxattn_grid_bwd_f32(const float *ref, int ref_dim, int ref_per_head, const float *offsets,
                         int V, int angle_mode, const float *kernel_idx,
                         const float *valid_ratios, const float *grad_grid, int B, int Lq, int H,
                         int L, int P, float *grad_offsets, float *grad_ref_rows, void *stream);

/*
 * ---- training forward: forward + backward plan ----------------------------------------------
 * Same as the plain forward, and additionally (when the binned backward applies) the count pass and
 * the scans of the backward -- they only depend on the sampling locations -- ride in the forward
 * kernel's launch as extra workgroups and leave the backward's PLAN in `plan` (boxattn_plan_bytes()
 * bytes, 256-byte aligned, contents ignored on entry: per bin workgroup and block the first record
 * slot, the work-item list; 1.4 MB at BoxeR-R50 encoder shapes).  It must stay untouched until the
 * matching *_bwd_ws_* call.  *plan_built is 1 if a plan was built, 0 if the call was just a plain
 * forward (shape not eligible / buffer too small / NULL).
 * state / state_bytes: NULL / 0, or a device buffer of at least boxattn_state_bytes(dimensions, level tables) bytes,
 * 8-byte aligned, that the caller ZEROED ONCE and keeps for the calls it issues on ONE stream with ONE set of dimensions
 * and level shapes (ABI 8; until ABI 7 one buffer served every shape of its stream): counters, the riders' hand-off
 * tickets (every call leaves them zero), and the record ranges of the backward's one-pass fill (*_bwd_ws_*: for the
 * shapes that fill takes, the training forward carries nothing of the backward and builds no plan -- *plan_built = 0).
 * A non-NULL state that is misaligned or too small is an error (hipErrorInvalidValue), not "no state".  The library
 * remembers (host side) which shape a buffer served last and zeroes a buffer that turns up with another one.
 * Without a state the tickets live in `plan` and a zero-fill launch (~5 us) precedes the forward kernel.
 * The FIRST 1 KiB of the state buffer holds 64 pairs of uint64 counters (then 64 bytes of counters of the one-pass
 * fill: {calls, blocks recomputed because they outgrew their range}; the tickets follow)
 * that the window-staged forward of the encoder case only ever ADDS to: {sample points it had to fetch from
 * global memory because their footprint missed the staged window, sample points inside the window test}.  A
 * caller may read them whenever it likes (e.g. an asynchronous copy after a call; deltas between two reads; a
 * caller that wants them per shape keeps one state buffer per shape and stream, as boxer_amd.ops does)
 * and, when most points miss -- sampling locations that are not local to their query, e.g. uniformly random
 * ones -- pass BOXATTN_HINT_NOT_LOCAL in `hints` of its next calls: forward and point gradients then run on
 * the row-gather kernels (same results; measured 14.5 against 12.8 Gpts/s on uniformly random locations,
 * 17 against 25 on BoxeR's).  boxer_amd.ops does exactly that.
 */
#define BOXATTN_HINT_NOT_LOCAL 1
/* The state buffer passed with this call was zeroed by the caller since its last call (typically: a newly allocated
 * one, possibly at an address a released buffer had): the library drops what it remembers (host side) of that address,
 * so that the backward's first call on it runs the two-pass passes (and plans the ranges) instead of recomputing every
 * block of a state it believes planned.  Speed only: a zeroed state gives the same results either way. */
#define BOXATTN_HINT_FRESH_STATE 2
size_t boxattn_state_bytes(int B, int S, int H, int C, int L, int Lq, int P, const int64_t *shapes_host,
                           const int64_t *lsi_host);
size_t boxattn_plan_bytes(int is_bf16, int B, int S, int H, int C, int L, int Lq, int P,
                          const int64_t *shapes_host, const int64_t *lsi_host);
int boxattn_fwd_train_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                          const float *loc, const float *attn, int B, int S, int H, int C, int L,
                          int Lq, int P, float *out, const int64_t *shapes_host,
                          const int64_t *lsi_host, void *plan, size_t plan_bytes, void *state,
                          size_t state_bytes, int hints, int *plan_built, void *stream);
int boxattn_fwd_train_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                           const float *loc, const float *attn, int B, int S, int H, int C, int L,
                           int Lq, int P, uint16_t *out, const int64_t *shapes_host,
                           const int64_t *lsi_host, void *plan, size_t plan_bytes, void *state,
                           size_t state_bytes, int hints, int *plan_built, void *stream);
int instattn_fwd_train_f32(const float *value, const int64_t *shapes, const int64_t *lsi,
                           const float *loc, const float *spatial_w, const float *level_w, int B,
                           int S, int H, int C, int L, int Lq, int P, float *out, float *mask_out,
                           const int64_t *shapes_host, const int64_t *lsi_host, void *plan,
                           size_t plan_bytes, void *state, size_t state_bytes, int hints, int *plan_built,
                           void *stream);
int instattn_fwd_train_bf16(const uint16_t *value, const int64_t *shapes, const int64_t *lsi,
                            const float *loc, const float *spatial_w, const float *level_w, int B,
                            int S, int H, int C, int L, int Lq, int P, uint16_t *out,
                            uint16_t *mask_out, const int64_t *shapes_host,
                            const int64_t *lsi_host, void *plan, size_t plan_bytes, void *state,
                            size_t state_bytes, int hints, int *plan_built, void *stream);

/*
 * Kernel-variant override for tests and A/B benchmarks (process-global, not thread-safe):
 *   0 = automatic choice (default), 1 = force the generic kernels (any C),
 *   2 = first-generation fast kernels with fp atomics (error if the shape does not qualify;
 *       never the binned backward),
 *   3 = like 0, but an error if the binned backward is not eligible.
 * Returns the previous value.
 */
int boxattn_set_variant(int variant);

/*
 * ---- pointwise work around the operator (opt-in, beyond the reference's native module) ------
 * The modules' softmax over the L*P attention logits of every (query, head) and the zeroing of
 * padded value rows (reference e2edet/module/box_attention.py:222-231), as single passes:
 *   boxattn_softmax_fwd_*: logits (rows, n) float32 / bfloat16 -> attn (rows, n) float32, n <= 64
 *   boxattn_softmax_bwd_*: grad_logits = attn * (grad_attn - sum_j attn_j grad_attn_j), written
 *                          in the logits' type
 *   boxattn_value_prep_*:  value (rows, d) float32 / bfloat16 -> bfloat16 with the rows whose mask
 *                          byte is non-zero set to 0 (mask may be NULL); d % 8 == 0
 */
int boxattn_softmax_fwd_f32(const float *logits, long long rows, int n, float *attn, void *stream);
int boxattn_softmax_fwd_bf16(const uint16_t *logits, long long rows, int n, float *attn,
                             void *stream);
int boxattn_softmax_bwd_f32(const float *attn, const float *grad_attn, long long rows, int n,
                            float *grad_logits, void *stream);
int boxattn_softmax_bwd_bf16(const float *attn, const float *grad_attn, long long rows, int n,
                             uint16_t *grad_logits, void *stream);
int boxattn_value_prep_f32(const float *value, const unsigned char *mask, long long rows, int d,
                           uint16_t *out, void *stream);
int boxattn_value_prep_bf16(const uint16_t *value, const unsigned char *mask, long long rows, int d,
                            uint16_t *out, void *stream);

/*
 * Tuning options for A/B runs (process-wide, relaxed atomics; 0 = default).  Returns the
 * previous value, -1 for an unknown key.  (The key numbers of earlier ABI versions are kept; the keys
 * of kernels that were removed are gone.)
 *  10  binned backward: records per work item (multiple of 64; default: from the number of sample
 *      points, 128 ... 1024).  Set it before boxattn_plan_bytes / boxattn_bwd_workspace_bytes: the
 *      layouts depend on it.
 *  11  window-staged kernels of the encoder case (box attention, bf16 or float32 storage, Lq == S, C = 32,
 *      2x2 points, <= 4 levels; forward and point gradients): 0 library default (on), 1 off
 *      (row-gather kernels -- the parity cross-check of the two kernel families; faster for uniformly random
 *      sampling locations, which the BOXATTN_HINT_NOT_LOCAL hint selects per call), 2 on
 *  15  binning passes of the backward as riders: 0 default -- the fill pass inside the point-gradient
 *      launch, chunked blocks summed inside the accumulate launch (wherever the riders run -- maps of up to 3 072
 *      blocks per (image, head) -- or the problem has < 65 536 sample points per (image, head); by a combine launch
 *      behind it otherwise); box attention on the matrix-core accumulates fills its bins in ONE pass from the ranges in
 *      the caller's state buffer (*_bwd_ws_*), everything else counts and scans inside the training forward's launch --,
 *      1 off (launches of their own), 2 on with the combine always a launch of its own, 3 on with the combine always
 *      inside, 4 on with the two-pass binning of ABI 7 for everything (count riders in the training forward, plan
 *      hand-over) instead of the one-pass fill
 *  19  float32, 32 channels per head: the grad_value accumulate -- 0 default: the bf16 matrix cores on exact
 *      three-term splits of rows and weights (float32-accurate, 16-byte records; instance attention from
 *      65 536 points per (image, head) slice up), 1: VALU list walk (4-byte records; the parity cross-check),
 *      2: v_mfma_f32_32x32x2_f32 (16-byte records; box attention only, instance attention takes the VALU walk).
 *      Set before boxattn_bwd_workspace_bytes / *_fwd_train_*.
 *  20  where the riders sit in their host kernel's grid: (s_count + 1) | (s_fill + 1) << 4 -- a group of 8
 *      rider workgroups every 2^s groups of 8 workgroups (s = 0: all in front) -- | v << 8: 64 v bin workgroups
 *      (= riders) in all; 0 = defaults (all in front; 256, or one per ~600 (query, slice) pairs -- 256 ... 768 -- for the
 *      one-pass fill at encoder sizes).  Set before boxattn_plan_bytes.
 *  (ABI 8 removed 12 / 13 -- window margins --, 17 and 21 -- staged forward / staged float32 kernels off: 11 = 1
 *  switches every window-staged kernel off.)
 */
int boxattn_set_option(int key, int value);
/* Number of boxattn_set_variant / boxattn_set_option calls so far: lets a binding cache the size queries
 * (boxattn_plan_bytes, boxattn_bwd_workspace_bytes: pure functions of their arguments and the switches). */
int boxattn_options_epoch(void);

/*
 * Debugging aid of the window-staged kernels: builds with -DBOXATTN_DENSE_DEBUG=2 write per-wave time
 * stamps to this device buffer (tools/gpu_dense_trace.py).  Ignored by regular builds.
 */
void boxattn_set_debug_buffer(float *device_buffer);

/*
 * Kernel timing for benchmarks (process-global, not thread-safe).  Between _begin and _end the
 * library brackets the launches of its main kernels with hipEvents recorded on the launch
 * stream, in BOXATTN_PROFILE_SLOTS slots:
 *   0  forward sampling kernel (training forward: + the count / scan riders of its launch)
 *   1  backward, point gradients (grad_loc / grad_weight; + the fill riders of its launch) -- or the
 *      whole atomic backward kernel when the binned algorithm is not used
 *   2  backward, grad_value accumulate kernel (+ the in-launch combine of chunked blocks)
 *   3  backward, binning passes run as launches of their own (count / scan / fill)
 *   4  backward, combine pass over the partial tiles of split blocks as a launch of its own
 *   5  unused
 * Zero-fills and the bf16 conversion pass are not included.  _end synchronises the events and
 * writes, per slot, the summed duration in milliseconds and the number of launches into the two
 * BOXATTN_PROFILE_SLOTS-element arrays.  At most 4096 launches per slot are kept.
 */
#define BOXATTN_PROFILE_SLOTS 6
int boxattn_profile_begin(void);
int boxattn_profile_end(double *ms_sum, int *launches);

#ifdef __cplusplus
}
#endif
#endif /* BOXATTN_H_ */
