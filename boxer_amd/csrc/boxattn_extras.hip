// C ABI, part two (declared in include/boxattn.h): the opt-in extras around the operator that are
// plain elementwise kernels -- reference windows + box offsets -> sampling grid (first step;
// the modules' `_where_to_attend`, e2edet/module/box_attention.py:63-81, 196-214, 304-338) and the softmax /
// mask-fill + cast passes (N3; box_attention.py:222-231).  The operator itself is boxattn_capi.hip.
#include "../../include/boxattn.h"

#include <hip/hip_runtime.h>

#include "boxattn_pointwise.h"

using namespace boxattn;

namespace {
inline bool aligned(const void *p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
inline int finish() { return (int)hipGetLastError(); }

int grid_dims(int ref_dim, int ref_per_head, int V, int angle_mode, int B, int Lq, int H,
              int L, int P, GridDims &d)
{
    if (B < 0 || Lq < 0 || H <= 0 || L <= 0 || P <= 0 || angle_mode < 0 || angle_mode > 2)
        return 0;
    if (V != (angle_mode == 1 ? 5 : 4) || ref_dim < (angle_mode ? 5 : 4)) return 0;
    d = GridDims{Lq, H, L, P, V, ref_dim, ref_per_head ? 1 : 0, angle_mode};
    return ((size_t)B * Lq == 0) ? 2 : 1;                       // 2: nothing to do
}
}  // namespace

extern "C" {

int boxattn_grid_fwd_f32(const float *ref, int ref_dim, int ref_per_head, const float *offsets,
                         int V, int angle_mode, const float *kernel_idx,
                         const float *valid_ratios, int B, int Lq, int H, int L, int P,
                         float *grid, void *stream)
{
    GridDims d{};
    const int ok = grid_dims(ref_dim, ref_per_head, V, angle_mode, B, Lq, H, L, P, d);
    if (ok == 2) return 0;
    if (ok != 1 || !ref || !offsets || !kernel_idx || !grid) return (int)hipErrorInvalidValue;
    const size_t n_pts = (size_t)B * Lq * H * L * P;
    const size_t blocks = (n_pts + 255) / 256;
    if (blocks > 0x7fffffffu) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(grid_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       ref, offsets, kernel_idx, valid_ratios, d, n_pts, grid);
    return finish();
}

int boxattn_grid_bwd_f32(const float *ref, int ref_dim, int ref_per_head, const float *offsets,
                         int V, int angle_mode, const float *kernel_idx,
                         const float *valid_ratios, const float *grad_grid, int B, int Lq, int H,
                         int L, int P, float *grad_offsets, float *grad_ref_rows, void *stream)
{
    GridDims d{};
    const int ok = grid_dims(ref_dim, ref_per_head, V, angle_mode, B, Lq, H, L, P, d);
    if (ok == 2) return 0;
    if (ok != 1 || !ref || !offsets || !kernel_idx || !grad_grid || !grad_offsets)
        return (int)hipErrorInvalidValue;
    const size_t n_rows = (size_t)B * Lq * H * L;
    const size_t blocks = (n_rows * 4 + 255) / 256;
    if (blocks > 0x7fffffffu) return (int)hipErrorInvalidValue;
    hipLaunchKernelGGL(grid_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                       ref, offsets, kernel_idx, valid_ratios, grad_grid, d, n_rows, grad_offsets,
                       grad_ref_rows);
    return finish();
}

}  // extern "C"

// ---- pointwise work around the operator ---------------------------------
// lanes per row of the vector kernels (rows of 4, 8, 16, 32 or 64 values, 16-byte aligned
// tensors), 0: the one-thread-per-row kernels
template <typename T>
static int softmax_group(int n, const T *typed, const float *f32)
{
    if (n % 4 != 0 || (n / 4 & (n / 4 - 1)) != 0 || n > 64) return 0;
    return aligned(typed, 16) && aligned(f32, 16) ? n / 4 : 0;
}

template <typename T>
static int softmax_fwd(const T *logits, long long rows, int n, float *attn, hipStream_t st)
{
    if (rows < 0 || n <= 0 || n > 64) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    if (!logits || !attn) return (int)hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((rows + 255) / 256);
    const int g = softmax_group(n, logits, attn);
    const size_t total = (size_t)rows * n;
    const unsigned vblocks = (unsigned)((total / 4 + 255) / 256);
#define BOXATTN_SOFTMAX_VEC(G) \
    hipLaunchKernelGGL((softmax_vec_fwd_kernel<T, G>), dim3(vblocks), dim3(256), 0, st, logits, total, attn)
    if (g == 1) BOXATTN_SOFTMAX_VEC(1);
    else if (g == 2) BOXATTN_SOFTMAX_VEC(2);
    else if (g == 4) BOXATTN_SOFTMAX_VEC(4);
    else if (g == 8) BOXATTN_SOFTMAX_VEC(8);
    else if (g == 16) BOXATTN_SOFTMAX_VEC(16);
#undef BOXATTN_SOFTMAX_VEC
    else if (n <= 16)
        hipLaunchKernelGGL((softmax_rows_fwd_kernel<T, 16>), dim3(blocks), dim3(256), 0, st, logits,
                           (size_t)rows, n, attn);
    else
        hipLaunchKernelGGL((softmax_rows_fwd_kernel<T, 64>), dim3(blocks), dim3(256), 0, st, logits,
                           (size_t)rows, n, attn);
    return finish();
}
template <typename T>
static int softmax_bwd(const float *attn, const float *grad_attn, long long rows, int n,
                       T *grad_logits, hipStream_t st)
{
    if (rows < 0 || n <= 0 || n > 64) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    if (!attn || !grad_attn || !grad_logits) return (int)hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((rows + 255) / 256);
    const int g = aligned(grad_attn, 16) ? softmax_group(n, grad_logits, attn) : 0;
    const size_t total = (size_t)rows * n;
    const unsigned vblocks = (unsigned)((total / 4 + 255) / 256);
#define BOXATTN_SOFTMAX_VEC(G) \
    hipLaunchKernelGGL((softmax_vec_bwd_kernel<T, G>), dim3(vblocks), dim3(256), 0, st, attn, \
                       grad_attn, total, grad_logits)
    if (g == 1) BOXATTN_SOFTMAX_VEC(1);
    else if (g == 2) BOXATTN_SOFTMAX_VEC(2);
    else if (g == 4) BOXATTN_SOFTMAX_VEC(4);
    else if (g == 8) BOXATTN_SOFTMAX_VEC(8);
    else if (g == 16) BOXATTN_SOFTMAX_VEC(16);
#undef BOXATTN_SOFTMAX_VEC
    else if (n <= 16)
        hipLaunchKernelGGL((softmax_rows_bwd_kernel<T, 16>), dim3(blocks), dim3(256), 0, st, attn,
                           grad_attn, (size_t)rows, n, grad_logits);
    else
        hipLaunchKernelGGL((softmax_rows_bwd_kernel<T, 64>), dim3(blocks), dim3(256), 0, st, attn,
                           grad_attn, (size_t)rows, n, grad_logits);
    return finish();
}

template <typename T>
static int value_prep(const T *value, const unsigned char *mask, long long rows, int d,
                      uint16_t *out, hipStream_t st)
{
    if (rows < 0 || d <= 0 || d % 8 != 0) return (int)hipErrorInvalidValue;
    if (rows == 0) return 0;
    if (!value || !out || !aligned(value, 16) || !aligned(out, 16)) return (int)hipErrorInvalidValue;
    const size_t n8 = (size_t)rows * d / 8;
    hipLaunchKernelGGL((value_mask_cast_kernel<T>), dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0,
                       st, value, mask, (size_t)rows, d, out);
    return finish();
}

extern "C" {

int boxattn_softmax_fwd_f32(const float *logits, long long rows, int n, float *attn, void *stream)
{
    return softmax_fwd<float>(logits, rows, n, attn, (hipStream_t)stream);
}
int boxattn_softmax_fwd_bf16(const uint16_t *logits, long long rows, int n, float *attn, void *stream)
{
    return softmax_fwd<bf16_t>(logits, rows, n, attn, (hipStream_t)stream);
}
int boxattn_softmax_bwd_f32(const float *attn, const float *grad_attn, long long rows, int n,
                            float *grad_logits, void *stream)
{
    return softmax_bwd<float>(attn, grad_attn, rows, n, grad_logits, (hipStream_t)stream);
}
int boxattn_softmax_bwd_bf16(const float *attn, const float *grad_attn, long long rows, int n,
                             uint16_t *grad_logits, void *stream)
{
    return softmax_bwd<bf16_t>(attn, grad_attn, rows, n, grad_logits, (hipStream_t)stream);
}

int boxattn_value_prep_f32(const float *value, const unsigned char *mask, long long rows, int d,
                           uint16_t *out, void *stream)
{
    return value_prep<float>(value, mask, rows, d, out, (hipStream_t)stream);
}
int boxattn_value_prep_bf16(const uint16_t *value, const unsigned char *mask, long long rows, int d,
                            uint16_t *out, void *stream)
{
    return value_prep<bf16_t>(value, mask, rows, d, out, (hipStream_t)stream);
}

}  // extern "C"
