// The elementwise kernels around the operator (one translation unit: boxattn_extras.hip): reference windows +
// box offsets -> sampling grid and back (first step), softmax over the L*P logits of a
// (query, head) each way, value mask-fill + bf16 cast (N3).  The device helpers they share with the sampling
// kernels (grid_box, grid_point, grid_grad_*) live in boxattn_grid.h.
#pragma once
#include "boxattn_grid.h"

namespace boxattn {

__global__ __launch_bounds__(256) void grid_fwd_kernel(const float *__restrict__ ref,
                                                       const float *__restrict__ offsets,
                                                       const float *__restrict__ kidx,
                                                       const float *__restrict__ vr, GridDims d,
                                                       size_t n_pts, float *__restrict__ grid)
{
#pragma clang fp contract(off)
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;      // (row n, point p)
    if (i >= n_pts) return;
    const size_t n = i / (unsigned)d.P;
    const int p = (int)(i - n * (unsigned)d.P);
    const GridBox g = grid_box(ref, offsets, vr, d, n);
    reinterpret_cast<float2 *>(grid)[i] = grid_point(g, kidx, p, vr != nullptr, d.angle_mode);
}

// grad_offsets (N, V) and, if asked for, the per-row gradient of the reference window
// grad_ref_rows (N, 5) = d/d(cx, cy, w, h, angle)_ref (summed over levels / heads by the
// caller): four lanes per row (a DPP quad), points strided over them, quad sum at the end.
__global__ __launch_bounds__(256) void grid_bwd_kernel(const float *__restrict__ ref,
                                                       const float *__restrict__ offsets,
                                                       const float *__restrict__ kidx,
                                                       const float *__restrict__ vr,
                                                       const float *__restrict__ grad_grid,
                                                       GridDims d, size_t n_rows,
                                                       float *__restrict__ grad_offsets,
                                                       float *__restrict__ grad_ref_rows)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t n = t / 4;
    const int j = (int)(t % 4);
    const bool live = n < n_rows;
    if (!live) n = n_rows - 1;                                  // keep the quad together
    const GridBox g = grid_box(ref, offsets, vr, d, n);
    const float2 *gg = reinterpret_cast<const float2 *>(grad_grid) + n * (unsigned)d.P;
    GridGrad a{0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p = j; p < d.P; p += 4) grid_grad_add(a, g, kidx, p, gg[p]);
    a.cx = group_sum<4>(a.cx); a.cy = group_sum<4>(a.cy);
    a.w = group_sum<4>(a.w); a.h = group_sum<4>(a.h); a.t = group_sum<4>(a.t);
    if (!(live && j == 0)) return;
    grid_grad_store(a, g, offsets + n * (unsigned)d.V, d, grad_offsets + n * (unsigned)d.V,
                    grad_ref_rows ? grad_ref_rows + n * 5 : nullptr);
}


// ---------------------------------------------------------------------------------------
// Pointwise work around the operator (reference e2edet/module/box_attention.py:222-231),
// SURVEY.md 8(f) N3:
//   * attention weights = softmax over the L*P logits of a (query, head), computed in float32
//     whatever the logits' type (float32 or the bfloat16 of an autocast projection) -- one pass
//     instead of cast + softmax (+ cast); its backward grad_logits = a (g - sum_j a_j g_j),
//     written in the logits' type;
//   * value rows of padded pixels (v_mask) zeroed and cast to bfloat16 in the same pass
//     (`value.masked_fill(v_mask[..., None], 0)` followed by the op's bf16 conversion).
// Rows whose length is 4 * 2^k (k <= 4; BoxeR: 16 = 4 levels x 2x2 points): 2^k lanes per row,
// four consecutive values per lane, so a wave reads and writes whole contiguous runs
// (element index = 4 * thread) and the row max / sum are cross-lane butterflies.  Any other
// length <= 64: one thread per row (strided accesses; the rare shapes).
// ---------------------------------------------------------------------------------------
template <typename T> __device__ __forceinline__ float pw_ld(const T *p);
template <> __device__ __forceinline__ float pw_ld<float>(const float *p) { return *p; }
template <> __device__ __forceinline__ float pw_ld<bf16_t>(const bf16_t *p) { return bf16_bits_to_f32(*p); }
template <typename T> __device__ __forceinline__ void pw_st(T *p, float v);
template <> __device__ __forceinline__ void pw_st<float>(float *p, float v) { *p = v; }
template <> __device__ __forceinline__ void pw_st<bf16_t>(bf16_t *p, float v) { *p = f32_to_bf16(v); }

template <typename T, int NMAX>      // n <= NMAX: the row stays in registers
__global__ __launch_bounds__(256) void softmax_rows_fwd_kernel(const T *__restrict__ logits,
                                                               size_t rows, int n,
                                                               float *__restrict__ attn)
{
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const T *src = logits + r * (size_t)n;
    float *dst = attn + r * (size_t)n;
    float v[NMAX];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
        v[i] = i < n ? pw_ld<T>(src + i) : -INFINITY;
        m = fmaxf(m, v[i]);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
        v[i] = i < n ? __expf(v[i] - m) : 0.f;
        sum += v[i];
    }
    const float inv = 1.f / sum;
#pragma unroll
    for (int i = 0; i < NMAX; ++i)
        if (i < n) dst[i] = v[i] * inv;
}

template <typename T, int NMAX>
__global__ __launch_bounds__(256) void softmax_rows_bwd_kernel(const float *__restrict__ attn,
                                                               const float *__restrict__ grad_attn,
                                                               size_t rows, int n,
                                                               T *__restrict__ grad_logits)
{
    const size_t r = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows) return;
    const float *a = attn + r * (size_t)n, *g = grad_attn + r * (size_t)n;
    T *dst = grad_logits + r * (size_t)n;
    float av[NMAX], gv[NMAX];
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < NMAX; ++i) {
        av[i] = i < n ? a[i] : 0.f;
        gv[i] = i < n ? g[i] : 0.f;
        dot += av[i] * gv[i];
    }
#pragma unroll
    for (int i = 0; i < NMAX; ++i)
        if (i < n) pw_st<T>(dst + i, av[i] * (gv[i] - dot));
}

template <int G> __device__ __forceinline__ float group_max(float v)
{
#pragma unroll
    for (int o = 1; o < G; o <<= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
template <int G> __device__ __forceinline__ float group_add(float v)
{
#pragma unroll
    for (int o = 1; o < G; o <<= 1) v += __shfl_xor(v, o, 64);
    return v;
}

template <typename T, int G>         // n = 4 G values per row, G lanes per row
__global__ __launch_bounds__(256) void softmax_vec_fwd_kernel(const T *__restrict__ logits,
                                                              size_t total, float *__restrict__ attn)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const bool live = i < total;             // rows are whole multiples of the group: uniform per row
    float v[4];
    VecIO<T, 4>::ld(logits + (live ? i : 0), v);
    const float m = group_max<G>(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])));
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = __expf(v[k] - m);
    const float inv = 1.f / group_add<G>((v[0] + v[1]) + (v[2] + v[3]));
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] *= inv;
    if (live) VecIO<float, 4>::st(attn + i, v);
}

template <typename T, int G>
__global__ __launch_bounds__(256) void softmax_vec_bwd_kernel(const float *__restrict__ attn,
                                                              const float *__restrict__ grad_attn,
                                                              size_t total, T *__restrict__ grad_logits)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const bool live = i < total;
    float a[4], g[4];
    VecIO<float, 4>::ld(attn + (live ? i : 0), a);
    VecIO<float, 4>::ld(grad_attn + (live ? i : 0), g);
    const float dot = group_add<G>((a[0] * g[0] + a[1] * g[1]) + (a[2] * g[2] + a[3] * g[3]));
#pragma unroll
    for (int k = 0; k < 4; ++k) a[k] *= g[k] - dot;
    if (live) VecIO<T, 4>::st(grad_logits + i, a);
}

// value (rows, d) of type T -> bfloat16, rows with mask != 0 zeroed; 8 channels per thread
template <typename T>
__global__ __launch_bounds__(256) void value_mask_cast_kernel(const T *__restrict__ value,
                                                              const unsigned char *__restrict__ mask,
                                                              size_t rows, int d,
                                                              bf16_t *__restrict__ out)
{
    const size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    if (i >= rows * (size_t)d) return;
    const size_t r = i / (unsigned)d;
    const bool dead = mask && mask[r];
    float v[8];
    if constexpr (std::is_same<T, bf16_t>::value) {
        VecIO<bf16_t, 8>::ld(value + i, v);
    } else {
        float lo[4], hi[4];
        VecIO<float, 4>::ld(value + i, lo);
        VecIO<float, 4>::ld(value + i + 4, hi);
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = lo[k]; v[4 + k] = hi[k]; }
    }
    if (dead) {
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = 0.f;
    }
    VecIO<bf16_t, 8>::st(out + i, v);
}

}  // namespace boxattn
