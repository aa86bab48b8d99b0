/*
 * boxattn_oracle.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Plain-C CPU restatement of the box-attention / instance-attention operator of
 * kienduynguyen/BoxeR.  It is the checker for the HIP path: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * path (boxer_amd/) never links, imports or falls back to anything in oracle/.
 *
 * What it follows (reference file:line, relative to /root/reference):
 *   forward, box      e2edet/module/ops/src/box_attn/box_attn_kernel.cuh:34-97, 274-349
 *   backward, box     e2edet/module/ops/src/box_attn/box_attn_kernel.cuh:100-184, 352-472
 *   forward, instance e2edet/module/ops/src/instance_attn/instance_attn_kernel.cuh:282-364
 *   backward, inst.   e2edet/module/ops/src/instance_attn/instance_attn_kernel.cuh:98-187, 367-505
 * The reference has no CPU implementation of the op (box_attn.h:53 raises), and its CUDA
 * sources cannot be built in this image (no nvcc; THC/THCAtomics.cuh is gone from torch
 * 2.10), so there is no oracle/_ref build.  Parity is pinned instead against fp64 golden
 * vectors produced by the reference's own test oracle (tests/box_attn_test.py:9-42,
 * tests/instance_attn_test.py:11-63) -- see tests/golden/make_goldens.py.
 *
 * Layouts (all row-major, contiguous):
 *   value   (B, S, H, C)        S = sum_l H_l*W_l, level l starts at row lsi[l]
 *   shapes  (L, 2) int64        (H_l, W_l)
 *   lsi     (L,)   int64
 *   loc     (B, Lq, H, L, P, 2) normalised [0,1]; [...,0] = x (width), [...,1] = y (height)
 *   attn    (B, Lq, H, L, P)
 *   out     (B, Lq, H, C)
 *   mask    (B, Lq, P, H, C)    instance only
 *
 * Every entry point exists for double (_f64) and float (_f32).  The float flavour performs
 * the arithmetic in float in the same order as the reference kernels.  Backward outputs
 * must be zero-filled by the caller (the reference contract: at::zeros / zeros_like,
 * box_attn.cu:44,105-107); grad_value is accumulated into.
 *
 * Parallelism: OpenMP over (b, h) -- distinct heads never touch the same grad_value
 * element, so no atomics are needed and results are deterministic.
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

#ifdef _OPENMP
#include <omp.h>
#endif

int boxattn_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

void boxattn_oracle_set_num_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

#define DEFINE_ORACLE(T, SUF, FLOORF)                                                        \
                                                                                             \
/* One sample point.  Mirrors *_im2col_bilinear / *_col2im_bilinear: four guarded corner  */ \
/* reads, weights hh*hw, hh*lw, lh*hw, lh*lw.                                             */ \
typedef struct {                                                                             \
    int inside;             /* passes the (-1, H) x (-1, W) window test                   */ \
    int ok[4];              /* corner k lies inside the feature map                       */ \
    ptrdiff_t off[4];       /* element offset of corner k (channel 0 of head m)           */ \
    T w[4];                 /* bilinear weight of corner k                                */ \
    T lh, lw, hh, hw;                                                                        \
} pt_##SUF;                                                                                  \
                                                                                             \
static pt_##SUF locate_##SUF(T loc_x, T loc_y, int Hl, int Wl, int H, int C, int m)          \
{                                                                                            \
    pt_##SUF p;                                                                              \
    const T h_im = loc_y * (T)Hl - (T)0.5;                                                   \
    const T w_im = loc_x * (T)Wl - (T)0.5;                                                   \
    p.inside = (h_im > (T)-1 && w_im > (T)-1 && h_im < (T)Hl && w_im < (T)Wl);               \
    if (!p.inside) return p;                                                                 \
    const int h_low = (int)FLOORF(h_im);                                                     \
    const int w_low = (int)FLOORF(w_im);                                                     \
    const int h_high = h_low + 1, w_high = w_low + 1;                                        \
    p.lh = h_im - (T)h_low;                                                                  \
    p.lw = w_im - (T)w_low;                                                                  \
    p.hh = (T)1 - p.lh;                                                                      \
    p.hw = (T)1 - p.lw;                                                                      \
    const ptrdiff_t ws = (ptrdiff_t)H * C, hs = ws * Wl, base = (ptrdiff_t)m * C;            \
    p.ok[0] = (h_low >= 0 && w_low >= 0);                                                    \
    p.ok[1] = (h_low >= 0 && w_high <= Wl - 1);                                              \
    p.ok[2] = (h_high <= Hl - 1 && w_low >= 0);                                              \
    p.ok[3] = (h_high <= Hl - 1 && w_high <= Wl - 1);                                        \
    p.off[0] = h_low * hs + w_low * ws + base;                                               \
    p.off[1] = h_low * hs + w_high * ws + base;                                              \
    p.off[2] = h_high * hs + w_low * ws + base;                                              \
    p.off[3] = h_high * hs + w_high * ws + base;                                             \
    p.w[0] = p.hh * p.hw;                                                                    \
    p.w[1] = p.hh * p.lw;                                                                    \
    p.w[2] = p.lh * p.hw;                                                                    \
    p.w[3] = p.lh * p.lw;                                                                    \
    return p;                                                                                \
}                                                                                            \
                                                                                             \
int boxattn_oracle_fwd_##SUF(const T *value, const int64_t *shapes, const int64_t *lsi,      \
                             const T *loc, const T *attn, int B, int S, int H, int C,        \
                             int L, int Lq, int P, T *out)                                   \
{                                                                                            \
    _Pragma("omp parallel for collapse(2) schedule(static)")                                 \
    for (int b = 0; b < B; ++b)                                                              \
        for (int m = 0; m < H; ++m)                                                          \
            for (int q = 0; q < Lq; ++q) {                                                   \
                const size_t qh = ((size_t)b * Lq + q) * H + m;                              \
                T *o = out + qh * C;                                                         \
                for (int c = 0; c < C; ++c) o[c] = 0;                                        \
                for (int l = 0; l < L; ++l) {                                                \
                    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];          \
                    const T *v = value + ((size_t)b * S + (size_t)lsi[l]) * H * C;           \
                    for (int p = 0; p < P; ++p) {                                            \
                        const size_t i = (qh * L + l) * P + p;                               \
                        const pt_##SUF s = locate_##SUF(loc[2 * i], loc[2 * i + 1], Hl, Wl,  \
                                                        H, C, m);                            \
                        if (!s.inside) continue;                                             \
                        const T a = attn[i];                                                 \
                        for (int c = 0; c < C; ++c) {                                        \
                            const T v1 = s.ok[0] ? v[s.off[0] + c] : (T)0;                   \
                            const T v2 = s.ok[1] ? v[s.off[1] + c] : (T)0;                   \
                            const T v3 = s.ok[2] ? v[s.off[2] + c] : (T)0;                   \
                            const T v4 = s.ok[3] ? v[s.off[3] + c] : (T)0;                   \
                            const T val = s.w[0] * v1 + s.w[1] * v2 + s.w[2] * v3 +          \
                                          s.w[3] * v4;                                       \
                            o[c] += val * a;                                                 \
                        }                                                                    \
                    }                                                                        \
                }                                                                            \
            }                                                                                \
    return 0;                                                                                \
}                                                                                            \
                                                                                             \
int boxattn_oracle_bwd_##SUF(const T *value, const int64_t *shapes, const int64_t *lsi,      \
                             const T *loc, const T *attn, const T *grad_out, int B, int S,   \
                             int H, int C, int L, int Lq, int P, T *grad_value,              \
                             T *grad_loc, T *grad_attn)                                      \
{                                                                                            \
    _Pragma("omp parallel for collapse(2) schedule(static)")                                 \
    for (int b = 0; b < B; ++b)                                                              \
        for (int m = 0; m < H; ++m)                                                          \
            for (int q = 0; q < Lq; ++q) {                                                   \
                const size_t qh = ((size_t)b * Lq + q) * H + m;                              \
                const T *g = grad_out + qh * C;                                              \
                for (int l = 0; l < L; ++l) {                                                \
                    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];          \
                    const size_t vo = ((size_t)b * S + (size_t)lsi[l]) * H * C;              \
                    const T *v = value + vo;                                                 \
                    T *gv = grad_value + vo;                                                 \
                    for (int p = 0; p < P; ++p) {                                            \
                        const size_t i = (qh * L + l) * P + p;                               \
                        const pt_##SUF s = locate_##SUF(loc[2 * i], loc[2 * i + 1], Hl, Wl,  \
                                                        H, C, m);                            \
                        if (!s.inside) continue;                                             \
                        const T a = attn[i];                                                 \
                        T ga = 0, gx = 0, gy = 0;                                            \
                        for (int c = 0; c < C; ++c) {                                        \
                            const T t = g[c] * a;                                            \
                            T gh = 0, gw = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;                \
                            if (s.ok[0]) {                                                   \
                                v1 = v[s.off[0] + c];                                        \
                                gh -= s.hw * v1; gw -= s.hh * v1;                            \
                                gv[s.off[0] + c] += s.w[0] * t;                              \
                            }                                                                \
                            if (s.ok[1]) {                                                   \
                                v2 = v[s.off[1] + c];                                        \
                                gh -= s.lw * v2; gw += s.hh * v2;                            \
                                gv[s.off[1] + c] += s.w[1] * t;                              \
                            }                                                                \
                            if (s.ok[2]) {                                                   \
                                v3 = v[s.off[2] + c];                                        \
                                gh += s.hw * v3; gw -= s.lh * v3;                            \
                                gv[s.off[2] + c] += s.w[2] * t;                              \
                            }                                                                \
                            if (s.ok[3]) {                                                   \
                                v4 = v[s.off[3] + c];                                        \
                                gh += s.lw * v4; gw += s.lh * v4;                            \
                                gv[s.off[3] + c] += s.w[3] * t;                              \
                            }                                                                \
                            const T val = s.w[0] * v1 + s.w[1] * v2 + s.w[2] * v3 +          \
                                          s.w[3] * v4;                                       \
                            ga += g[c] * val;                                                \
                            gx += (T)Wl * gw * t;                                            \
                            gy += (T)Hl * gh * t;                                            \
                        }                                                                    \
                        grad_attn[i] = ga;                                                   \
                        grad_loc[2 * i] = gx;                                                \
                        grad_loc[2 * i + 1] = gy;                                            \
                    }                                                                        \
                }                                                                            \
            }                                                                                \
    return 0;                                                                                \
}                                                                                            \
                                                                                             \
int instattn_oracle_fwd_##SUF(const T *value, const int64_t *shapes, const int64_t *lsi,     \
                              const T *loc, const T *spatial_w, const T *level_w, int B,     \
                              int S, int H, int C, int L, int Lq, int P, T *out, T *mask)    \
{                                                                                            \
    _Pragma("omp parallel for collapse(2) schedule(static)")                                 \
    for (int b = 0; b < B; ++b)                                                              \
        for (int m = 0; m < H; ++m)                                                          \
            for (int q = 0; q < Lq; ++q) {                                                   \
                const size_t qh = ((size_t)b * Lq + q) * H + m;                              \
                T *o = out + qh * C;                                                         \
                T *mk = mask + ((size_t)b * Lq + q) * P * H * C + (size_t)m * C;             \
                for (int c = 0; c < C; ++c) o[c] = 0;                                        \
                for (int p = 0; p < P; ++p)                                                  \
                    for (int c = 0; c < C; ++c) mk[(size_t)p * H * C + c] = 0;               \
                for (int l = 0; l < L; ++l) {                                                \
                    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];          \
                    const T *v = value + ((size_t)b * S + (size_t)lsi[l]) * H * C;           \
                    for (int p = 0; p < P; ++p) {                                            \
                        const size_t i = (qh * L + l) * P + p;                               \
                        const pt_##SUF s = locate_##SUF(loc[2 * i], loc[2 * i + 1], Hl, Wl,  \
                                                        H, C, m);                            \
                        if (!s.inside) continue;                                             \
                        const T as = spatial_w[i], al = level_w[i];                          \
                        for (int c = 0; c < C; ++c) {                                        \
                            const T v1 = s.ok[0] ? v[s.off[0] + c] : (T)0;                   \
                            const T v2 = s.ok[1] ? v[s.off[1] + c] : (T)0;                   \
                            const T v3 = s.ok[2] ? v[s.off[2] + c] : (T)0;                   \
                            const T v4 = s.ok[3] ? v[s.off[3] + c] : (T)0;                   \
                            const T val = s.w[0] * v1 + s.w[1] * v2 + s.w[2] * v3 +          \
                                          s.w[3] * v4;                                       \
                            o[c] += val * as;                                                \
                            mk[(size_t)p * H * C + c] += val * al;                           \
                        }                                                                    \
                    }                                                                        \
                }                                                                            \
            }                                                                                \
    return 0;                                                                                \
}                                                                                            \
                                                                                             \
int instattn_oracle_bwd_##SUF(const T *value, const int64_t *shapes, const int64_t *lsi,     \
                              const T *loc, const T *spatial_w, const T *level_w,            \
                              const T *grad_out, const T *grad_mask, int B, int S, int H,    \
                              int C, int L, int Lq, int P, T *grad_value, T *grad_loc,       \
                              T *grad_spatial, T *grad_level)                                \
{                                                                                            \
    _Pragma("omp parallel for collapse(2) schedule(static)")                                 \
    for (int b = 0; b < B; ++b)                                                              \
        for (int m = 0; m < H; ++m)                                                          \
            for (int q = 0; q < Lq; ++q) {                                                   \
                const size_t qh = ((size_t)b * Lq + q) * H + m;                              \
                const T *g = grad_out + qh * C;                                              \
                const T *gm = grad_mask + ((size_t)b * Lq + q) * P * H * C + (size_t)m * C;  \
                for (int l = 0; l < L; ++l) {                                                \
                    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];          \
                    const size_t vo = ((size_t)b * S + (size_t)lsi[l]) * H * C;              \
                    const T *v = value + vo;                                                 \
                    T *gv = grad_value + vo;                                                 \
                    for (int p = 0; p < P; ++p) {                                            \
                        const size_t i = (qh * L + l) * P + p;                               \
                        const pt_##SUF s = locate_##SUF(loc[2 * i], loc[2 * i + 1], Hl, Wl,  \
                                                        H, C, m);                            \
                        if (!s.inside) continue;                                             \
                        const T as = spatial_w[i], al = level_w[i];                          \
                        T gs = 0, gl = 0, gx = 0, gy = 0;                                    \
                        for (int c = 0; c < C; ++c) {                                        \
                            const T tg = g[c], tm = gm[(size_t)p * H * C + c];               \
                            const T t = tg * as + tm * al;                                   \
                            T gh = 0, gw = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;                \
                            if (s.ok[0]) {                                                   \
                                v1 = v[s.off[0] + c];                                        \
                                gh -= s.hw * v1; gw -= s.hh * v1;                            \
                                gv[s.off[0] + c] += s.w[0] * t;                              \
                            }                                                                \
                            if (s.ok[1]) {                                                   \
                                v2 = v[s.off[1] + c];                                        \
                                gh -= s.lw * v2; gw += s.hh * v2;                            \
                                gv[s.off[1] + c] += s.w[1] * t;                              \
                            }                                                                \
                            if (s.ok[2]) {                                                   \
                                v3 = v[s.off[2] + c];                                        \
                                gh += s.hw * v3; gw -= s.lh * v3;                            \
                                gv[s.off[2] + c] += s.w[2] * t;                              \
                            }                                                                \
                            if (s.ok[3]) {                                                   \
                                v4 = v[s.off[3] + c];                                        \
                                gh += s.lw * v4; gw += s.lh * v4;                            \
                                gv[s.off[3] + c] += s.w[3] * t;                              \
                            }                                                                \
                            const T val = s.w[0] * v1 + s.w[1] * v2 + s.w[2] * v3 +          \
                                          s.w[3] * v4;                                       \
                            gs += tg * val;                                                  \
                            gl += tm * val;                                                  \
                            gx += (T)Wl * gw * t;                                            \
                            gy += (T)Hl * gh * t;                                            \
                        }                                                                    \
                        grad_spatial[i] = gs;                                                \
                        grad_level[i] = gl;                                                  \
                        grad_loc[2 * i] = gx;                                                \
                        grad_loc[2 * i + 1] = gy;                                            \
                    }                                                                        \
                }                                                                            \
            }                                                                                \
    return 0;                                                                                \
}

DEFINE_ORACLE(double, f64, floor)
DEFINE_ORACLE(float, f32, floorf)
