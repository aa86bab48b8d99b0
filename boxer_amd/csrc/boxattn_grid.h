// Reference windows + predicted offsets -> sampling grid, and its backward: everything of the
// modules' `_where_to_attend` after the box-offset projection (reference
// e2edet/module/box_attention.py:63-81, 196-214, 304-338) as one kernel each way instead of
// ~15 elementwise passes (4 of them over the (B,Lq,H,L,P,2) grid).
//
//   ref     (B, Lq, D) or (B, Lq, H, D)   (cx, cy, w, h[, angle, ...]), D >= 4
//   offsets (B, Lq, H, L, V)              V = 4, or 5 with a learned rotation
//   kidx    (P, 2)                        the module's `kernel_indices` buffer
//   vr      (B, L, 2) or null             `v_valid_ratios` (B,1,1,L,1,2)
//   box     = ref[:4] + offsets[:4] / 8 * (w, h, w, h)_ref
//   theta   = none (angle_mode 0) | (ref[4] + offsets[4] / 16) * 2 pi (1) | ref[4] (2)
//   grid[n, p] = (c + R(theta) (kidx_p * relu(size))) * vr        n = (b, q, h, l)
//
// Same operation order as the torch code (no FMA contraction), so without rotation the grid
// is bit-identical; with rotation the device sin / cos differ from torch's by an ulp.
#pragma once
#include "boxattn_device.h"

namespace boxattn {

struct GridDims {
    int Lq, H, L, P, V;
    int ref_dim, ref_per_head, angle_mode;
};

struct GridBox { float cx, cy, w, h, sn, cs, vx, vy, rw, rh; };

__device__ __forceinline__ GridBox grid_box(const float *__restrict__ ref,
                                            const float *__restrict__ offsets,
                                            const float *__restrict__ vr, const GridDims &d,
                                            size_t n)
{
#pragma clang fp contract(off)
    const int l = (int)(n % (unsigned)d.L);
    const size_t bqh = n / (unsigned)d.L;
    const int h = (int)(bqh % (unsigned)d.H);
    const size_t bq = bqh / (unsigned)d.H;
    const size_t b = bq / (unsigned)d.Lq;
    const float *r = ref + (d.ref_per_head ? bqh : bq) * (unsigned)d.ref_dim;
    const float *o = offsets + n * (unsigned)d.V;
    GridBox g;
    g.rw = r[2];
    g.rh = r[3];
    g.cx = r[0] + o[0] / 8.f * g.rw;
    g.cy = r[1] + o[1] / 8.f * g.rh;
    g.w = g.rw + o[2] / 8.f * g.rw;
    g.h = g.rh + o[3] / 8.f * g.rh;
    g.sn = 0.f;
    g.cs = 1.f;
    if (d.angle_mode == 1) sincosf((r[4] + o[4] / 16.f) * 2.f * 3.14159265358979323846f, &g.sn, &g.cs);
    else if (d.angle_mode == 2) sincosf(r[4], &g.sn, &g.cs);
    g.vx = vr ? vr[(b * d.L + l) * 2] : 1.f;
    g.vy = vr ? vr[(b * d.L + l) * 2 + 1] : 1.f;
    (void)h;
    return g;
}

// grid point p of a decoded box (the body of grid_fwd_kernel, shared with the sampling kernels
// that take boxes instead of a grid: same operations, same order, no FMA contraction)
__device__ __forceinline__ float2 grid_point(const GridBox &g, const float *__restrict__ kidx,
                                             int p, bool has_vr, int angle_mode)
{
#pragma clang fp contract(off)
    const float lx = kidx[2 * p] * fmaxf(g.w, 0.f), ly = kidx[2 * p + 1] * fmaxf(g.h, 0.f);
    float gx, gy;
    if (angle_mode) {
        gx = g.cx + (lx * g.cs - ly * g.sn);
        gy = g.cy + (lx * g.sn + ly * g.cs);
    } else {
        gx = g.cx + lx;
        gy = g.cy + ly;
    }
    if (has_vr) {
        gx = gx * g.vx;
        gy = gy * g.vy;
    }
    return make_float2(gx, gy);
}

// one point's share of the box gradient (the loop body of grid_bwd_kernel)
struct GridGrad { float cx, cy, w, h, t; };
__device__ __forceinline__ void grid_grad_add(GridGrad &a, const GridBox &g, const float *__restrict__ kidx,
                                              int p, float2 q)
{
    const float wr = fmaxf(g.w, 0.f), hr = fmaxf(g.h, 0.f);
    const float gx = q.x * g.vx, gy = q.y * g.vy;           // d / d (unscaled grid)
    const float kx = kidx[2 * p], ky = kidx[2 * p + 1];
    a.cx += gx;
    a.cy += gy;
    a.w += kx * (gx * g.cs + gy * g.sn);                     // d / d relu(w)
    a.h += ky * (gy * g.cs - gx * g.sn);                     // d / d relu(h)
    const float lx = kx * wr, ly = ky * hr;
    a.t += gx * (-lx * g.sn - ly * g.cs) + gy * (lx * g.cs - ly * g.sn);
}
// ... and the row's results from the summed shares (the tail of grid_bwd_kernel)
__device__ __forceinline__ void grid_grad_store(GridGrad a, const GridBox &g, const float *__restrict__ o,
                                                const GridDims &d, float *__restrict__ go,
                                                float *__restrict__ gr)
{
    a.w = g.w > 0.f ? a.w : 0.f;                             // relu'
    a.h = g.h > 0.f ? a.h : 0.f;
    go[0] = a.cx * g.rw / 8.f;
    go[1] = a.cy * g.rh / 8.f;
    go[2] = a.w * g.rw / 8.f;
    go[3] = a.h * g.rh / 8.f;
    const float two_pi = 2.f * 3.14159265358979323846f;
    if (d.V == 5) go[4] = d.angle_mode == 1 ? a.t * two_pi / 16.f : 0.f;
    if (gr) {
        gr[0] = a.cx;
        gr[1] = a.cy;
        gr[2] = a.cx * o[0] / 8.f + a.w * (1.f + o[2] / 8.f);
        gr[3] = a.cy * o[1] / 8.f + a.h * (1.f + o[3] / 8.f);
        gr[4] = d.angle_mode == 1 ? a.t * two_pi : (d.angle_mode == 2 ? a.t : 0.f);
    }
}

}  // namespace boxattn
