// Generic kernels: any channel count C, any L / P / H, float / double / bf16 storage.
// They are the correctness backstop (and the fp64 gradcheck path); the hot BoxeR shapes
// (C a multiple of 4, H*C/4 lanes per query) take the kernels in boxattn_fast.h.
//
// Work mapping
//   forward : one thread per (b, q, head, channel), like the reference forward
//             (box_attn_kernel.cuh:274-349) -- channel-contiguous lanes give coalesced
//             corner reads.  Instance flavour loops points outermost so one thread owns
//             mask_out[b,q,p,head,c] and accumulates it in a register (the reference uses
//             a global atomicAdd there, instance_attn_kernel.cuh:354-355).
//   backward: one wavefront per (b, q, head); lanes stride over channels; per sample point
//             the per-channel partials of grad_loc / grad_weight are summed with a wave64
//             shuffle reduction (the reference: shared memory + serial thread-0 sum,
//             box_attn_kernel.cuh:443-463) and grad_value is scattered with hardware fp
//             atomics.
#pragma once
#include "boxattn_device.h"

namespace boxattn {

template <typename ST, bool INST>
__global__ __launch_bounds__(256) void fwd_generic_kernel(
    const ST *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const typename Storage<ST>::compute *__restrict__ loc,
    const typename Storage<ST>::compute *__restrict__ w_sp,
    const typename Storage<ST>::compute *__restrict__ w_lv, int S, int H, int C, int L, int Lq,
    int P, ST *__restrict__ out, ST *__restrict__ mask, size_t n)
{
    typedef typename Storage<ST>::compute T;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
        const int c = (int)(idx % C);
        const size_t qh = idx / C;
        const int m = (int)(qh % H);
        const size_t bq = qh / H;
        const size_t b = bq / Lq;
        const size_t HC = (size_t)H * C;
        const ST *vb = value + b * S * HC + (size_t)m * C + c;
        const size_t pt0 = qh * L * P;
        T acc = 0;
        if constexpr (!INST) {
            for (int l = 0; l < L; ++l) {
                const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
                const ST *vl = vb + (size_t)lsi[l] * HC;
                for (int p = 0; p < P; ++p) {
                    const size_t i = pt0 + (size_t)l * P + p;
                    const Sample<T> s = locate<T>(loc[2 * i], loc[2 * i + 1], Hl, Wl);
                    if (!s.inside) continue;
                    const T v1 = s.ok[0] ? Storage<ST>::ld(vl + (size_t)s.pix[0] * HC) : (T)0;
                    const T v2 = s.ok[1] ? Storage<ST>::ld(vl + (size_t)s.pix[1] * HC) : (T)0;
                    const T v3 = s.ok[2] ? Storage<ST>::ld(vl + (size_t)s.pix[2] * HC) : (T)0;
                    const T v4 = s.ok[3] ? Storage<ST>::ld(vl + (size_t)s.pix[3] * HC) : (T)0;
                    const T val = s.hh * s.hw * v1 + s.hh * s.lw * v2 + s.lh * s.hw * v3 +
                                  s.lh * s.lw * v4;
                    acc += val * w_sp[i];
                }
            }
        } else {
            ST *mk = mask + bq * P * HC + (size_t)m * C + c;
            for (int p = 0; p < P; ++p) {
                T macc = 0;
                for (int l = 0; l < L; ++l) {
                    const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
                    const ST *vl = vb + (size_t)lsi[l] * HC;
                    const size_t i = pt0 + (size_t)l * P + p;
                    const Sample<T> s = locate<T>(loc[2 * i], loc[2 * i + 1], Hl, Wl);
                    if (!s.inside) continue;
                    const T v1 = s.ok[0] ? Storage<ST>::ld(vl + (size_t)s.pix[0] * HC) : (T)0;
                    const T v2 = s.ok[1] ? Storage<ST>::ld(vl + (size_t)s.pix[1] * HC) : (T)0;
                    const T v3 = s.ok[2] ? Storage<ST>::ld(vl + (size_t)s.pix[2] * HC) : (T)0;
                    const T v4 = s.ok[3] ? Storage<ST>::ld(vl + (size_t)s.pix[3] * HC) : (T)0;
                    const T val = s.hh * s.hw * v1 + s.hh * s.lw * v2 + s.lh * s.hw * v3 +
                                  s.lh * s.lw * v4;
                    acc += val * w_sp[i];
                    macc += val * w_lv[i];
                }
                Storage<ST>::st(mk + (size_t)p * HC, macc);
            }
        }
        Storage<ST>::st(out + idx, acc);
    }
}

// GT = type grad_value is accumulated in (float for bf16 storage, else the compute type).
template <typename ST, bool INST>
__global__ __launch_bounds__(256) void bwd_generic_kernel(
    const ST *__restrict__ value, const int64_t *__restrict__ shapes,
    const int64_t *__restrict__ lsi, const typename Storage<ST>::compute *__restrict__ loc,
    const typename Storage<ST>::compute *__restrict__ w_sp,
    const typename Storage<ST>::compute *__restrict__ w_lv, const ST *__restrict__ grad_out,
    const ST *__restrict__ grad_mask, int S, int H, int C, int L, int Lq, int P,
    typename Storage<ST>::compute *__restrict__ grad_value,
    typename Storage<ST>::compute *__restrict__ grad_loc,
    typename Storage<ST>::compute *__restrict__ grad_sp,
    typename Storage<ST>::compute *__restrict__ grad_lv, size_t n_qh)
{
    typedef typename Storage<ST>::compute T;
    const int lane = threadIdx.x & (kWave - 1);
    const size_t wave = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) / kWave;
    const size_t n_waves = (size_t)gridDim.x * blockDim.x / kWave;
    const size_t HC = (size_t)H * C;
    for (size_t qh = wave; qh < n_qh; qh += n_waves) {
        const int m = (int)(qh % H);
        const size_t bq = qh / H;
        const size_t b = bq / Lq;
        const size_t voff = b * S * HC + (size_t)m * C;
        const ST *go = grad_out + qh * C;
        const size_t pt0 = qh * L * P;
        for (int l = 0; l < L; ++l) {
            const int Hl = (int)shapes[2 * l], Wl = (int)shapes[2 * l + 1];
            const size_t lo = voff + (size_t)lsi[l] * HC;
            for (int p = 0; p < P; ++p) {
                const size_t i = pt0 + (size_t)l * P + p;
                const Sample<T> s = locate<T>(loc[2 * i], loc[2 * i + 1], Hl, Wl);
                T g_sp = 0, g_lv = 0, g_x = 0, g_y = 0;
                if (s.inside) {                                  // wave-uniform
                    const T a_sp = w_sp[i];
                    const T a_lv = INST ? w_lv[i] : (T)0;
                    const ST *gm = INST ? grad_mask + (bq * P + p) * HC + (size_t)m * C : nullptr;
                    const T w1 = s.hh * s.hw, w2 = s.hh * s.lw, w3 = s.lh * s.hw,
                            w4 = s.lh * s.lw;
                    for (int c = lane; c < C; c += kWave) {
                        const T tg = Storage<ST>::ld(go + c);
                        const T tm = INST ? Storage<ST>::ld(gm + c) : (T)0;
                        const T t = INST ? tg * a_sp + tm * a_lv : tg * a_sp;
                        T gh = 0, gw = 0, v1 = 0, v2 = 0, v3 = 0, v4 = 0;
                        if (s.ok[0]) {
                            const size_t o = lo + (size_t)s.pix[0] * HC + c;
                            v1 = Storage<ST>::ld(value + o);
                            gh -= s.hw * v1; gw -= s.hh * v1;
                            atomic_add(grad_value + o, w1 * t);
                        }
                        if (s.ok[1]) {
                            const size_t o = lo + (size_t)s.pix[1] * HC + c;
                            v2 = Storage<ST>::ld(value + o);
                            gh -= s.lw * v2; gw += s.hh * v2;
                            atomic_add(grad_value + o, w2 * t);
                        }
                        if (s.ok[2]) {
                            const size_t o = lo + (size_t)s.pix[2] * HC + c;
                            v3 = Storage<ST>::ld(value + o);
                            gh += s.hw * v3; gw -= s.lh * v3;
                            atomic_add(grad_value + o, w3 * t);
                        }
                        if (s.ok[3]) {
                            const size_t o = lo + (size_t)s.pix[3] * HC + c;
                            v4 = Storage<ST>::ld(value + o);
                            gh += s.lw * v4; gw += s.lh * v4;
                            atomic_add(grad_value + o, w4 * t);
                        }
                        const T val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4;
                        g_sp += tg * val;
                        if (INST) g_lv += tm * val;
                        g_x += (T)Wl * gw * t;
                        g_y += (T)Hl * gh * t;
                    }
                    g_sp = wave_sum(g_sp);
                    g_x = wave_sum(g_x);
                    g_y = wave_sum(g_y);
                    if (INST) g_lv = wave_sum(g_lv);
                }
                if (lane == 0) {                 // also defines the outputs of skipped points
                    grad_sp[i] = g_sp;
                    grad_loc[2 * i] = g_x;
                    grad_loc[2 * i + 1] = g_y;
                    if (INST) grad_lv[i] = g_lv;
                }
            }
        }
    }
}

// Zero-fill (bytes a multiple of 4... any size: dword body, byte tail).  A kernel, not hipMemsetAsync:
// a memset node at the head of a captured HIP graph left the kernel nodes behind it reading stale
// copies of tensors written outside the graph between replays (ROCm 7.2, MI355X: found with the
// training forward's ticket clear, tests/test_gpu_parity.py::test_forward_backward_under_graph_capture).
__global__ __launch_bounds__(256) void zero_fill_kernel(unsigned char *__restrict__ p, size_t bytes, int streaming)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x, i0 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t head = (16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15;       // bytes up to 16-byte alignment
    const size_t h = head < bytes ? head : bytes, n16 = (bytes - h) / 16;
    for (size_t i = i0; i < h; i += stride) p[i] = 0;
    uint4 *q = reinterpret_cast<uint4 *>(p + h);
    if (streaming) {                       // big fills (a sparse map's grad_value): non-temporal, nobody reads the zeros soon
        typedef unsigned int zf_u32x4 __attribute__((ext_vector_type(4)));
        for (size_t i = i0; i < n16; i += stride)
            __builtin_nontemporal_store(zf_u32x4{0u, 0u, 0u, 0u}, reinterpret_cast<zf_u32x4 *>(q + i));
    } else {
        for (size_t i = i0; i < n16; i += stride) q[i] = make_uint4(0u, 0u, 0u, 0u);
    }
    for (size_t i = h + n16 * 16 + i0; i < bytes; i += stride) p[i] = 0;
}

// float32 accumulation buffer -> bf16 grad_value
__global__ __launch_bounds__(256) void cvt_f32_to_bf16_kernel(const float *__restrict__ src,
                                                              bf16_t *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    const size_t n4 = n / 4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const float4 v = reinterpret_cast<const float4 *>(src)[i];
        uint2 t;
        t.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
        t.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
        reinterpret_cast<uint2 *>(dst)[i] = t;
    }
    for (size_t i = n4 * 4 + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        dst[i] = f32_to_bf16(src[i]);
}

}  // namespace boxattn
