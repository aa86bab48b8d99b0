// Device-side helpers shared by the box-attention kernels (gfx950 / CDNA4, wave64).
//
// Semantics follow the reference sampling helpers
// (e2edet/module/ops/src/box_attn/box_attn_kernel.cuh:34-97 forward, :100-184 backward):
// pixel coordinate = loc * size - 0.5, window test (-1, size), four guarded corners with
// weights hh*hw, hh*lw, lh*hw, lh*lw.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace boxattn {

typedef uint16_t bf16_t;          // raw bfloat16 storage
constexpr int kWave = 64;         // CDNA wavefront
constexpr int kMaxLevels = 16;    // level table kept in LDS by the fast kernels

// ---------------------------------------------------------------------------------------
// storage <-> compute conversion
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float bf16_bits_to_f32(uint32_t bits16) {
    return __uint_as_float(bits16 << 16);
}
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {      // round-to-nearest-even
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40u);   // quiet NaN
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

template <typename ST> struct Storage;                         // storage type -> compute type
template <> struct Storage<float> {
    typedef float compute;
    static __device__ __forceinline__ float ld(const float *p) { return *p; }
    static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
};
template <> struct Storage<double> {
    typedef double compute;
    static __device__ __forceinline__ double ld(const double *p) { return *p; }
    static __device__ __forceinline__ void st(double *p, double v) { *p = v; }
};
template <> struct Storage<bf16_t> {
    typedef float compute;
    static __device__ __forceinline__ float ld(const bf16_t *p) { return bf16_bits_to_f32(*p); }
    static __device__ __forceinline__ void st(bf16_t *p, float v) { *p = f32_to_bf16(v); }
};

// Two fp32 -> packed bf16 pair, round-to-nearest-even, one v_cvt_pk_bf16_f32.
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}

// Vector access of VEC consecutive channels (VEC*sizeof(ST) is 8 or 16 bytes, aligned).
template <typename ST, int VEC> struct VecIO;
template <> struct VecIO<float, 4> {
    static __device__ __forceinline__ void ld(const float *p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void st(float *p, const float (&v)[4]) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct VecIO<bf16_t, 4> {
    static __device__ __forceinline__ void ld(const bf16_t *p, float (&v)[4]) {
        const uint2 t = *reinterpret_cast<const uint2 *>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
    }
    static __device__ __forceinline__ void st(bf16_t *p, const float (&v)[4]) {
        uint2 t;
        t.x = pack_bf16x2(v[0], v[1]);
        t.y = pack_bf16x2(v[2], v[3]);
        *reinterpret_cast<uint2 *>(p) = t;
    }
};
template <> struct VecIO<bf16_t, 8> {
    static __device__ __forceinline__ void ld(const bf16_t *p, float (&v)[8]) {
        const uint4 t = *reinterpret_cast<const uint4 *>(p);
        v[0] = __uint_as_float(t.x << 16); v[1] = __uint_as_float(t.x & 0xffff0000u);
        v[2] = __uint_as_float(t.y << 16); v[3] = __uint_as_float(t.y & 0xffff0000u);
        v[4] = __uint_as_float(t.z << 16); v[5] = __uint_as_float(t.z & 0xffff0000u);
        v[6] = __uint_as_float(t.w << 16); v[7] = __uint_as_float(t.w & 0xffff0000u);
    }
    static __device__ __forceinline__ void st(bf16_t *p, const float (&v)[8]) {
        uint4 t;
        t.x = pack_bf16x2(v[0], v[1]);
        t.y = pack_bf16x2(v[2], v[3]);
        t.z = pack_bf16x2(v[4], v[5]);
        t.w = pack_bf16x2(v[6], v[7]);
        *reinterpret_cast<uint4 *>(p) = t;
    }
};

// ---------------------------------------------------------------------------------------
// atomics: hardware global_atomic_add_f32 / _f64 (no CAS loop).  Memory from hipMalloc /
// the torch caching allocator is coarse-grained, where these are valid.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ void atomic_add(float *p, float v) { unsafeAtomicAdd(p, v); }
__device__ __forceinline__ void atomic_add(double *p, double v) { unsafeAtomicAdd(p, v); }

// ---------------------------------------------------------------------------------------
// one sample point
// ---------------------------------------------------------------------------------------
template <typename T> struct Sample {
    T lh, lw, hh, hw;
    int pix[4];      // pixel index (row*W+col) of the 4 corners inside the level, CLAMPED into
                     // the map so it is always safe to load from; use ok[] to discard
    bool ok[4];      // corner lies inside the map
    bool inside;     // passes the reference window test
    int y0, x0;      // unclamped top-left corner (floor of the pixel coordinate), >= -1
};

template <typename T>
__device__ __forceinline__ Sample<T> locate(T x, T y, int Hl, int Wl) {
    Sample<T> s;
    T h_im, w_im;
    {
        // two roundings (mul, then sub), as in the reference (`loc_h * spatial_h - 0.5`, where
        // the double constant keeps nvcc from fusing): no FMA contraction for the pixel
        // coordinate, so the bilinear cell is chosen exactly like the reference chooses it.
        // Everywhere else contraction stays on (the hot loops want v_fma / v_pk_fma).
#pragma clang fp contract(off)
        h_im = y * (T)Hl - (T)0.5;
        w_im = x * (T)Wl - (T)0.5;
    }
    s.inside = (h_im > (T)-1) && (w_im > (T)-1) && (h_im < (T)Hl) && (w_im < (T)Wl) &&
               Hl > 0 && Wl > 0;     // an empty level has nothing to read (keeps pix >= 0)
    // NaN / out-of-window points: keep the index arithmetic finite
    const T hs = s.inside ? h_im : (T)0;
    const T ws = s.inside ? w_im : (T)0;
    const T hf = floor(hs), wf = floor(ws);
    const int h_low = (int)hf, w_low = (int)wf;
    s.y0 = h_low;
    s.x0 = w_low;
    s.lh = hs - hf;
    s.lw = ws - wf;
    s.hh = (T)1 - s.lh;
    s.hw = (T)1 - s.lw;
    const bool h0 = h_low >= 0, h1 = h_low + 1 <= Hl - 1;
    const bool w0 = w_low >= 0, w1 = w_low + 1 <= Wl - 1;
    s.ok[0] = s.inside && h0 && w0;
    s.ok[1] = s.inside && h0 && w1;
    s.ok[2] = s.inside && h1 && w0;
    s.ok[3] = s.inside && h1 && w1;
    const int r0 = h0 ? h_low : 0, r1 = h1 ? h_low + 1 : Hl - 1;
    const int c0 = w0 ? w_low : 0, c1 = w1 ? w_low + 1 : Wl - 1;
    s.pix[0] = r0 * Wl + c0;
    s.pix[1] = r0 * Wl + c1;
    s.pix[2] = r1 * Wl + c0;
    s.pix[3] = r1 * Wl + c1;
    return s;
}

// ---------------------------------------------------------------------------------------
// cheap integer helpers (runtime divisors would otherwise cost ~30 VALU per division)
// ---------------------------------------------------------------------------------------
// n / d and n % d for 0 <= n < 2^24 with rcp = 1.0f / d: float estimate + one correction.
__device__ __forceinline__ void divmod_small(int n, int d, float rcp, int &q, int &r) {
    q = (int)((float)n * rcp);
    r = n - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; r -= d; }
}
// n / d and n % d for n < 2^31 with magic = floor(2^32 / d) from the host (0xFFFFFFFF for d = 1):
// the multiply-high estimate is the quotient or one less, fixed by one compare.
__device__ __forceinline__ void divmod_magic(unsigned n, unsigned d, unsigned magic, unsigned &q,
                                             unsigned &r) {
    q = __umulhi(n, magic);
    r = n - q * d;
    if (r >= d) { ++q; r -= d; }
}
// ---------------------------------------------------------------------------------------
// cross-lane sums
// ---------------------------------------------------------------------------------------
// Sum over aligned groups of G consecutive lanes, result in every lane of the group.
// G <= 16 stays inside one DPP row (16 lanes): quad_perm / row_half_mirror / row_ror adds,
// no LDS traffic.  Larger groups finish with ds_bpermute (__shfl_xor).
template <int G> __device__ __forceinline__ float group_sum(float v) {
    if constexpr (G >= 2)
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                 0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
    if constexpr (G >= 4)
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                 0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
    if constexpr (G >= 8)
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                 0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
    if constexpr (G >= 16)
        v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(
                 0, __builtin_bit_cast(int, v), 0x128, 0xF, 0xF, true));  // row_ror:8
    if constexpr (G >= 32) v += __shfl_xor(v, 16, kWave);
    if constexpr (G >= 64) v += __shfl_xor(v, 32, kWave);
    return v;
}

// Ordering point for LDS traffic inside ONE wavefront (single-wave workgroups): a wave's DS
// operations execute in issue order, so only the compiler has to be kept from reordering.
// Unlike __syncthreads() this does not drain vmcnt, i.e. global loads issued earlier stay in
// flight across it (__syncthreads() carries an s_waitcnt vmcnt(0) and would expose their
// full latency at every phase boundary).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o, kWave);
    return v;
}

}  // namespace boxattn
