// Layout checks for the two instructions the window-staged forward is built on (run on the GPU box:
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma4 tools/microbench_mfma4.hip && /tmp/mfma4):
//  1. v_mfma_f32_4x4x4_16B_bf16: 16 independent 4x4x4 products, block = lane / 4.  Hypothesis: lane l holds
//     row i = l % 4 of A (4 consecutive k), column j = l % 4 of B (4 consecutive k) and column j = l % 4 of D
//     (rows i = 0..3 in its 4 result registers).
//  2. ds_read_b64_tr_b16: inside a 16-lane group, element j of lane i comes from the address supplied by lane
//     4 j + (i >> 2), sub-element i & 3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef short i16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
static __device__ __host__ unsigned short f2bf(float f) { union { float f; unsigned u; } x; x.f = f; return (unsigned short)(x.u >> 16); }
static float bf2f(unsigned short b) { union { float f; unsigned u; } x; x.u = (unsigned)b << 16; return x.f; }

__global__ void k_mfma(const unsigned short *a, const unsigned short *b, float *d)
{
    const int l = threadIdx.x;
    i16x4 av, bv;
    for (int k = 0; k < 4; ++k) { av[k] = (short)a[l * 4 + k]; bv[k] = (short)b[l * 4 + k]; }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(av, bv, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) d[l * 4 + i] = c[i];
}
__global__ void k_tr(unsigned *out)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
    const int l = threadIdx.x;
    for (int i = l; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    // lane l supplies the address of 4 consecutive elements starting at element 64 * l + 4 (its own row)
    typedef __attribute__((address_space(3))) i16x4 lds_vec;
    const i16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_vec *)(lds + 64 * l + 4));
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main()
{
    unsigned short ha[256], hb[256]; float hd[256];
    for (int i = 0; i < 256; ++i) { ha[i] = f2bf((float)((i * 7) % 13 - 6)); hb[i] = f2bf((float)((i * 5) % 11 - 5)); }
    unsigned short *da, *db; float *dd; unsigned *dt;
    hipMalloc(&da, 512); hipMalloc(&db, 512); hipMalloc(&dd, 1024); hipMalloc(&dt, 1024);
    hipMemcpy(da, ha, 512, hipMemcpyHostToDevice); hipMemcpy(db, hb, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, da, db, dd);
    hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i) {
            const int blk = l / 4, j = l % 4;
            float want = 0.f;
            for (int k = 0; k < 4; ++k) want += bf2f(ha[(blk * 4 + i) * 4 + k]) * bf2f(hb[(blk * 4 + j) * 4 + k]);
            if (std::fabs(want - hd[l * 4 + i]) > 1e-3f) { if (bad < 8) printf("mfma lane %d reg %d: got %g want %g\n", l, i, hd[l * 4 + i], want); ++bad; }
        }
    printf("mfma 4x4x4 layout hypothesis: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    unsigned ht[256];
    hipLaunchKernelGGL(k_tr, dim3(1), dim3(64), 0, 0, dt);
    hipMemcpy(ht, dt, 1024, hipMemcpyDeviceToHost);
    bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) {
            const int g = l / 16, i = l % 16, src = 16 * g + 4 * j + (i >> 2);
            const unsigned want = (unsigned)(64 * src + 4 + (i & 3));
            if (ht[l * 4 + j] != want) { if (bad < 8) printf("tr lane %d elem %d: got %u want %u\n", l, j, ht[l * 4 + j], want); ++bad; }
        }
    printf("ds_read_b64_tr_b16 hypothesis: %s (%d mismatches)\n", bad ? "WRONG" : "ok", bad);
    return 0;
}
