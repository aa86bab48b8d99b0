// Issue probe: do scalar and vector instructions of DIFFERENT waves on one SIMD issue side by side, or does a SIMD
// retire ~1 instruction per 4 cycles whatever the mix?  Three loop bodies (64 instructions each, unrolled, independent
// chains): V = v_fma_f32 only, S = s_add_u32 / s_mul_i32 only, M = the two alternating (32 + 32); at 1, 2, 4, 8 waves per
// SIMD (one workgroup of 256 x W threads per CU).  Prints cycles per instruction and SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/issue_probe tools/probes/issue_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ void probe(float *out, int iters, unsigned seed)
{
    float v0 = threadIdx.x, v1 = 1.f, v2 = 2.f, v3 = 3.f, v4 = 4.f, v5 = 5.f, v6 = 6.f, v7 = 7.f;
    unsigned s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3, s4 = seed + 4, s5 = seed + 5, s6 = seed + 6, s7 = seed + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (MODE == 0) {
                asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3\n"
                             "v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7\n"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+v"(v4), "+v"(v5), "+v"(v6), "+v"(v7));
            } else if (MODE == 3) {         // ONE dependent chain
                asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n"
                             "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %0, %0, %0, %0\n"
                             : "+v"(v0));
            } else if (MODE == 4) {         // TWO dependent chains, interleaved
                asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n"
                             "v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n"
                             : "+v"(v0), "+v"(v1));
            } else if (MODE == 1) {
                asm volatile("s_add_u32 %0, %0, %0\n s_add_u32 %1, %1, %1\n s_add_u32 %2, %2, %2\n s_add_u32 %3, %3, %3\n"
                             "s_add_u32 %4, %4, %4\n s_add_u32 %5, %5, %5\n s_add_u32 %6, %6, %6\n s_add_u32 %7, %7, %7\n"
                             : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : : "scc");
            } else {
                asm volatile("v_fma_f32 %0, %0, %0, %0\n s_add_u32 %4, %4, %4\n v_fma_f32 %1, %1, %1, %1\n s_add_u32 %5, %5, %5\n"
                             "v_fma_f32 %2, %2, %2, %2\n s_add_u32 %6, %6, %6\n v_fma_f32 %3, %3, %3, %3\n s_add_u32 %7, %7, %7\n"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3), "+s"(s4), "+s"(s5), "+s"(s6), "+s"(s7) : : "scc");
            }
        }
    }
    if (v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7 == 1.2345f || (s0 ^ s1 ^ s2 ^ s3 ^ s4 ^ s5 ^ s6 ^ s7) == 0x12345u) out[0] = v0;
}

int main()
{
    float *out; hipMalloc(&out, 64);
    int cus = 256, clk_khz = 2400000;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("%d CUs, %d kHz\n", cus, clk_khz);
    const int iters = 20000;
    const char *names[5] = {"V (v_fma, 8 chains)", "S (s_add only)", "M (alternating)", "D (v_fma, ONE chain)", "D2 (v_fma, 2 chains)"};
    for (int mode = 0; mode < 5; ++mode)
        for (int w : {1, 2, 4, 8}) {
            // w waves per SIMD: a workgroup of 256 threads = one wave per SIMD; w workgroups per CU
            const int blocks = cus * w;
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            auto launch = [&]() {
                if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
                else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
                else if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
                else if (mode == 3) hipLaunchKernelGGL(probe<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
                else hipLaunchKernelGGL(probe<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 1u);
            };
            launch();
            hipEventRecord(a); launch(); hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            const double instr_per_simd = (double)iters * 64 * w;
            printf("%-22s %d waves/SIMD: %8.1f us  %.2f cycles per instruction and SIMD (at the nominal clock)\n", names[mode], w,
                   ms * 1e3, ms * 1e-3 * clk_khz * 1e3 / instr_per_simd);
        }
    return 0;
}
