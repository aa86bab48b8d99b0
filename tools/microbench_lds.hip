// LDS primitive rates on MI355X with a lean loop (address math = one add + and per op).
// hipcc --offload-arch=gfx950 -O3 -o tools/microbench_lds.bin tools/microbench_lds.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int kTileBytes = 32768;       // 32 KB tile per block

// OP: 0 ds_add_f32, 1 ds_add_u32, 2 ds_add_u64, 3 ds_add_rtn_u32, 4 ds_read_b128, 5 ds_write_b128,
//     6 ds_read_b32+ds_write_b32 (non-atomic RMW), 7 ds_add_f64, 8 ds_read_b64, 9 ds_max_f32 (probe)
// PAT: 0 = every lane its own dword, consecutive lanes consecutive addresses (conflict-free)
//      1 = 8-lane groups on a random 128-B row (lane m -> row + 16m [+4j]) (our scatter pattern)
//      2 = 32-lane groups on a random 128-B row (lane c -> row + 4c)
template <int OP, int PAT>
__global__ __launch_bounds__(256) void k(float *out, int iters)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    unsigned char *smem = (unsigned char *)__builtin_assume_aligned(smem_raw, 16);
    for (int i = threadIdx.x * 4; i < kTileBytes; i += blockDim.x * 4) *(float *)(smem + i) = 0.f;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63;
    unsigned width = (OP == 2 || OP == 7 || OP == 8) ? 8 : (OP == 4 || OP == 5) ? 16 : 4;
    unsigned grp, off;
    if (PAT == 0) { grp = threadIdx.x; off = 0; }
    else if (PAT == 1) { grp = threadIdx.x >> 3; off = (lane & 7) * 16; }
    else { grp = threadIdx.x >> 5; off = (lane & 31) * 4; }
    unsigned state = grp * 2654435761u + blockIdx.x * 40503u + 1u;
    float facc = 0.f;
    unsigned uacc = 0;
    for (int i = 0; i < iters; ++i) {
        state = state * 1664525u + 1013904223u;
        unsigned a;
        if (PAT == 0) a = ((threadIdx.x + (state >> 20)) * width) & (kTileBytes - 1) & ~(width - 1);
        else a = (((state >> 12) * 128u) & (kTileBytes - 1)) + off;
        if (OP == 4 || OP == 5) a &= ~15u;
        unsigned char *p = smem + a;
        if (OP == 0) __hip_atomic_fetch_add((float *)p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (OP == 1) __hip_atomic_fetch_add((unsigned *)p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (OP == 2) __hip_atomic_fetch_add((unsigned long long *)(smem + (a & ~7u)), 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (OP == 3) uacc += __hip_atomic_fetch_add((unsigned *)p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (OP == 4) { float4 v = *(float4 *)p; facc += v.x + v.w; }
        if (OP == 5) { *(float4 *)p = make_float4(facc, 1.f, 2.f, (float)i); }
        if (OP == 6) { float v = *(float *)p; *(float *)p = v + 1.0f; }
        if (OP == 7) __hip_atomic_fetch_add((double *)(smem + (a & ~7u)), 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (OP == 8) { float2 v = *(float2 *)(smem + (a & ~7u)); facc += v.x + v.y; }
        if (OP == 9) __hip_atomic_fetch_max((float *)p, (float)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __syncthreads();
    out[blockIdx.x * blockDim.x + threadIdx.x] = facc + (float)uacc + *(float *)(smem + threadIdx.x * 4);
}

template <int OP, int PAT> void run(const char *label, float *out)
{
    const int blocks = 1024, iters = 4096;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<OP, PAT>), dim3(blocks), dim3(256), kTileBytes, 0, out, iters);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CHECK(hipEventRecord(a));
        hipLaunchKernelGGL((k<OP, PAT>), dim3(blocks), dim3(256), kTileBytes, 0, out, iters);
        CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    CHECK(hipGetLastError());
    const double lane_ops = (double)blocks * 256 * iters;
    // per CU: 4 blocks (1024 blocks / 256 CUs); clocks at 2.4 GHz
    const double clk_per_wave_op = best * 1e-3 * 2.4e9 / (lane_ops / 64 / 256);
    printf("%-46s %8.3f ms  %8.1f G lane-ops/s  %6.2f clk/wave-op/CU\n", label, best, lane_ops / best / 1e6, clk_per_wave_op);
}

int main()
{
    float *out; CHECK(hipMalloc(&out, 1024 * 256 * 4));
    run<0, 0>("ds_add_f32      linear", out);
    run<0, 1>("ds_add_f32      8-lane rows", out);
    run<0, 2>("ds_add_f32      32-lane rows", out);
    run<1, 0>("ds_add_u32      linear", out);
    run<1, 1>("ds_add_u32      8-lane rows", out);
    run<1, 2>("ds_add_u32      32-lane rows", out);
    run<3, 0>("ds_add_rtn_u32  linear", out);
    run<3, 2>("ds_add_rtn_u32  32-lane rows", out);
    run<2, 0>("ds_add_u64      linear", out);
    run<2, 2>("ds_add_u64      32-lane rows(8B)", out);
    run<7, 0>("ds_add_f64      linear", out);
    run<9, 0>("ds_max_f32      linear", out);
    run<6, 0>("read+write b32  linear (non-atomic RMW)", out);
    run<6, 2>("read+write b32  32-lane rows", out);
    run<4, 0>("ds_read_b128    linear", out);
    run<4, 1>("ds_read_b128    8-lane rows", out);
    run<8, 0>("ds_read_b64     linear", out);
    run<5, 0>("ds_write_b128   linear", out);
    run<5, 1>("ds_write_b128   8-lane rows", out);
    return 0;
}
