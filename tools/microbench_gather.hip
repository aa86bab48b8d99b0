// Ceiling of the vector-L1 gather path for the operator's access pattern: random ROWS of 64 B
// (bf16, C = 32: 4 lanes x 16 B) or 128 B (fp32: 8 lanes x 16 B) from a window that fits the L1
// (100 % hits) or the L2, 16 independent loads in flight per lane, ~2 VALU instructions per load.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench_gather.hip -o /tmp/mbg && /tmp/mbg
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// LPR lanes per row (16 B each); window_rows must be a power of two
template <int LPR, int U>
__global__ __launch_bounds__(256) void k_rows(const char *buf, unsigned *out, unsigned window_rows, int iters)
{
    const unsigned lane_in_row = threadIdx.x % LPR;
    const unsigned grp = (blockIdx.x * blockDim.x + threadIdx.x) / LPR;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(buf), 0, window_rows * LPR * 16, 0x00020000);
    unsigned row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) row[u] = hash32(grp * 131u + u * 7919u);
    u32x4 acc = {0, 0, 0, 0};
    const unsigned mask = window_rows - 1;
    for (int i = 0; i < iters; ++i) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned off = ((row[u] + i * 2654435761u) & mask) * (LPR * 16) + lane_in_row * 16;
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

// The same gather with VALU work behind every load (W packed FMAs per 16 loaded bytes -- the
// forward kernel has ~12 VALU instructions per load) and a register budget like the real
// kernel's (launch bounds 256 -> up to 8 waves per SIMD; PAD registers cut that down).
template <int LPR, int U, int W>
__global__ __launch_bounds__(256) void k_rows_valu(const char *buf, float *out, unsigned window_rows, int iters, float wgt)
{
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const unsigned lane_in_row = threadIdx.x % LPR;
    const unsigned grp = (blockIdx.x * blockDim.x + threadIdx.x) / LPR;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(buf), 0, window_rows * LPR * 16, 0x00020000);
    unsigned row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) row[u] = hash32(grp * 131u + u * 7919u);
    f32x2 acc[4] = {{0, 0}, {0, 0}, {0, 0}, {0, 0}};
    const unsigned mask = window_rows - 1;
    for (int i = 0; i < iters; ++i) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned off = ((row[u] + i * 2654435761u) & mask) * (LPR * 16) + lane_in_row * 16;
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const f32x2 w2 = {wgt + u, wgt + u};
#pragma unroll
            for (int k = 0; k < W; ++k) {                      // bf16-style unpack + packed FMA
                const unsigned wd = v[u][k & 3] + k;
                const f32x2 x = {__uint_as_float(wd << 16), __uint_as_float(wd & 0xffff0000u)};
                acc[k & 3] = __builtin_elementwise_fma(w2, x, acc[k & 3]);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc[0].x + acc[0].y + acc[1].x + acc[1].y + acc[2].x + acc[2].y + acc[3].x + acc[3].y;
}

// The operator's real layout: value (pixel, head, C) -> row (pixel p, head h) at p*512 + h*64
// (bf16, H = 8, C = 32); group g of a wave reads head g % 8 of a random pixel of the window.
template <int U>
__global__ __launch_bounds__(256) void k_rows_heads(const char *buf, unsigned *out, unsigned window_pix, int iters)
{
    const unsigned lane_in_row = threadIdx.x % 4;
    const unsigned grp = (blockIdx.x * blockDim.x + threadIdx.x) / 4;
    const unsigned h = grp % 8;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<char *>(buf), 0, window_pix * 512, 0x00020000);
    unsigned row[U];
#pragma unroll
    for (int u = 0; u < U; ++u) row[u] = hash32(grp * 131u + u * 7919u);
    u32x4 acc = {0, 0, 0, 0};
    const unsigned mask = window_pix - 1;
    for (int i = 0; i < iters; ++i) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned off = ((row[u] + i * 2654435761u) & mask) * 512 + h * 64 + lane_in_row * 16;
            v[u] = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

template <typename F> float time_ms(F launch, int reps = 5)
{
    hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    launch(); (void)hipDeviceSynchronize();
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
        (void)hipEventRecord(a); launch(); (void)hipEventRecord(b); (void)hipEventSynchronize(b);
        float ms; (void)hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
    }
    return best;
}

int main()
{
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    printf("device: %s, CUs %d, clock %d MHz\n", p.name, p.multiProcessorCount, p.clockRate / 1000);
    const size_t bytes = 64u << 20;
    char *buf; unsigned *out;
    CHECK(hipMalloc(&buf, bytes)); CHECK(hipMemset(buf, 1, bytes));
    const int blocks = 8192, iters = 64;
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    const double cu_clk = (double)p.multiProcessorCount * p.clockRate * 1e3;   // CU-cycles per second
    for (unsigned window_rows : {128u, 2048u, 65536u, 524288u}) {
        {
            constexpr int LPR = 4, U = 16;
            float t = time_ms([&] { hipLaunchKernelGGL((k_rows<LPR, U>), dim3(blocks), dim3(256), 0, 0, buf, out, window_rows, iters); });
            const double rows = (double)blocks * 256 / LPR * U * iters;
            printf("64-B rows  (4 lanes x b128), window %7u rows (%6.0f KiB): %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU  %6.1f G rows/s\n",
                   window_rows, window_rows * 64 / 1024.0, t, rows * 64 / t / 1e9, rows * 64 / (t * 1e-3) / cu_clk, rows / t / 1e6);
        }
        {
            constexpr int LPR = 8, U = 16;
            float t = time_ms([&] { hipLaunchKernelGGL((k_rows<LPR, U>), dim3(blocks), dim3(256), 0, 0, buf, out, window_rows, iters); });
            const double rows = (double)blocks * 256 / LPR * U * iters;
            printf("128-B rows (8 lanes x b128), window %7u rows (%6.0f KiB): %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU  %6.1f G rows/s\n",
                   window_rows, window_rows * 128 / 1024.0, t, rows * 128 / t / 1e9, rows * 128 / (t * 1e-3) / cu_clk, rows / t / 1e6);
        }
    }
    // head-strided rows: (pixel, head) layout of the operator's value tensor
    for (unsigned window_pix : {16u, 64u, 256u, 4096u, 32768u}) {
        constexpr int U = 16;
        float t = time_ms([&] { hipLaunchKernelGGL((k_rows_heads<U>), dim3(blocks), dim3(256), 0, 0, buf, out, window_pix, iters); });
        const double rows = (double)blocks * 256 / 4 * U * iters;
        printf("64-B rows at pixel*512 + head*64, window %6u pixels (%6.0f KiB): %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU  %6.1f G rows/s\n",
               window_pix, window_pix * 512 / 1024.0, t, rows * 64 / t / 1e9, rows * 64 / (t * 1e-3) / cu_clk, rows / t / 1e6);
    }
    // gather + VALU: do the two overlap?  (64-B rows, L1-resident window)
    {
        constexpr int LPR = 4, U = 16;
        const double rows = (double)blocks * 256 / LPR * U * iters;
#define RUN_W(W)                                                                                  \
        {                                                                                         \
            float t = time_ms([&] { hipLaunchKernelGGL((k_rows_valu<LPR, U, W>), dim3(blocks), dim3(256), 0, 0, buf, (float *)out, 128u, iters, 0.5f); }); \
            printf("64-B rows + %2d x (2 unpack + 1 pk_fma) per load, window 8 KiB: %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU   (VALU alone would be %.3f ms)\n", \
                   W, t, rows * 64 / t / 1e9, rows * 64 / (t * 1e-3) / cu_clk,                   \
                   (double)blocks * 4 * U * iters * (3.0 * W + 3) * 4 / (p.multiProcessorCount * 4.0) / (p.clockRate * 1e3) * 1e3); \
        }
        RUN_W(0) RUN_W(1) RUN_W(2) RUN_W(4) RUN_W(8)
    }
    // contiguous (fully coalesced) 16 B per lane for comparison: every wave reads 1 KiB runs
    {
        constexpr int LPR = 64, U = 16;
        float t = time_ms([&] { hipLaunchKernelGGL((k_rows<LPR, U>), dim3(blocks), dim3(256), 0, 0, buf, out, 16u, iters); });
        const double rows = (double)blocks * 256 / LPR * U * iters;
        printf("1-KiB rows (64 lanes x b128, coalesced), window 16 KiB: %7.3f ms  %6.2f TB/s  %5.1f B/clk/CU\n",
               t, rows * 1024 / t / 1e9, rows * 1024 / (t * 1e-3) / cu_clk);
    }
    return 0;
}
