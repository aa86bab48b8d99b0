// Micro-benchmarks that price the building blocks of the box-attention backward on MI355X:
//   1. global fp32 atomics: agent scope vs workgroup scope (L2-resident), random 128-B rows
//      vs XCD-partitioned rows;
//   2. LDS ds_add_f32 and ds_read_b128 under the access patterns of an LDS-tiled scatter;
//   3. 128-byte row gathers from an L2/MALL-resident buffer (the forward's access pattern).
// Build & run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/microbench tools/microbench.hip && /tmp/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 0xf; }

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}

// ---- 1. global atomics: each half-wave adds to one random 128-B row (32 floats) ---------
// MODE 0: agent scope (default atomicAdd semantics, hardware fadd)
// MODE 1: workgroup scope (stays in this XCD's L2) -- rows restricted to this XCD's partition
// MODE 2: agent scope, rows restricted to this XCD's partition (isolates scope vs locality)
template <int MODE>
__global__ __launch_bounds__(256) void k_global_atomics(float *buf, unsigned n_rows, int iters)
{
    const unsigned lane32 = threadIdx.x & 31;
    const unsigned hw = (blockIdx.x * blockDim.x + threadIdx.x) >> 5;      // half-wave id
    const unsigned x = xcc_id();
    const unsigned rows_per_xcd = n_rows / 8;
    for (int i = 0; i < iters; ++i) {
        unsigned r = hash32(hw * 9781u + i * 6271u + 12345u);
        if (MODE == 0) r = r % n_rows; else r = x * rows_per_xcd + r % rows_per_xcd;
        float *p = buf + (size_t)r * 32 + lane32;
        if (MODE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else           __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// ---- 2. LDS atomics / reads -------------------------------------------------------------
// A tile of ROWS rows x STRIDE floats in LDS.  Lane layout of the fast kernels: 8 lanes per
// point (4 channels each), 8 points per wave; per point 4 corner rows.
// PATTERN 0: ds_add_f32, lane's 4 channels contiguous (c = 4m + j)
// PATTERN 1: ds_add_f32, lane's channels strided  (c = m + 8j)
// PATTERN 2: ds_read_b128 (value gather), accumulate to keep it live
template <int PATTERN, int STRIDE>
__global__ __launch_bounds__(256) void k_lds(float *out, int iters, int rows)
{
    extern __shared__ __attribute__((aligned(16))) float tile[];
    for (int i = threadIdx.x; i < rows * STRIDE; i += blockDim.x) tile[i] = 0.f;
    __syncthreads();
    const unsigned lane = threadIdx.x & 63, m = lane & 7, pt = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        const unsigned r = hash32(pt * 7919u + i * 104729u) % (unsigned)rows;   // random row per point
        float *row = tile + r * STRIDE;
        if (PATTERN == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) __hip_atomic_fetch_add(row + 4 * m + j, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else if (PATTERN == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j) __hip_atomic_fetch_add(row + m + 8 * j, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        } else {
            const float4 v = *reinterpret_cast<const float4 *>(row + 4 * m);
            acc += v.x + v.y + v.z + v.w;
        }
    }
    __syncthreads();
    if (PATTERN == 2) out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    else if (threadIdx.x < 32) out[blockIdx.x * 32 + threadIdx.x] = tile[threadIdx.x];
}

// ---- 3. row gathers from global: 8 lanes x float4 per 128-B row, random rows ------------
template <int UNROLL>
__global__ __launch_bounds__(256) void k_gather(const float *buf, float *out, unsigned n_rows, int iters, unsigned window)
{
    const unsigned m = threadIdx.x & 7, pt = (blockIdx.x * blockDim.x + threadIdx.x) >> 3;
    float acc = 0.f;
    // rows drawn from a window around the point's own position: models spatial locality
    const unsigned base = (unsigned)(((unsigned long long)pt * n_rows) / (gridDim.x * blockDim.x / 8));
    for (int i = 0; i < iters; i += UNROLL) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            unsigned r = (base + hash32(pt * 31u + (i + u) * 977u) % window) % n_rows;
            v[u] = *reinterpret_cast<const float4 *>(buf + (size_t)r * 32 + 4 * m);
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <typename F> float time_ms(F launch, int reps = 5)
{
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    launch(); CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int i = 0; i < reps; ++i) {
        CHECK(hipEventRecord(a)); launch(); CHECK(hipEventRecord(b)); CHECK(hipEventSynchronize(b));
        float ms; CHECK(hipEventElapsedTime(&ms, a, b)); if (ms < best) best = ms;
    }
    CHECK(hipGetLastError());
    return best;
}

int main()
{
    hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s, CUs %d, clock %d MHz\n", prop.name, prop.multiProcessorCount, prop.clockRate / 1000);

    // grad_value-sized buffer: 2 images x 13294 px x 8 heads rows of 32 floats = 27 MB
    const unsigned n_rows = 2 * 13294 * 8;
    float *buf, *out; CHECK(hipMalloc(&buf, (size_t)n_rows * 128)); CHECK(hipMalloc(&out, 64 << 20));
    CHECK(hipMemset(buf, 0, (size_t)n_rows * 128));

    {   // global atomics: total lane-atomics = blocks*256*iters
        const int blocks = 2048, iters = 256;
        const double n = (double)blocks * 256 * iters;
        float t0 = time_ms([&] { hipLaunchKernelGGL(k_global_atomics<0>, dim3(blocks), dim3(256), 0, 0, buf, n_rows, iters); });
        float t2 = time_ms([&] { hipLaunchKernelGGL(k_global_atomics<2>, dim3(blocks), dim3(256), 0, 0, buf, n_rows, iters); });
        float t1 = time_ms([&] { hipLaunchKernelGGL(k_global_atomics<1>, dim3(blocks), dim3(256), 0, 0, buf, n_rows, iters); });
        printf("global fadd agent-scope   random rows        : %8.3f ms  %7.1f G lane-atomics/s\n", t0, n / t0 / 1e6);
        printf("global fadd agent-scope   XCD-partitioned    : %8.3f ms  %7.1f G lane-atomics/s\n", t2, n / t2 / 1e6);
        printf("global fadd wg-scope (L2) XCD-partitioned    : %8.3f ms  %7.1f G lane-atomics/s\n", t1, n / t1 / 1e6);
    }
    {   // LDS: 256 CUs x 4 blocks
        const int blocks = 1024, iters = 2048;
        const double pts = (double)blocks * 32 * iters;     // points (8 lanes each)
        const int rows = 256;
#define RUN_LDS(PAT, STR, label)                                                                     \
        {   size_t sh = (size_t)rows * STR * 4;                                                       \
            float t = time_ms([&] { hipLaunchKernelGGL((k_lds<PAT, STR>), dim3(blocks), dim3(256), sh, 0, out, iters, rows); }); \
            printf("LDS %-42s: %8.3f ms  %7.1f G row-ops/s  (%6.1f G lane-elems/s)\n", label, t, pts / t / 1e6, pts * 32 / t / 1e6); }
        RUN_LDS(0, 32, "ds_add_f32 contiguous-4, stride 32")
        RUN_LDS(0, 33, "ds_add_f32 contiguous-4, stride 33")
        RUN_LDS(1, 32, "ds_add_f32 strided-8,    stride 32")
        RUN_LDS(1, 33, "ds_add_f32 strided-8,    stride 33")
        RUN_LDS(1, 40, "ds_add_f32 strided-8,    stride 40")
        RUN_LDS(2, 32, "ds_read_b128 row gather, stride 32")
        RUN_LDS(2, 36, "ds_read_b128 row gather, stride 36")
    }
    {   // global row gathers
        const int blocks = 4096, iters = 64;
        const double rows = (double)blocks * 32 * iters;
        for (unsigned window : {64u, 4096u, n_rows}) {
            float t1 = time_ms([&] { hipLaunchKernelGGL(k_gather<1>, dim3(blocks), dim3(256), 0, 0, buf, out, n_rows, iters, window); });
            float t4 = time_ms([&] { hipLaunchKernelGGL(k_gather<4>, dim3(blocks), dim3(256), 0, 0, buf, out, n_rows, iters, window); });
            float t8 = time_ms([&] { hipLaunchKernelGGL(k_gather<8>, dim3(blocks), dim3(256), 0, 0, buf, out, n_rows, iters, window); });
            printf("global 128B-row gather window %8u rows: unroll1 %7.3f ms %6.2f TB/s | unroll4 %7.3f ms %6.2f TB/s | unroll8 %7.3f ms %6.2f TB/s\n",
                   window, t1, rows * 128 / t1 / 1e9, t4, rows * 128 / t4 / 1e9, t8, rows * 128 / t8 / 1e9);
        }
    }
    // sanity: wg-scope atomics -- do the sums survive to memory?  (checked on a fresh buffer)
    {
        CHECK(hipMemset(buf, 0, (size_t)n_rows * 128));
        const int blocks = 2048, iters = 64;
        hipLaunchKernelGGL(k_global_atomics<1>, dim3(blocks), dim3(256), 0, 0, buf, n_rows, iters);
        CHECK(hipDeviceSynchronize());
        std::vector<float> h((size_t)n_rows * 32);
        CHECK(hipMemcpy(h.data(), buf, h.size() * 4, hipMemcpyDeviceToHost));
        double s = 0; for (float v : h) s += v;
        printf("wg-scope atomics checksum: got %.0f expected %.0f\n", s, (double)blocks * 256 * iters);
    }
    return 0;
}
