// L2 channel probe: does "head = XCD" (every workgroup of an XCD gathers rows of ONE head: rows of `row` bytes at a
// stride of 8 rows, fixed offset) use all of an XCD's L2 channels?  Gathers L2-resident rows in three address patterns
// and prints the rate of each:
//   0  head = blockIdx % 8 (the XCD under round-robin dispatch)      1  head random per row
//   2  head = (pixel + blockIdx) % 8 (every XCD touches all offsets, footprint as in 0 per pixel)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probes/l2_channel_probe tools/probes/l2_channel_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ unsigned mix(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int ROW>      // bytes per row: 128 (float32, 32 channels) or 64 (bf16)
__global__ __launch_bounds__(256) void probe(const char *base, unsigned n_pix, int mode, int iters, uint4 *sink)
{
    constexpr int LPR = ROW / 16, STRIDE = ROW * 8;
    const int lane = threadIdx.x & 63, slot = lane % LPR, grp = lane / LPR;
    const unsigned wid = blockIdx.x * 4 + threadIdx.x / 64;
    uint4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        uint4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned r = mix(wid * 0x9e3779b9u + (unsigned)(it * 8 + u) * 64u + (unsigned)grp);
            const unsigned pix = r % n_pix;
            unsigned head = blockIdx.x % 8;
            if (mode == 1) head = (r >> 20) & 7;
            if (mode == 2) head = (pix + blockIdx.x) % 8;
            v[u] = *reinterpret_cast<const uint4 *>(base + (size_t)pix * STRIDE + head * ROW + slot * 16);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) { acc.x ^= v[u].x; acc.y ^= v[u].y; acc.z ^= v[u].z; acc.w ^= v[u].w; }
    }
    if (acc.x == 0x12345678u) sink[0] = acc;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 400;
    for (int row : {128, 64}) {
        for (unsigned n_pix : {1024u, 2048u, 8192u, 32768u}) {
            const size_t bytes = (size_t)n_pix * row * 8;
            char *buf; uint4 *sink;
            hipMalloc(&buf, bytes); hipMalloc(&sink, 64);
            hipMemset(buf, 1, bytes);
            for (int mode = 0; mode < 3; ++mode) {
                hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
                const int blocks = 2048;
                auto launch = [&]() {
                    if (row == 128) hipLaunchKernelGGL(probe<128>, dim3(blocks), dim3(256), 0, 0, buf, n_pix, mode, iters, sink);
                    else hipLaunchKernelGGL(probe<64>, dim3(blocks), dim3(256), 0, 0, buf, n_pix, mode, iters, sink);
                };
                launch(); launch();
                hipEventRecord(a);
                for (int k = 0; k < 5; ++k) launch();
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
                const double gathered = (double)blocks * 256 * 16 * 8 * iters;
                printf("row %3d B  array %6.1f MB (%5.2f MB a head)  mode %d: %7.1f us  %6.2f TB/s\n", row, bytes / 1e6,
                       bytes / 8e6, mode, ms * 1e3, gathered / ms / 1e9);
            }
            hipFree(buf); hipFree(sink);
        }
    }
    return 0;
}
