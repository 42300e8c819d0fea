// Streaming-read bandwidth probe (MI355X): how many bytes per CU must be in flight to reach the HBM read rate?
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 scripts/probe/read_bw.hip -o /tmp/read_bw && /tmp/read_bw
// Every wave reads its contiguous slice with U independent 16-B-per-lane loads per iteration (U KB in flight per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void read_kernel(const u32x4* __restrict__ src, size_t vec_per_wave, unsigned* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const u32x4* p = src + wave * vec_per_wave + lane;
    u32x4 acc = {0, 0, 0, 0};
    for (size_t i = 0; i < vec_per_wave; i += 64 * U) {
        u32x4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __builtin_nontemporal_load(p + i + 64 * u);
#pragma unroll
        for (int u = 0; u < U; ++u) acc ^= v[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[wave] = 1;   // never true: keeps the loads alive
}

// The aggregate kernel's pattern: 8 waves share a stream of 64-KB tiles, each reading its own 8-KB eighth of every tile
// (8 KB contiguous, then a 64-KB stride); `tiles` tiles per stream, tile t+1 requested while tile t is "processed".
__global__ __launch_bounds__(256) void tile_kernel(const u32x4* __restrict__ src, int tiles, unsigned* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const size_t stream = wave >> 3, sub = wave & 7;
    const u32x4* p = src + (stream * tiles * 4096) + sub * 512 + lane;   // 64 KB = 4096 vectors, 8 KB = 512
    u32x4 acc = {0, 0, 0, 0};
    u32x4 a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = __builtin_nontemporal_load(p + 64 * u);
    for (int t = 0; t < tiles; t += 2) {
#pragma unroll
        for (int u = 0; u < 8; ++u) b[u] = __builtin_nontemporal_load(p + (size_t)(t + 1) * 4096 + 64 * u);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= a[u];
        if (t + 2 < tiles) {
#pragma unroll
            for (int u = 0; u < 8; ++u) a[u] = __builtin_nontemporal_load(p + (size_t)(t + 2) * 4096 + 64 * u);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= b[u];
    }
    if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) out[wave] = 1;
}

static void run_tiles(const u32x4* src, int wgs, int tiles, unsigned* out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(tile_kernel, dim3(wgs), dim3(256), 0, 0, src, tiles, out);
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(tile_kernel, dim3(wgs), dim3(256), 0, 0, src, tiles, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double moved = (double)wgs * 4 * tiles * 8192;
    printf("tile pattern: wgs=%d tiles/wave=%d: %.3f ms  %.2f TB/s (%.0f MB)\n", wgs, tiles, ms / reps, moved / (ms / reps * 1e-3) / 1e12, moved / 1e6);
}

template <int U>
static void run(const u32x4* src, size_t bytes, int wgs, unsigned* out) {
    const size_t waves = (size_t)wgs * 4;
    size_t vec_per_wave = bytes / 16 / waves;
    vec_per_wave -= vec_per_wave % (64 * U);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(read_kernel<U>, dim3(wgs), dim3(256), 0, 0, src, vec_per_wave, out);
    hipEventRecord(e0, 0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(read_kernel<U>, dim3(wgs), dim3(256), 0, 0, src, vec_per_wave, out);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double moved = (double)vec_per_wave * 16 * waves;
    printf("U=%2d (%2d KB/wave) wgs=%5d (%4.1f waves/CU, %5.0f KB in flight/CU): %.3f ms  %.2f TB/s\n", U, U, wgs, wgs * 4 / 256.0,
           U * wgs * 4 / 256.0, ms / reps, moved / (ms / reps * 1e-3) / 1e12);
}

int main() {
    const size_t bytes = 640ull << 20;    // past the 256-MiB Infinity Cache, about the aggregate's input
    u32x4* src; unsigned* out;
    hipMalloc(&src, bytes); hipMalloc(&out, 1 << 22);
    hipMemset(src, 1, bytes);
    for (int wgs : {256, 512, 1024, 2048}) {
        run<1>(src, bytes, wgs, out);
        run<2>(src, bytes, wgs, out);
        run<4>(src, bytes, wgs, out);
        run<8>(src, bytes, wgs, out);
        run<16>(src, bytes, wgs, out);
    }
    run_tiles(src, 512, 32, out);     // the aggregate's grid at batch 64: 2 x 4 splits x 64 clouds
    run_tiles(src, 1024, 16, out);
    run_tiles(src, 256, 64, out);
    return 0;
}
