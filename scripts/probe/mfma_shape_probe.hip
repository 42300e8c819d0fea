// MFMA shape probe (MI355X): does v_mfma_f32_16x16x32_f16 deliver more FLOP/s than v_mfma_f32_32x32x16_f16 under the clock the
// chip holds on random data (MI355X_MICROARCH.md, DVFS give-back item 7), in the shape conv5 uses them -- 2 waves per SIMD,
// 512-thread workgroups, A fragments re-read from LDS every k-step, one accumulation chain of 48 (96) MFMAs per chunk?
//   hipcc -O3 --offload-arch=gfx950 scripts/probe/mfma_shape_probe.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// conv5's f32-equivalent inner loop: per chunk 16 k-steps x 3 products on one 32x32 accumulator, A (hi, lo) from LDS
__global__ __launch_bounds__(512) void k32(const u32x4* __restrict__ w, const u32x4* __restrict__ x, float* __restrict__ out, int chunks) {
    __shared__ u32x4 lds[2048];      // 32 KB: one chunk of hi + lo fragments
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = w[i];
    f16x8 xh[16], xl[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        xh[s] = __builtin_bit_cast(f16x8, x[(blockIdx.x * 32 + s) * 64 + lane]);
        xl[s] = __builtin_bit_cast(f16x8, x[(blockIdx.x * 32 + 16 + s) * 64 + lane]);
    }
    __syncthreads();
    f32x16 tot;
#pragma unroll
    for (int r = 0; r < 16; ++r) tot[r] = 0.f;
    for (int c = 0; c < chunks; ++c) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const f16x8 ah = __builtin_bit_cast(f16x8, lds[(2 * s) * 64 + lane]);
            const f16x8 al = __builtin_bit_cast(f16x8, lds[(2 * s + 1) * 64 + lane]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl[s], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh[s], acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) tot[r] += acc[r];
    }
    float v = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) v += tot[r];
    out[blockIdx.x * 512 + threadIdx.x] = v;
}

// the same work as 16x16x32: the 32 ch x 32 pt chunk tile = 2 x 2 tiles of 16 x 16, 8 k-steps of 32, 3 products:
// 96 MFMAs per chunk on four accumulators; A fragments (2 channel groups x hi, lo) from LDS, the same 32 KB per chunk
__global__ __launch_bounds__(512) void k16(const u32x4* __restrict__ w, const u32x4* __restrict__ x, float* __restrict__ out, int chunks) {
    __shared__ u32x4 lds[2048];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = w[i];
    f16x8 xh[2][8], xl[2][8];        // [point group][k-step]
#pragma unroll
    for (int s = 0; s < 8; ++s)
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            xh[p][s] = __builtin_bit_cast(f16x8, x[(blockIdx.x * 32 + 2 * s + p) * 64 + lane]);
            xl[p][s] = __builtin_bit_cast(f16x8, x[(blockIdx.x * 32 + 16 + 2 * s + p) * 64 + lane]);
        }
    __syncthreads();
    f32x4 tot = {0.f, 0.f, 0.f, 0.f};
    for (int c = 0; c < chunks; ++c) {
        f32x4 acc[2][2];
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int p = 0; p < 2; ++p) acc[g][p] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 8; ++s) {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const f16x8 ah = __builtin_bit_cast(f16x8, lds[((2 * s + g) * 2) * 64 + lane]);
                const f16x8 al = __builtin_bit_cast(f16x8, lds[((2 * s + g) * 2 + 1) * 64 + lane]);
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    acc[g][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, xh[p][s], acc[g][p], 0, 0, 0);
                    acc[g][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xl[p][s], acc[g][p], 0, 0, 0);
                    acc[g][p] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, xh[p][s], acc[g][p], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int p = 0; p < 2; ++p) tot += acc[g][p];
    }
    out[blockIdx.x * 512 + threadIdx.x] = tot[0] + tot[1] + tot[2] + tot[3];
}

int main() {
    const int blocks = 1024, chunks = 32 * 8;     // 8 x conv5's chunk count per workgroup: ~2 ms per launch
    std::vector<_Float16> hw(2048 * 8), hx((size_t)blocks * 32 * 64 * 8);
    srand(1);
    for (auto& v : hw) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 0.25f);
    for (auto& v : hx) v = (_Float16)((rand() / (float)RAND_MAX - 0.5f) * 2.0f);
    u32x4 *dw, *dx;
    float* dout;
    hipMalloc(&dw, hw.size() * 2);
    hipMalloc(&dx, hx.size() * 2);
    hipMalloc(&dout, (size_t)blocks * 512 * 4);
    hipMemcpy(dw, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dx, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const double flops = 2.0 * blocks * 8 * (double)chunks * 32 * 32 * 256 * 3;   // executed (3 products)
    for (int rep = 0; rep < 3; ++rep)
        for (int which = 0; which < 2; ++which) {
            for (int i = 0; i < 20; ++i) {
                if (which == 0) hipLaunchKernelGGL(k32, dim3(blocks), dim3(512), 0, 0, dw, dx, dout, chunks);
                else hipLaunchKernelGGL(k16, dim3(blocks), dim3(512), 0, 0, dw, dx, dout, chunks);
            }
            hipDeviceSynchronize();
            hipEventRecord(e0);
            const int iters = 100;
            for (int i = 0; i < iters; ++i) {
                if (which == 0) hipLaunchKernelGGL(k32, dim3(blocks), dim3(512), 0, 0, dw, dx, dout, chunks);
                else hipLaunchKernelGGL(k16, dim3(blocks), dim3(512), 0, 0, dw, dx, dout, chunks);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms;
            hipEventElapsedTime(&ms, e0, e1);
            printf("%s  %.4f ms/launch  %.1f TFLOP/s executed (fp16 MFMA, 2 waves/SIMD, A from LDS)\n", which ? "16x16x32" : "32x32x16",
                   ms / iters, flops / (ms / iters * 1e-3) / 1e12);
        }
    return 0;
}
