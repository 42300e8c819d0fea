// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 (fp8 e4m3 x fp8 e4m3): operand lane maps and scale semantics.
// A (32 x 64) and B (64 x 32) arrive as float matrices; the kernel converts them to fp8 with the map under test
// (lane l: row/col = l & 31, k = 32 * (l >> 5) + j, byte j of the 32-byte fragment) and writes D (32 x 32).
#include <hip/hip_runtime.h>
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ unsigned char to_fp8(float v) {   // e4m3fn via the hardware conversion (v_cvt_pk_fp8_f32)
    const int packed = __builtin_amdgcn_cvt_pk_fp8_f32(v, 0.0f, 0, false);
    return (unsigned char)(packed & 0xff);
}

extern "C" __global__ void probe(const float* A, const float* B, float* D, int scale_a, int scale_b) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    i32x8 a, b;
    for (int w = 0; w < 8; ++w) {
        unsigned int wa = 0, wb = 0;
        for (int q = 0; q < 4; ++q) {
            const int k = 32 * h + 4 * w + q;
            wa |= (unsigned int)to_fp8(A[r * 64 + k]) << (8 * q);
            wb |= (unsigned int)to_fp8(B[k * 32 + r]) << (8 * q);
        }
        a[w] = (int)wa;
        b[w] = (int)wb;
    }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, scale_a, 0, scale_b);
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        D[row * 32 + r] = c[i];
    }
}

extern "C" int run_probe(const float* A, const float* B, float* D, int scale_a, int scale_b, void* stream) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, D, scale_a, scale_b);
    return (int)hipGetLastError();
}
