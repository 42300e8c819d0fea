// Probe of the fp6 (e2m3) path of v_mfma_scale_f32_32x32x64_f8f6f4 and of v_cvt_scalef32_pk32_fp6_f16 on gfx950:
//  (1) bit layout of the 32 x 6-bit register group the conversion produces (one-hot inputs);
//  (2) operand lane map: lane l supplies row / column l & 31, k = 32 * (l >> 5) + element (exact-integer product);
//  (3) scale operands: per-lane E8M0 bytes, byte chosen by op_sel.
// Build + run on the GPU box: see run_fp6_probe.py.
#include <hip/hip_runtime.h>
typedef _Float16 h32 __attribute__((ext_vector_type(32)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

extern "C" __global__ void onehot(int* out) {   // lane j < 32: element j = 1.0, the rest 0
    const int l = threadIdx.x;
    h32 v;
    for (int i = 0; i < 32; ++i) v[i] = (i == (l & 31)) ? (_Float16)1.0f : (_Float16)0.0f;
    const i32x6 r = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v, 1.0f);
    for (int i = 0; i < 6; ++i) out[l * 6 + i] = r[i];
}

template <int OPA, int OPB>
__device__ f32x16 run(const i32x8& a, const i32x8& b, int sa, int sb) {
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 2, 2, OPA, sa, OPB, sb);
}

// A (32 x 64), B (64 x 32) floats holding values exact in e2m3; scale_a / scale_b: 64 ints (one per lane)
extern "C" __global__ void probe(const float* A, const float* B, float* D, const int* scale_a, const int* scale_b, int opsel) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    h32 va, vb;
    for (int j = 0; j < 32; ++j) {
        va[j] = (_Float16)A[r * 64 + 32 * h + j];
        vb[j] = (_Float16)B[(32 * h + j) * 32 + r];
    }
    const i32x6 pa = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(va, 1.0f);
    const i32x6 pb = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(vb, 1.0f);
    const i32x8 a = {pa[0], pa[1], pa[2], pa[3], pa[4], pa[5], 0, 0};
    const i32x8 b = {pb[0], pb[1], pb[2], pb[3], pb[4], pb[5], 0, 0};
    f32x16 c;
    if (opsel == 0) c = run<0, 0>(a, b, scale_a[l], scale_b[l]);
    else if (opsel == 1) c = run<1, 1>(a, b, scale_a[l], scale_b[l]);
    else if (opsel == 2) c = run<2, 2>(a, b, scale_a[l], scale_b[l]);
    else c = run<3, 3>(a, b, scale_a[l], scale_b[l]);
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        D[row * 32 + r] = c[i];
    }
}

extern "C" int run_onehot(int* out, void* stream) {
    hipLaunchKernelGGL(onehot, dim3(1), dim3(64), 0, (hipStream_t)stream, out);
    return (int)hipGetLastError();
}
extern "C" int run_probe(const float* A, const float* B, float* D, const int* sa, const int* sb, int opsel, void* stream) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, (hipStream_t)stream, A, B, D, sa, sb, opsel);
    return (int)hipGetLastError();
}
