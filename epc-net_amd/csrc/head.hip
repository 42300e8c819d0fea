// Descriptor heads.
//   EPC-Net  : loupe.py:295-331 + models/epc-net.py:153  (three small kernels, ~0.9 % of the FLOPs)
//   EPC-Net-L: models/epc-net-l.py:95-98                  (one kernel)
#include "common.h"

// packed head stage (floats): [C 1024*64][H KH*256][bn_s 256][bn_t 256][Wg 256*256][gbn_s 256][gbn_t 256]
// with KH = 65536 / groups.

// (V = aggregate - a_sum * centres and its per-cluster sums of squares `colss` (32 chunks of 32 features per cloud) come
// from epc_vlad_aggregate_fwd, conv5_vlad.hip.)

// ---- H1b: intra-norm over the 1024 features of each cluster, global norm, group fold ---------------------------
// In inference the grouped projection sum_g BN(v_g W) equals s * ((sum_g v_g) W) + G t (one shared W, affine BN),
// so the G group slices are folded before the GEMM: U[fi*64 + k] = sum_g Vn[g*FPG + fi][k].
// grid (FPG/16 slabs, clouds) x 256 threads: thread (k, r) owns fi = 16*slab + r + 4m, m < 4.
template <int GROUPS>
__global__ __launch_bounds__(256) void vlad_fold_kernel(const float* __restrict__ V, const float* __restrict__ colss,
                                                        float* __restrict__ U) {
    constexpr int FPG = 1024 / GROUPS;
    __shared__ float cn[64];
    __shared__ float gsum[64];
    const int slab = blockIdx.x, cloud = blockIdx.y;
    const int k = threadIdx.x & 63, r = threadIdx.x >> 6;
    if (r == 0) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 32; ++g) t += colss[((size_t)cloud * 32 + g) * 64 + k];
        const float inv = 1.0f / sqrtf(fmaxf(t, 1e-12f));
        cn[k] = inv;
        gsum[k] = t * inv * inv;  // squared norm of the normalised column (1 unless the column is ~0)
    }
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 64; ++q) tot += gsum[q];
    const float sc = cn[k] * (1.0f / sqrtf(fmaxf(tot, 1e-12f)));
    const float* v = V + (size_t)cloud * 1024 * 64;
    float* uo = U + (size_t)cloud * FPG * 64;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int fi = 16 * slab + r + 4 * m;
        float u = 0.f;
#pragma unroll
        for (int g = 0; g < GROUPS; ++g) u += v[(size_t)(g * FPG + fi) * 64 + k] * sc;
        uo[(size_t)fi * 64 + k] = u;
    }
}

// ---- H2: Yp[slice][row][256] = U[row][slice*HSL .. +HSL] @ H[slice*HSL .. +HSL][256]  (split-K, f32 MFMA) --------
// HSL = 128 k per slice: a wave's serial work is 128 f32 MFMAs (4 us) and 2 rounds of 8 k-steps of loads; with 64 rows
// (clouds) there are only two row tiles per column tile, so the K split is what fills the chip (1024 waves).
#ifndef HSL
#define HSL 128
#endif
// One wave per (32-column tile, K slice, 64-row group).  A (row on lane) is read as float4 along K: k-step (q,c) takes
// k = 8q + c from lane-half 0 and k = 8q + 4 + c from lane-half 1; B rows are read to match.
__global__ __launch_bounds__(64) void hidden_gemm_kernel(const float* __restrict__ U, const float* __restrict__ H,
                                                         int rows, int kh, float* __restrict__ Yp) {
    const int lane = threadIdx.x, j = lane & 31, h = lane >> 5;
    const int nb = blockIdx.x * 32;
    const int slice = blockIdx.y;
    const int m0 = blockIdx.z * 64;
    const int k0 = slice * HSL;
    f32x16 acc[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
    const int r0 = m0 + j, r1 = m0 + 32 + j;
    const float* a0p = U + (size_t)min(r0, rows - 1) * kh + k0 + 4 * h;
    const float* a1p = U + (size_t)min(r1, rows - 1) * kh + k0 + 4 * h;
    const float z0 = r0 < rows ? 1.f : 0.f, z1 = r1 < rows ? 1.f : 0.f;
    const float* bp = H + (size_t)(k0 + 4 * h) * 256 + nb + j;
#pragma unroll 8
    for (int q = 0; q < HSL / 8; ++q) {
        const float4 a0 = *reinterpret_cast<const float4*>(a0p + 8 * q);
        const float4 a1 = *reinterpret_cast<const float4*>(a1p + 8 * q);
        const float b0 = bp[(size_t)(8 * q + 0) * 256], b1 = bp[(size_t)(8 * q + 1) * 256];
        const float b2 = bp[(size_t)(8 * q + 2) * 256], b3 = bp[(size_t)(8 * q + 3) * 256];
        acc[0] = mfma32(a0.x * z0, b0, acc[0]);
        acc[1] = mfma32(a1.x * z1, b0, acc[1]);
        acc[0] = mfma32(a0.y * z0, b1, acc[0]);
        acc[1] = mfma32(a1.y * z1, b1, acc[1]);
        acc[0] = mfma32(a0.z * z0, b2, acc[0]);
        acc[1] = mfma32(a1.z * z1, b2, acc[1]);
        acc[0] = mfma32(a0.w * z0, b3, acc[0]);
        acc[1] = mfma32(a1.w * z1, b3, acc[1]);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + 32 * t + mfma_row(r, h);
            if (row < rows) Yp[((size_t)slice * rows + row) * 256 + nb + j] = acc[t][r];
        }
}

// ---- H3: sum slices, BN (folded, summed over groups), context gating, final L2 -------------------------------
// 1024 threads per cloud: thread (c = tid & 255, part = tid >> 8) takes a quarter of the slice sum and a quarter of the
// 256-long gating dot product, so a thread's chain of dependent L2 round trips is 4 + 4 deep instead of 8 + 32 (the
// kernel has one workgroup per cloud: it is latency, not bandwidth, that it pays for).
__global__ __launch_bounds__(1024) void head_finish_kernel(const float* __restrict__ Yp, int slices, int rows,
                                                           const float* __restrict__ hp, int groups,
                                                           const int32_t* __restrict__ status,
                                                           float* __restrict__ out) {
    __shared__ float part_y[4][256];
    __shared__ float v[256];
    __shared__ float red[4];
    const int cloud = blockIdx.x, c = threadIdx.x & 255, part = threadIdx.x >> 8;
    const float* bn_s = hp;
    const float* bn_t = hp + 256;
    const float* Wg = hp + 512;
    const float* g_s = Wg + 65536;
    const float* g_t = g_s + 256;
    // slices part, part + 4, ...: 8 independent loads in flight per thread
    float y = 0.f;
    int s = part;
    for (; s + 28 < slices; s += 32) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = Yp[((size_t)(s + 4 * u) * rows + cloud) * 256 + c];
#pragma unroll
        for (int u = 0; u < 8; ++u) y += t[u];
    }
    for (; s < slices; s += 4) y += Yp[((size_t)s * rows + cloud) * 256 + c];
    part_y[part][c] = y;
    __syncthreads();
    const float val = ((part_y[0][c] + part_y[1][c]) + (part_y[2][c] + part_y[3][c])) * bn_s[c] + (float)groups * bn_t[c];
    if (part == 0) v[c] = val;
    __syncthreads();
    float g = 0.f;
#pragma unroll 16
    for (int k = 64 * part; k < 64 * part + 64; ++k) g += v[k] * Wg[k * 256 + c];
    part_y[part][c] = g;
    __syncthreads();
    g = (part_y[0][c] + part_y[1][c]) + (part_y[2][c] + part_y[3][c]);
    g = g * g_s[c] + g_t[c];
    const float o = val * (1.0f / (1.0f + expf(-g)));
    float ss = o * o;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
    if (part == 0 && (c & 63) == 0) red[c >> 6] = ss;
    __syncthreads();
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    // a flagged cloud (EPC_STATUS_*: non-finite coordinate, fp16 range) leaves as NaN, never as a wrong finite vector
    const bool poisoned = status && status[cloud] != 0;
    if (part == 0) out[(size_t)cloud * 256 + c] = poisoned ? __int_as_float(0x7fc00000) : o * (1.0f / sqrtf(fmaxf(tot, 1e-12f)));
}

// ---- EPC-Net-L head: fc1 (1024->256, folded BN) + ReLU + L2.  packed: [Wf 1024*256][bf 256] ------------------
// 1024 threads per cloud: thread (c, part) takes a quarter of the 1024-long dot product (see head_finish_kernel).
__global__ __launch_bounds__(1024) void fc_head_kernel(const float* __restrict__ pooled,
                                                       const float* __restrict__ pack,
                                                       const int32_t* __restrict__ status, float* __restrict__ out) {
    __shared__ float m[1024];
    __shared__ float part_y[4][256];
    __shared__ float red[4];
    const int cloud = blockIdx.x, c = threadIdx.x & 255, part = threadIdx.x >> 8;
    m[threadIdx.x] = pooled[(size_t)cloud * 1024 + threadIdx.x];
    __syncthreads();
    float y = 0.f;
#pragma unroll 16
    for (int k = 256 * part; k < 256 * part + 256; ++k) y += m[k] * pack[k * 256 + c];
    part_y[part][c] = y;
    __syncthreads();
    y = fmaxf(((part_y[0][c] + part_y[1][c]) + (part_y[2][c] + part_y[3][c])) + pack[1024 * 256 + c], 0.f);
    float ss = y * y;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
    if (part == 0 && (c & 63) == 0) red[c >> 6] = ss;
    __syncthreads();
    const float tot = (red[0] + red[1]) + (red[2] + red[3]);
    const bool poisoned = status && status[cloud] != 0;
    if (part == 0) out[(size_t)cloud * 256 + c] = poisoned ? __int_as_float(0x7fc00000) : y * (1.0f / sqrtf(fmaxf(tot, 1e-12f)));
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

extern "C" size_t epc_vlad_head_workspace_bytes(int num_clouds, int groups) {
    if (num_clouds <= 0 || groups <= 0 || 64 % groups) return 0;
    const size_t kh = 65536 / groups;
    const size_t u = align_up((size_t)num_clouds * kh * sizeof(float), 256);
    const size_t yp = align_up((kh / HSL) * (size_t)num_clouds * 256 * sizeof(float), 256);
    return u + yp;
}

extern "C" int epc_vlad_head_fwd(const float* V, const float* colss, const void* packed_head, int groups,
                                 int num_clouds, float* out, const int32_t* status, void* workspace,
                                 size_t workspace_bytes, void* stream) {
    EPC_CHECK_ARG(V && colss && packed_head && out && workspace, "null pointer");
    EPC_CHECK_ARG(groups > 0 && 64 % groups == 0, "GROUPS must divide 64");
    EPC_CHECK_ARG(num_clouds >= 0, "bad shape");
    if (num_clouds == 0) return EPC_OK;
    if (workspace_bytes < epc_vlad_head_workspace_bytes(num_clouds, groups)) {
        epc_set_error("epc_vlad_head_fwd: workspace too small");
        return EPC_ENOMEM;
    }
    const int kh = 65536 / groups;
    const float* hp = (const float*)packed_head;
    const float* Hw = hp + 65536;   // past the centres (epc_vlad_aggregate_fwd's operand)
    const float* tail = Hw + (size_t)kh * 256;
    char* wsp = (char*)workspace;
    float* U = (float*)wsp;
    wsp += align_up((size_t)num_clouds * kh * sizeof(float), 256);
    float* Yp = (float*)wsp;
    hipStream_t st = (hipStream_t)stream;
    EPC_CHECK_ARG(num_clouds <= 65535, "too many clouds per call");
    switch (groups) {
#define EPC_VF(G)                                                                                                  \
    case G:                                                                                                        \
        hipLaunchKernelGGL(vlad_fold_kernel<G>, dim3(1024 / G / 16, num_clouds), dim3(256), 0, st, V, colss, U);   \
        break;
        EPC_VF(1) EPC_VF(2) EPC_VF(4) EPC_VF(8) EPC_VF(16)
#undef EPC_VF
        default:
            epc_set_error("epc_vlad_head_fwd: GROUPS must be one of 1,2,4,8,16");
            return EPC_EINVAL;
    }
    EPC_CHECK_LAUNCH();
    const int slices = kh / HSL;
    hipLaunchKernelGGL(hidden_gemm_kernel, dim3(8, slices, (num_clouds + 63) / 64), dim3(64), 0, st, U, Hw,
                       num_clouds, kh, Yp);
    EPC_CHECK_LAUNCH();
    hipLaunchKernelGGL(head_finish_kernel, dim3(num_clouds), dim3(1024), 0, st, Yp, slices, num_clouds, tail,
                       groups, status, out);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_fc_head_fwd(const float* pooled, const void* packed_fc, int num_clouds, float* out,
                               const int32_t* status, void* stream) {
    EPC_CHECK_ARG(pooled && packed_fc && out, "null pointer");
    EPC_CHECK_ARG(num_clouds >= 0, "bad shape");
    if (num_clouds == 0) return EPC_OK;
    hipLaunchKernelGGL(fc_head_kernel, dim3(num_clouds), dim3(1024), 0, (hipStream_t)stream, pooled,
                       (const float*)packed_fc, status, out);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}
