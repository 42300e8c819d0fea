// ProxyConv block (models/epc-net.py:66-83 and its three repeats) as ONE kernel per block:
//
//   xm  = (sum_{j in nbr(i)} x_j) / k         gather over the kNN index lists (the reference multiplies a dense
//                                             (N,N) 0/1 mask: 2.15 GFLOP per cloud and block; here 5.2 M adds)
//   t   = xm - x
//   t   = relu(bn(conv_a(t)))                 64x64, bf16x3 MFMA (f32-accurate), BN folded into the weights
//   t   = relu(bn(conv_b(t)))                 B operand = the previous accumulators (no LDS round trip)
//   out = t + xm                  -> concat buffer slice (models/epc-net.py:134)
//   x'  = relu(bn(conv_{b+1}(out)))           the next block's leading conv, fused (grid-wide dependency sits
//                                             only at the gather, so one launch per block is the minimum)
//
// Geometry: 512 threads = 8 waves; each wave owns 32 consecutive points (one MFMA column tile: point = lane&31).
// Transposed orientation out^T[ch][pt] = W^T x^T so that a layer's accumulators (channel in the register index,
// point on the lane) are directly the next layer's B operands.
// The gather runs 16 lanes x float4 per point (full 256-B rows, 4 points per wave-instruction), results go
// through a per-wave LDS staging tile to switch between the row layout and the MFMA layout.
#include "common.h"

#define BLK_THREADS 512
#define BLK_WAVES 8
#define ST_STRIDE 68  // floats per staged row: 64 + 4 pad -> conflict-free b128 reads in both layouts
#define BLK_PACK EPC_BLOCK_PACK_FLOATS
#define BLK_LDS_FLOATS (BLK_PACK + BLK_WAVES * 32 * ST_STRIDE)

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

__device__ __forceinline__ void acc_init_bias(f32x16& acc, const float* bias32, int h) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 b = ld4(bias32 + 8 * g + 4 * h);
        acc[4 * g + 0] = b.x;
        acc[4 * g + 1] = b.y;
        acc[4 * g + 2] = b.z;
        acc[4 * g + 3] = b.w;
    }
}

__device__ __forceinline__ void relu16(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.f);
}

__device__ __forceinline__ bf16x8 ldfrag(const float* p) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
}

// The three 64x64 layers run on the bf16 MFMA in split (bf16x3) arithmetic like conv5: 24 MFMAs of 32
// cycles per layer instead of 64 f32 MFMAs of 64 cycles.  Weight fragments: pack.hip fold_pack_block_bf16_kernel.
// 64->64 layer whose B operand comes from a staged [pt][64] row: k-step s = channels 16s + 8h .. +7 of the lane's point.
__device__ __forceinline__ void layer_split64(const float* lw, const float* lbias, const bf16x8 (&bh)[4],
                                              const bf16x8 (&bl)[4], f32x16 (&acc)[2], int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        acc_init_bias(acc[t], lbias + 32 * t, h);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const bf16x8 ah = ldfrag(lw + (((t * 4 + s) * 2 + 0) * 64 + lane) * 4);
            const bf16x8 al = ldfrag(lw + (((t * 4 + s) * 2 + 1) * 64 + lane) * 4);
            acc[t] = mfma_bf16(al, bh[s], acc[t]);
            acc[t] = mfma_bf16(ah, bl[s], acc[t]);
            acc[t] = mfma_bf16(ah, bh[s], acc[t]);
        }
    }
}

// 64->64 layer whose B operand is the previous layer's accumulators: k-step (tin, s') = registers 8s'..8s'+7 of tile tin.
__device__ __forceinline__ void layer_acc64(const float* lw, const float* lbias, const f32x16 (&in)[2],
                                            f32x16 (&acc)[2], int lane) {
    const int h = lane >> 5;
    bf16x8 fh[4], fl[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = in[st >> 1][8 * (st & 1) + q];
        split8(v, fh[st], fl[st]);
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        acc_init_bias(acc[t], lbias + 32 * t, h);
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const bf16x8 ah = ldfrag(lw + (((t * 4 + st) * 2 + 0) * 64 + lane) * 4);
            const bf16x8 al = ldfrag(lw + (((t * 4 + st) * 2 + 1) * 64 + lane) * 4);
            acc[t] = mfma_bf16(al, fh[st], acc[t]);
            acc[t] = mfma_bf16(ah, fl[st], acc[t]);
            acc[t] = mfma_bf16(ah, fh[st], acc[t]);
        }
    }
}

// accumulators (channel in register, point on lane) -> staged [pt][64] rows
__device__ __forceinline__ void acc_to_stage(float* st, const f32x16 (&acc)[2], int lane) {
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            st4(st + j * ST_STRIDE + 32 * t + 8 * g + 4 * h,
                make_float4(acc[t][4 * g], acc[t][4 * g + 1], acc[t][4 * g + 2], acc[t][4 * g + 3]));
}

// staged [pt][64] row of the lane's point -> split B fragments (k-step s = channels 16s + 8h .. +7)
__device__ __forceinline__ void stage_to_bop(const float* st, bf16x8 (&bh)[4], bf16x8 (&bl)[4], int lane) {
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float4 a = ld4(st + j * ST_STRIDE + 16 * s + 8 * h), b = ld4(st + j * ST_STRIDE + 16 * s + 8 * h + 4);
        float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        split8(v, bh[s], bl[s]);
    }
}

__global__ __launch_bounds__(BLK_THREADS) void proxyconv_block_kernel(
    const float* __restrict__ x, const float* __restrict__ xyz, const int32_t* __restrict__ idx,
    const int32_t* __restrict__ cnt, const float* __restrict__ kth, int cap, const float* __restrict__ pack,
    int has_next, int total_points, int n, float kdiv, float* __restrict__ out, int out_stride, int out_off,
    float* __restrict__ x_next) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int o = tid * 4; o < BLK_PACK; o += BLK_THREADS * 4) st4(lds + o, ld4(pack + o));
    __syncthreads();
    const float* wa = lds;
    const float* ba = lds + 4096;
    const float* wb = lds + 4160;
    const float* bb = lds + 4160 + 4096;
    const float* wn = lds + 8320;
    const float* bn = lds + 8320 + 4096;
    float* st = lds + BLK_PACK + wave * 32 * ST_STRIDE;

    // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD group), so
    // giving each group a CONTIGUOUS range of tiles keeps all tiles of a cloud -- which gather from the same 1 MB of
    // x rows -- behind one L2 instead of eight.  Speed only: any mapping is correct.
    int bid = blockIdx.x;
#ifndef BLK_NO_XCD_REMAP
    {
        const int nb = gridDim.x, q = nb >> 3, r = nb & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;  // bijective for any grid size
    }
#endif
    const int g0 = (bid * BLK_WAVES + wave) * 32;
    if (g0 >= total_points) return;  // no further workgroup barriers below
    const int cloud_base = (g0 / n) * n;
    const int p = lane >> 4, q = lane & 15;
    const float4* x4 = reinterpret_cast<const float4*>(x);

    // ---- gather-mean, 4 points per pass ----
    float4 xm[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int g = g0 + 4 * s + p;
        const int c = cnt[g];
        const bool ovf = c > cap;
        // every row holds >= 20 valid entries (cnt >= 20 by construction); ties beyond 20 are the rare tail
        const int4* il = reinterpret_cast<const int4*>(idx + (size_t)g * cap);
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!ovf) {
            int nb[EPC_KNN_SELECT];
#pragma unroll
            for (int m4 = 0; m4 < EPC_KNN_SELECT / 4; ++m4) {
                const int4 t = il[m4];
                nb[4 * m4] = t.x;
                nb[4 * m4 + 1] = t.y;
                nb[4 * m4 + 2] = t.z;
                nb[4 * m4 + 3] = t.w;
            }
            float4 v[EPC_KNN_SELECT];
#pragma unroll
            for (int m = 0; m < EPC_KNN_SELECT; ++m) v[m] = x4[(size_t)(cloud_base + nb[m]) * 16 + q];
#pragma unroll
            for (int m = 0; m < EPC_KNN_SELECT; ++m) {  // ascending j, one rounding per add
                acc.x += v[m].x;
                acc.y += v[m].y;
                acc.z += v[m].z;
                acc.w += v[m].w;
            }
            for (int m = EPC_KNN_SELECT; m < c; ++m) {
                const float4 w = x4[(size_t)(cloud_base + idx[(size_t)g * cap + m]) * 16 + q];
                acc.x += w.x;
                acc.y += w.y;
                acc.z += w.z;
                acc.w += w.w;
            }
        }
        if (__any(ovf)) {
            // more than `cap` entries satisfy a_ij >= kth (ties / zero-padded cloud): exact scan of the row
            if (ovf) {
                const float* pc = xyz + (size_t)cloud_base * 3;
                const int i = g - cloud_base;
                const float xi = pc[3 * i], yi = pc[3 * i + 1], zi = pc[3 * i + 2];
                const float sqi = sq3(xi, yi, zi);
                const float kv = kth[g];
                for (int j = 0; j < n; ++j) {
                    const float xj = pc[3 * j], yj = pc[3 * j + 1], zj = pc[3 * j + 2];
                    const float a = neg_sq_dist(sqi, xi, yi, zi, xj, yj, zj, sq3(xj, yj, zj));
                    if (a >= kv) {
                        const float4 v = x4[(size_t)(cloud_base + j) * 16 + q];
                        acc.x += v.x;
                        acc.y += v.y;
                        acc.z += v.z;
                        acc.w += v.w;
                    }
                }
            }
        }
        acc.x /= kdiv;
        acc.y /= kdiv;
        acc.z /= kdiv;
        acc.w /= kdiv;
        xm[s] = acc;
        const float4 xi4 = x4[(size_t)g * 16 + q];
        st4(st + (4 * s + p) * ST_STRIDE + 4 * q,
            make_float4(acc.x - xi4.x, acc.y - xi4.y, acc.z - xi4.z, acc.w - xi4.w));
    }

    // ---- conv_a, conv_b ----
    bf16x8 bh[4], bl[4];
    f32x16 a1[2], a2[2];
    stage_to_bop(st, bh, bl, lane);
    layer_split64(wa, ba, bh, bl, a1, lane);
    relu16(a1[0]);
    relu16(a1[1]);
    layer_acc64(wb, bb, a1, a2, lane);
    relu16(a2[0]);
    relu16(a2[1]);

    // ---- out = t + xm (row layout, coalesced 256-B rows) ----
    acc_to_stage(st, a2, lane);
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        float* row = st + (4 * s + p) * ST_STRIDE + 4 * q;
        const float4 t = ld4(row);
        const float4 o = make_float4(t.x + xm[s].x, t.y + xm[s].y, t.z + xm[s].z, t.w + xm[s].w);
        st4(out + (size_t)(g0 + 4 * s + p) * out_stride + out_off + 4 * q, o);
        st4(row, o);
    }
    if (!has_next) return;

    // ---- next block's leading conv ----
    stage_to_bop(st, bh, bl, lane);
    layer_split64(wn, bn, bh, bl, a1, lane);
    relu16(a1[0]);
    relu16(a1[1]);
    acc_to_stage(st, a1, lane);
#pragma unroll
    for (int s = 0; s < 8; ++s)
        st4(x_next + (size_t)(g0 + 4 * s + p) * 64 + 4 * q, ld4(st + (4 * s + p) * ST_STRIDE + 4 * q));
}

// conv1 (models/epc-net.py:66-69): 3 -> 64, folded BN, ReLU.  16 lanes x float4 per point.
__global__ __launch_bounds__(256) void conv1_kernel(const float* __restrict__ xyz, const float* __restrict__ pack,
                                                    int total_points, float* __restrict__ x) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int g = t >> 4, q = t & 15;
    if (g >= total_points) return;
    const float px = xyz[3 * (size_t)g], py = xyz[3 * (size_t)g + 1], pz = xyz[3 * (size_t)g + 2];
    const float4 w0 = ld4(pack + 4 * q), w1 = ld4(pack + 64 + 4 * q), w2 = ld4(pack + 128 + 4 * q);
    const float4 b = ld4(pack + 192 + 4 * q);
    float4 y;
    y.x = fmaxf(((px * w0.x + py * w1.x) + pz * w2.x) + b.x, 0.f);
    y.y = fmaxf(((px * w0.y + py * w1.y) + pz * w2.y) + b.y, 0.f);
    y.z = fmaxf(((px * w0.z + py * w1.z) + pz * w2.z) + b.z, 0.f);
    y.w = fmaxf(((px * w0.w + py * w1.w) + pz * w2.w) + b.w, 0.f);
    st4(x + (size_t)g * 64 + 4 * q, y);
}

extern "C" int epc_conv1_fwd(const float* xyz, const void* packed_conv1, int num_points_total, float* x,
                             void* stream) {
    EPC_CHECK_ARG(xyz && packed_conv1 && x, "null pointer");
    EPC_CHECK_ARG(num_points_total >= 0, "bad shape");
    if (num_points_total == 0) return EPC_OK;
    const long threads = (long)num_points_total * 16;
    hipLaunchKernelGGL(conv1_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       xyz, (const float*)packed_conv1, num_points_total, x);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_proxyconv_block_fwd(const float* x, const float* xyz, const int32_t* idx, const int32_t* cnt,
                                       const float* kth, int cap, const void* packed_block, int has_next,
                                       int num_clouds, int n, int knn, float* out, int out_stride, int out_off,
                                       float* x_next, void* stream) {
    EPC_CHECK_ARG(x && xyz && idx && cnt && kth && packed_block && out, "null pointer");
    EPC_CHECK_ARG(!has_next || x_next, "x_next required when has_next");
    EPC_CHECK_ARG(cap == EPC_KNN_CAP, "neighbour-list capacity must be EPC_KNN_CAP (32)");
    EPC_CHECK_ARG(n > 0 && n % 32 == 0, "num_points must be a multiple of 32");
    EPC_CHECK_ARG(knn > 0, "KNN divisor must be positive");
    EPC_CHECK_ARG(out_stride % 4 == 0 && out_off % 4 == 0 && out_off + 64 <= out_stride, "bad output slice");
    if (num_clouds <= 0) return num_clouds == 0 ? EPC_OK : EPC_EINVAL;
    const long total = (long)num_clouds * n;
    EPC_CHECK_ARG(total < (1L << 31), "too many points");
    static const size_t lds_bytes = BLK_LDS_FLOATS * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(proxyconv_block_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
        epc_set_error("epc_proxyconv_block_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return EPC_EHIP;
    }
    const unsigned blocks = (unsigned)((total + BLK_WAVES * 32 - 1) / (BLK_WAVES * 32));
    hipLaunchKernelGGL(proxyconv_block_kernel, dim3(blocks), dim3(BLK_THREADS), lds_bytes, (hipStream_t)stream, x,
                       xyz, idx, cnt, kth, cap, (const float*)packed_block, has_next, (int)total, n, (float)knn,
                       out, out_stride, out_off, x_next);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}
