// Static xyz kNN graph in index form -- replaces utils/tf_util.py:647-666 (pairwise_distance_mask), which
// materialises a dense (B,N,N) float mask (64 MB per 4096-point cloud).
//
// One thread per query point; candidates stream through LDS as (x,y,z,|p|^2) float4 tiles that every lane
// reads at the same address (LDS broadcast, conflict-free).  Pass 1 keeps the 20 largest a_ij of the row in
// registers (sorted, v_max/v_min insertion network, skipped wave-wide when no lane improves) and yields
// kth = the 20th largest with multiplicity (tf.nn.top_k + reduce_min).  Pass 2 re-scans and emits every
// j with a_ij >= kth in ascending order (greater_equal, tf_util.py:664), counting past the list capacity so the
// consumer knows when a row needs the exact scan path.  VALU/LDS bound; no HBM traffic beyond xyz and the lists.
#include "common.h"

#define KNN_THREADS 256
#define KNN_TILE 1024

template <int KSEL>
__global__ __launch_bounds__(KNN_THREADS) void knn_topk_kernel(const float* __restrict__ xyz, int n, int cap,
                                                               int32_t* __restrict__ idx,
                                                               int32_t* __restrict__ cnt,
                                                               float* __restrict__ kth_out) {
    __shared__ float4 tile[KNN_TILE];
    const int cloud = blockIdx.y;
    const int i = blockIdx.x * KNN_THREADS + threadIdx.x;
    const bool valid = i < n;
    const float* pc = xyz + (size_t)cloud * n * 3;
    float xi = 0.f, yi = 0.f, zi = 0.f;
    if (valid) {
        xi = pc[3 * i + 0];
        yi = pc[3 * i + 1];
        zi = pc[3 * i + 2];
    }
    const float sqi = sq3(xi, yi, zi);

    float top[KSEL];
#pragma unroll
    for (int s = 0; s < KSEL; ++s) top[s] = -INFINITY;

    // ---- pass 1: k-th largest ----
    for (int t0 = 0; t0 < n; t0 += KNN_TILE) {
        __syncthreads();
        for (int c = threadIdx.x; c < KNN_TILE; c += KNN_THREADS) {
            const int j = t0 + c;
            float4 v = make_float4(0.f, 0.f, 0.f, INFINITY);
            if (j < n) {
                v.x = pc[3 * j + 0];
                v.y = pc[3 * j + 1];
                v.z = pc[3 * j + 2];
                v.w = sq3(v.x, v.y, v.z);
            }
            tile[c] = v;
        }
        __syncthreads();
        const int lim = min(KNN_TILE, n - t0);
#pragma unroll 4
        for (int c = 0; c < lim; ++c) {
            const float4 q = tile[c];
            float v = neg_sq_dist(sqi, xi, yi, zi, q.x, q.y, q.z, q.w);
            if (v > top[KSEL - 1]) {
#pragma unroll
                for (int s = 0; s < KSEL; ++s) {
                    const float hi = fmaxf(top[s], v);
                    v = fminf(top[s], v);
                    top[s] = hi;
                }
            }
        }
    }
    const float kth = top[KSEL - 1];

    // ---- pass 2: emit {j : a_ij >= kth} ascending ----
    int count = 0;
    int32_t* my = idx + ((size_t)cloud * n + (valid ? i : 0)) * cap;
    for (int t0 = 0; t0 < n; t0 += KNN_TILE) {
        __syncthreads();
        for (int c = threadIdx.x; c < KNN_TILE; c += KNN_THREADS) {
            const int j = t0 + c;
            float4 v = make_float4(0.f, 0.f, 0.f, INFINITY);
            if (j < n) {
                v.x = pc[3 * j + 0];
                v.y = pc[3 * j + 1];
                v.z = pc[3 * j + 2];
                v.w = sq3(v.x, v.y, v.z);
            }
            tile[c] = v;
        }
        __syncthreads();
        const int lim = min(KNN_TILE, n - t0);
#pragma unroll 4
        for (int c = 0; c < lim; ++c) {
            const float4 q = tile[c];
            const float v = neg_sq_dist(sqi, xi, yi, zi, q.x, q.y, q.z, q.w);
            if (v >= kth) {
                if (valid && count < cap) my[count] = t0 + c;
                ++count;
            }
        }
    }
    if (valid) {
        cnt[(size_t)cloud * n + i] = count;
        kth_out[(size_t)cloud * n + i] = kth;
    }
}

__global__ __launch_bounds__(256) void knn_mask_kernel(const float* __restrict__ xyz,
                                                       const float* __restrict__ kth, int n,
                                                       float* __restrict__ mask) {
    const int cloud = blockIdx.z;
    const int i = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float* pc = xyz + (size_t)cloud * n * 3;
    const float xi = pc[3 * i], yi = pc[3 * i + 1], zi = pc[3 * i + 2];
    const float xj = pc[3 * j], yj = pc[3 * j + 1], zj = pc[3 * j + 2];
    const float a = neg_sq_dist(sq3(xi, yi, zi), xi, yi, zi, xj, yj, zj, sq3(xj, yj, zj));
    mask[((size_t)cloud * n + i) * n + j] = (a >= kth[(size_t)cloud * n + i]) ? 1.0f : 0.0f;
}

extern "C" int epc_knn_topk(const float* xyz, int num_clouds, int n, int cap, int32_t* idx, int32_t* cnt,
                            float* kth, void* stream) {
    EPC_CHECK_ARG(xyz && idx && cnt && kth, "null pointer");
    EPC_CHECK_ARG(num_clouds >= 0 && n >= EPC_KNN_SELECT, "need num_points >= 20 (tf.nn.top_k k=20)");
    EPC_CHECK_ARG(cap >= EPC_KNN_SELECT, "list capacity must be >= 20");
    if (num_clouds == 0) return EPC_OK;
    dim3 grid((n + KNN_THREADS - 1) / KNN_THREADS, num_clouds);
    hipLaunchKernelGGL(knn_topk_kernel<EPC_KNN_SELECT>, grid, dim3(KNN_THREADS), 0, (hipStream_t)stream, xyz, n,
                       cap, idx, cnt, kth);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_knn_mask(const float* xyz, const float* kth, int num_clouds, int n, float* mask,
                            void* stream) {
    EPC_CHECK_ARG(xyz && kth && mask, "null pointer");
    EPC_CHECK_ARG(num_clouds >= 0 && n > 0 && n <= 65535, "bad shape");
    if (num_clouds == 0) return EPC_OK;
    dim3 grid((n + 255) / 256, n, num_clouds);
    hipLaunchKernelGGL(knn_mask_kernel, grid, dim3(256), 0, (hipStream_t)stream, xyz, kth, n, mask);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}
