// Descriptor retrieval -- replaces sklearn KDTree(database).query(q, k=25) of evaluate.py:463,481.
// One wave per query.  Exact Euclidean distances sum_c (q_c - d_c)^2 in f32 (no ||q||^2+||d||^2-2q.d cancellation),
// kept in LDS; then k rounds of wave-wide arg-min (ties -> lower database index), so the result is the sorted
// k-nearest list.  Database sizes on this path are 10^2..10^4 rows (Oxford runs hold ~400 submaps each).
#include "common.h"

__global__ __launch_bounds__(64) void pairwise_topk_kernel(const float* __restrict__ db, int num_db,
                                                           const float* __restrict__ queries, int dim, int k,
                                                           int32_t* __restrict__ idx, float* __restrict__ dist) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* qv = lds;          // dim
    float* d2 = lds + dim;    // num_db
    const int lane = threadIdx.x;
    const int qi = blockIdx.x;
    for (int c = lane; c < dim; c += 64) qv[c] = queries[(size_t)qi * dim + c];
    __syncthreads();
    for (int d = lane; d < num_db; d += 64) {
        const float* row = db + (size_t)d * dim;
        float acc = 0.f;
        for (int c = 0; c < dim; c += 4) {
            const float4 v = *reinterpret_cast<const float4*>(row + c);
            const float e0 = qv[c] - v.x, e1 = qv[c + 1] - v.y, e2 = qv[c + 2] - v.z, e3 = qv[c + 3] - v.w;
            acc += e0 * e0;
            acc += e1 * e1;
            acc += e2 * e2;
            acc += e3 * e3;
        }
        d2[d] = acc;
    }
    __syncthreads();
    for (int r = 0; r < k; ++r) {
        float best = INFINITY;
        int bi = 0x7fffffff;
        for (int d = lane; d < num_db; d += 64) {
            const float v = d2[d];
            if (v < best) {  // ascending d within a lane: strict < keeps the lower index on ties
                best = v;
                bi = d;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(best, off);
            const int oi = __shfl_xor(bi, off);
            if (ov < best || (ov == best && oi < bi)) {
                best = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            idx[(size_t)qi * k + r] = bi;
            dist[(size_t)qi * k + r] = sqrtf(best);
            d2[bi] = INFINITY;
        }
        __syncthreads();
    }
}

extern "C" int epc_pairwise_topk(const float* database, int num_db, const float* queries, int num_q, int dim,
                                 int k, int32_t* idx, float* dist, void* stream) {
    EPC_CHECK_ARG(database && queries && idx && dist, "null pointer");
    EPC_CHECK_ARG(dim > 0 && dim % 4 == 0, "descriptor dim must be a multiple of 4");
    EPC_CHECK_ARG(k > 0 && k <= num_db, "need 0 < k <= num_db");
    EPC_CHECK_ARG(num_q >= 0, "bad shape");
    const size_t lds_bytes = ((size_t)dim + num_db) * sizeof(float);
    EPC_CHECK_ARG(lds_bytes <= 160 * 1024, "database shard too large for one LDS-resident pass (shard it)");
    if (num_q == 0) return EPC_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pairwise_topk_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
        epc_set_error("epc_pairwise_topk: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return EPC_EHIP;
    }
    hipLaunchKernelGGL(pairwise_topk_kernel, dim3(num_q), dim3(64), lds_bytes, (hipStream_t)stream, database,
                       num_db, queries, dim, k, idx, dist);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}
