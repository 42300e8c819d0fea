// Tail of the training step as single launches: the VLAD normalisations (loupe.py:292-298) and the lazy quadruplet loss
// (models/epc-net.py:269-284), forward and backward.  In eager torch arithmetic these were ~100 launches of 4-5 us each
// per step on a few KB..MB of data (a kernel costs at least 4.6 us even inside a HIP graph); here each is one kernel.
#include "common.h"

// ----------------------------------------------------------------------------------------------------------------
// VLAD normalisations: v = raw - a_sum (x) w2 (loupe.py:284,292); u = l2_normalize(v, over the F axis) per (cloud,
// cluster) (:295); out = l2_normalize(flatten(u)) per cloud (:297-298).  raw, out: (B, F, C) f32 with C == 64; a_sum:
// (B, C); w2: (F, C).  One workgroup of 1024 threads per cloud: thread = (cluster c = tid & 63, feature group tid >> 6).
//   r1[b][c] = rsqrt(max(sum_f v^2, 1e-12)),  r2[b] = rsqrt(max(sum_{f,c} (v r1)^2, 1e-12))
// Backward (out saved): with T_c = sum_f do*o, Q_c = sum_f o^2, S = sum_c T_c (0 if the outer clamp was active):
//   du*u summed over f = T_c - S Q_c =: S_c (0 if the inner clamp was active),
//   dv = r1 (r2 (do - o S) - (o / r2) S_c);   d raw = dv;   d a_sum[c] = -sum_f dv w2[f][c];   the w2 gradient
//   (-sum_b a_sum[b][c] dv[b]) is left to the caller (it crosses clouds).
// ----------------------------------------------------------------------------------------------------------------
#define VN_THREADS 1024
#define VN_GROUPS (VN_THREADS / 64)
// Round 6: VN_SLICES workgroups per cloud.  Each runs the WHOLE first pass (the column sums: the cloud's 256 - 768 KB, from L2 for all but
// the first) -- the same sums in the same order, so nothing has to cross workgroups -- and writes one slice of the second pass: one
// workgroup per cloud left 18 CUs moving a megabyte each (19 + 20 us); 8 x 18 workgroups: 8 + 9 us.
#define VN_SLICES 8
#define L2_EPS 1e-12f
#ifndef VN_UNROLL
#define VN_UNROLL 8     // rows of a column in flight per thread and round trip (16 and 32 measured in round 6: the step 2.02 -> 2.05 / 2.07 ms)
#endif

__device__ __forceinline__ float group_sum(float v, float (*red)[64], int c, int g) {  // sum over the 16 feature groups
    __syncthreads();
    red[g][c] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < VN_GROUPS; ++q) t += red[q][c];
    return t;
}

__global__ __launch_bounds__(VN_THREADS) void vlad_normalize_fwd_kernel(const float* __restrict__ raw,
                                                                        const float* __restrict__ a_sum,
                                                                        const float* __restrict__ w2, int F,
                                                                        float* __restrict__ out, float* __restrict__ r1,
                                                                        float* __restrict__ r2) {
    __shared__ float red[VN_GROUPS][64];
    __shared__ float s_tot[64];
    const int b = blockIdx.x, sl = blockIdx.y, c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const float* pr = raw + (size_t)b * F * 64;
    const float as = a_sum[b * 64 + c];
    float ss = 0.f;
#pragma unroll VN_UNROLL
    for (int f = g; f < F; f += VN_GROUPS) {
        const float v = pr[f * 64 + c] - as * w2[f * 64 + c];
        ss += v * v;
    }
    ss = group_sum(ss, red, c, g);
    const float rc = 1.0f / sqrtf(fmaxf(ss, L2_EPS));
    // sum over clusters of (v rc)^2 = sum_c rc^2 ss_c
    if (g == 0) s_tot[c] = rc * rc * ss;
    __syncthreads();
    float tot = 0.f;
#pragma unroll
    for (int q = 0; q < 64; ++q) tot += s_tot[q];
    const float rb = 1.0f / sqrtf(fmaxf(tot, L2_EPS));
    float* po = out + (size_t)b * F * 64;
    const int f0 = sl * (F / VN_SLICES), f1 = sl + 1 == VN_SLICES ? F : f0 + F / VN_SLICES;   // this workgroup's slice of the second pass
#pragma unroll VN_UNROLL
    for (int f = f0 + g; f < f1; f += VN_GROUPS) {
        const float v = pr[f * 64 + c] - as * w2[f * 64 + c];
        po[f * 64 + c] = (v * rc) * rb;
    }
    if (sl == 0) {
        if (g == 0) r1[b * 64 + c] = rc;
        if (threadIdx.x == 0) r2[b] = rb;
    }
}

__global__ __launch_bounds__(VN_THREADS) void vlad_normalize_bwd_kernel(const float* __restrict__ dout,
                                                                        const float* __restrict__ out,
                                                                        const float* __restrict__ r1,
                                                                        const float* __restrict__ r2,
                                                                        const float* __restrict__ w2, int F,
                                                                        float* __restrict__ draw,
                                                                        float* __restrict__ da_sum) {
    __shared__ float red[VN_GROUPS][64];
    __shared__ float s_t[64];
    const int b = blockIdx.x, c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const float* pd = dout + (size_t)b * F * 64;
    const float* po = out + (size_t)b * F * 64;
    float T = 0.f, Q = 0.f, DW = 0.f, OW = 0.f;   // + sum_f d w2, sum_f o w2: d a_sum follows from the four sums (dv is linear in d and o)
#pragma unroll VN_UNROLL
    for (int f = g; f < F; f += VN_GROUPS) {
        const float o = po[f * 64 + c], d = pd[f * 64 + c], w = w2[f * 64 + c];
        T += d * o;
        Q += o * o;
        DW += d * w;
        OW += o * w;
    }
    T = group_sum(T, red, c, g);
    Q = group_sum(Q, red, c, g);
    DW = group_sum(DW, red, c, g);
    OW = group_sum(OW, red, c, g);
    if (g == 0) s_t[c] = T;
    __syncthreads();
    float S = 0.f;
#pragma unroll
    for (int q = 0; q < 64; ++q) S += s_t[q];
    const float rb = r2[b], rc = r1[b * 64 + c];
    if (rb >= 0.99e6f) S = 0.f;                                   // outer clamp active: no projection term
    const float Sc = (rc >= 0.99e6f) ? 0.f : (T - S * Q);         // inner clamp active: no projection term
    const float inv_rb = 1.0f / rb;
    float* pw = draw + (size_t)b * F * 64;
    const int sl = blockIdx.y;
    const int f0 = sl * (F / VN_SLICES), f1 = sl + 1 == VN_SLICES ? F : f0 + F / VN_SLICES;
#pragma unroll VN_UNROLL
    for (int f = f0 + g; f < f1; f += VN_GROUPS) {
        const float o = po[f * 64 + c], d = pd[f * 64 + c];
        pw[f * 64 + c] = rc * (rb * (d - o * S) - (o * inv_rb) * Sc);
    }
    // d a_sum[c] = -sum_f dv w2 = -rc (rb (DW - S OW) - (Sc / rb) OW)
    if (sl == 0 && g == 0) da_sum[b * 64 + c] = -(rc * (rb * (DW - S * OW) - (inv_rb * Sc) * OW));
}

extern "C" int epc_vlad_normalize_fwd(const float* raw, const float* a_sum, const float* w2, int num_clouds, int F, int C,
                                      float* out, float* r1, float* r2, void* stream) {
    EPC_CHECK_ARG(raw && a_sum && w2 && out && r1 && r2, "null pointer");
    EPC_CHECK_ARG(num_clouds > 0 && F > 0 && C == 64, "cluster_size must be 64");
    hipLaunchKernelGGL(vlad_normalize_fwd_kernel, dim3(num_clouds, VN_SLICES), dim3(VN_THREADS), 0, (hipStream_t)stream, raw, a_sum, w2,
                       F, out, r1, r2);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_vlad_normalize_bwd(const float* dout, const float* out, const float* r1, const float* r2,
                                      const float* w2, int num_clouds, int F, int C, float* draw, float* da_sum,
                                      void* stream) {
    EPC_CHECK_ARG(dout && out && r1 && r2 && w2 && draw && da_sum, "null pointer");
    EPC_CHECK_ARG(num_clouds > 0 && F > 0 && C == 64, "cluster_size must be 64");
    hipLaunchKernelGGL(vlad_normalize_bwd_kernel, dim3(num_clouds, VN_SLICES), dim3(VN_THREADS), 0, (hipStream_t)stream, dout, out, r1,
                       r2, w2, F, draw, da_sum);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// lazy_quadruplet_loss (models/epc-net.py:269-284; best_pos_distance :160-167):
//   best_b = min_p |pos_p - q|^2;  L1_b = max_n max(m1 + best_b - |neg_n - q|^2, 0);
//   L2_b = max_n max(m2 + best_b - |neg_n - other|^2, 0);  loss = mean_b L1_b + mean_b L2_b.
// Descriptors (B, P, D) etc. are a few KB: ONE workgroup, one wave per distance.  The forward records, per tuple, the
// arg-min positive and the two arg-max negatives (-1 when the hinge is inactive) for the backward, which writes all four
// gradients (zeros included) in one launch.  Ties take the lowest index (a measure-zero event for real descriptors).
// ----------------------------------------------------------------------------------------------------------------
#define QL_THREADS 256
#define QL_MAX_VECS 64   // positives / negatives per tuple

__device__ __forceinline__ float sqdist_wave(const float* a, const float* b, int D, int lane) {
    float s = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float t = a[d] - b[d];
        s += t * t;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    return s;
}

__global__ __launch_bounds__(QL_THREADS) void quadruplet_loss_fwd_kernel(const float* __restrict__ q,
                                                                         const float* __restrict__ pos,
                                                                         const float* __restrict__ neg,
                                                                         const float* __restrict__ other, int B, int P,
                                                                         int Nn, int D, float m1, float m2,
                                                                         float* __restrict__ loss,
                                                                         int* __restrict__ sel) {
    __shared__ float dp[QL_MAX_VECS], dq[QL_MAX_VECS], dn[QL_MAX_VECS];
    __shared__ float s_acc[2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_acc[0] = s_acc[1] = 0.f;
    for (int b = 0; b < B; ++b) {
        const float* qb = q + (size_t)b * D;
        const float* ob = other + (size_t)b * D;
        __syncthreads();
        for (int v = wave; v < P + 2 * Nn; v += QL_THREADS / 64) {
            if (v < P) {
                const float d = sqdist_wave(pos + ((size_t)b * P + v) * D, qb, D, lane);
                if (lane == 0) dp[v] = d;
            } else if (v < P + Nn) {
                const float d = sqdist_wave(neg + ((size_t)b * Nn + (v - P)) * D, qb, D, lane);
                if (lane == 0) dq[v - P] = d;
            } else {
                const float d = sqdist_wave(neg + ((size_t)b * Nn + (v - P - Nn)) * D, ob, D, lane);
                if (lane == 0) dn[v - P - Nn] = d;
            }
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int pbest = 0;
            for (int p = 1; p < P; ++p)
                if (dp[p] < dp[pbest]) pbest = p;
            const float best = dp[pbest];
            int n1 = 0, n2 = 0;
            float h1 = fmaxf(m1 + best - dq[0], 0.f), h2 = fmaxf(m2 + best - dn[0], 0.f);
            for (int n = 1; n < Nn; ++n) {
                const float a1 = fmaxf(m1 + best - dq[n], 0.f), a2 = fmaxf(m2 + best - dn[n], 0.f);
                if (a1 > h1) h1 = a1, n1 = n;
                if (a2 > h2) h2 = a2, n2 = n;
            }
            s_acc[0] += h1;
            s_acc[1] += h2;
            sel[3 * b + 0] = pbest;
            sel[3 * b + 1] = h1 > 0.f ? n1 : -1;
            sel[3 * b + 2] = h2 > 0.f ? n2 : -1;
        }
    }
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = s_acc[0] / (float)B + s_acc[1] / (float)B;
}

__global__ __launch_bounds__(QL_THREADS) void quadruplet_loss_bwd_kernel(const float* __restrict__ q,
                                                                         const float* __restrict__ pos,
                                                                         const float* __restrict__ neg,
                                                                         const float* __restrict__ other,
                                                                         const int* __restrict__ sel,
                                                                         const float* __restrict__ dloss, int B, int P,
                                                                         int Nn, int D, float* __restrict__ dq,
                                                                         float* __restrict__ dpos,
                                                                         float* __restrict__ dneg,
                                                                         float* __restrict__ dother) {
    const float w = dloss[0] / (float)B;
    for (int b = 0; b < B; ++b) {
        const int pb = sel[3 * b], n1 = sel[3 * b + 1], n2 = sel[3 * b + 2];
        const float k = (n1 >= 0 ? 1.f : 0.f) + (n2 >= 0 ? 1.f : 0.f);   // how many hinge terms carry best_b
        const float* qb = q + (size_t)b * D;
        const float* ob = other + (size_t)b * D;
        const float* pp = pos + ((size_t)b * P + pb) * D;
        for (int d = threadIdx.x; d < D; d += QL_THREADS) {
            const float gp = 2.f * (pp[d] - qb[d]) * k * w;              // d best / d pos_p*  (and - d best / d q)
            float gq = -gp, go = 0.f;
            if (n1 >= 0) gq += 2.f * (neg[((size_t)b * Nn + n1) * D + d] - qb[d]) * w;       // -|neg - q|^2
            if (n2 >= 0) go = 2.f * (neg[((size_t)b * Nn + n2) * D + d] - ob[d]) * w;        // -|neg - other|^2
            dq[(size_t)b * D + d] = gq;
            dother[(size_t)b * D + d] = go;
            for (int p = 0; p < P; ++p) dpos[((size_t)b * P + p) * D + d] = p == pb ? gp : 0.f;
            for (int n = 0; n < Nn; ++n) {
                float g = 0.f;
                if (n == n1) g -= 2.f * (neg[((size_t)b * Nn + n) * D + d] - qb[d]) * w;
                if (n == n2) g -= 2.f * (neg[((size_t)b * Nn + n) * D + d] - ob[d]) * w;
                dneg[((size_t)b * Nn + n) * D + d] = g;
            }
        }
    }
}

extern "C" int epc_lazy_quadruplet_loss_fwd(const float* q, const float* pos, const float* neg, const float* other, int B,
                                            int P, int Nn, int D, float m1, float m2, float* loss, int32_t* sel,
                                            void* stream) {
    EPC_CHECK_ARG(q && pos && neg && other && loss && sel, "null pointer");
    EPC_CHECK_ARG(B > 0 && D > 0 && P > 0 && Nn > 0 && P <= QL_MAX_VECS && Nn <= QL_MAX_VECS, "bad tuple shape");
    hipLaunchKernelGGL(quadruplet_loss_fwd_kernel, dim3(1), dim3(QL_THREADS), 0, (hipStream_t)stream, q, pos, neg, other, B, P,
                       Nn, D, m1, m2, loss, sel);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_lazy_quadruplet_loss_bwd(const float* q, const float* pos, const float* neg, const float* other,
                                            const int32_t* sel, const float* dloss, int B, int P, int Nn, int D,
                                            float* dq, float* dpos, float* dneg, float* dother, void* stream) {
    EPC_CHECK_ARG(q && pos && neg && other && sel && dloss && dq && dpos && dneg && dother, "null pointer");
    EPC_CHECK_ARG(B > 0 && D > 0 && P > 0 && Nn > 0 && P <= QL_MAX_VECS && Nn <= QL_MAX_VECS, "bad tuple shape");
    hipLaunchKernelGGL(quadruplet_loss_bwd_kernel, dim3(1), dim3(QL_THREADS), 0, (hipStream_t)stream, q, pos, neg, other, sel,
                       dloss, B, P, Nn, D, dq, dpos, dneg, dother);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// Distillation terms (kd_train.py:330-340, 376-383): square_error_sum / square_error_mean of two equally shaped tensors --
// the student's and the teacher's descriptors ("soft labels", rows x 256) and, with GAMMA != 0, their l2-normalised point
// features (rows x 1024: 302 MB each at 18 x 4096 rows).  loss = scale * sum (a - b)^2 (scale = 1, or 1 / n for the mean).
// Forward: ONE read of a and b -- every workgroup reduces a contiguous slice to a partial (lanes, then waves, in a fixed order),
// the finish adds the partials in ascending order in double: bit-reproducible.  Backward: da = (2 scale dloss) (a - b), one pass;
// b (the teacher's output) gets no gradient (kd_train.py feeds it through a placeholder).
// ----------------------------------------------------------------------------------------------------------------
#define SQ_WG_ELEMS (256 * 4 * 16)   // elements per workgroup: 16 float4 per thread

__global__ __launch_bounds__(256) void sq_err_partial_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                             float* __restrict__ partial) {
    __shared__ float red[4];
    const long base = (long)blockIdx.x * SQ_WG_ELEMS;
    float s = 0.f;
#pragma unroll 4
    for (int u = 0; u < 16; ++u) {
        const long o = base + ((long)u * 256 + threadIdx.x) * 4;
        if (o + 3 < n) {
            const float4 x = *reinterpret_cast<const float4*>(a + o), y = *reinterpret_cast<const float4*>(b + o);
            const float d0 = x.x - y.x, d1 = x.y - y.y, d2 = x.z - y.z, d3 = x.w - y.w;
            s += (d0 * d0 + d1 * d1) + (d2 * d2 + d3 * d3);
        } else {
            for (long e = o; e < n && e < o + 4; ++e) {
                const float d = a[e] - b[e];
                s += d * d;
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ __launch_bounds__(256) void sq_err_finish_kernel(const float* __restrict__ partial, int count, float scale,
                                                            float* __restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int p = threadIdx.x; p < count; p += 256) s += (double)partial[p];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] * (double)scale);
}

__global__ __launch_bounds__(256) void sq_err_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b, long n,
                                                         float scale2, const float* __restrict__ dloss,
                                                         float* __restrict__ da) {
    const float w = scale2 * dloss[0];
    const long o = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (o + 3 < n) {
        const float4 x = *reinterpret_cast<const float4*>(a + o), y = *reinterpret_cast<const float4*>(b + o);
        *reinterpret_cast<float4*>(da + o) = make_float4(w * (x.x - y.x), w * (x.y - y.y), w * (x.z - y.z), w * (x.w - y.w));
    } else {
        for (long e = o; e < n && e < o + 4; ++e) da[e] = w * (a[e] - b[e]);
    }
}

extern "C" size_t epc_sq_err_partial_floats(long n) { return n > 0 ? (size_t)((n + SQ_WG_ELEMS - 1) / SQ_WG_ELEMS) : 0; }

extern "C" int epc_sq_err_fwd(const float* a, const float* b, long n, int mean, float* loss, float* partials,
                              size_t partial_floats, void* stream) {
    EPC_CHECK_ARG(a && b && loss && partials, "null pointer");
    EPC_CHECK_ARG(n > 0 && n < (1L << 40), "bad element count");
    EPC_CHECK_ARG(((reinterpret_cast<size_t>(a) | reinterpret_cast<size_t>(b)) & 15) == 0, "a and b must be 16-byte aligned");
    const size_t wgs = epc_sq_err_partial_floats(n);
    EPC_CHECK_ARG(partial_floats >= wgs, "partial buffer too small (epc_sq_err_partial_floats)");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(sq_err_partial_kernel, dim3((unsigned)wgs), dim3(256), 0, st, a, b, n, partials);
    hipLaunchKernelGGL(sq_err_finish_kernel, dim3(1), dim3(256), 0, st, partials, (int)wgs, mean ? 1.0f / (float)n : 1.0f, loss);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_sq_err_bwd(const float* a, const float* b, long n, int mean, const float* dloss, float* da, void* stream) {
    EPC_CHECK_ARG(a && b && dloss && da, "null pointer");
    EPC_CHECK_ARG(n > 0 && n < (1L << 40), "bad element count");
    EPC_CHECK_ARG(((reinterpret_cast<size_t>(a) | reinterpret_cast<size_t>(b) | reinterpret_cast<size_t>(da)) & 15) == 0,
                  "a, b and da must be 16-byte aligned");
    const long quads = (n + 3) / 4;
    hipLaunchKernelGGL(sq_err_bwd_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, b, n,
                       mean ? 2.0f / (float)n : 2.0f, dloss, da);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// The VLAD feature gradient (backward of loupe.py:255-291 with respect to the point features), a product with K = 128 and a
// 302-MB output at the training tuple's size:
//     df[b][n][f] = sum_k a[b][n][k] dvlad[b][f][k]  +  sum_k dz[b][n][k] Wc[f][k]             (rows = B n_points, F = 1024, k < 64)
// The generic tile GEMM took 184 us for it (plus 35 us of torch.cat to build its [a | dz] and [dvlad^T ; Wc^T] operands): 4608
// workgroups of four short k-tiles each, every one re-reading and re-splitting operand tiles other workgroups have split already,
// for an output that takes 67 us to write.  Here the wave's 32 rows of [a | dz] are RESIDENT as bf16 hi + lo fragments (64
// registers, read and split once), the cloud's right operand is packed once per step into fragment order (vlad_df_pack_kernel:
// 512 KB per cloud) and streams through a double-buffered 16-KB LDS chunk of 32 output columns shared by the workgroup's waves (global ->
// registers under the previous chunk's products -> LDS), one barrier per chunk; the wave writes whole 128-byte row segments.
// Measured at 18 x 4096 rows (scripts/time_df.py): 114 us against 160-175 us for torch.cat + the generic product, the same bits.
// Four-wave workgroups (three per CU at 146 registers; the tail form: two per CU, its tiles beyond a whole round cut into column parts --
// tail_split, 168 -> 136 us): eight-wave ones at 128 registers spilled and took 136-165 us.  Arithmetic: two bf16 pieces per operand, three products
// (epc_gemm_f32_fast's); PIECES = 1: one bf16 value per operand.
// ----------------------------------------------------------------------------------------------------------------
#define VDF_WAVES 4
#define VDF_CHUNK_U4 (8 * 2 * 64)    // u32x4 per 32-column chunk: [k-step 8][piece 2][lane 64] = 16 KB
#ifndef VDF_NT
#define VDF_NT 1                      // 32-column chunks per LDS stage (2 = a row's 64 columns leave as one 256-byte run: measured slower, 160 vs 136 us)
#endif

// Bp[b][chunk c][k-step s][piece][lane (i, h)][8 bf16]: value j = B[k = 16 s + 8 h + j][f = 32 c + i], B = [dvlad[b]^T ; Wc^T]
__global__ __launch_bounds__(256) void vlad_df_pack_kernel(const float* __restrict__ dvlad, const float* __restrict__ Wc, int F,
                                                           u32x4* __restrict__ Bp) {
    const int c = blockIdx.x, b = blockIdx.y;
    for (int e = threadIdx.x; e < 8 * 64; e += 256) {
        const int s = e >> 6, l = e & 63, i = l & 31, h = l >> 5;
        const int f = 32 * c + i, k0 = 16 * s + 8 * h;
        const float* src = k0 < 64 ? dvlad + ((size_t)b * F + f) * 64 + k0 : Wc + (size_t)f * 64 + (k0 - 64);
        const float4 x = *reinterpret_cast<const float4*>(src), y = *reinterpret_cast<const float4*>(src + 4);
        const float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
        bf16x8 hi, lo;
        split8(v, hi, lo);
        u32x4* dst = Bp + ((size_t)b * (F / 32) + c) * VDF_CHUNK_U4 + (size_t)s * 128 + l;
        dst[0] = __builtin_bit_cast(u32x4, hi);
        dst[64] = __builtin_bit_cast(u32x4, lo);
    }
}

// TAIL (conv5's features, F = 1024): the gradient does not leave as df but as du, the gradient of conv5's BatchNorm OUTPUT -- the
// backward of f = l2_normalize(relu(bn(z5))) (models/epc-net.py:136-148) applied to the accumulators while they are in registers:
//     u = relu(z5 s + t),  f = u rn,  du = [f > 0] rn (df - f t_row),   t_row = sum_c df f  (the row's dot product)
// together with the two column sums the BatchNorm backward needs (sum du, sum du zhat), as one partial per workgroup.  t_row is
// known BEFORE df exists: sum_c df_c f_c = sum_k a_k (f . dvlad_k) + sum_k dz_k (f . Wc_k) = sum_k (a_k da_k + dz_k zc_k), 64-wide dot
// products of tensors the assignment's backward holds anyway (epc_assign_softmax_bwd's rowdot).  What this removes from the step
// is a whole pass over (df, z5) -- epc_bn_relu_rownorm_bwd's sums kernel, 154 us at 18 x 4096 rows -- for one extra read of z5
// here; the apply pass that follows reads (du, z5) where it read (df, z5).
struct VdfTail {
    const float* z5;     // (rows, F) conv5's pre-activation
    const float* rn;     // (rows) reciprocal row norms of the forward (>= 0.99e6: the clamped zero row, du = df rn)
    const float* trow;   // (rows) t_row
    const float *mean, *var, *gamma, *beta;   // conv5's BatchNorm (F each)
    float eps;
    float* partials;     // [workgroups][2][F]
};

template <int PIECES, bool TAIL>
__global__ __launch_bounds__(64 * VDF_WAVES, TAIL ? 2 : 3) void vlad_df_kernel(const float* __restrict__ a, const float* __restrict__ dz,
                                                                    const u32x4* __restrict__ Bp, int n_points, int F,
                                                                    float* __restrict__ df, VdfTail tail, int whole, int parts) {
    __shared__ u32x4 Bs[2][VDF_NT * VDF_CHUNK_U4];   // 2 x 16 KB
    __shared__ __attribute__((aligned(16))) float coef[TAIL ? 4 : 1][TAIL ? 1024 : 1];   // per column: s, t, mean, rstd
    __shared__ __attribute__((aligned(16))) float rowc[TAIL ? VDF_WAVES : 1][2][32];      // per wave and row: rn, rn * t_row (0 when clamped)
    __shared__ float psum[2][TAIL ? VDF_WAVES : 1][2][32];                               // per stage parity and wave: the two sums of 32 columns
    // (the wave index through readfirstlane: the tile's row base is then a scalar, and the sixteen rows a lane loads or stores per
    // stage are sixteen scalar bases + ONE vector offset instead of sixteen 64-bit vector addresses)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    // Work split (round 5): workgroup = (a 128-row tile of a cloud, a RANGE of the column stages).  The first `whole` tiles (a multiple of
    // the co-resident workgroup count) take every stage; each of the rest is cut into `parts` workgroups (tail_split): the last,
    // partly filled round costs a fraction of a tile's time.  The sums' partials stay one per tile.
    const int tpc = ((n_points + 31) / 32 + VDF_WAVES - 1) / VDF_WAVES;   // 128-row tiles per cloud
    const int wg = (int)blockIdx.x < whole ? (int)blockIdx.x : whole + ((int)blockIdx.x - whole) / parts;
    const int b = wg / tpc;
    const int tile = (wg % tpc) * VDF_WAVES + wave;            // 32-row tile of the cloud
    const int r0 = tile * 32;
    const bool live = r0 < n_points;                            // (a wave past the cloud's end only helps staging)
    const int rl = min(r0 + i, n_points - 1);
    const size_t row = (size_t)b * n_points + rl;
    const int stages = F / (32 * VDF_NT);
    const int st_begin = (int)blockIdx.x < whole ? 0 : (((int)blockIdx.x - whole) % parts) * (stages / parts);
    const int st_end = (int)blockIdx.x < whole ? stages : st_begin + stages / parts;
    const u32x4* src = Bp + (size_t)b * (F / 32) * VDF_CHUNK_U4;
    // a stage's fragments travel global -> registers -> LDS in two steps, so that the loads of stage st + 1 are in flight under the
    // products and stores of stage st (a plain copy loop waits for its loads before the wave's first MFMA: 32 exposed round trips)
    constexpr int PER = VDF_NT * VDF_CHUNK_U4 / (64 * VDF_WAVES);
    static_assert(VDF_NT * VDF_CHUNK_U4 % (64 * VDF_WAVES) == 0, "a stage is a whole number of 16-byte pieces per thread");
    u32x4 pre[PER];
    auto request = [&](int st) {
#pragma unroll
        for (int u = 0; u < PER; ++u) pre[u] = src[(size_t)st * VDF_NT * VDF_CHUNK_U4 + tid + u * 64 * VDF_WAVES];
    };
    auto deposit = [&](int buf) {
#pragma unroll
        for (int u = 0; u < PER; ++u) Bs[buf][tid + u * 64 * VDF_WAVES] = pre[u];
    };
    request(st_begin);   // (while the A rows travel)
    // A[m = row i][k]: lane (i, h) of k-step s holds k = 16 s + 8 h .. + 7: s < 4 from a, s >= 4 from dz
    bf16x8 ah[8], al[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const float* p = (s < 4 ? a : dz) + row * 64 + 16 * (s & 3) + 8 * h;
        const float4 x = *reinterpret_cast<const float4*>(p), y = *reinterpret_cast<const float4*>(p + 4);
        float v[8] = {x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w};
        if (r0 + i >= n_points) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = 0.f;
        }
        if constexpr (PIECES == 2) {
            split8(v, ah[s], al[s]);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) ah[s][j] = (__bf16)v[j];
        }
    }
    if constexpr (TAIL) {
        for (int c = tid; c < F; c += 64 * VDF_WAVES) {
            const float rs = 1.0f / sqrtf(tail.var[c] + tail.eps), sc = rs * tail.gamma[c];
            coef[0][c] = sc, coef[1][c] = tail.beta[c] - tail.mean[c] * sc, coef[2][c] = tail.mean[c], coef[3][c] = rs;
        }
        if (h == 0) {
            const bool in = r0 + i < n_points;
            const float rnv = in ? tail.rn[row] : 0.f, tv = in ? tail.trow[row] : 0.f;
            rowc[wave][0][i] = rnv;
            rowc[wave][1][i] = rnv >= 0.99e6f ? 0.f : rnv * tv;
        }
    }
    deposit(st_begin & 1);
    __syncthreads();
    // rows of the tile as scalar bases: row (r & 3) + 8 (r >> 2) of the tile, + 4 h rows and the column in the vector offset
    const size_t tile_base = ((size_t)b * n_points + r0) * F;   // (scalar: b, r0)
    const unsigned voff = 4u * h * (unsigned)F + i;
    float zn[TAIL ? 16 : 1];
    auto zload = [&](int st) {
        if constexpr (TAIL) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float* zb = tail.z5 + tile_base + (size_t)((r & 3) + 8 * (r >> 2)) * F + 32 * st;
                zn[r] = zb[voff];
            }
        }
    };
    if constexpr (TAIL) {
        if (live) zload(st_begin);
    }
    for (int st = st_begin; st < st_end; ++st) {
        const int buf = st & 1;
        if (st + 1 < st_end) request(st + 1);
        float zv[TAIL ? 16 : 1];
        if constexpr (TAIL) {
            // this stage's z5 values were requested a stage ago (an HBM round trip is longer than a stage's products); request the next
#pragma unroll
            for (int r = 0; r < 16; ++r) zv[r] = zn[r];
            if (live && st + 1 < st_end) zload(st + 1);
            // the previous stage's column sums: four waves -> one partial (the barrier at the end of that stage published them)
            if (st > st_begin && tid < 64) {
                const int q = tid >> 5, c = tid & 31;
                const float v = (psum[buf ^ 1][0][q][c] + psum[buf ^ 1][1][q][c]) + (psum[buf ^ 1][2][q][c] + psum[buf ^ 1][3][q][c]);
                tail.partials[((size_t)wg * 2 + q) * F + 32 * (st - 1) + c] = v;
            }
        }
        __builtin_amdgcn_sched_barrier(0);   // keep the requests AHEAD of this stage's work (hipcc sinks loads to their use otherwise)
        if (live) {
            f32x16 acc[VDF_NT];
#pragma unroll
            for (int nt = 0; nt < VDF_NT; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s)
#pragma unroll
                for (int nt = 0; nt < VDF_NT; ++nt) {
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, Bs[buf][nt * VDF_CHUNK_U4 + s * 128 + lane]);
                    if constexpr (PIECES == 2) {
                        const bf16x8 bl = __builtin_bit_cast(bf16x8, Bs[buf][nt * VDF_CHUNK_U4 + s * 128 + 64 + lane]);
                        acc[nt] = mfma_bf16(al[s], bh, acc[nt]);
                        acc[nt] = mfma_bf16(ah[s], bl, acc[nt]);
                    }
                    acc[nt] = mfma_bf16(ah[s], bh, acc[nt]);
                }
            // D[m = row][n = column]: lane = column, register r = row mfma_row(r, h): 128-byte row segments per store.  (Through a per-wave
            // LDS image as float4 rows -- four store instructions instead of sixteen -- measured 121 against 114 us: not store-issue-bound.)
            if constexpr (TAIL) {
                static_assert(!TAIL || VDF_NT == 1, "the tail epilogue is written for one 32-column chunk per stage");
                const int c = 32 * st + i;
                const float cs = coef[0][c], ct = coef[1][c], mu = coef[2][c], rs = coef[3][c];
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rr = mfma_row(r, h);
                    const float rnv = rowc[wave][0][rr], rt = rowc[wave][1][rr];
                    const float f = fmaxf(zv[r] * cs + ct, 0.f) * rnv;       // bn_value(z, affine), the forward's own expression
                    const float du = rnv * acc[0][r] - rt * f;
                    const float d = f > 0.f ? du : 0.f;
                    s1 += d;
                    s2 += d * ((zv[r] - mu) * rs);
                    acc[0][r] = d;
                }
                s1 += __shfl_xor(s1, 32);
                s2 += __shfl_xor(s2, 32);
                if (h == 0) psum[buf][wave][0][i] = s1, psum[buf][wave][1][i] = s2;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = mfma_row(r, h);
                if (r0 + rr < n_points) {
                    float* ob = df + tile_base + (size_t)((r & 3) + 8 * (r >> 2)) * F + 32 * VDF_NT * st;
#pragma unroll
                    for (int nt = 0; nt < VDF_NT; ++nt) ob[voff + 32 * nt] = acc[nt][r];
                }
            }
        } else if constexpr (TAIL) {
            if (h == 0) psum[buf][wave][0][i] = 0.f, psum[buf][wave][1][i] = 0.f;
        }
        if (st + 1 < st_end) deposit(buf ^ 1);   // (the other buffer: its last readers passed the barrier at the end of stage st - 1)
        __syncthreads();
    }
    if constexpr (TAIL) {
        if (tid < 64) {
            const int q = tid >> 5, c = tid & 31, pb = (st_end - 1) & 1;
            const float v = (psum[pb][0][q][c] + psum[pb][1][q][c]) + (psum[pb][2][q][c] + psum[pb][3][q][c]);
            tail.partials[((size_t)wg * 2 + q) * F + 32 * (st_end - 1) + c] = v;
        }
    }
}

extern "C" size_t epc_vlad_df_packed_bytes(int num_clouds, int F) {
    return num_clouds > 0 && F > 0 ? (size_t)num_clouds * (F / 32) * VDF_CHUNK_U4 * sizeof(u32x4) : 0;
}

extern "C" int epc_vlad_df(const float* a, const float* dz, const float* dvlad, const float* Wc, int num_clouds, int n_points, int F,
                           int pieces, void* packed, size_t packed_bytes, float* df, void* stream) {
    EPC_CHECK_ARG(a && dz && dvlad && Wc && packed && df, "null pointer");
    EPC_CHECK_ARG(num_clouds > 0 && n_points > 0 && F >= 32 * VDF_NT && F % (32 * VDF_NT) == 0 && (pieces == 2 || pieces == 1), "bad shape (64 clusters, F a multiple of 64; pieces: 2 or 1)");
    EPC_CHECK_ARG(num_clouds <= 65535, "too many clouds");
    EPC_CHECK_ARG(packed_bytes >= epc_vlad_df_packed_bytes(num_clouds, F), "packed buffer too small (epc_vlad_df_packed_bytes)");
    EPC_CHECK_ARG(((reinterpret_cast<size_t>(a) | reinterpret_cast<size_t>(dz) | reinterpret_cast<size_t>(dvlad) | reinterpret_cast<size_t>(Wc) |
                    reinterpret_cast<size_t>(packed) | reinterpret_cast<size_t>(df)) & 15) == 0,
                  "tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(vlad_df_pack_kernel, dim3(F / 32, num_clouds), dim3(256), 0, st, dvlad, Wc, F, (u32x4*)packed);
    const int tiles = (n_points + 31) / 32;
    const int wgs = (tiles + VDF_WAVES - 1) / VDF_WAVES * num_clouds;   // (three workgroups per CU: every tile whole)
    const dim3 grid(wgs);
    if (pieces == 2) hipLaunchKernelGGL((vlad_df_kernel<2, false>), grid, dim3(64 * VDF_WAVES), 0, st, a, dz, (const u32x4*)packed, n_points, F, df, VdfTail{}, wgs, 1);
    else hipLaunchKernelGGL((vlad_df_kernel<1, false>), grid, dim3(64 * VDF_WAVES), 0, st, a, dz, (const u32x4*)packed, n_points, F, df, VdfTail{}, wgs, 1);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" size_t epc_vlad_df_tail_partial_floats(int num_clouds, int n_points) {
    if (num_clouds <= 0 || n_points <= 0) return 0;
    const size_t tiles = (n_points + 31) / 32;
    return (size_t)num_clouds * ((tiles + VDF_WAVES - 1) / VDF_WAVES) * 2 * 1024;
}

extern "C" int epc_vlad_df_tail(const float* a, const float* dz, const float* dvlad, const float* Wc, int num_clouds, int n_points,
                                int pieces, void* packed, size_t packed_bytes, const float* z5, const float* rn, const float* trow,
                                const float* mean, const float* var, const float* gamma, const float* beta, float eps, float* du,
                                float* dbeta_dgamma, float* partials, size_t partial_floats, void* stream) {
    constexpr int F = 1024;
    EPC_CHECK_ARG(a && dz && dvlad && Wc && packed && z5 && rn && trow && mean && var && gamma && beta && du && dbeta_dgamma && partials,
                  "null pointer");
    EPC_CHECK_ARG(num_clouds > 0 && num_clouds <= 65535 && n_points > 0 && n_points % 32 == 0 && (pieces == 2 || pieces == 1) &&
                      (long)num_clouds * n_points * F < (1L << 32),
                  "bad shape (n_points a multiple of 32; rows * 1024 < 2^32)");
    EPC_CHECK_ARG(packed_bytes >= epc_vlad_df_packed_bytes(num_clouds, F), "packed buffer too small (epc_vlad_df_packed_bytes)");
    EPC_CHECK_ARG(partial_floats >= epc_vlad_df_tail_partial_floats(num_clouds, n_points), "partials buffer too small");
    EPC_CHECK_ARG(((reinterpret_cast<size_t>(a) | reinterpret_cast<size_t>(dz) | reinterpret_cast<size_t>(dvlad) | reinterpret_cast<size_t>(Wc) |
                    reinterpret_cast<size_t>(packed) | reinterpret_cast<size_t>(du) | reinterpret_cast<size_t>(partials) |
                    reinterpret_cast<size_t>(dbeta_dgamma)) & 15) == 0,
                  "tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(vlad_df_pack_kernel, dim3(F / 32, num_clouds), dim3(256), 0, st, dvlad, Wc, F, (u32x4*)packed);
    const int tiles = (n_points + 31) / 32;
    const int wgs = (tiles + VDF_WAVES - 1) / VDF_WAVES * num_clouds;
    int whole, parts;
    tail_split(wgs, 2 * epc_device_cu_count(), F / (32 * VDF_NT), whole, parts);   // (two workgroups per CU: launch bounds of the tail form)
    const dim3 grid(whole + (wgs - whole) * parts);
    const VdfTail tail{z5, rn, trow, mean, var, gamma, beta, eps, partials};
    if (pieces == 2) hipLaunchKernelGGL((vlad_df_kernel<2, true>), grid, dim3(64 * VDF_WAVES), 0, st, a, dz, (const u32x4*)packed, n_points, F, du, tail, whole, parts);
    else hipLaunchKernelGGL((vlad_df_kernel<1, true>), grid, dim3(64 * VDF_WAVES), 0, st, a, dz, (const u32x4*)packed, n_points, F, du, tail, whole, parts);
    epc_partial_sum_wide_launch(partials, wgs, 2 * F, dbeta_dgamma, stream);   // [0, F): sum du; [F, 2F): sum du zhat
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// EPC-Net-L's global max over a cloud's points in training mode (models/epc-net-l.py:88-92: tf_util.max_pool2d with the kernel covering
// all N points; utils/tf_util.py:349-372): out[b][c] = max_n x[b][n][c] and the row that holds it (the FIRST one on ties: the element
// tf.nn.max_pool's gradient routes to), then dx = dy at that row, zero elsewhere.  A workgroup owns 64 channels of one cloud (256-byte
// row segments), four row phases meet in LDS.  NaN propagates (a NaN entry wins).
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_points_fwd_kernel(const float* __restrict__ x, int n, int C, float* __restrict__ out,
                                                                 int32_t* __restrict__ arg) {
    __shared__ float sv[4][64];
    __shared__ int si[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6, b = blockIdx.y;
    const bool on = c < C;
    const float* p = x + (size_t)b * n * C + (on ? c : 0);
    float best = -INFINITY;
    int bi = 0;
    bool nan = false;
    for (int r = part; r < n; r += 4) {
        const float v = p[(size_t)r * C];
        if (v != v && !nan) nan = true, best = v, bi = r;
        if (!nan && v > best) best = v, bi = r;
    }
    sv[part][threadIdx.x & 63] = best, si[part][threadIdx.x & 63] = bi;
    __syncthreads();
    if (part == 0 && on) {
        const int l = threadIdx.x & 63;
        float v = sv[0][l];
        int idx = si[0][l];
        for (int q = 1; q < 4; ++q) {
            const float w = sv[q][l];
            const int wi = si[q][l];
            const bool vn = v != v, wn = w != w;
            if ((wn && (!vn || wi < idx)) || (!vn && !wn && (w > v || (w == v && wi < idx)))) v = w, idx = wi;
        }
        out[(size_t)b * C + c] = v;
        arg[(size_t)b * C + c] = idx;
    }
}

__global__ __launch_bounds__(256) void maxpool_points_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ arg, int n, int C,
                                                                 long total, float* __restrict__ dx) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;      // one (cloud, channel) per thread
    if (e >= total) return;
    const long b = e / C, c = e % C;
    dx[((size_t)b * n + arg[e]) * C + c] = dy[e];
}

extern "C" int epc_maxpool_points_fwd(const float* x, int num_clouds, int n, int C, float* out, int32_t* arg, void* stream) {
    EPC_CHECK_ARG(x && out && arg, "null pointer");
    EPC_CHECK_ARG(num_clouds > 0 && num_clouds <= 65535 && n > 0 && C > 0, "bad shape");
    hipLaunchKernelGGL(maxpool_points_fwd_kernel, dim3((C + 63) / 64, num_clouds), dim3(256), 0, (hipStream_t)stream, x, n, C, out, arg);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_maxpool_points_bwd(const float* dy, const int32_t* arg, int num_clouds, int n, int C, float* dx, void* stream) {
    EPC_CHECK_ARG(dy && arg && dx, "null pointer");
    EPC_CHECK_ARG(num_clouds > 0 && n > 0 && C > 0, "bad shape");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dx, 0, (size_t)num_clouds * n * C * sizeof(float), st) != hipSuccess) {
        epc_set_error("epc_maxpool_points_bwd: hipMemsetAsync failed");
        return EPC_EHIP;
    }
    const long total = (long)num_clouds * C;
    hipLaunchKernelGGL(maxpool_points_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, dy, arg, n, C, total, dx);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// Two small sums of the VLAD head that were torch compositions (mul + reduce + neg; reshape + reduce):
//   * cluster_weights2's gradient (loupe.py:284,292: vlad - a_sum (x) w2):  dw2[f][c] = - sum_b draw[b][f][c] a_sum[b][c], clouds in
//     ascending order (what epc_vlad_normalize_bwd leaves to its caller: it crosses clouds);
//   * the sum over the G group rows behind the grouped hidden projection (loupe.py:326-328): y[b][o] = sum_g x[b G + g][o], and its
//     backward dx[b G + g][o] = dy[b][o].
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void vlad_w2_grad_kernel(const float* __restrict__ draw, const float* __restrict__ a_sum, int B, long per,
                                                           float* __restrict__ dw2) {
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;      // four consecutive clusters of one feature row
    if (e >= per) return;
    const int c = (int)(e & 63);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int b = 0; b < B; ++b) {
        const float4 d = *reinterpret_cast<const float4*>(draw + (size_t)b * per + e);
        const float4 a = *reinterpret_cast<const float4*>(a_sum + (size_t)b * 64 + c);
        s.x += d.x * a.x, s.y += d.y * a.y, s.z += d.z * a.z, s.w += d.w * a.w;
    }
    *reinterpret_cast<float4*>(dw2 + e) = make_float4(-s.x, -s.y, -s.z, -s.w);
}

__global__ __launch_bounds__(256) void group_sum_kernel(const float* __restrict__ x, int G, int O, long total, int bwd, float* __restrict__ y) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    if (!bwd) {            // e = (b, o)
        const long b = e / O, o = e % O;
        float s = 0.f;
        for (int g = 0; g < G; ++g) s += x[(b * G + g) * O + o];
        y[e] = s;
    } else {               // e = (b G + g, o): x is dy (B, O)
        const long row = e / O, o = e % O;
        y[e] = x[(row / G) * O + o];
    }
}

extern "C" int epc_vlad_w2_grad(const float* draw, const float* a_sum, int num_clouds, int F, int C, float* dw2, void* stream) {
    EPC_CHECK_ARG(draw && a_sum && dw2, "null pointer");
    EPC_CHECK_ARG(num_clouds > 0 && F > 0 && C == 64, "cluster_size must be 64");
    EPC_CHECK_ARG(((reinterpret_cast<size_t>(draw) | reinterpret_cast<size_t>(a_sum) | reinterpret_cast<size_t>(dw2)) & 15) == 0,
                  "tensors must be 16-byte aligned");
    const long per = (long)F * 64;
    hipLaunchKernelGGL(vlad_w2_grad_kernel, dim3((unsigned)((per / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, draw, a_sum, num_clouds,
                       per, dw2);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_group_sum_fwd(const float* x, int rows_out, int G, int O, float* y, void* stream) {
    EPC_CHECK_ARG(x && y && rows_out > 0 && G > 0 && O > 0, "bad argument");
    const long total = (long)rows_out * O;
    hipLaunchKernelGGL(group_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, G, O, total, 0, y);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_group_sum_bwd(const float* dy, int rows_out, int G, int O, float* dx, void* stream) {
    EPC_CHECK_ARG(dy && dx && rows_out > 0 && G > 0 && O > 0, "bad argument");
    const long total = (long)rows_out * G * O;
    hipLaunchKernelGGL(group_sum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dy, G, O, total, 1, dx);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// The VLAD tail behind the hidden projection (loupe.py:323-331 + :61-101) as ONE launch each way -- nine and ten launches of 2-8 us
// on (B G, O) and (B, O) tensors before (column moments + finish + apply, the group sum, the gating product, its BatchNorm's three,
// the gate):
//   y = bn1(h) over the B G rows (training mode: batch mean / population variance);   v[b] = sum_g y[b G + g]        (:323, :326-328)
//   gl = v Wg;   gt = bn2(gl) over the B rows;   out = v * sigmoid(gt)                                                  (:75-100)
// ONE workgroup of 1024 threads.  The column-wise parts: thread (column c = tid % O, slice q = tid / O), every reduction over rows on
// the slices in parallel, meeting in LDS in slice order (bit-reproducible).  The (B, O) x (O, O) products on the matrix pipe: the B <= 32
// rows are ONE 32-row tile in LDS (rows beyond B zero), a wave takes 32-column tiles, operands split into bf16 pieces exactly as the
// per-op GEMMs split them (forward three pieces / six products, backward two / three) -- f32-accurate in BOTH arithmetics of the step:
// products with at most 32 rows stay float32 under "bf16" too (oracle/epcnet_oracle_torch.py: bf16_product_rule; the per-op path ran
// them on plain FMAs).  A first version on scalar FMAs with v broadcast from LDS took 106 / 146 us: one CU's LDS pipe.
// O in {64, 128, 256}; B <= 32.
// ----------------------------------------------------------------------------------------------------------------
#include "train_chain_common.h"
#define HT_THREADS 1024
#define HT_ROWS 32

// sum over the slices of red[slice][c] (slices in order) -> every thread of column c
__device__ __forceinline__ float ht_meet(float part, float* red, int c, int q, int slices, int O) {
    __syncthreads();
    red[q * O + c] = part;
    __syncthreads();
    float t = 0.f;
    for (int s = 0; s < slices; ++s) t += red[s * O + c];
    return t;
}

// out[b][n] (+)= sum_k A[b][k] Bm(k, n) for the 32-row tile A (LDS, row stride O) and an (O, O) global matrix W: TRANS = false: Bm(k, n) =
// W[k][n]; true: Bm(k, n) = W[n][k].  Wave w takes the 32-column tiles w, w + waves, ...; the result lands in `dst` (LDS, row stride O),
// added to what is there when ACCUM.  P bf16 pieces per operand.
template <int P, bool TRANS, bool ACCUM>
__device__ __forceinline__ void ht_product(const float* A, const float* __restrict__ W, int O, float* dst) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = blockDim.x >> 6, i = lane & 31, hh = lane >> 5;
    for (int nt = wave; nt < O / 32; nt += nwaves) {
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll 4
        for (int s = 0; s < O / 16; ++s) {
            float a8[8], b8[8];
            ch_ld8(A + i * O + 16 * s + 8 * hh, a8);
            if (TRANS) {
                ch_ld8(W + (size_t)(32 * nt + i) * O + 16 * s + 8 * hh, b8);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) b8[j] = W[(size_t)(16 * s + 8 * hh + j) * O + 32 * nt + i];
            }
            bf16x8 ap[P], bp[P];
            ch_split<P>(a8, ap), ch_split<P>(b8, bp);
            acc = ch_prod<P>(ap, bp, acc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* d = dst + mfma_row(r, hh) * O + 32 * nt + i;
            *d = ACCUM ? *d + acc[r] : acc[r];
        }
    }
}

template <int P>
__global__ __launch_bounds__(HT_THREADS) void hidden_tail_fwd_kernel(const float* __restrict__ h, int B, int G, int O, const float* __restrict__ gamma1,
                                                                    const float* __restrict__ beta1, const float* __restrict__ Wg,
                                                                    const float* __restrict__ gamma2, const float* __restrict__ beta2, float eps,
                                                                    float bessel1, float bessel2, float* __restrict__ mean1,
                                                                    float* __restrict__ var1, float* __restrict__ var1u, float* __restrict__ v_out,
                                                                    float* __restrict__ gl_out, float* __restrict__ mean2, float* __restrict__ var2,
                                                                    float* __restrict__ var2u, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // v[32][O] | gl[32][O] | red[slices][O]
    const int tid = threadIdx.x, c = tid % O, q = tid / O, slices = HT_THREADS / O, R = B * G;
    float* vs = lds;
    float* acc = lds + (size_t)HT_ROWS * O;
    float* red = acc + (size_t)HT_ROWS * O;
    // ---- bn1: batch moments of column c over the R rows (two passes: mean, then the centred second moment) ----
    // (holding a slice's rows of h in registers across the three passes was built and measured: 24.8 us against 22.5 -- not kept)
    float p = 0.f;
#pragma unroll 8
    for (int r = q; r < R; r += slices) p += h[(size_t)r * O + c];   // (unrolled: a row per iteration would be one dependent L2 round trip each)
    const float m1 = ht_meet(p, red, c, q, slices, O) / (float)R;
    p = 0.f;
#pragma unroll 8
    for (int r = q; r < R; r += slices) {
        const float d = h[(size_t)r * O + c] - m1;
        p += d * d;
    }
    const float v1 = ht_meet(p, red, c, q, slices, O) / (float)R;
    const float s1 = (1.0f / sqrtf(v1 + eps)) * gamma1[c], t1 = beta1[c] - m1 * s1;   // y = z s + t (bn_affine's expression)
    if (q == 0) mean1[c] = m1, var1[c] = v1, var1u[c] = v1 * bessel1;   // (the fused slim op feeds the Bessel-corrected variance to the moving average)
    // ---- the group sums (rows beyond B: zeros -- the products' tile is 32 rows) ----
#pragma unroll 2
    for (int b = q; b < HT_ROWS; b += slices) {
        float t = 0.f;
        if (b < B) {
#pragma unroll 4
            for (int g = 0; g < G; ++g) t += h[(size_t)(b * G + g) * O + c] * s1 + t1;
            v_out[(size_t)b * O + c] = t;
        }
        vs[b * O + c] = t;
    }
    __syncthreads();
    ht_product<P, false, false>(vs, Wg, O, acc);   // gl = v Wg
    __syncthreads();
    // ---- bn2 over the B rows, the gate ----
    p = 0.f;
    for (int b = q; b < B; b += slices) p += acc[b * O + c];
    const float m2 = ht_meet(p, red, c, q, slices, O) / (float)B;
    p = 0.f;
    for (int b = q; b < B; b += slices) {
        const float d = acc[b * O + c] - m2;
        p += d * d;
    }
    const float v2 = ht_meet(p, red, c, q, slices, O) / (float)B;
    const float s2 = (1.0f / sqrtf(v2 + eps)) * gamma2[c], t2 = beta2[c] - m2 * s2;
    if (q == 0) mean2[c] = m2, var2[c] = v2, var2u[c] = v2 * bessel2;
    for (int b = q; b < B; b += slices) {
        const float gl = acc[b * O + c], gt = gl * s2 + t2;
        gl_out[(size_t)b * O + c] = gl;
        out[(size_t)b * O + c] = vs[b * O + c] * (1.f / (1.f + expf(-gt)));
    }
}

// Backward: from dout (B, O) and the forward's h, mean1, var1, v, gl, mean2, var2:
//   s = sigmoid(bn2(gl));  dv = dout s;  dgt = dout v s (1 - s);  bn2': dgl = gamma2 rstd2 (dgt - dbeta2 / B - gt^ dgamma2 / B);
//   dWg = v^T dgl;  dv += dgl Wg^T;  dy[b G + g] = dv[b];  bn1': dh = gamma1 rstd1 (dy - dbeta1 / R - z^ dgamma1 / R)
template <int P>
__global__ __launch_bounds__(HT_THREADS) void hidden_tail_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ h, int B, int G, int O,
                                                                    const float* __restrict__ gamma1, const float* __restrict__ mean1,
                                                                    const float* __restrict__ var1, const float* __restrict__ v,
                                                                    const float* __restrict__ gl, const float* __restrict__ Wg,
                                                                    const float* __restrict__ gamma2, const float* __restrict__ beta2,
                                                                    const float* __restrict__ mean2, const float* __restrict__ var2, float eps,
                                                                    float* __restrict__ dh, float* __restrict__ dgamma1,
                                                                    float* __restrict__ dbeta1, float* __restrict__ dWg, float* __restrict__ dgamma2,
                                                                    float* __restrict__ dbeta2) {
    extern __shared__ __attribute__((aligned(16))) float lds[];   // vs[32][O] | dgl[32][O] | dv[32][O] | red[slices][O]
    const int tid = threadIdx.x, c = tid % O, q = tid / O, slices = HT_THREADS / O, R = B * G;
    float* vs = lds;
    float* dgls = lds + (size_t)HT_ROWS * O;
    float* dvs = dgls + (size_t)HT_ROWS * O;
    float* red = dvs + (size_t)HT_ROWS * O;
    const float rs2 = 1.0f / sqrtf(var2[c] + eps), m2 = mean2[c], s2 = rs2 * gamma2[c], t2 = beta2[c] - m2 * s2;
    // ---- the gate's backward; bn2's two sums (rows beyond B: zeros) ----
    float pb = 0.f, pg = 0.f;
#pragma unroll 4
    for (int b = q; b < HT_ROWS; b += slices) {
        float vv = 0.f, dvv = 0.f, dgt = 0.f;
        if (b < B) {
            const float glv = gl[(size_t)b * O + c], gt = glv * s2 + t2, s = 1.f / (1.f + expf(-gt)), d = dout[(size_t)b * O + c];
            vv = v[(size_t)b * O + c];
            dvv = d * s;
            dgt = d * vv * (s * (1.f - s));
            pb += dgt, pg += dgt * ((glv - m2) * rs2);
        }
        vs[b * O + c] = vv, dvs[b * O + c] = dvv, dgls[b * O + c] = dgt;   // (dgt for now)
    }
    const float db2 = ht_meet(pb, red, c, q, slices, O), dg2 = ht_meet(pg, red, c, q, slices, O);
    if (q == 0) dbeta2[c] = db2, dgamma2[c] = dg2;
#pragma unroll 4
    for (int b = q; b < B; b += slices) {
        const float zh = (gl[(size_t)b * O + c] - m2) * rs2;
        dgls[b * O + c] = s2 * (dgls[b * O + c] - db2 / (float)B - zh * (dg2 / (float)B));
    }
    __syncthreads();
    // ---- dWg = v^T dgl: 32 x 32 output tiles over the waves, the 32 rows are two k-steps ----
    {
        const int lane = tid & 63, wave = tid >> 6, nwaves = HT_THREADS / 64, i = lane & 31, hh = lane >> 5, T = O / 32;
        for (int t = wave; t < T * T; t += nwaves) {
            const int mt = t / T, nt = t % T;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                float a8[8], b8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) a8[j] = vs[(16 * s + 8 * hh + j) * O + 32 * mt + i], b8[j] = dgls[(16 * s + 8 * hh + j) * O + 32 * nt + i];
                bf16x8 ap[P], bp[P];
                ch_split<P>(a8, ap), ch_split<P>(b8, bp);
                acc = ch_prod<P>(ap, bp, acc);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) dWg[(size_t)(32 * mt + mfma_row(r, hh)) * O + 32 * nt + i] = acc[r];
        }
    }
    ht_product<P, true, true>(dgls, Wg, O, dvs);   // dv += dgl Wg^T
    __syncthreads();
    // ---- the group sum's transpose and bn1's backward over the R rows ----
    const float rs1 = 1.0f / sqrtf(var1[c] + eps), m1 = mean1[c], k1g = rs1 * gamma1[c];
    pb = 0.f, pg = 0.f;
#pragma unroll 8
    for (int r = q; r < R; r += slices) {
        const float dy = dvs[(r / G) * O + c];
        pb += dy, pg += dy * ((h[(size_t)r * O + c] - m1) * rs1);
    }
    const float db1 = ht_meet(pb, red, c, q, slices, O), dg1 = ht_meet(pg, red, c, q, slices, O);
    if (q == 0) dbeta1[c] = db1, dgamma1[c] = dg1;
#pragma unroll 8
    for (int r = q; r < R; r += slices) {
        const float dy = dvs[(r / G) * O + c], zh = (h[(size_t)r * O + c] - m1) * rs1;
        dh[(size_t)r * O + c] = k1g * (dy - db1 / (float)R - zh * (dg1 / (float)R));
    }
}

static bool ht_shape_ok(int B, int G, int O) { return B > 0 && B <= HT_ROWS && G > 0 && (O == 64 || O == 128 || O == 256); }

extern "C" int epc_hidden_tail_ok(int B, int G, int O) { return ht_shape_ok(B, G, O) ? 1 : 0; }

extern "C" int epc_hidden_tail_fwd(const float* h, int B, int G, int O, const float* gamma1, const float* beta1, const float* Wg,
                                   const float* gamma2, const float* beta2, float eps, float bessel1, float bessel2, float* mean1,
                                   float* var1, float* var1u, float* v, float* gl, float* mean2, float* var2, float* var2u, float* out,
                                   void* stream) {
    EPC_CHECK_ARG(h && gamma1 && beta1 && Wg && gamma2 && beta2 && mean1 && var1 && var1u && v && gl && mean2 && var2 && var2u && out, "null pointer");
    EPC_CHECK_ARG(ht_shape_ok(B, G, O), "shape not covered (epc_hidden_tail_ok)");
    const size_t lds = ((size_t)2 * HT_ROWS * O + (size_t)(HT_THREADS / O) * O) * sizeof(float);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(hidden_tail_fwd_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        epc_set_error("%s: hipFuncSetAttribute failed", __func__);
        return EPC_EHIP;
    }
    hipLaunchKernelGGL(hidden_tail_fwd_kernel<3>, dim3(1), dim3(HT_THREADS), lds, (hipStream_t)stream, h, B, G, O, gamma1, beta1, Wg, gamma2, beta2, eps,
                       bessel1, bessel2, mean1, var1, var1u, v, gl, mean2, var2, var2u, out);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_hidden_tail_bwd(const float* dout, const float* h, int B, int G, int O, const float* gamma1, const float* mean1, const float* var1,
                                   const float* v, const float* gl, const float* Wg, const float* gamma2, const float* beta2, const float* mean2,
                                   const float* var2, float eps, float* dh, float* dgamma1, float* dbeta1, float* dWg,
                                   float* dgamma2, float* dbeta2, void* stream) {
    EPC_CHECK_ARG(dout && h && gamma1 && mean1 && var1 && v && gl && Wg && gamma2 && beta2 && mean2 && var2 && dh && dgamma1 && dbeta1 && dWg &&
                      dgamma2 && dbeta2,
                  "null pointer");
    EPC_CHECK_ARG(ht_shape_ok(B, G, O), "shape not covered (epc_hidden_tail_ok)");
    EPC_CHECK_ARG((reinterpret_cast<size_t>(Wg) & 15) == 0, "Wg must be 16-byte aligned");
    const size_t lds = ((size_t)3 * HT_ROWS * O + (size_t)(HT_THREADS / O) * O) * sizeof(float);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(hidden_tail_bwd_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
        epc_set_error("%s: hipFuncSetAttribute failed", __func__);
        return EPC_EHIP;
    }
    hipLaunchKernelGGL(hidden_tail_bwd_kernel<2>, dim3(1), dim3(HT_THREADS), lds, (hipStream_t)stream, dout, h, B, G, O, gamma1, mean1, var1, v, gl, Wg,
                       gamma2, beta2, mean2, var2, eps, dh, dgamma1, dbeta1, dWg, dgamma2, dbeta2);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}
