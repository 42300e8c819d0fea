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
#define L2_EPS 1e-12f

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
    const int b = blockIdx.x, c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const float* pr = raw + (size_t)b * F * 64;
    const float as = a_sum[b * 64 + c];
    float ss = 0.f;
#pragma unroll 8
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
#pragma unroll 8
    for (int f = g; f < F; f += VN_GROUPS) {
        const float v = pr[f * 64 + c] - as * w2[f * 64 + c];
        po[f * 64 + c] = (v * rc) * rb;
    }
    if (g == 0) r1[b * 64 + c] = rc;
    if (threadIdx.x == 0) r2[b] = rb;
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
    float T = 0.f, Q = 0.f;
#pragma unroll 8
    for (int f = g; f < F; f += VN_GROUPS) {
        const float o = po[f * 64 + c], d = pd[f * 64 + c];
        T += d * o;
        Q += o * o;
    }
    T = group_sum(T, red, c, g);
    Q = group_sum(Q, red, c, g);
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
    float da = 0.f;
#pragma unroll 8
    for (int f = g; f < F; f += VN_GROUPS) {
        const float o = po[f * 64 + c], d = pd[f * 64 + c];
        const float dv = rc * (rb * (d - o * S) - (o * inv_rb) * Sc);
        pw[f * 64 + c] = dv;
        da -= dv * w2[f * 64 + c];
    }
    da = group_sum(da, red, c, g);
    if (g == 0) da_sum[b * 64 + c] = da;
}

extern "C" int epc_vlad_normalize_fwd(const float* raw, const float* a_sum, const float* w2, int num_clouds, int F, int C,
                                      float* out, float* r1, float* r2, void* stream) {
    EPC_CHECK_ARG(raw && a_sum && w2 && out && r1 && r2, "null pointer");
    EPC_CHECK_ARG(num_clouds > 0 && F > 0 && C == 64, "cluster_size must be 64");
    hipLaunchKernelGGL(vlad_normalize_fwd_kernel, dim3(num_clouds), dim3(VN_THREADS), 0, (hipStream_t)stream, raw, a_sum, w2,
                       F, out, r1, r2);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_vlad_normalize_bwd(const float* dout, const float* out, const float* r1, const float* r2,
                                      const float* w2, int num_clouds, int F, int C, float* draw, float* da_sum,
                                      void* stream) {
    EPC_CHECK_ARG(dout && out && r1 && r2 && w2 && draw && da_sum, "null pointer");
    EPC_CHECK_ARG(num_clouds > 0 && F > 0 && C == 64, "cluster_size must be 64");
    hipLaunchKernelGGL(vlad_normalize_bwd_kernel, dim3(num_clouds), dim3(VN_THREADS), 0, (hipStream_t)stream, dout, out, r1,
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
