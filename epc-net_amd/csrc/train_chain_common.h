// Device helpers shared by the launch chain (train_chain.hip) and the persistent chain (train_chain_persist.hip): tile algebra,
// the pooled training-mode BatchNorm, the moment / sum partials.  See train_chain.hip's header for the design.
#pragma once
#include "common.h"

// Geometry.  A workgroup takes `wg_rows` consecutive rows (= one partial), a multiple of 32; its waves take the 32-row tiles
// wave, wave + nw, ... .  wg_rows is chosen on the host so that the grid is ONE workgroup per CU whenever the rows allow it
// (epc_chain_wg_rows: tiles / CUs rounded up -- 9 tiles for the 73 728 rows of an 18 x 4096 tuple on 256 CUs, 11 for 22 clouds):
// fixed 256-row workgroups were 288 on 256 CUs, i.e. 32 CUs with twice the work of the others, and these kernels are one dependent
// chain per wave -- the kernel took as long as its busiest CU.  One tile per wave where the registers allow (forward: up to 12 waves),
// otherwise strided tiles (backward: up to 8 waves).
#define CH_MAX_TILES 16      // tiles per workgroup (wg_rows <= 512)
#define CH_FWD_MAX_WAVES 8     // the row-streaming forward layer: two waves per SIMD (at three its weight fragments spill)
#define CH_GATHER_MAX_WAVES 12 // the gather layer: three waves per SIMD, one tile each -- bytes in flight are what it is short of
#define CH_BWD_MAX_WAVES 8

typedef short ch_s16x4 __attribute__((ext_vector_type(4)));
typedef short ch_s16x8 __attribute__((ext_vector_type(8)));

struct ChBnAffine {
    float s, t;
};
// y = z * s + t: the expression the forward and every recomputed ReLU mask share bit for bit (bn_affine of train_ops.hip)
__device__ __forceinline__ ChBnAffine ch_bn_affine(float mean, float var, float gamma, float beta, float eps) {
    ChBnAffine a;
    a.s = (1.0f / sqrtf(var + eps)) * gamma;
    a.t = beta - mean * a.s;
    return a;
}

template <int P>
__device__ __forceinline__ void ch_split(const float (&v)[8], bf16x8 (&p)[P]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        p[0][j] = (__bf16)v[j];
        if constexpr (P >= 2) {
            const float r1 = v[j] - (float)p[0][j];
            p[1][j] = (__bf16)r1;
            if constexpr (P >= 3) p[2][j] = (__bf16)(r1 - (float)p[1][j]);
        }
    }
}
// a (P pieces) times b (P pieces), smallest terms first: six products for P = 3, three for P = 2, one for P = 1
template <int P>
__device__ __forceinline__ f32x16 ch_prod(const bf16x8 (&a)[P], const bf16x8 (&b)[P], f32x16 acc) {
    if constexpr (P == 3) {
        acc = mfma_bf16(a[2], b[0], acc);
        acc = mfma_bf16(a[0], b[2], acc);
        acc = mfma_bf16(a[1], b[1], acc);
    }
    if constexpr (P >= 2) {
        acc = mfma_bf16(a[1], b[0], acc);
        acc = mfma_bf16(a[0], b[1], acc);
    }
    return mfma_bf16(a[0], b[0], acc);
}

__device__ __forceinline__ void ch_ld8(const float* p, float (&v)[8]) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
}
__device__ __forceinline__ void ch_st8(float* p, const float (&v)[8]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

// ---- a training-mode BatchNorm whose batch moments come from a producer's partials (or are given) -----------------------
struct ChBn {
    const float* stats;   // partials [parts][3][64]: sums of (v - p), (v - p)^2 and the pivot p, of the product WITHOUT `bias`; null: mean / var given
    int parts;
    int part_rows;        // rows per partial (the last one shorter)
    const float* bias;    // added to the pooled mean (may be null)
    float* mean;          // pooled: written by workgroup 0;  given: read
    float* var;
    const float* gamma;
    const float* beta;
};

// Pools the partials' moments: thread (column quad cq = tid & 15, slice ps = tid >> 4) merges partials ps, ps + 16, ... around the
// pivot of its first one, the 16 slices are rebased onto slice 0's pivot and merged in order -- double precision throughout:
//   sum (v - p0) = sum (v - pw) + n (pw - p0);   sum (v - p0)^2 = sum (v - pw)^2 + 2 (pw - p0) sum (v - pw) + n (pw - p0)^2
// mean = p0 + A / rows (+ bias), population variance = B / rows - (A / rows)^2 (tf.nn.moments).  Every workgroup computes the same
// bits.  In two halves so that the caller can put its own row loads in between: ch_bn_begin requests a thread's partials in as few
// round trips as registers allow (CH_POOL_CHUNK per trip), folds them and parks the slice in `scratch` (4 * 16 * 64 doubles);
// ch_bn_finish (after the caller has ISSUED its loads) merges the slices and leaves the layer's s, t in coef[0..1][64] (LDS).  A
// pooled prologue costs ~7 us of dependent steps when written naively (scripts/time_chain.py): the affine parameters are requested
// first, the partials in one or two trips, and the rows travel under the merge.
#define CH_POOL_CHUNK 8    // moment partials per slice and round trip (24 float4 in flight per thread): 256 partials in two trips
struct ChBnRegs {          // per-thread state between the two halves (threads 0..63: the column's affine parameters / given moments)
    float gamma, beta, bias, mean, var;
};
__device__ __forceinline__ ChBnRegs ch_bn_begin(const ChBn& bn, int rows, double* scratch) {
    // 16 slices when the workgroup has 256 threads or more (the first 256 pool), fewer in the small grids of short inputs
    const int tid = threadIdx.x, cq = tid & 15, ps = tid >> 4;
    const int nslices = min((int)blockDim.x >> 4, 16);
    ChBnRegs r;
    r.gamma = r.beta = r.bias = r.mean = r.var = 0.f;
    if (tid < 64) {
        r.gamma = bn.gamma[tid], r.beta = bn.beta[tid];
        if (bn.bias) r.bias = bn.bias[tid];
        if (!bn.stats) r.mean = bn.mean[tid], r.var = bn.var[tid];
    }
    if (!bn.stats || ps >= nslices) return r;
    double N[4] = {0, 0, 0, 0}, A[4] = {0, 0, 0, 0}, B[4] = {0, 0, 0, 0}, PL[4] = {0, 0, 0, 0};
    bool first = true;
    for (int t0 = ps; t0 < bn.parts; t0 += nslices * CH_POOL_CHUNK) {
        float4 v1[CH_POOL_CHUNK], v2[CH_POOL_CHUNK], vp[CH_POOL_CHUNK];
#pragma unroll
        for (int u = 0; u < CH_POOL_CHUNK; ++u) {
            const int t = min(t0 + nslices * u, bn.parts - 1);
            const float* p = bn.stats + (size_t)t * 192 + 4 * cq;
#ifdef CH_ABL_NOLOAD
            v1[u] = v2[u] = vp[u] = make_float4((float)t, 1.f, 2.f, (float)cq);
#else
            v1[u] = *reinterpret_cast<const float4*>(p), v2[u] = *reinterpret_cast<const float4*>(p + 64);
            vp[u] = *reinterpret_cast<const float4*>(p + 128);
#endif
        }
#pragma unroll
        for (int u = 0; u < CH_POOL_CHUNK; ++u) {
            const int t = t0 + nslices * u;
            if (t < bn.parts) {
                const double nt = (double)min(bn.part_rows, rows - t * bn.part_rows);
                const float s1[4] = {v1[u].x, v1[u].y, v1[u].z, v1[u].w}, s2[4] = {v2[u].x, v2[u].y, v2[u].z, v2[u].w},
                            pv[4] = {vp[u].x, vp[u].y, vp[u].z, vp[u].w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
#ifdef CH_ABL_NOMATH
                    N[q] += nt, A[q] += (double)(s1[q] + s2[q] + pv[q]);
#else
                    if (first) PL[q] = (double)pv[q];
                    const double d = (double)pv[q] - PL[q];
                    N[q] += nt;
                    A[q] += (double)s1[q] + nt * d;
                    B[q] += (double)s2[q] + 2.0 * d * (double)s1[q] + nt * d * d;
#endif
                }
                first = false;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        scratch[(0 * 16 + ps) * 64 + 4 * cq + q] = N[q];
        scratch[(1 * 16 + ps) * 64 + 4 * cq + q] = A[q];
        scratch[(2 * 16 + ps) * 64 + 4 * cq + q] = B[q];
        scratch[(3 * 16 + ps) * 64 + 4 * cq + q] = PL[q];
    }
    return r;
}
// second half: merge (when pooled), the moments to s_mean / s_var (LDS) and -- workgroup 0 -- to bn.mean / bn.var, then s, t of the
// BatchNorm (+ReLU) applied to an operand as it is loaded into coef[0..1][64].  Ends with a barrier.
__device__ __forceinline__ void ch_bn_finish(const ChBn& bn, const ChBnRegs& r, int rows, float eps, const double* scratch,
                                             float* s_mean, float* s_var, float (*coef)[64]) {
    const int tid = threadIdx.x;
    __syncthreads();   // the slices are parked (and whatever else the caller wrote to LDS before: its weight fragments)
    if (tid < 64) {
        float mean = r.mean, var = r.var;
        if (bn.stats) {
            const double p0 = scratch[(3 * 16 + 0) * 64 + tid];
            double a = scratch[(1 * 16 + 0) * 64 + tid], b = scratch[(2 * 16 + 0) * 64 + tid];
            const int nslices = min((int)blockDim.x >> 4, 16);
            for (int s = 1; s < nslices; ++s) {
                const double n = scratch[(0 * 16 + s) * 64 + tid];
                if (n > 0.0) {
                    const double as = scratch[(1 * 16 + s) * 64 + tid], bs = scratch[(2 * 16 + s) * 64 + tid];
                    const double d = scratch[(3 * 16 + s) * 64 + tid] - p0;
                    a += as + n * d;
                    b += bs + 2.0 * d * as + n * d * d;
                }
            }
            const double m1 = a / (double)rows;
            mean = (float)(p0 + m1 + (double)r.bias);
            var = (float)fmax(b / (double)rows - m1 * m1, 0.0);
            if (blockIdx.x == 0) bn.mean[tid] = mean, bn.var[tid] = var;
        }
        s_mean[tid] = mean, s_var[tid] = var;
        const ChBnAffine a2 = ch_bn_affine(mean, var, r.gamma, r.beta, eps);
        coef[0][tid] = a2.s, coef[1][tid] = a2.t;
    }
    __syncthreads();
}

#define CH_POOL_CHUNK_S 16 // sum partials per slice and round trip (32 float4 in flight per thread: ONE round trip for 256 partials)
// The two column sums of a BatchNorm backward (sum dy [mask], sum dy [mask] zhat) pooled from a producer's partials
// [parts][2][64]: thread (cq, ps) adds partials ps, ps + 16, ... in double, the slices meet in order.  scratch: 2 * 16 * 64
// doubles.  Results in s_sum[2][64] (LDS); workgroup 0 also writes dbeta = sum 0, dgamma = sum 1.  Ends with a barrier.
__device__ __forceinline__ void ch_pool_sums(const float* __restrict__ psums, int parts, double* scratch, float (*s_sum)[64],
                                             float* dbeta, float* dgamma) {
    const int tid = threadIdx.x, cq = tid & 15, ps = tid >> 4;
    const int nslices = min((int)blockDim.x >> 4, 16);
    double a[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
    for (int t0 = ps; ps < nslices && t0 < parts; t0 += nslices * CH_POOL_CHUNK_S) {
        float4 v[CH_POOL_CHUNK_S][2];
#pragma unroll
        for (int u = 0; u < CH_POOL_CHUNK_S; ++u) {
            const int t = min(t0 + nslices * u, parts - 1);
            const float* p = psums + (size_t)t * 128 + 4 * cq;
            v[u][0] = *reinterpret_cast<const float4*>(p), v[u][1] = *reinterpret_cast<const float4*>(p + 64);
        }
#pragma unroll
        for (int u = 0; u < CH_POOL_CHUNK_S; ++u)
            if (t0 + nslices * u < parts) {
#pragma unroll
                for (int k = 0; k < 2; ++k)
                    a[k][0] += (double)v[u][k].x, a[k][1] += (double)v[u][k].y, a[k][2] += (double)v[u][k].z, a[k][3] += (double)v[u][k].w;
            }
    }
    if (ps < nslices) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q) scratch[(k * 16 + ps) * 64 + 4 * cq + q] = a[k][q];
    }
    __syncthreads();
    for (int o = tid; o < 128; o += blockDim.x) {   // (a 64-thread workgroup of a short input takes both sums in turn)
        const int k = o >> 6, c = o & 63;
        double t = 0.0;
        for (int s = 0; s < nslices; ++s) t += scratch[(k * 16 + s) * 64 + c];
        s_sum[k][c] = (float)t;
        if (blockIdx.x == 0) (k ? dgamma : dbeta)[c] = (float)t;
    }
    __syncthreads();
}

// The moment partial of a workgroup from its waves' (sum, sum of squares, pivot, rows): wave 0 rebases the others onto its own pivot
// in wave order (fixed order: registers, lane halves, waves) and stores [3][64] at `out`.
//   sum (v - p0) = sum (v - pw) + n (pw - p0),  sum (v - p0)^2 = sum (v - pw)^2 + 2 (pw - p0) sum (v - pw) + n (pw - p0)^2
// sred: [waves][3][64], snrows: [waves].  Called by every thread; contains the barrier.
__device__ __forceinline__ void ch_store_stats(float (&s1)[2], float (&s2)[2], const float (&piv)[2], int my_rows, float (*sred)[3][64],
                                               int* snrows, float* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        s1[nt] += __shfl_xor(s1[nt], 32);
        s2[nt] += __shfl_xor(s2[nt], 32);
    }
    if (wave > 0 && h == 0) {
        sred[wave][0][i] = s1[0], sred[wave][0][32 + i] = s1[1];
        sred[wave][1][i] = s2[0], sred[wave][1][32 + i] = s2[1];
        sred[wave][2][i] = piv[0], sred[wave][2][32 + i] = piv[1];
    }
    if (lane == 0) snrows[wave] = my_rows;
    __syncthreads();
    if (wave == 0 && h == 0) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int c = 32 * nt + i;
            float t1 = s1[nt], t2 = s2[nt];
            for (int w = 1; w < nw; ++w) {
                const float n_w = (float)snrows[w];
                if (n_w > 0.f) {
                    const float dp = sred[w][2][c] - piv[nt];
                    t1 += sred[w][0][c] + n_w * dp;
                    t2 += sred[w][1][c] + (2.0f * dp) * sred[w][0][c] + n_w * dp * dp;
                }
            }
            out[0 * 64 + c] = t1, out[1 * 64 + c] = t2, out[2 * 64 + c] = piv[nt];
        }
    }
}

#define CH_STG_STRIDE 68   // floats per row of the staging tile (272 B: conflict-free float4 rows both ways)
#define CH_STG_FLOATS (32 * CH_STG_STRIDE)
struct ChBnGiven {   // a BatchNorm with known batch moments (null mean: none)
    const float *mean, *var, *gamma, *beta;
};
#define CH_IMG_BYTES 4096   // one piece of the transposition image: 32 rows x 128 B
// The producer's column sums go through a small per-wave LDS tile, eight channels at a time (64 per-lane accumulators -- a lane
// holds ONE row's 32 channels -- would cost the kernel its second wave per SIMD): word of (quantity q, row, channel ch) below;
// a wave's 32-lane half reads 32 distinct banks (lane = (row group g, q, ch): bank = 8 q + 16 g + 8 r + ch mod 32).
#define CH_SUMT_WORDS 640
__device__ __forceinline__ int ch_sumt_word(int qn, int row, int ch) { return qn * 328 + row * 8 + (row >> 3) * 16 + ch; }
__device__ __forceinline__ int ch_img_off(int row, int chunk) {   // byte offset of 16-byte chunk `chunk` (0..7) of row `row`
    return 128 * row + 16 * (chunk ^ (((row >> 1) & 1) << 2) ^ (((row >> 2) & 1) << 1));
}
// the A / B fragment (k = rows 16 s2 + 8 h .. + 7, m or n = channel 32 t + (lane & 31)) of one piece, read transposed
__device__ __forceinline__ bf16x8 ch_tr_frag(const char* img, int t, int s2, int lane) {
    const int g16 = lane >> 4, l16 = lane & 15, qq = l16 >> 2, pp = l16 & 3;
    const int chunk = 4 * t + 2 * (g16 & 1) + (pp >> 1);
    const int r0 = 16 * s2 + 8 * (g16 >> 1);
    typedef __attribute__((address_space(3))) ch_s16x4* lds_ptr;
    const ch_s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + ch_img_off(r0 + qq, chunk) + 8 * (pp & 1)));
    const ch_s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + ch_img_off(r0 + 4 + qq, chunk) + 8 * (pp & 1)));
    const ch_s16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(bf16x8, v);
}
