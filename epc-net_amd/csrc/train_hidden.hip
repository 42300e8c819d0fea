// The grouped hidden projection of the VLAD head in the training step (loupe.py:302-322: vlad (B G, C F / G) @ hidden1_weights (C F / G, O),
// B G = 72 or 88 rows, C F / G = 16 384, O = 256) and its two gradients -- three SKINNY products around a 16-MB weight matrix that the
// generic tile GEMM (train_ops.hip) ran as 3 x 4 tiles x split-K slices: 34 + 10 us forward, 16 + 17 us backward at 18 clouds for products
// whose bytes are 3 us each.  Here every product is ONE pass over the weight matrix (or over dW) by 256 workgroups of four waves:
//
//   forward   Y  (M, 256)  = X (M, K) W (K, 256)       hp_fwd_kernel: workgroup = 256 consecutive k x 64 columns; X's slice (M rows of 1 KB)
//                                                       lands in LDS by LDS-DMA, W's 64-KB block streams into registers (all loads in flight
//                                                       at once: one HBM round trip), a wave takes 64 of the 256 k; the four waves meet in
//                                                       LDS in order; the K / 256 slice partials are added in a fixed order (hp_reduce_kernel)
//   dX (M, K)    = dY (M, 256) W^T                      hp_dx_kernel: workgroup = 64 rows of W (a contiguous 64-KB block), dY in LDS; no partials
//   dW (K, 256)  = X^T dY                               hp_dw_kernel: workgroup = 64 rows of dW (contiguous), contraction over the M rows; dY in LDS
//
// Arithmetic: the split-bf16 forms of the tile GEMM (common.h) -- P pieces per operand: 1 = one bf16 value (params["TRAIN_PRECISION"] =
// "bf16"), 2 = three products (the backward products of the default arithmetic), 3 = six products (its forward products: f32-accurate);
// f32 accumulation on the matrix pipe (v_mfma_f32_32x32x16_bf16), partial sums met in a fixed order: the same bits every run.
// Columns are INTERLEAVED over the two accumulators of a lane (tile t of a pair holds columns 2 i + t): one 8-byte access per lane and row
// where the operand or the result is row-major in the 256 columns.
// M: 64 .. 128 rows (what the tuple sizes of the training step give: 16 .. 32 clouds x 4 groups), a multiple of 4; K a multiple of 256.
#include "train_head_common.h"

#define HP_PITCH 260          // floats per staged row: 1 KB + 16 bytes (a lane's 32-byte fragment reads of 32 rows spread over the banks)
#define HP_MT 4               // row tiles of 32 at most

template <int P>
struct HpFrag {
    bf16x8 p[P];
};
template <int P>
__device__ __forceinline__ void hp_split(const float (&v)[8], HpFrag<P>& f) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float r = v[j];
#pragma unroll
        for (int q = 0; q < P; ++q) {
            f.p[q][j] = (__bf16)r;
            r -= (float)f.p[q][j];
        }
    }
}
// (the products: piece x of a with piece s - x of b for s = P - 1 .. 0, smallest terms first -- 1, 3 or 6 MFMAs, interleaved over two
// accumulators in the kernels)

// rows 0 .. mpad - 1 of a row-major matrix (1 KB of each: 256 floats from column c0), rows past M - 1 clamped, into LDS at HP_PITCH
__device__ __forceinline__ void hp_stage_rows(const float* __restrict__ src, long row_stride, int M, int mpad, unsigned lds_base, int wave,
                                              int lane) {
    for (int r = wave; r < mpad; r += 4)
        glds16(src + (size_t)min(r, M - 1) * row_stride, 16u * lane, lds_base + 4u * HP_PITCH * r);
}
__device__ __forceinline__ void hp_ld8(const float* p, float (&v)[8]) {
    const float4 x = ld4(p), y = ld4(p + 4);
    v[0] = x.x, v[1] = x.y, v[2] = x.z, v[3] = x.w, v[4] = y.x, v[5] = y.y, v[6] = y.z, v[7] = y.w;
}

// ---- forward ---------------------------------------------------------------------------------------------------------------------
// grid (4 column tiles of 64, K / 256 slices); partial[slice][M][256]
template <int P>
__global__ __launch_bounds__(256, 1) void hp_fwd_kernel(const float* __restrict__ X, const float* __restrict__ W, int M, int K,
                                                        float* __restrict__ part) {
    extern __shared__ __attribute__((aligned(16))) float hp_lds[];
    const unsigned lds_base = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)hp_lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int MT = (M + 31) >> 5;
    const int n0 = 64 * blockIdx.x, k0 = 256 * blockIdx.y, kw = k0 + 64 * wave;
    hp_stage_rows(X + k0, K, M, 32 * MT, lds_base, wave, lane);
    // W[kw + 16 ks + 8 h + j][n0 + 2 i + t]: the wave's 16 KB, every load in flight before the first use
    float2 w2[4][8];
    {
        const float* wp = W + (size_t)(kw + 8 * h) * 256 + n0 + 2 * i;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) w2[ks][j] = *reinterpret_cast<const float2*>(wp + (size_t)(16 * ks + j) * 256);
    }
    f32x16 acc[HP_MT][2];
#pragma unroll
    for (int mt = 0; mt < HP_MT; ++mt)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][t][r] = 0.f;
    // the LDS-DMA rows are invisible to the compiler's counters; they were issued BEFORE W's 32 loads and the counter retires in order:
    // at most 32 outstanding = every staged row has landed, W still travelling under the barrier and the first fragments' splits
    asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        HpFrag<P> b[2];
        {
            float v0[8], v1[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v0[j] = w2[ks][j].x, v1[j] = w2[ks][j].y;
            hp_split<P>(v0, b[0]);
            hp_split<P>(v1, b[1]);
        }
#pragma unroll
        for (int mt = 0; mt < HP_MT; ++mt) {
            if (mt < MT) {
                float v[8];
                hp_ld8(hp_lds + (32 * mt + i) * HP_PITCH + 64 * wave + 16 * ks + 8 * h, v);
                HpFrag<P> a;
                hp_split<P>(v, a);
#pragma unroll
                for (int s = P - 1; s >= 0; --s)
#pragma unroll
                    for (int x = 0; x <= s; ++x) {
                        acc[mt][0] = mfma_bf16(a.p[x], b[0].p[s - x], acc[mt][0]);
                        acc[mt][1] = mfma_bf16(a.p[x], b[1].p[s - x], acc[mt][1]);
                    }
            }
        }
    }
    __syncthreads();                                      // (the staged rows are dead: the four waves' accumulators meet in their place)
    // red[wave][mt][r][t][lane]
#pragma unroll
    for (int mt = 0; mt < HP_MT; ++mt)
        if (mt < MT)
#pragma unroll
            for (int r = 0; r < 16; ++r)
#pragma unroll
                for (int t = 0; t < 2; ++t) hp_lds[(((wave * HP_MT + mt) * 16 + r) * 2 + t) * 64 + lane] = acc[mt][t][r];
    __syncthreads();
    float* out = part + (size_t)blockIdx.y * M * 256 + n0;
    for (int e = tid; e < MT * 16 * 64; e += 256) {       // e = (mt, r, lane)
        const int l = e & 63, r = (e >> 6) & 15, mt = e >> 10;
        const int row = 32 * mt + mfma_row(r, l >> 5);
        float2 s = make_float2(0.f, 0.f);
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float* q = hp_lds + (((w * HP_MT + mt) * 16 + r) * 2) * 64 + l;
            s.x += q[0], s.y += q[64];
        }
        if (row < M) *reinterpret_cast<float2*>(out + (size_t)row * 256 + 2 * (l & 31)) = s;
    }
}

// Y[e] = the sum over the slices of part[s][e], ascending s within each of four groups of slices (one group per wave, every load of a
// group in flight at once), the four groups added in order: one memory round trip instead of eight (18 workgroups of
// h16_partial_reduce_kernel took 4.8 us for 4.7 MB).  per = M x 256 floats; a workgroup takes 64 float4 columns.
__global__ __launch_bounds__(256) void hp_reduce_kernel(const float* __restrict__ part, int S, long per, float* __restrict__ Y) {
    __shared__ float4 meet[3][64];
    const int c = threadIdx.x & 63, g = threadIdx.x >> 6;
    const long e = ((long)blockIdx.x * 64 + c) * 4;
    const int per_group = (S + 3) >> 2, s0 = g * per_group, s1 = min(s0 + per_group, S);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e < per) {
        for (int s = s0; s < s1; s += 16) {
            float4 v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = s + u < s1 ? ld4(part + (size_t)(s + u) * per + e) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int u = 0; u < 16; ++u) acc.x += v[u].x, acc.y += v[u].y, acc.z += v[u].z, acc.w += v[u].w;
        }
    }
    if (g > 0) meet[g - 1][c] = acc;
    __syncthreads();
    if (g == 0 && e < per) {
#pragma unroll
        for (int q = 0; q < 3; ++q) acc.x += meet[q][c].x, acc.y += meet[q][c].y, acc.z += meet[q][c].z, acc.w += meet[q][c].w;
        st4(Y + e, acc);
    }
}

// ---- dX = dY W^T ------------------------------------------------------------------------------------------------------------------
// grid K / 64; wave w: W rows k0 + 32 (w & 1) + i (the output's columns), row tiles 2 (w >> 1), 2 (w >> 1) + 1
template <int P>
__global__ __launch_bounds__(256, 1) void hp_dx_kernel(const float* __restrict__ dY, const float* __restrict__ W, int M, int K,
                                                       float* __restrict__ dX) {
    extern __shared__ __attribute__((aligned(16))) float hp_lds[];
    const unsigned lds_base = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)hp_lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int MT = (M + 31) >> 5;
    const int kr = 64 * blockIdx.x + 32 * (wave & 1) + i;
    hp_stage_rows(dY, 256, M, 32 * MT, lds_base, wave, lane);
    float4 wr[16][2];                                     // W[kr][16 ks + 8 h .. + 7]: the lane's whole row share, in flight at once
    {
        const float* wp = W + (size_t)kr * 256 + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) wr[ks][0] = ld4(wp + 16 * ks), wr[ks][1] = ld4(wp + 16 * ks + 4);
    }
    f32x16 acc[2];
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int mt0 = 2 * (wave >> 1);
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
        HpFrag<P> b;
        {
            const float v[8] = {wr[ks][0].x, wr[ks][0].y, wr[ks][0].z, wr[ks][0].w, wr[ks][1].x, wr[ks][1].y, wr[ks][1].z, wr[ks][1].w};
            hp_split<P>(v, b);
        }
        HpFrag<P> a[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float v[8];
            hp_ld8(hp_lds + (32 * min(mt0 + q, MT - 1) + i) * HP_PITCH + 16 * ks + 8 * h, v);
            hp_split<P>(v, a[q]);
        }
#pragma unroll
        for (int s = P - 1; s >= 0; --s)
#pragma unroll
            for (int x = 0; x <= s; ++x) {
                acc[0] = mfma_bf16(a[0].p[x], b.p[s - x], acc[0]);
                acc[1] = mfma_bf16(a[1].p[x], b.p[s - x], acc[1]);
            }
    }
    // D: lane (i, h), register r of q = dX[row 32 (mt0 + q) + mfma_row(r, h)][column kr]
#pragma unroll
    for (int q = 0; q < 2; ++q)
        if (mt0 + q < MT)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = 32 * (mt0 + q) + mfma_row(r, h);
                if (row < M) dX[(size_t)row * K + kr] = acc[q][r];
            }
}

// ---- dW = X^T dY ------------------------------------------------------------------------------------------------------------------
// grid K / 64; wave w: dW rows k0 + 32 (w & 1) + i, columns 128 (w >> 1) + 64 pr + 2 i + t (pr, t < 2); contraction over the M rows
template <int P>
__global__ __launch_bounds__(256, 1) void hp_dw_kernel(const float* __restrict__ X, const float* __restrict__ dY, int M, int K,
                                                       float* __restrict__ dW) {
    extern __shared__ __attribute__((aligned(16))) float hp_lds[];
    const unsigned lds_base = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)hp_lds;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int KS = (M + 15) >> 4;                         // k-steps of 16 rows (at most 8)
    const int kr = 64 * blockIdx.x + 32 * (wave & 1) + i;
    const int nw = 128 * (wave >> 1);
    hp_stage_rows(dY, 256, M, 16 * KS, lds_base, wave, lane);
    float xr[8][8];                                       // X[16 ks + 8 h + j][kr]: 128-byte runs per row over the lanes; zero past M
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = 16 * ks + 8 * h + j;
            xr[ks][j] = X[(size_t)min(row, M - 1) * K + kr];      // (unconditional: 64 loads in flight; a guarded load is waited for one by one)
        }
#pragma unroll
    for (int ks = 0; ks < 8; ++ks)
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (16 * ks + 8 * h + j >= M) xr[ks][j] = 0.f;
    f32x16 acc[2][2];
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pr][t][r] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        if (ks < KS) {
            HpFrag<P> a;
            hp_split<P>(xr[ks], a);
#pragma unroll
            for (int pr = 0; pr < 2; ++pr) {
                float v0[8], v1[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float2 d = *reinterpret_cast<const float2*>(hp_lds + (16 * ks + 8 * h + j) * HP_PITCH + nw + 64 * pr + 2 * i);
                    v0[j] = d.x, v1[j] = d.y;
                }
                HpFrag<P> b0, b1;
                hp_split<P>(v0, b0);
                hp_split<P>(v1, b1);
#pragma unroll
                for (int s = P - 1; s >= 0; --s)
#pragma unroll
                    for (int x = 0; x <= s; ++x) {
                        acc[pr][0] = mfma_bf16(a.p[x], b0.p[s - x], acc[pr][0]);
                        acc[pr][1] = mfma_bf16(a.p[x], b1.p[s - x], acc[pr][1]);
                    }
            }
        }
    }
    // D: lane (i, h), register r of (pr, t) = dW[row k0 + 32 (w & 1) + mfma_row(r, h)][column nw + 64 pr + 2 i + t]
    float* o = dW + (size_t)(64 * blockIdx.x + 32 * (wave & 1)) * 256 + nw + 2 * i;
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            *reinterpret_cast<float2*>(o + (size_t)mfma_row(r, h) * 256 + 64 * pr) = make_float2(acc[pr][0][r], acc[pr][1][r]);
}

// ---- C ABI ---------------------------------------------------------------------------------------------------------------------
extern "C" int epc_hidden_proj_ok(int M, int K, int N) { return M >= 64 && M <= 32 * HP_MT && M % 4 == 0 && N == 256 && K >= 256 && K % 256 == 0; }

extern "C" size_t epc_hidden_proj_scratch_bytes(int M, int K) {
    return M > 0 && K > 0 ? (size_t)(K / 256) * M * 256 * sizeof(float) : 0;
}

template <typename Kern>
static int hp_set_lds(Kern kern, size_t lds, const char* who) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        epc_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
        return EPC_EHIP;
    }
    return EPC_OK;
}
#define HP_LAUNCH(kern, grid, lds, ...)                                                                   \
    do {                                                                                                  \
        int rc_ = EPC_OK;                                                                                 \
        if (pieces == 1) {                                                                                \
            if ((rc_ = hp_set_lds(kern<1>, lds, __func__)) != EPC_OK) return rc_;                         \
            hipLaunchKernelGGL(kern<1>, grid, dim3(256), lds, st, __VA_ARGS__);                           \
        } else if (pieces == 2) {                                                                         \
            if ((rc_ = hp_set_lds(kern<2>, lds, __func__)) != EPC_OK) return rc_;                         \
            hipLaunchKernelGGL(kern<2>, grid, dim3(256), lds, st, __VA_ARGS__);                           \
        } else {                                                                                          \
            if ((rc_ = hp_set_lds(kern<3>, lds, __func__)) != EPC_OK) return rc_;                         \
            hipLaunchKernelGGL(kern<3>, grid, dim3(256), lds, st, __VA_ARGS__);                           \
        }                                                                                                 \
    } while (0)

// Y (M, 256) = X (M, K) W (K, 256); pieces: bf16 pieces per operand (1, 2, 3)
extern "C" int epc_hidden_proj_fwd(const float* X, const float* W, int M, int K, int pieces, float* Y, void* scratch, size_t scratch_bytes,
                                   void* stream) {
    EPC_CHECK_ARG(X && W && Y && scratch, "null pointer");
    EPC_CHECK_ARG(epc_hidden_proj_ok(M, K, 256), "shape not covered (epc_hidden_proj_ok)");
    EPC_CHECK_ARG(pieces >= 1 && pieces <= 3, "pieces must be 1, 2 or 3");
    EPC_CHECK_ARG(scratch_bytes >= epc_hidden_proj_scratch_bytes(M, K), "scratch too small (epc_hidden_proj_scratch_bytes)");
    EPC_CHECK_ARG(h16_aligned16(X) && h16_aligned16(W) && h16_aligned16(Y) && h16_aligned16(scratch), "tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int MT = (M + 31) / 32, S = K / 256;
    const size_t stage = (size_t)32 * MT * HP_PITCH * sizeof(float), meet = (size_t)4 * HP_MT * 16 * 2 * 64 * sizeof(float);
    const size_t lds = stage > meet ? stage : meet;
    HP_LAUNCH(hp_fwd_kernel, dim3(4, S), lds, X, W, M, K, (float*)scratch);
    const long per = (long)M * 256;
    hipLaunchKernelGGL(hp_reduce_kernel, dim3((unsigned)((per / 4 + 63) / 64)), dim3(256), 0, st, (const float*)scratch, S, per, Y);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// dX (M, K) = dY (M, 256) W^T (dX may be NULL) and dW (K, 256) = X^T dY (dW may be NULL)
extern "C" int epc_hidden_proj_bwd(const float* X, const float* W, const float* dY, int M, int K, int pieces, float* dX, float* dW,
                                   void* stream) {
    EPC_CHECK_ARG(X && W && dY, "null pointer");
    EPC_CHECK_ARG(epc_hidden_proj_ok(M, K, 256), "shape not covered (epc_hidden_proj_ok)");
    EPC_CHECK_ARG(pieces >= 1 && pieces <= 3, "pieces must be 1, 2 or 3");
    EPC_CHECK_ARG(h16_aligned16(X) && h16_aligned16(W) && h16_aligned16(dY) && h16_aligned16(dX) && h16_aligned16(dW), "tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int MT = (M + 31) / 32;
    if (dX) {
        const size_t lds = (size_t)32 * MT * HP_PITCH * sizeof(float);
        HP_LAUNCH(hp_dx_kernel, dim3(K / 64), lds, dY, W, M, K, dX);
    }
    if (dW) {
        const size_t lds = (size_t)16 * ((M + 15) / 16) * HP_PITCH * sizeof(float);
        HP_LAUNCH(hp_dw_kernel, dim3(K / 64), lds, X, dY, M, K, dW);
    }
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}
