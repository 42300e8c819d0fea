// Training-step operators (config c3: train.py:251-277).  The fused inference kernels fold BatchNorm into the weights;
// training needs batch statistics between layers (utils/tf_util.py:454-491: moments over all B*N rows), so the step is
// built from per-layer operators, each with its backward twin:
//
//   epc_gemm_f32            C = op(A) op(B) (+bias)   exact f32 MFMA; NN (y = x W), NT (dx = dy W^T), TN (dW = x^T dy,
//                           split-K with f32 atomics), batched (VLAD aggregate fwd/bwd, loupe.py:286-292)
//   epc_col_moments         per-channel mean / population variance over rows (tf.nn.moments), two-pass, deterministic
//   epc_bn_apply_fwd/bwd    y = act(gamma (z-mean) rstd + beta) and its backward (dgamma, dbeta, dz)
//   epc_neighbour_mean_fwd/bwd   xm_i = sum_{j in nbr(i)} x_j / k (models/epc-net.py:70-71) and its transpose
//   epc_rownorm_fwd/bwd     tf.nn.l2_normalize over the channel axis (models/epc-net.py:148)
//   epc_softmax64_fwd/bwd   tf.nn.softmax over the 64 clusters (loupe.py:272)
//   epc_adam_step           tf.train.AdamOptimizer update (train.py:273), one fused pass per tensor
//
// Arithmetic is plain f32 throughout (the reference trains in fp32).
#include <type_traits>
#include "common.h"

// ----------------------------------------------------------------------------------------------------------------
// GEMM: 64x64 output tile per 256-thread workgroup (4 waves x one 32x32 MFMA tile), K in steps of 32 through LDS.
// A(m,k) = A[m*sAm + k*sAk], B(k,n) = B[k*sBk + n*sBn]: the four transpose forms are stride choices; tiles are
// loaded with the contiguous index on the lanes and stored k-major in LDS (+1 pad: conflict-free both ways).
// ----------------------------------------------------------------------------------------------------------------
#define G_BM 64
#define G_BN 64
#define G_BK 32

struct GemmArgs {
    const float* A;
    const float* B;
    float* C;
    const float* bias;
    int M, N, K;
    long sAm, sAk, sBk, sBn;
    int ldc;
    long bA, bB, bC;
    int splitk, accumulate;
    float* stats;   // optional (split kernel, splitk == 1, batch == 1): per row tile the column sums of A.B and of (A.B)^2
    float* partial; // optional (splitk > 1): slice blockIdx.z stores its (M, N) partial product here instead of adding it to C
                    // atomically; splitk_reduce_kernel then adds the slices in ascending order (deterministic split-K)
    float a_scale = 1.f, b_scale = 1.f, descale = 1.f;   // PIECES == 4 (split-fp16): powers of two, descale = 1 / (a_scale b_scale)
    // Range guard of the split-fp16 form: the PIECES == 4 kernel ORs 1 into *range_flag when a scaled operand value is
    // not finite or leaves fp16's range (flag_mode 1); the six-product kernel launched behind it returns at once unless the word is
    // set (flag_mode 2) -- an in-stream, in-graph fallback with no host round trip.  Null: no guard.
    unsigned int* range_flag = nullptr;
    int flag_mode = 0;
    int b16 = 0;   // B is stored as bf16 (strides in elements): the training head's dz5 as the right operand of dW5 = cat^T dz5
};

__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    __shared__ float As[G_BK][G_BM + 1];
    __shared__ float Bs[G_BK][G_BN + 1];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * G_BN, m0 = blockIdx.y * G_BM;
    const int batch = blockIdx.z / g.splitk, ks = blockIdx.z % g.splitk;
    const float* A = g.A + (size_t)batch * g.bA;
    const float* B = g.B + (size_t)batch * g.bB;
    float* C = g.C + (size_t)batch * g.bC;
    int kchunk = (g.K + g.splitk - 1) / g.splitk;
    kchunk = (kchunk + G_BK - 1) / G_BK * G_BK;
    const int k0 = ks * kchunk, k1 = min(g.K, k0 + kchunk);
    const bool a_kc = g.sAk == 1, b_nc = g.sBn == 1;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    for (int kt = k0; kt < k1; kt += G_BK) {
#pragma unroll
        for (int u = 0; u < (G_BM * G_BK) / 256; ++u) {
            const int e = tid + 256 * u;
            const int kk = a_kc ? e % G_BK : e / G_BM, mm = a_kc ? e / G_BK : e % G_BM;
            const int gm = m0 + mm, gk = kt + kk;
            // (loads from clamped indices, zeroed afterwards: a guarded load is an exec-masked branch with a full wait at its
            // join -- the 16 loads of a k-tile went out one memory round trip at a time: 27-44 us for the step's three
            // 18-row products)
            const float v = A[(size_t)min(gm, g.M - 1) * g.sAm + (size_t)min(gk, k1 - 1) * g.sAk];
            As[kk][mm] = (gm < g.M && gk < k1) ? v : 0.f;
        }
#pragma unroll
        for (int u = 0; u < (G_BN * G_BK) / 256; ++u) {
            const int e = tid + 256 * u;
            const int nn = b_nc ? e % G_BN : e / G_BK, kk = b_nc ? e / G_BN : e % G_BK;
            const int gn = n0 + nn, gk = kt + kk;
            const float v = B[(size_t)min(gk, k1 - 1) * g.sBk + (size_t)min(gn, g.N - 1) * g.sBn];
            Bs[kk][nn] = (gn < g.N && gk < k1) ? v : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < G_BK / 2; ++s) acc = mfma32(As[2 * s + h][32 * wm + i], Bs[2 * s + h][32 * wn + i], acc);
        __syncthreads();
    }
    const int col = n0 + 32 * wn + i;
    if (col < g.N) {
        const float bv = (g.bias && ks == 0) ? g.bias[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = m0 + 32 * wm + mfma_row(r, h);
            if (row < g.M) {
                float* p = C + (size_t)row * g.ldc + col;
                const float v = acc[r] + bv;
                if (g.partial)
                    g.partial[((size_t)blockIdx.z * g.M + row) * g.N + col] = acc[r];
                else if (g.splitk > 1)
                    atomicAdd(p, v);
                else
                    *p = g.accumulate ? *p + v : v;
            }
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Products with at most 32 rows (the 18-row context-gating layer of a training tuple, loupe.py:84-100, forward and input
// gradient): the tile kernels above run them as FOUR workgroups walking K in eight barrier-separated steps -- 27-46 us of
// latency for 2 MFLOP.  Here A (M x K) sits in LDS, a workgroup owns 16 output columns and 16 interleaved K slices
// (thread = column x slice: all of its B values are requested at once), every thread keeps M running sums in plain f32 FMAs
// and the slices meet in LDS in order.  One pass, deterministic.
// ----------------------------------------------------------------------------------------------------------------
#define SM_MAX_M 32
#define SM_COLS 16
#define SM_SLICES 16
template <int MM>   // M rounded up to a multiple of 8: the rows past M are zeros in LDS, so the inner loop carries no row test
__global__ __launch_bounds__(256) void gemm_small_m_kernel(GemmArgs g) {
    extern __shared__ float sm_lds[];   // A as [MM][K], then red[SM_SLICES][MM][SM_COLS]
    float* xs = sm_lds;
    float* red = sm_lds + (size_t)MM * g.K;
    const int tid = threadIdx.x, c = tid & (SM_COLS - 1), sl = tid / SM_COLS;
    const int n = blockIdx.x * SM_COLS + c;
    const int total = g.M * g.K;
    if (g.sAk == 1 && g.sAm == g.K && (g.K & 3) == 0 && (reinterpret_cast<size_t>(g.A) & 15) == 0) {   // dense rows: float4 copies
        const float4* src = reinterpret_cast<const float4*>(g.A);
        float4* dst = reinterpret_cast<float4*>(xs);
#pragma unroll 4
        for (int e = tid; e < total / 4; e += 256) dst[e] = src[e];
    } else {
#pragma unroll 8
        for (int e = tid; e < total; e += 256) xs[e] = g.A[(size_t)(e / g.K) * g.sAm + (size_t)(e % g.K) * g.sAk];
    }
    for (int e = total + tid; e < MM * g.K; e += 256) xs[e] = 0.f;
    __syncthreads();
    float acc[MM];
#pragma unroll
    for (int m = 0; m < MM; ++m) acc[m] = 0.f;
    const float* bp = g.B + (size_t)min(n, g.N - 1) * g.sBn;
    for (int k0 = sl; k0 < g.K; k0 += 16 * SM_SLICES) {   // 16 of the thread's B values in flight per round (K = 256: one round)
        float b[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) b[u] = bp[(size_t)min(k0 + u * SM_SLICES, g.K - 1) * g.sBk];   // clamped, zeroed below
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int k = k0 + u * SM_SLICES;
            const float bv = k < g.K ? b[u] : 0.f;
            const float* xk = xs + min(k, g.K - 1);
#pragma unroll
            for (int m = 0; m < MM; ++m) acc[m] += xk[(size_t)m * g.K] * bv;
        }
    }
#pragma unroll
    for (int m = 0; m < MM; ++m) red[(sl * MM + m) * SM_COLS + c] = acc[m];
    __syncthreads();
    for (int o = tid; o < g.M * SM_COLS; o += 256) {
        const int m = o / SM_COLS, cc = o % SM_COLS, col = blockIdx.x * SM_COLS + cc;
        if (col >= g.N) continue;
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < SM_SLICES; ++q) t += red[(q * MM + m) * SM_COLS + cc];
        if (g.bias) t += g.bias[col];
        float* p = g.C + (size_t)m * g.ldc + col;
        *p = g.accumulate ? *p + t : t;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// GEMM on the bf16 MFMA with f32-level accuracy ("bf16x6"): every f32 operand is carried as THREE bf16 pieces
// p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1) (24 significant bits) and a product is evaluated as
// a2*b0 + a0*b2 + a1*b1 + a1*b0 + a0*b1 + a0*b0 (smallest terms first, f32 accumulation; the dropped terms are below
// 2^-24 relative).  Six 32-cycle MFMAs per 16-deep k-step replace eight 64-cycle f32 MFMAs: 2.7x fewer matrix-pipe
// cycles at the same accuracy.  (The two-piece form the inference path uses is 2^-16 per product: fine under a VLAD
// sum, but it showed in the deepest gradients -- conv1: 9e-3 relative against a 5e-3 bar.)
// Same interface and stride generality (NN / NT / TN, batched, split-K) as gemm_f32_kernel, which keeps the shapes
// with a side below 64.  Tile (64*WM) x (64*WN) x 32, 256 threads = 2 x 2 waves of (32*WM) x (32*WN).  Operands are
// split while they are staged: each thread fetches lane-fragments (8 consecutive k of one row / column: two float4
// when k is the contiguous axis, 8 lane-coalesced scalars otherwise) and writes the three pieces as ready MFMA
// fragments ([32-row block][k-step][piece][lane]).
// ----------------------------------------------------------------------------------------------------------------
#define S_BK 32

__device__ __forceinline__ void split8x3(const float (&v)[8], bf16x8& p0, bf16x8& p1, bf16x8& p2) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        p0[j] = (__bf16)v[j];
        const float r1 = v[j] - (float)p0[j];
        p1[j] = (__bf16)r1;
        p2[j] = (__bf16)(r1 - (float)p1[j]);
    }
}

// Staging is split in two so that the global loads of k-tile t+1 are in flight while the MFMAs of tile t run:
// fetch_fragments() only loads (8 floats per lane-fragment, kept in registers), store_fragments() splits them into
// bf16 pieces and writes the ready MFMA fragments.  PIECES = 3: f32-accurate (six products), 2: backward GEMMs (three
// products), 1: plain bf16 operands with f32 accumulation (one product; BASELINE.json's "bf16" training configuration).
template <int BLOCKS>  // 32-row blocks of this operand tile (2 * WM or 2 * WN)
__device__ __forceinline__ void fetch_fragments(const float* __restrict__ P, long s_outer, long s_k, int outer0,
                                                int outer_lim, int kt, int k_lim, float (&v)[(BLOCKS * 128) / 256][8],
                                                int tid, bool is16 = false) {
    const bool k_contig = s_k == 1;
    constexpr int ROWS = 32 * BLOCKS;
    if (is16) {   // (workgroup-uniform) the operand is stored as bf16: same fragments, two bytes per element
        const unsigned short* P16 = reinterpret_cast<const unsigned short*>(P);
        const bool inside = outer0 + ROWS <= outer_lim && kt + S_BK <= k_lim;
#pragma unroll
        for (int u = 0; u < (ROWS * 4) / 256; ++u) {
            const int f = tid + 256 * u;
            const int kg = k_contig ? (f & 3) : (f / ROWS), row = k_contig ? (f >> 2) : (f % ROWS);
            const int go = outer0 + row, gk = kt + 8 * kg;
            const unsigned short* src = P16 + (size_t)go * s_outer + (size_t)gk * s_k;
            if (inside && k_contig && ((reinterpret_cast<size_t>(src) & 15) == 0)) {
                const u32x4 w = *reinterpret_cast<const u32x4*>(src);
#pragma unroll
                for (int q = 0; q < 4; ++q) v[u][2 * q] = __uint_as_float(w[q] << 16), v[u][2 * q + 1] = __uint_as_float(w[q] & 0xffff0000u);
            } else if (inside) {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj) v[u][jj] = __uint_as_float((unsigned)src[(size_t)jj * s_k] << 16);
            } else {
#pragma unroll
                for (int jj = 0; jj < 8; ++jj)
                    v[u][jj] = (go < outer_lim && gk + jj < k_lim) ? __uint_as_float((unsigned)src[(size_t)jj * s_k] << 16) : 0.f;
            }
        }
        return;
    }
    // Interior tiles (every row and all 32 k of the tile inside the matrix: all but the edge workgroups / the K tail) take a
    // BRANCH-FREE fetch -- the test is workgroup-uniform, so it is one scalar branch.  The guarded form below puts every
    // load in an exec-masked branch of its own, and hipcc then waits `vmcnt(0)` at each join before it merges the loaded
    // values: the lane-fragments of a k-tile were fetched one memory round trip after the other (2.6 us per k-tile on the
    // K = 73 728 products) instead of all at once.
    const bool interior = outer0 + ROWS <= outer_lim && kt + S_BK <= k_lim;
    if (interior && k_contig && ((reinterpret_cast<size_t>(P) | (size_t)(s_outer * 4)) & 15) == 0) {
#pragma unroll
        for (int u = 0; u < (ROWS * 4) / 256; ++u) {
            const int f = tid + 256 * u;
            const float* src = P + (size_t)(outer0 + (f >> 2)) * s_outer + (size_t)(kt + 8 * (f & 3));
            const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
            v[u][0] = a.x, v[u][1] = a.y, v[u][2] = a.z, v[u][3] = a.w;
            v[u][4] = b.x, v[u][5] = b.y, v[u][6] = b.z, v[u][7] = b.w;
        }
        return;
    }
    if (interior && !k_contig) {
#pragma unroll
        for (int u = 0; u < (ROWS * 4) / 256; ++u) {
            const int f = tid + 256 * u;
            const float* src = P + (size_t)(outer0 + f % ROWS) * s_outer + (size_t)(kt + 8 * (f / ROWS)) * s_k;
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) v[u][jj] = src[(size_t)jj * s_k];
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < (ROWS * 4) / 256; ++u) {
        const int f = tid + 256 * u;  // ROWS x 4 lane-fragments (4 k-groups of 8)
        const int kg = k_contig ? (f & 3) : (f / ROWS), row = k_contig ? (f >> 2) : (f % ROWS);
        const int go = outer0 + row, gk = kt + 8 * kg;
        const float* src = P + (size_t)go * s_outer + (size_t)gk * s_k;
        if (go < outer_lim && gk + 8 <= k_lim && k_contig && ((reinterpret_cast<size_t>(src) & 15) == 0)) {
            const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
            v[u][0] = a.x, v[u][1] = a.y, v[u][2] = a.z, v[u][3] = a.w;
            v[u][4] = b.x, v[u][5] = b.y, v[u][6] = b.z, v[u][7] = b.w;
        } else {
#pragma unroll
            for (int jj = 0; jj < 8; ++jj) v[u][jj] = (go < outer_lim && gk + jj < k_lim) ? src[(size_t)jj * s_k] : 0.f;
        }
    }
}

// PIECES == 4: TWO fp16 pieces of the operand times `scale` (a power of two chosen by the caller so that the operand's values sit
// high in fp16's range; clamped to it), three products hi*lo + lo*hi + hi*hi on the fp16 MFMA: 2^-22 per product at half the
// matrix work of the three-piece bf16 form -- for products whose operands are bounded by construction (BatchNorm / l2-normalised
// activations against weights: conv5, the VLAD assignment).
template <int PIECES>
struct PieceCount {
    static constexpr int value = PIECES == 4 ? 2 : PIECES;
};

template <int BLOCKS, int PIECES>
__device__ __forceinline__ void store_fragments(const float (&v)[(BLOCKS * 128) / 256][8], bool k_contig,
                                                u32x4 (*dst)[2][PieceCount<PIECES>::value][64], int tid, float scale = 1.f,
                                                bool* out_of_range = nullptr) {
    constexpr int ROWS = 32 * BLOCKS;
    unsigned carry = 0;
#pragma unroll
    for (int u = 0; u < (ROWS * 4) / 256; ++u) {
        const int f = tid + 256 * u;
        const int kg = k_contig ? (f & 3) : (f / ROWS), row = k_contig ? (f >> 2) : (f % ROWS);
        const int l2 = (row & 31) + 32 * (kg & 1);
        if constexpr (PIECES == 4) {
            // No clamp: a scaled value beyond fp16's range becomes +-Inf in hi (and NaN in lo), a NaN stays NaN -- the product is then
            // garbage, the range word is set and the six-product kernel behind recomputes everything (range guard, below).  Detected on
            // the packed halves, two per dword: an exponent field of all ones (0x7C00) carries into the sign position when 0x0400 is added.
            f16x8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float xs = v[u][j] * scale;
                hi[j] = (_Float16)xs;
                lo[j] = (_Float16)(xs - (float)hi[j]);
            }
            const u32x4 hw = __builtin_bit_cast(u32x4, hi);
#pragma unroll
            for (int w = 0; w < 4; ++w) carry |= (hw[w] & 0x7C007C00u) + 0x04000400u;
            dst[row >> 5][kg >> 1][0][l2] = hw;
            dst[row >> 5][kg >> 1][1][l2] = __builtin_bit_cast(u32x4, lo);
        } else if constexpr (PIECES == 1) {
            bf16x8 p0;
#pragma unroll
            for (int j = 0; j < 8; ++j) p0[j] = (__bf16)v[u][j];
            dst[row >> 5][kg >> 1][0][l2] = __builtin_bit_cast(u32x4, p0);
        } else {
            bf16x8 p0, p1, p2;
            split8x3(v[u], p0, p1, p2);
            dst[row >> 5][kg >> 1][0][l2] = __builtin_bit_cast(u32x4, p0);
            dst[row >> 5][kg >> 1][1][l2] = __builtin_bit_cast(u32x4, p1);
            if constexpr (PIECES == 3) dst[row >> 5][kg >> 1][2][l2] = __builtin_bit_cast(u32x4, p2);
        }
    }
    if constexpr (PIECES == 4) {
        if (out_of_range) *out_of_range |= (carry & 0x80008000u) != 0;
    }
}

// SWAP: the MFMA operands trade places (D' = B^T A^T), so a lane holds ONE row of C and four consecutive columns per register
// quad: the epilogue is float4 stores -- a quarter of the store instructions of the lane = column layout, whose 64 dword stores
// per wave made the wide-output products (conv5 forward, the VLAD feature gradient: 302 MB written) store-issue-bound.  Used for
// plain outputs (no statistics epilogue, which wants a column per lane; no split-K; N, ldc multiples of 4).
#ifdef GEMM_STAMPS
// Diagnostic build only (scripts/gemm_stamps.py): per wave the shader cycles of each phase of the k-tile loop, summed over its
// k-tiles -- [store (split + LDS writes), barrier 1, fetch issue, MFMA phase (LDS reads + products), barrier 2, epilogue, total].
__device__ unsigned int gemm_stamp_buf[32768][8];
extern "C" int epc_debug_gemm_stamps(void* host, size_t bytes) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(gemm_stamp_buf), bytes < sizeof(gemm_stamp_buf) ? bytes : sizeof(gemm_stamp_buf)) == hipSuccess ? 0 : -3;
}
#define GS_T(var) __builtin_amdgcn_sched_barrier(0); const unsigned long long var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#define GS_ADD(dst, a, b) dst += (unsigned)((b) - (a))
#else
#define GS_T(var)
#define GS_ADD(dst, a, b)
#endif

// (the split-fp16 instantiation of the 128 x 128 tile sits at the edge of three waves per SIMD -- 163 registers; the range guard's
// state pushed it to 179 = two waves and 189 -> 238 us on conv5's forward product -- so it is held there explicitly)
template <int WM, int WN, int PIECES, bool SWAP = false, int G_PF = 1>
__global__ __launch_bounds__(256, (PIECES == 4 && WM == 2 && WN == 2) ? 3 : 1) void gemm_split_kernel(GemmArgs g) {
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int NP = PieceCount<PIECES>::value;
    __shared__ u32x4 As[2 * WM][2][NP][64];  // 4 KB per WM per piece
    __shared__ u32x4 Bs[2 * WN][2][NP][64];
    if (g.flag_mode == 2 && __hip_atomic_load(g.range_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;   // fallback not needed
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int n0 = blockIdx.x * BN, m0 = blockIdx.y * BM;
    bool range_bad = false;
    const int batch = blockIdx.z / g.splitk, ks = blockIdx.z % g.splitk;
    const float* A = g.A + (size_t)batch * g.bA;
    const float* B = g.b16 ? reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(g.B) + (size_t)batch * g.bB)
                           : g.B + (size_t)batch * g.bB;
    float* C = g.C + (size_t)batch * g.bC;
    int kchunk = (g.K + g.splitk - 1) / g.splitk;
    kchunk = (kchunk + S_BK - 1) / S_BK * S_BK;
    const int k0 = ks * kchunk, k1 = min(g.K, k0 + kchunk);
    const bool a_kc = g.sAk == 1, b_kc = g.sBk == 1;

    f32x16 acc[WM][WN];
#pragma unroll
    for (int rb = 0; rb < WM; ++rb)
#pragma unroll
        for (int cb = 0; cb < WN; ++cb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][cb][r] = 0.f;

    // operand values of the next G_PF k-tiles, in registers (WM + WN lane-fragments per thread and tile): a tile's global loads
    // are issued G_PF iterations before store_fragments() consumes them.  One tile ahead leaves a load one MFMA phase
    // (768 cycles with three products) to land and every iteration then waits out the rest of the memory latency -- 2.6 us
    // per k-tile on the K = 73 728 products; two ahead cover it (G_PF = 2: the split-K products; it costs 64 registers, i.e. a
    // wave per SIMD, which the short-K products -- whose loads the other workgroups of the CU cover -- do not get back).
    float va[G_PF][WM][8], vb[G_PF][WN][8];
#pragma unroll
    for (int pf = 0; pf < G_PF; ++pf)
        if (k0 + pf * S_BK < k1) {
            fetch_fragments<2 * WM>(A, g.sAm, g.sAk, m0, g.M, k0 + pf * S_BK, k1, va[pf], tid);
            fetch_fragments<2 * WN>(B, g.sBn, g.sBk, n0, g.N, k0 + pf * S_BK, k1, vb[pf], tid, g.b16 != 0);
        }
#ifdef GEMM_STAMPS
    unsigned st_store = 0, st_b1 = 0, st_fetch = 0, st_mfma = 0, st_b2 = 0;
    GS_T(t_begin);
#endif
    auto k_tile = [&](int kt, auto slotc) {
        constexpr int slot = decltype(slotc)::value;
        GS_T(t0);
        store_fragments<2 * WM, PIECES>(va[slot], a_kc, As, tid, g.a_scale, &range_bad);
        store_fragments<2 * WN, PIECES>(vb[slot], b_kc, Bs, tid, g.b_scale, &range_bad);
        GS_T(t1);
        __syncthreads();
        GS_T(t2);
        if (kt + G_PF * S_BK < k1) {  // in flight under the MFMAs of this tile and of the G_PF - 1 after it
            fetch_fragments<2 * WM>(A, g.sAm, g.sAk, m0, g.M, kt + G_PF * S_BK, k1, va[slot], tid);
            fetch_fragments<2 * WN>(B, g.sBn, g.sBk, n0, g.N, kt + G_PF * S_BK, k1, vb[slot], tid, g.b16 != 0);
        }
        GS_T(t3);
        GS_ADD(st_store, t0, t1);
        GS_ADD(st_b1, t1, t2);
        GS_ADD(st_fetch, t2, t3);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            bf16x8 a[WM][NP], b[WN][NP];   // (fp16 pieces travel in the same 16-byte registers)
#pragma unroll
            for (int pc = 0; pc < NP; ++pc) {
#pragma unroll
                for (int rb = 0; rb < WM; ++rb) a[rb][pc] = __builtin_bit_cast(bf16x8, As[WM * wm + rb][s][pc][lane]);
#pragma unroll
                for (int cb = 0; cb < WN; ++cb) b[cb][pc] = __builtin_bit_cast(bf16x8, Bs[WN * wn + cb][s][pc][lane]);
            }
#pragma unroll
            for (int rb = 0; rb < WM; ++rb)
#pragma unroll
                for (int cb = 0; cb < WN; ++cb) {
                    f32x16 c = acc[rb][cb];
                    auto mm = [&](int pa, int pb) {
                        if constexpr (PIECES == 4) {
                            const f16x8 fa = __builtin_bit_cast(f16x8, a[rb][pa]), fb = __builtin_bit_cast(f16x8, b[cb][pb]);
                            c = SWAP ? mfma_f16(fb, fa, c) : mfma_f16(fa, fb, c);
                        } else {
                            c = SWAP ? mfma_bf16(b[cb][pb], a[rb][pa], c) : mfma_bf16(a[rb][pa], b[cb][pb], c);
                        }
                    };
                    if constexpr (PIECES == 3) {
                        mm(2, 0);
                        mm(0, 2);
                        mm(1, 1);
                    }
                    if constexpr (PIECES >= 2) {
                        mm(1, 0);
                        mm(0, 1);
                    }
                    mm(0, 0);
                    acc[rb][cb] = c;
                }
        }
        GS_T(t4);
        __syncthreads();
        GS_T(t5);
        GS_ADD(st_mfma, t3, t4);
        GS_ADD(st_b2, t4, t5);
    };
    // (the register slot is a compile-time constant: a runtime index would send va / vb to scratch memory)
    for (int kt = k0; kt < k1; kt += G_PF * S_BK) {
        k_tile(kt, std::integral_constant<int, 0>{});
        if constexpr (G_PF > 1)
            if (kt + S_BK < k1) k_tile(kt + S_BK, std::integral_constant<int, 1>{});
    }
    if constexpr (PIECES == 4) {   // the operands were scaled by powers of two: exact to undo
        if (g.flag_mode == 1 && __any(range_bad) && lane == 0) atomicOr(g.range_flag, 1u);
#pragma unroll
        for (int rb = 0; rb < WM; ++rb)
#pragma unroll
            for (int cb = 0; cb < WN; ++cb)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rb][cb][r] *= g.descale;
    }
#ifdef GEMM_STAMPS
    GS_T(t_loop_end);
    struct StampWriter {
        unsigned a, b, c, d, e;
        unsigned long long t0, t1;
        int slot, lane;
        __device__ ~StampWriter() {
            const unsigned long long tend = __builtin_amdgcn_s_memtime();
            if (lane == 0 && slot < 32768) {
                gemm_stamp_buf[slot][0] = a, gemm_stamp_buf[slot][1] = b, gemm_stamp_buf[slot][2] = c, gemm_stamp_buf[slot][3] = d;
                gemm_stamp_buf[slot][4] = e, gemm_stamp_buf[slot][5] = (unsigned)(tend - t1), gemm_stamp_buf[slot][6] = (unsigned)(tend - t0);
                gemm_stamp_buf[slot][7] = (unsigned)(t1 - t0);
            }
        }
    } stamp_writer{st_store, st_b1, st_fetch, st_mfma, st_b2, t_begin, t_loop_end,
                   (int)(((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 4 + wave), lane};
#endif
    // Column statistics of the product for a training-mode BatchNorm that follows (epc_gemm_f32_stats): per row tile and column
    // the PIVOT p = the product's value in the tile's first row, and the sums of (v - p) and (v - p)^2 over the tile's valid rows
    // (of the product without the bias).  Shifted by a value of the column itself the sums stay at the scale of the column's
    // spread, so a channel with |mean| >> std loses nothing to cancellation (the sums were shifted by the bias only);
    // moments_finalize_kernel merges the tiles' (n, mean, M2) in double precision.  Fixed order: registers, lane halves, the
    // two row-waves.
    if (g.stats) {
        __shared__ float sstat[2][2][32 * WN];   // [column wave][sum, sum of squares][column]
        __shared__ float spiv[2][32 * WN];       // [column wave][column]: the tile's first row (held by row-wave 0)
        float s1[WN], s2[WN], piv[WN];
        if (wm == 0 && h == 0) {
#pragma unroll
            for (int cb = 0; cb < WN; ++cb) spiv[wn][32 * cb + i] = acc[0][cb][0];   // mfma_row(0, 0) = row m0 of the tile
        }
        __syncthreads();
#pragma unroll
        for (int cb = 0; cb < WN; ++cb) {
            piv[cb] = spiv[wn][32 * cb + i];
            s1[cb] = s2[cb] = 0.f;
#pragma unroll
            for (int rb = 0; rb < WM; ++rb)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = m0 + 32 * WM * wm + 32 * rb + mfma_row(r, h);
                    const float v = row < g.M ? acc[rb][cb][r] - piv[cb] : 0.f;
                    s1[cb] += v;
                    s2[cb] += v * v;
                }
            s1[cb] += __shfl_xor(s1[cb], 32);
            s2[cb] += __shfl_xor(s2[cb], 32);
            if (wm == 1 && h == 0) {
                sstat[wn][0][32 * cb + i] = s1[cb];
                sstat[wn][1][32 * cb + i] = s2[cb];
            }
        }
        __syncthreads();
        if (wm == 0 && h == 0) {
#pragma unroll
            for (int cb = 0; cb < WN; ++cb) {
                const int col = n0 + 32 * WN * wn + 32 * cb + i;
                if (col < g.N) {
                    g.stats[((size_t)blockIdx.y * 3 + 0) * g.N + col] = s1[cb] + sstat[wn][0][32 * cb + i];
                    g.stats[((size_t)blockIdx.y * 3 + 1) * g.N + col] = s2[cb] + sstat[wn][1][32 * cb + i];
                    g.stats[((size_t)blockIdx.y * 3 + 2) * g.N + col] = piv[cb];
                }
            }
        }
    }
    // With the statistics epilogue the lane = column layout stays (column sums are in-lane), but its plain stores -- 16 dword
    // store instructions per 32 x 32 tile -- go through a per-wave LDS tile (the operand buffers are free now) and leave as
    // float4 rows: 4 store instructions per tile, each writing eight whole 128-B rows.
    if constexpr (!SWAP && WM == 2 && (PIECES == 3 || PIECES == 4)) {
        if (g.stats && g.N % 4 == 0 && g.ldc % 4 == 0 && (reinterpret_cast<size_t>(C) & 15) == 0 &&
            (!g.bias || (reinterpret_cast<size_t>(g.bias) & 15) == 0)) {
            float* stg = reinterpret_cast<float*>(&As[0][0][0][0]) + wave * (32 * 36);   // 4 x 4.6 KB of the 24-KB A buffer
#pragma unroll
            for (int rb = 0; rb < WM; ++rb)
#pragma unroll
                for (int cb = 0; cb < WN; ++cb) {
                    __syncthreads();
#pragma unroll
                    for (int r = 0; r < 16; ++r) stg[mfma_row(r, h) * 36 + i] = acc[rb][cb][r];
                    __syncthreads();
                    const int col = n0 + 32 * WN * wn + 32 * cb + 4 * (lane & 7);
                    float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (g.bias && col < g.N) bv = *reinterpret_cast<const float4*>(g.bias + col);
#pragma unroll
                    for (int it = 0; it < 4; ++it) {
                        const int rl = (lane >> 3) + 8 * it, row = m0 + 32 * WM * wm + 32 * rb + rl;
                        if (row < g.M && col < g.N) {
                            float4 v = *reinterpret_cast<const float4*>(stg + rl * 36 + 4 * (lane & 7));
                            v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
                            *reinterpret_cast<float4*>(C + (size_t)row * g.ldc + col) = v;
                        }
                    }
                }
            return;
        }
    }
    if constexpr (SWAP) {   // register 4g + e of (rb, cb) = C[m0 + .. + 32 rb + i][n0 + .. + 32 cb + 8g + 4h + e]
#pragma unroll
        for (int rb = 0; rb < WM; ++rb) {
            const int row = m0 + 32 * WM * wm + 32 * rb + i;
            if (row >= g.M) continue;
#pragma unroll
            for (int cb = 0; cb < WN; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int col = n0 + 32 * WN * wn + 32 * cb + 8 * q + 4 * h;
                    if (col >= g.N) continue;
                    float4 v = make_float4(acc[rb][cb][4 * q], acc[rb][cb][4 * q + 1], acc[rb][cb][4 * q + 2], acc[rb][cb][4 * q + 3]);
                    if (g.bias) {
                        const float4 bv = *reinterpret_cast<const float4*>(g.bias + col);
                        v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
                    }
                    float4* p = reinterpret_cast<float4*>(C + (size_t)row * g.ldc + col);
                    if (g.accumulate) {
                        const float4 o = *p;
                        v.x += o.x, v.y += o.y, v.z += o.z, v.w += o.w;
                    }
                    *p = v;
                }
        }
        return;
    }
#pragma unroll
    for (int cb = 0; cb < WN; ++cb) {
        const int col = n0 + 32 * WN * wn + 32 * cb + i;
        if (col >= g.N) continue;
        const float bv = (g.bias && ks == 0) ? g.bias[col] : 0.f;
#pragma unroll
        for (int rb = 0; rb < WM; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + 32 * WM * wm + 32 * rb + mfma_row(r, h);
                if (row < g.M) {
                    float* p = C + (size_t)row * g.ldc + col;
                    const float v = acc[rb][cb][r] + bv;
                    if (g.partial)
                        g.partial[((size_t)blockIdx.z * g.M + row) * g.N + col] = acc[rb][cb][r];
                    else if (g.splitk > 1)
                        atomicAdd(p, v);
                    else
                        *p = g.accumulate ? *p + v : v;
                }
            }
    }
}

template <int WM, int WN>
static void launch_gemm_split(const GemmArgs& g, int batch, int pieces, hipStream_t st) {
    dim3 grid((g.N + 64 * WN - 1) / (64 * WN), (g.M + 64 * WM - 1) / (64 * WM), batch * g.splitk);
#ifndef EPC_GEMM_NO_SWAP
    const bool swap = !g.stats && !g.partial && g.splitk == 1 && g.N % 4 == 0 && g.ldc % 4 == 0 && g.bC % 4 == 0 &&
                      (reinterpret_cast<size_t>(g.C) & 15) == 0 && (!g.bias || (reinterpret_cast<size_t>(g.bias) & 15) == 0);
#else
    const bool swap = false;
#endif
    if (swap) {
        if (pieces == 1)
            hipLaunchKernelGGL((gemm_split_kernel<WM, WN, 1, true>), grid, dim3(256), 0, st, g);
        else if (pieces == 2)
            hipLaunchKernelGGL((gemm_split_kernel<WM, WN, 2, true>), grid, dim3(256), 0, st, g);
        else
            hipLaunchKernelGGL((gemm_split_kernel<WM, WN, 3, true>), grid, dim3(256), 0, st, g);
        return;
    }
#ifndef EPC_GEMM_NO_DEEP_PREFETCH
    const int kslice = (g.K + g.splitk - 1) / g.splitk;
    if (g.splitk > 1 && kslice >= 8 * S_BK) {   // deep-K slices (dW = x^T dy over all rows): loads two k-tiles ahead
        if (pieces == 1)
            hipLaunchKernelGGL((gemm_split_kernel<WM, WN, 1, false, 2>), grid, dim3(256), 0, st, g);
        else if (pieces == 2)
            hipLaunchKernelGGL((gemm_split_kernel<WM, WN, 2, false, 2>), grid, dim3(256), 0, st, g);
        else
            hipLaunchKernelGGL((gemm_split_kernel<WM, WN, 3, false, 2>), grid, dim3(256), 0, st, g);
        return;
    }
#endif
    if (pieces == 1)
        hipLaunchKernelGGL((gemm_split_kernel<WM, WN, 1>), grid, dim3(256), 0, st, g);
    else if (pieces == 2)
        hipLaunchKernelGGL((gemm_split_kernel<WM, WN, 2>), grid, dim3(256), 0, st, g);
    else if (pieces == 4)   // (the split-fp16 form exists with the statistics epilogue only: epc_gemm_f16x3_stats)
        hipLaunchKernelGGL((gemm_split_kernel<WM, WN, 4>), grid, dim3(256), 0, st, g);
    else
        hipLaunchKernelGGL((gemm_split_kernel<WM, WN, 3>), grid, dim3(256), 0, st, g);
}

static int gemm_impl(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long sAm, long sAk,
                     long sBk, long sBn, int ldc, int batch, long bA, long bB, long bC, int splitk, int accumulate,
                     int pieces, void* stream, float* partial = nullptr, size_t partial_floats = 0, int b16 = 0);

// C[b](m, n) = bias(n) + sum over the split-K slices IN ASCENDING ORDER of their partial products: the deterministic
// counterpart of the atomic epilogue (same bits on every run), one float4 of C per thread.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                                            float* __restrict__ C, int M, int N, int ldc, long bC, int splitk,
                                                            int accumulate) {
    const size_t per = (size_t)M * N;
    const size_t e = ((size_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= per) return;
    const int batch = blockIdx.y;
    const float* p = partial + (size_t)batch * splitk * per + e;
    const int row = (int)(e / N), col = (int)(e % N);
    float4 s = bias ? *reinterpret_cast<const float4*>(bias + col) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < splitk; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(p + (size_t)k * per);
        s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    float4* o = reinterpret_cast<float4*>(C + (size_t)batch * bC + (size_t)row * ldc + col);
    if (accumulate) {
        const float4 c = *o;
        s.x += c.x, s.y += c.y, s.z += c.z, s.w += c.w;
    }
    *o = s;
}

extern "C" int epc_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                            long sAm, long sAk, long sBk, long sBn, int ldc, int batch, long bA, long bB, long bC,
                            int splitk, int accumulate, void* stream) {
    return gemm_impl(A, B, C, bias, M, N, K, sAm, sAk, sBk, sBn, ldc, batch, bA, bB, bC, splitk, accumulate, 3, stream);
}

// The same GEMM with TWO bf16 pieces per operand (three products: 2^-16 relative per product).  For the backward
// GEMMs of the training step: they are linear in the incoming gradient, so the error stays at 1e-5 of the gradient
// (the forward keeps three pieces: it decides ReLU masks and BatchNorm statistics).
extern "C" int epc_gemm_f32_fast(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                                 long sAm, long sAk, long sBk, long sBn, int ldc, int batch, long bA, long bB, long bC,
                                 int splitk, int accumulate, void* stream) {
    return gemm_impl(A, B, C, bias, M, N, K, sAm, sAk, sBk, sBn, ldc, batch, bA, bB, bC, splitk, accumulate, 2, stream);
}

// One bf16 value per operand (a single product): BASELINE.json's "bf16" training configuration.
extern "C" int epc_gemm_bf16(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                             long sAm, long sAk, long sBk, long sBn, int ldc, int batch, long bA, long bB, long bC,
                             int splitk, int accumulate, void* stream) {
    return gemm_impl(A, B, C, bias, M, N, K, sAm, sAk, sBk, sBn, ldc, batch, bA, bB, bC, splitk, accumulate, 1, stream);
}

static int gemm_impl(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long sAm, long sAk,
                     long sBk, long sBn, int ldc, int batch, long bA, long bB, long bC, int splitk, int accumulate,
                     int pieces, void* stream, float* partial, size_t partial_floats, int b16) {
    EPC_CHECK_ARG(A && B && C, "null pointer");
    EPC_CHECK_ARG(!b16 || (M >= 64 && N >= 64 && K >= 32), "a bf16 right operand needs every side of the product at least 64 (K at least 32)");
    EPC_CHECK_ARG(M > 0 && N > 0 && K > 0 && batch > 0 && splitk >= 1 && ldc >= N, "bad shape");
    EPC_CHECK_ARG((long)batch * splitk <= 65535, "batch*splitk too large");
    hipStream_t st = (hipStream_t)stream;
    if (splitk == 1) partial = nullptr;
    if (partial) {
        EPC_CHECK_ARG(partial_floats >= (size_t)batch * splitk * M * N, "split-K workspace too small (batch * splitk * M * N floats)");
        EPC_CHECK_ARG(N % 4 == 0 && ldc % 4 == 0 && bC % 4 == 0, "deterministic split-K needs N, ldc and the batch stride of C in multiples of 4");
        // splitk_reduce_kernel moves float4s of the partials, of C and of the bias
        EPC_CHECK_ARG((reinterpret_cast<size_t>(C) & 15) == 0 && (reinterpret_cast<size_t>(partial) & 15) == 0 &&
                          (!bias || (reinterpret_cast<size_t>(bias) & 15) == 0),
                      "deterministic split-K needs 16-byte aligned C, bias and workspace");
    }
    if (splitk > 1 && !accumulate && !partial) {  // split-K partial sums are added atomically into a zeroed C
        if (ldc == N && (batch == 1 || bC == (long)M * N)) {
            if (hipMemsetAsync(C, 0, (size_t)batch * M * N * sizeof(float), st) != hipSuccess) {
                epc_set_error("epc_gemm_f32: hipMemsetAsync failed");
                return EPC_EHIP;
            }
        } else {
            EPC_CHECK_ARG(false, "split-K needs a dense C (ldc == N, contiguous batches) or accumulate=1");
        }
    }
    GemmArgs g{A, B, C, bias, M, N, K, sAm, sAk, sBk, sBn, ldc, bA, bB, bC, splitk, accumulate, nullptr, partial};
    g.b16 = b16;
    auto finish = [&]() -> int {
        if (partial) {
            dim3 rgrid((unsigned)(((size_t)M * N / 4 + 255) / 256), batch);
            hipLaunchKernelGGL(splitk_reduce_kernel, rgrid, dim3(256), 0, st, partial, bias, C, M, N, ldc, bC, splitk, accumulate);
            EPC_CHECK_LAUNCH();
        }
        return EPC_OK;
    };
    {   // at most 32 rows, one product, no split: A in LDS, plain f32 (gemm_small_m_kernel)
        const int mm = (M + 7) / 8 * 8;
        const size_t lds = ((size_t)mm * K + (size_t)SM_SLICES * mm * SM_COLS) * sizeof(float);
        if (M <= SM_MAX_M && batch == 1 && splitk == 1 && lds <= 64 * 1024 && !b16) {
            const dim3 sgrid((N + SM_COLS - 1) / SM_COLS);
            if (mm == 8) hipLaunchKernelGGL(gemm_small_m_kernel<8>, sgrid, dim3(256), lds, st, g);
            else if (mm == 16) hipLaunchKernelGGL(gemm_small_m_kernel<16>, sgrid, dim3(256), lds, st, g);
            else if (mm == 24) hipLaunchKernelGGL(gemm_small_m_kernel<24>, sgrid, dim3(256), lds, st, g);
            else hipLaunchKernelGGL(gemm_small_m_kernel<32>, sgrid, dim3(256), lds, st, g);
            EPC_CHECK_LAUNCH();
            return EPC_OK;
        }
    }
#ifndef EPC_GEMM_F32_ONLY
    // sides of at least 64: the split-bf16 kernel with the tile that fits; short sides stay on the f32 MFMA kernel
    if (M >= 64 && N >= 64 && K >= 32) {
        const bool bigm = M >= 128, bign = N >= 128;
        if (bigm && bign) launch_gemm_split<2, 2>(g, batch, pieces, st);
        else if (bigm) launch_gemm_split<2, 1>(g, batch, pieces, st);
        else if (bign) launch_gemm_split<1, 2>(g, batch, pieces, st);
        else launch_gemm_split<1, 1>(g, batch, pieces, st);
        EPC_CHECK_LAUNCH();
        return finish();
    }
#endif
    dim3 grid((N + G_BN - 1) / G_BN, (M + G_BM - 1) / G_BM, batch * splitk);
    hipLaunchKernelGGL(gemm_f32_kernel, grid, dim3(256), 0, st, g);
    EPC_CHECK_LAUNCH();
    return finish();
}

// Split-K WITHOUT atomics: every slice stores its partial product in the caller's workspace (batch * splitk * M * N floats) and a
// second launch adds the slices in ascending order -- the same bits on every run, which the FORWARD products of the training
// step want (a last-bit difference in an activation can flip a ReLU mask and move a whole gradient tensor by 1e-3).
// pieces: 3 = the f32-accurate arithmetic of epc_gemm_f32, 2 = epc_gemm_f32_fast, 1 = epc_gemm_bf16.
extern "C" int epc_gemm_splitk_det(const float* A, const float* B, float* C, const float* bias, int M, int N, int K,
                                   long sAm, long sAk, long sBk, long sBn, int ldc, int batch, long bA, long bB, long bC,
                                   int splitk, int accumulate, int pieces, float* workspace, size_t workspace_floats,
                                   void* stream) {
    EPC_CHECK_ARG(pieces >= 1 && pieces <= 3, "pieces must be 1, 2 or 3");
    EPC_CHECK_ARG(workspace || splitk == 1, "null workspace");
    return gemm_impl(A, B, C, bias, M, N, K, sAm, sAk, sBk, sBn, ldc, batch, bA, bB, bC, splitk, accumulate, pieces, stream,
                     workspace, workspace_floats);
}

// epc_gemm_splitk_det with the RIGHT operand stored as bf16 (strides and batch stride in elements): the training head's dW5 = cat^T dz5
// (train_head16.hip keeps dz5 in bf16).  pieces: 1 (A rounded to one bf16 value) or 2 (A in two pieces: the bf16 operand is exact).
extern "C" int epc_gemm_splitk_det_b16(const float* A, const void* B16, float* C, const float* bias, int M, int N, int K,
                                       long sAm, long sAk, long sBk, long sBn, int ldc, int batch, long bA, long bB, long bC,
                                       int splitk, int accumulate, int pieces, float* workspace, size_t workspace_floats,
                                       void* stream) {
    EPC_CHECK_ARG(pieces == 1 || pieces == 2, "pieces must be 1 or 2");
    EPC_CHECK_ARG(workspace || splitk == 1, "null workspace");
    return gemm_impl(A, reinterpret_cast<const float*>(B16), C, bias, M, N, K, sAm, sAk, sBk, sBn, ldc, batch, bA, bB, bC, splitk,
                     accumulate, pieces, stream, workspace, workspace_floats, 1);
}

// ---- y = x W + b together with the batch statistics a training-mode BatchNorm on y needs -------------------------------
// The GEMM's epilogue leaves, per row tile of `tile_rows` rows (the last one shorter) and column, (S1, S2, p): the sums of
// (v - p) and (v - p)^2 with p the tile's first-row value.  A tile's own moments are mean_t = p + S1 / n_t and
// M2_t = S2 - S1^2 / n_t; the tiles are pooled in double precision in a fixed order (population variance = M2 / rows,
// tf.nn.moments).
template <int COLS, int PARTS>
__global__ __launch_bounds__(COLS * PARTS) void moments_finalize_kernel(const float* __restrict__ stats, int tiles, int N, int rows,
                                                                        int tile_rows, const float* __restrict__ bias,
                                                                        float* __restrict__ mean, float* __restrict__ var,
                                                                        int group_rows = 0) {
    // COLS columns per workgroup; thread (column, part) takes tiles part, part + PARTS, ...; the parts meet in LDS in a fixed order:
    // deterministic.  Pooled moments in double precision, two sweeps over the tiles' (sum, sum of squares, pivot):
    //   mean = sum_t (n_t p_t + S1_t) / rows;   M2 = sum_t [ (S2_t - S1_t^2 / n_t) + n_t (p_t + S1_t / n_t - mean)^2 ]
    // -- no division per tile (1 / n_t is one constant for every full tile).  Few columns per workgroup ON PURPOSE: the partials
    // were written by other XCDs, every first touch misses L2, and ONE CU draws ~10 B per clock from beyond it -- a thin layer's
    // 221 KB of partials through one 64-column workgroup took 12 us (rocprofv3) whatever its arithmetic; eight workgroups of
    // 8 columns take a quarter of that.
    __shared__ double sm[PARTS][COLS];
    const int cl = threadIdx.x % COLS, part = threadIdx.x / COLS;
    const int c = blockIdx.x * COLS + cl;
    const bool on = c < N;
    const double inv_full = 1.0 / (double)tile_rows;
    // rows of tile t.  group_rows > 0: the tiles never straddle groups of that many rows (a cloud's points: train_head16.hip) -- every
    // group has ceil(group_rows / tile_rows) tiles, the last one shorter
    const int tpg = group_rows > 0 ? (group_rows + tile_rows - 1) / tile_rows : 0;
    auto n_of = [&](int t) {
        return group_rows > 0 ? min(tile_rows, group_rows - (t % tpg) * tile_rows) : min(tile_rows, rows - t * tile_rows);
    };
    // a sweep visits the thread's tiles MF_CHUNK at a time with ALL of a chunk's loads in flight at once
    constexpr int MF_CHUNK = 5;
    auto sweep = [&](auto&& f) {
        for (int t0 = part; t0 < tiles; t0 += MF_CHUNK * PARTS) {
            float v1[MF_CHUNK], v2[MF_CHUNK], vp[MF_CHUNK];
#pragma unroll
            for (int u = 0; u < MF_CHUNK; ++u) {
                const int t = min(t0 + u * PARTS, tiles - 1);
                v1[u] = stats[((size_t)t * 3 + 0) * N + c], v2[u] = stats[((size_t)t * 3 + 1) * N + c];
                vp[u] = stats[((size_t)t * 3 + 2) * N + c];
            }
#pragma unroll
            for (int u = 0; u < MF_CHUNK; ++u) {
                const int t = t0 + u * PARTS;
                if (t < tiles && n_of(t) > 0) f(n_of(t), (double)v1[u], (double)v2[u], (double)vp[u]);
            }
        }
    };
    double a = 0.0;
    if (on) sweep([&](int nt, double s1, double, double pv) { a += (double)nt * pv + s1; });
    sm[part][cl] = a;
    __syncthreads();
    double tot = 0.0;
    for (int q = 0; q < PARTS; ++q) tot += sm[q][cl];
    const double mu = tot / (double)rows;
    __syncthreads();
    double b = 0.0;
    if (on)
        sweep([&](int nt, double s1, double s2, double pv) {
            const double inv = nt == tile_rows ? inv_full : 1.0 / (double)nt;
            const double d = pv + s1 * inv - mu;
            b += fmax(s2 - s1 * s1 * inv, 0.0) + (double)nt * d * d;
        });
    sm[part][cl] = b;
    __syncthreads();
    if (part == 0 && on) {
        double m2 = 0.0;
        for (int q = 0; q < PARTS; ++q) m2 += sm[q][cl];
        mean[c] = (float)(mu + (bias ? (double)bias[c] : 0.0));
        var[c] = (float)fmax(m2 / (double)rows, 0.0);
    }
}
#define MF_COLS 8
#define MF_PARTS 64
static void launch_moments_finalize(const float* stats, int tiles, int N, int rows, int tile_rows, const float* bias, float* mean,
                                    float* var, hipStream_t st, int group_rows = 0) {
    hipLaunchKernelGGL((moments_finalize_kernel<MF_COLS, MF_PARTS>), dim3((N + MF_COLS - 1) / MF_COLS), dim3(MF_COLS * MF_PARTS), 0,
                       st, stats, tiles, N, rows, tile_rows, bias, mean, var, group_rows);
}

// library-internal (train_head16.hip leaves partials in the same (S1, S2, pivot) form)
int epc_moments_finalize_launch(const float* stats, int tiles, int N, int rows, int tile_rows, const float* bias, float* mean,
                                float* var, void* stream, int group_rows) {
    launch_moments_finalize(stats, tiles, N, rows, tile_rows, bias, mean, var, (hipStream_t)stream, group_rows);
    return EPC_OK;
}

extern "C" int epc_gemm_stats_tiles(int M) { return M >= 128 ? (M + 127) / 128 : (M + 63) / 64; }

struct BnAffine {  // y = z * s + t, the expression the forward and the backward mask must share bit for bit
    float s, t;
};
__device__ __forceinline__ BnAffine bn_affine(float mean, float var, float gamma, float beta, float eps) {
    BnAffine a;
    a.s = (1.0f / sqrtf(var + eps)) * gamma;
    a.t = beta - mean * a.s;
    return a;
}
__device__ __forceinline__ float bn_value(float z, const BnAffine& a) { return z * a.s + a.t; }

struct BnParams {   // a training-mode BatchNorm (+ReLU) applied to an operand AS IT IS LOADED: null mean = none
    const float *mean, *var, *gamma, *beta;
    float eps;
};
__global__ void linear_stats64_kernel(const float* __restrict__ x, BnParams xbn, const float* __restrict__ W,
                                      const float* __restrict__ bias, int rows, float* __restrict__ z,
                                      float* __restrict__ stats);   // (below)

static int gemm_stats_impl(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long sAm, long sAk,
                           long sBk, long sBn, int ldc, float* stats, size_t stats_floats, float* mean, float* var, int pieces,
                           float a_scale, float b_scale, void* stream);

extern "C" int epc_gemm_f32_stats(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long sAm,
                                  long sAk, long sBk, long sBn, int ldc, float* stats, size_t stats_floats, float* mean,
                                  float* var, void* stream) {
    return gemm_stats_impl(A, B, C, bias, M, N, K, sAm, sAk, sBk, sBn, ldc, stats, stats_floats, mean, var, 3, 1.f, 1.f, stream);
}

// The same with ONE bf16 value per operand (BASELINE.json configs[2]'s "bf16" training arithmetic): one product, f32 accumulate.
extern "C" int epc_gemm_bf16_stats(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long sAm,
                                   long sAk, long sBk, long sBn, int ldc, float* stats, size_t stats_floats, float* mean,
                                   float* var, void* stream) {
    return gemm_stats_impl(A, B, C, bias, M, N, K, sAm, sAk, sBk, sBn, ldc, stats, stats_floats, mean, var, 1, 1.f, 1.f, stream);
}

// The same product in the split-fp16 three-product arithmetic (2^-22 per product, half the matrix work): A * 2^a_scale_log2 and
// B * 2^b_scale_log2 are split into fp16 hi + lo, the product is un-scaled exactly.  For operands bounded by construction --
// choose the scales so that typical magnitudes land in [2^-3, 2^12].  RANGE GUARD: a scaled operand value that is
// NaN, infinite or beyond fp16's range (|x| >= 2^(16 - a_scale_log2): nothing bounds a BatchNorm output -- z-hat reaches
// sqrt(rows), gamma is learned) sets a device word, and the six-product split-bf16 kernel launched right behind -- a no-op
// otherwise -- recomputes C and the statistics in float32's range: the result is then exactly epc_gemm_f32_stats', NaN / Inf
// propagate as there, and a diverged step shows a NaN loss instead of a finite wrong one.  The word is the LAST float of `stats`
// (stats_floats >= epc_gemm_stats_tiles(M) * 3 * N + 1); it is zeroed by a memset in front of the product.
extern "C" int epc_gemm_f16x3_stats(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long sAm,
                                    long sAk, long sBk, long sBn, int ldc, int a_scale_log2, int b_scale_log2, float* stats,
                                    size_t stats_floats, float* mean, float* var, void* stream) {
    EPC_CHECK_ARG(a_scale_log2 >= -30 && a_scale_log2 <= 30 && b_scale_log2 >= -30 && b_scale_log2 <= 30, "scale exponents out of range");
    EPC_CHECK_ARG(stats_floats >= (size_t)epc_gemm_stats_tiles(M) * 3 * N + 1, "statistics buffer too small (epc_gemm_stats_tiles(M) * 3 * N + 1 floats: the last one is the range word)");
    return gemm_stats_impl(A, B, C, bias, M, N, K, sAm, sAk, sBk, sBn, ldc, stats, stats_floats, mean, var, 4,
                           ldexpf(1.f, a_scale_log2), ldexpf(1.f, b_scale_log2), stream);
}

static int gemm_stats_impl(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long sAm, long sAk,
                           long sBk, long sBn, int ldc, float* stats, size_t stats_floats, float* mean, float* var, int pieces,
                           float a_scale, float b_scale, void* stream) {
    EPC_CHECK_ARG(A && B && C && stats && mean && var, "null pointer");
    EPC_CHECK_ARG(M >= 64 && N >= 64 && K >= 32 && ldc >= N, "the statistics epilogue exists in the split-bf16 kernel: M, N >= 64, K >= 32");
    const int tiles = epc_gemm_stats_tiles(M);
    EPC_CHECK_ARG(stats_floats >= (size_t)tiles * 3 * N, "statistics buffer too small (epc_gemm_stats_tiles(M) * 3 * N floats)");
    hipStream_t st = (hipStream_t)stream;
#ifndef EPC_NO_THIN_FORWARD
    if (pieces == 3 && N == 64 && K == 64 && sAm == 64 && sAk == 1 && sBk == 64 && sBn == 1 && ldc == 64 &&
        ((reinterpret_cast<size_t>(A) | reinterpret_cast<size_t>(C)) & 15) == 0) {
        // the thin layers: one pass, one partial per 256 rows (fewer than the tiles the caller sized `stats` for)
        const int wgs = (M + 255) / 256;
        hipLaunchKernelGGL(linear_stats64_kernel, dim3(wgs), dim3(256), 0, st, A, BnParams{nullptr, nullptr, nullptr, nullptr, 0.f}, B,
                           bias, M, C, stats);
        EPC_CHECK_LAUNCH();
        launch_moments_finalize(stats, wgs, 64, M, 256 /* LS_ROWS_PER_WG */, bias, mean, var, st);
        EPC_CHECK_LAUNCH();
        return EPC_OK;
    }
#endif
    GemmArgs g{A, B, C, bias, M, N, K, sAm, sAk, sBk, sBn, ldc, 0, 0, 0, 1, 0, stats, nullptr};
    g.a_scale = a_scale, g.b_scale = b_scale, g.descale = 1.0f / (a_scale * b_scale);
    const bool bigm = M >= 128, bign = N >= 128;
    auto launch = [&](const GemmArgs& ga, int pc) {
        if (bigm && bign) launch_gemm_split<2, 2>(ga, 1, pc, st);
        else if (bigm) launch_gemm_split<2, 1>(ga, 1, pc, st);
        else if (bign) launch_gemm_split<1, 2>(ga, 1, pc, st);
        else launch_gemm_split<1, 1>(ga, 1, pc, st);
    };
    if (pieces == 4) {   // split-fp16 with the range guard: flag word = the last float of `stats`
        g.range_flag = reinterpret_cast<unsigned int*>(stats + (size_t)tiles * 3 * N);
        g.flag_mode = 1;
        if (hipMemsetAsync(g.range_flag, 0, sizeof(unsigned int), st) != hipSuccess) {
            epc_set_error("epc_gemm_f16x3_stats: hipMemsetAsync failed");
            return EPC_EHIP;
        }
        launch(g, 4);
        EPC_CHECK_LAUNCH();
        GemmArgs f = g;   // the six-product form in float32's range, run only when the word is set
        f.a_scale = f.b_scale = f.descale = 1.f;
        f.flag_mode = 2;
        launch(f, 3);
    } else {
        launch(g, pieces);
    }
    EPC_CHECK_LAUNCH();
    launch_moments_finalize(stats, tiles, N, M, M >= 128 ? 128 : 64, bias, mean, var, st);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// ---- the same for the thin layers (64 -> 64: conv*_a, conv*_b, conv2..4 of models/epc-net.py:66-132) ---------------------------
// One pass over the rows, no k loop: W (16 KB) is split once per workgroup into B fragments in LDS, a wave takes 32 rows at a
// time -- their 64 channels are the A fragments as loaded (lane = row, eight consecutive channels per k-step) -- and runs the
// six products of the f32-accurate arithmetic; D = [row][out channel] puts a column on every lane, so the column sums of the
// statistics are in-lane.  One partial per workgroup of 256 rows (the general kernel: one per 128-row tile, two barriers and
// an LDS round trip per k-tile for a K of 64): 23.5 -> 13 us per layer at 73 728 rows, and half the partials to finalize.
#define LS_TILES_PER_WAVE 2
#define LS_ROWS_PER_WG (4 * 32 * LS_TILES_PER_WAVE)
static_assert(LS_ROWS_PER_WG == 256, "epc_gemm_f32_stats passes 256 to moments_finalize_kernel");

// One partial (sum, sum of squares, pivot) per workgroup; moments_finalize_kernel merges them.  (Until round 3 the workgroup that
// finished LAST merged them itself, behind a counter and agent-scope loads: one launch less, but that tail -- an atomic round
// trip, then 72 dependent uncached loads per thread -- took 15 of the kernel's 31 us; the separate 1024-thread finish takes 4.)
// xbn: the layer's input is relu(bn(x)) of the PREVIOUS layer's pre-activation x, formed as the rows are loaded (conv_a -> conv_b
// of a block, models/epc-net.py:77-85: the activation between them is never written).
__global__ __launch_bounds__(256) void linear_stats64_kernel(const float* __restrict__ x, BnParams xbn,
                                                             const float* __restrict__ W, const float* __restrict__ bias,
                                                             int rows, float* __restrict__ z, float* __restrict__ stats) {
    __shared__ u32x4 Wf[2][4][3][64];                                  // [out tile][k-step][piece][lane]: 24 KB
    __shared__ __attribute__((aligned(16))) float sred[3][3][64];      // waves 1..3: [wave][sum, sum of squares, pivot][column]
    __shared__ __attribute__((aligned(16))) float xcoef[2][64];        // xbn: s, t per input channel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const bool with_bn = xbn.mean != nullptr;
    if (with_bn && tid < 64) {
        const BnAffine a = bn_affine(xbn.mean[tid], xbn.var[tid], xbn.gamma[tid], xbn.beta[tid], xbn.eps);
        xcoef[0][tid] = a.s, xcoef[1][tid] = a.t;
    }
    // B[k = in][n = out] = W[k][n]: lane (n = 32 nt + i, k group h) of k-step s holds W[16 s + 8 h .. + 7][n]
    for (int f = tid; f < 2 * 4 * 64; f += 256) {
        const int l = f & 63, s4 = (f >> 6) & 3, nt = f >> 8;
        const float* src = W + (size_t)(16 * s4 + 8 * (l >> 5)) * 64 + 32 * nt + (l & 31);
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = src[(size_t)q * 64];
        bf16x8 p0, p1, p2;
        split8x3(v, p0, p1, p2);
        Wf[nt][s4][0][l] = __builtin_bit_cast(u32x4, p0);
        Wf[nt][s4][1][l] = __builtin_bit_cast(u32x4, p1);
        Wf[nt][s4][2][l] = __builtin_bit_cast(u32x4, p2);
    }
    __syncthreads();
    // column statistics shifted by a PIVOT (the wave's first row of that column): sums of (v - p), (v - p)^2 stay at the scale of
    // the column's spread whatever its mean; waves are rebased onto wave 0's pivot when they meet
    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, piv[2] = {0.f, 0.f};
    const float b0 = bias ? bias[i] : 0.f, b1 = bias ? bias[32 + i] : 0.f;
    // every tile's rows are requested before the first is used (a wave is alone on its SIMD in these grids: 1152 waves on 1024
    // SIMDs -- nothing else covers its memory latency)
    float4 xr[LS_TILES_PER_WAVE][4][2];
#pragma unroll
    for (int t = 0; t < LS_TILES_PER_WAVE; ++t) {
        const int row = min(blockIdx.x * LS_ROWS_PER_WG + (wave * LS_TILES_PER_WAVE + t) * 32 + i, rows - 1);
        const size_t o = (size_t)row * 64 + 8 * h;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            xr[t][s4][0] = *reinterpret_cast<const float4*>(x + o + 16 * s4);
            xr[t][s4][1] = *reinterpret_cast<const float4*>(x + o + 16 * s4 + 4);
        }
    }
#pragma unroll
    for (int t = 0; t < LS_TILES_PER_WAVE; ++t) {
        const int base = blockIdx.x * LS_ROWS_PER_WG + (wave * LS_TILES_PER_WAVE + t) * 32;   // wave-uniform
        if (base >= rows) break;
        const bool ok = base + i < rows;
        bf16x8 a0[4], a1[4], a2[4];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            float4 u0 = xr[t][s4][0], u1 = xr[t][s4][1];
            float v[8] = {u0.x, u0.y, u0.z, u0.w, u1.x, u1.y, u1.z, u1.w};
            if (with_bn) {   // channels 16 s4 + 8 h .. + 7 of the row: the forward's own expression (bn_value), then the ReLU
                const float4 c0 = *reinterpret_cast<const float4*>(&xcoef[0][16 * s4 + 8 * h]), c1 = *reinterpret_cast<const float4*>(&xcoef[0][16 * s4 + 8 * h + 4]);
                const float4 t0 = *reinterpret_cast<const float4*>(&xcoef[1][16 * s4 + 8 * h]), t1 = *reinterpret_cast<const float4*>(&xcoef[1][16 * s4 + 8 * h + 4]);
                const float cs[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w}, ct[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q] * cs[q] + ct[q], 0.f);
            }
            if (!ok) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = 0.f;
            }
            split8x3(v, a0[s4], a1[s4], a2[s4]);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const bf16x8 w0 = __builtin_bit_cast(bf16x8, Wf[nt][s4][0][lane]), w1 = __builtin_bit_cast(bf16x8, Wf[nt][s4][1][lane]),
                             w2 = __builtin_bit_cast(bf16x8, Wf[nt][s4][2][lane]);
                acc = mfma_bf16(a2[s4], w0, acc);   // smallest terms first (gemm_split_kernel's order)
                acc = mfma_bf16(a0[s4], w2, acc);
                acc = mfma_bf16(a1[s4], w1, acc);
                acc = mfma_bf16(a1[s4], w0, acc);
                acc = mfma_bf16(a0[s4], w1, acc);
                acc = mfma_bf16(a0[s4], w0, acc);
            }
            const float bv = nt ? b1 : b0;
            if (t == 0) piv[nt] = __shfl(acc[0], i);   // row `base` of the wave's first tile (mfma_row(0, 0) = 0: lanes h = 0), always < rows
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = base + mfma_row(r, h);
                if (rr < rows) {
                    const float v = acc[r];     // (statistics of the product WITHOUT the bias: the finish adds it to the mean)
                    const float d = v - piv[nt];
                    s1[nt] += d;
                    s2[nt] += d * d;
                    z[(size_t)rr * 64 + 32 * nt + i] = v + bv;
                }
            }
        }
    }
    // fixed order: registers (above), lane halves, waves 0..3 (each rebased onto wave 0's pivot:
    // sum (v - p0) = sum (v - pw) + n (pw - p0),  sum (v - p0)^2 = sum (v - pw)^2 + 2 (pw - p0) sum (v - pw) + n (pw - p0)^2)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        s1[nt] += __shfl_xor(s1[nt], 32);
        s2[nt] += __shfl_xor(s2[nt], 32);
    }
    if (wave > 0 && h == 0) {
        sred[wave - 1][0][i] = s1[0], sred[wave - 1][0][32 + i] = s1[1];
        sred[wave - 1][1][i] = s2[0], sred[wave - 1][1][32 + i] = s2[1];
        sred[wave - 1][2][i] = piv[0], sred[wave - 1][2][32 + i] = piv[1];
    }
    __syncthreads();
    if (wave == 0 && h == 0) {
        const int wg_rows = min(LS_ROWS_PER_WG, rows - (int)blockIdx.x * LS_ROWS_PER_WG);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int c = 32 * nt + i;
            float t1 = s1[nt], t2 = s2[nt];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float nw = (float)max(0, min(32 * LS_TILES_PER_WAVE, wg_rows - w * 32 * LS_TILES_PER_WAVE));   // the wave's valid rows
                const float dp = nw > 0.f ? sred[w - 1][2][c] - piv[nt] : 0.f;
                t1 += sred[w - 1][0][c] + nw * dp;
                t2 += sred[w - 1][1][c] + (2.0f * dp) * sred[w - 1][0][c] + nw * dp * dp;
            }
            stats[((size_t)blockIdx.x * 3 + 0) * 64 + c] = t1;
            stats[((size_t)blockIdx.x * 3 + 1) * 64 + c] = t2;
            stats[((size_t)blockIdx.x * 3 + 2) * 64 + c] = piv[nt];
        }
    }
}


// ----------------------------------------------------------------------------------------------------------------
// Column reductions over the rows of (rows, C) tensors: every workgroup reduces a 256-row x 64-column panel (float4 per lane,
// 16 row groups) to a partial; colreduce_finish_kernel (16 columns per workgroup: 64 groups take every 64th partial each
// and meet in LDS in group order) adds the partials in a fixed order and writes the result -- deterministic.  (Until round 3
// the workgroup that finished a column panel LAST did this itself behind a counter: one launch less, but its tail -- an
// atomic round trip and five rounds of dependent uncached loads -- was half of the kernel's 20 us on the 64-channel layers.)
// The workspace keeps its layout (CR_COUNTERS unused words, then the partials); nothing in it needs to be zero any more.
//   kind 0: sum_r x                         -> out0 = scale * sum                      (bias gradients)
//   kind 1: sum_r (x - x0), sum_r (x - x0)^2 with x0 = row 0 of the column (one pass; the shift keeps the second
//           moment free of cancellation)   -> out0 = mean, out1 = POPULATION variance (tf.nn.moments)
//   kind 2: sum_r dyr, sum_r dyr * zhat, dyr = dy * [BN(z) > 0 or no relu], zhat = (z - mean) * rstd
//                                           -> out0 = dbeta, out1 = dgamma            (BatchNorm backward)
// C must be a multiple of 4 (every BatchNorm site of the network has 64, 256 or 1024 channels).
// ----------------------------------------------------------------------------------------------------------------
#ifndef CR_ROWS
#define CR_ROWS 256      // rows per workgroup
#endif
#define CR_COUNTERS 64   // counter slots at the head of the workspace: C <= 4096


template <int KIND>
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                        const float* __restrict__ mean, const float* __restrict__ var,
                                                        const float* __restrict__ gamma, const float* __restrict__ beta,
                                                        float eps, int relu, int rows, int C, float scale,
                                                        float* __restrict__ partial) {
    constexpr int NQ = KIND == 0 ? 1 : 2;
    __shared__ float red[NQ][16][64];
    __shared__ __attribute__((aligned(16))) float coef[4][64];   // kind 2: s, t (ReLU mask), mean, rstd per column
    const int tid = threadIdx.x, l16 = tid & 15, rg = tid >> 4;
    // grid = (column panels, row panels): the column panels of one row panel are neighbours in dispatch order, so a 4-KB row
    // of a 1024-wide tensor is read by workgroups that run together (row panels fastest left every row to be visited by 16
    // workgroups at 16 different times: 2.7 TB/s on conv5's sums)
    const int bx = blockIdx.y, by = blockIdx.x;
    const int c = by * 64 + 4 * l16;
    const int r0 = bx * CR_ROWS, r1 = min(rows, r0 + CR_ROWS);
    const int nb = gridDim.y;
    if (KIND == 2) {   // the square roots and divisions once per column, not once per thread
        if (tid < 64 && by * 64 + tid < C) {
            const int cc = by * 64 + tid;
            const BnAffine a = bn_affine(mean[cc], var[cc], gamma[cc], beta[cc], eps);
            coef[0][tid] = a.s, coef[1][tid] = a.t, coef[2][tid] = mean[cc], coef[3][tid] = 1.0f / sqrtf(var[cc] + eps);
        }
        __syncthreads();
    }
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        float sh[4] = {0.f, 0.f, 0.f, 0.f}, mu[4], rs[4];
        BnAffine af[4];
        if (KIND == 1) {
            const float4 v = *reinterpret_cast<const float4*>(x + c);
            sh[0] = v.x, sh[1] = v.y, sh[2] = v.z, sh[3] = v.w;
        }
        if (KIND == 2) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                af[q].s = coef[0][4 * l16 + q], af[q].t = coef[1][4 * l16 + q];
                mu[q] = coef[2][4 * l16 + q], rs[q] = coef[3][4 * l16 + q];
            }
        }
#pragma unroll 4
        for (int r = r0 + rg; r < r1; r += 16) {
            const size_t o = (size_t)r * C + c;
            const float4 v = *reinterpret_cast<const float4*>(x + o);
            const float in[4] = {v.x, v.y, v.z, v.w};
            if (KIND == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) s0[q] += in[q];
            } else if (KIND == 1) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float d = in[q] - sh[q];
                    s0[q] += d;
                    s1[q] += d * d;
                }
            } else {
                const float4 g = *reinterpret_cast<const float4*>(dy + o);
                const float gd[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float d = (relu && !(bn_value(in[q], af[q]) > 0.f)) ? 0.f : gd[q];
                    s0[q] += d;
                    s1[q] += d * ((in[q] - mu[q]) * rs[q]);
                }
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        red[0][rg][4 * l16 + q] = s0[q];
        if (NQ == 2) red[1][rg][4 * l16 + q] = s1[q];
    }
    __syncthreads();
    if (tid < 64 * NQ) {
        const int q = tid >> 6, l = tid & 63;
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[q][g][l];
        if (by * 64 + l < C) partial[((size_t)q * nb + bx) * C + by * 64 + l] = t;
    }
}

// 16 columns per workgroup (4 float4 columns x 64 groups): the partials come from other XCDs (L2 misses), and several small
// workgroups on several CUs draw them faster than one large one (moments_finalize_kernel).
#define CF_COLS 16
template <int KIND>
__global__ __launch_bounds__(256) void colreduce_finish_kernel(const float* __restrict__ partial, const float* __restrict__ x,
                                                               int nb, int C, float scale, float* __restrict__ out0,
                                                               float* __restrict__ out1) {
    constexpr int NQ = KIND == 0 ? 1 : 2;
    __shared__ float red[NQ][64][CF_COLS];
    const int tid = threadIdx.x, l4 = tid & 3, rg = tid >> 2;
    const int c = blockIdx.x * CF_COLS + 4 * l4;
    float t0[4] = {0.f, 0.f, 0.f, 0.f}, t1[4] = {0.f, 0.f, 0.f, 0.f};
    if (c < C) {
        const float* p0 = partial + c;
        const float* p1 = partial + (size_t)nb * C + c;
#pragma unroll 5
        for (int b = rg; b < nb; b += 64) {
            const float4 u0 = *reinterpret_cast<const float4*>(p0 + (size_t)b * C);
            t0[0] += u0.x, t0[1] += u0.y, t0[2] += u0.z, t0[3] += u0.w;
            if (NQ == 2) {
                const float4 u1 = *reinterpret_cast<const float4*>(p1 + (size_t)b * C);
                t1[0] += u1.x, t1[1] += u1.y, t1[2] += u1.z, t1[3] += u1.w;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        red[0][rg][4 * l4 + q] = t0[q];
        if (NQ == 2) red[NQ - 1][rg][4 * l4 + q] = t1[q];
    }
    __syncthreads();
    const int l = tid % CF_COLS, qn = tid / CF_COLS;
    const int cc = blockIdx.x * CF_COLS + l;
    float a = 0.f;
    if (qn < NQ && cc < C) {
#pragma unroll 16
        for (int g = 0; g < 64; ++g) a += red[qn][g][l];
    }
    __syncthreads();
    if (qn < NQ && cc < C) red[qn][0][l] = a;
    __syncthreads();
    if (tid < CF_COLS && cc < C) {
        const float a0 = red[0][0][l], a1 = NQ == 2 ? red[NQ - 1][0][l] : 0.f;
        if (KIND == 0) {
            out0[cc] = a0 * scale;
        } else if (KIND == 1) {
            const float m1 = a0 * scale;                       // E[x - x0]
            out0[cc] = x[cc] + m1;
            out1[cc] = fmaxf(a1 * scale - m1 * m1, 0.f);
        } else {
            out0[cc] = a0;
            out1[cc] = a1;
        }
    }
}

extern "C" size_t epc_colreduce_workspace_bytes(int rows, int C) {
    const size_t nb = (rows + CR_ROWS - 1) / CR_ROWS;
    return CR_COUNTERS * sizeof(unsigned int) + 3 * nb * (size_t)C * sizeof(float);   // (linear_stats64: sum, sum of squares, pivot)
}

static int colreduce_check(const char* who, int rows, int C, const void* workspace, size_t workspace_bytes) {
    EPC_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0 && C <= 64 * CR_COUNTERS, "C must be a multiple of 4, at most 4096");
    EPC_CHECK_ARG(workspace && (reinterpret_cast<size_t>(workspace) & 15) == 0, "workspace must be 16-byte aligned");
    if (workspace_bytes < epc_colreduce_workspace_bytes(rows, C)) {
        epc_set_error(who);
        return EPC_ENOMEM;
    }
    return EPC_OK;
}

// y = x W + b for a 64 -> 64 layer TOGETHER with the batch moments of y (linear_stats64_kernel + the finish).
// `workspace`: the column-reduction workspace (epc_colreduce_workspace_bytes(rows, 64)).
static int linear_stats64_impl(const float* x, BnParams xbn, const float* W, const float* bias, int rows, float* z, float* mean,
                               float* var, void* workspace, size_t workspace_bytes, void* stream);

extern "C" int epc_linear_stats64(const float* x, const float* W, const float* bias, int rows, float* z, float* mean, float* var,
                                  void* workspace, size_t workspace_bytes, void* stream) {
    return linear_stats64_impl(x, BnParams{nullptr, nullptr, nullptr, nullptr, 0.f}, W, bias, rows, z, mean, var, workspace,
                               workspace_bytes, stream);
}

// The same with the layer's input formed on the fly: x_in = relu(batch_norm(x_pre)) with the given batch moments and affine
// parameters of the layer that produced x_pre (conv_a -> conv_b inside a block: the activation between them is never written).
extern "C" int epc_linear_stats64_bn(const float* x_pre, const float* in_mean, const float* in_var, const float* in_gamma,
                                     const float* in_beta, float eps, const float* W, const float* bias, int rows, float* z,
                                     float* mean, float* var, void* workspace, size_t workspace_bytes, void* stream) {
    EPC_CHECK_ARG(in_mean && in_var && in_gamma && in_beta, "null pointer");
    return linear_stats64_impl(x_pre, BnParams{in_mean, in_var, in_gamma, in_beta, eps}, W, bias, rows, z, mean, var, workspace,
                               workspace_bytes, stream);
}

static int linear_stats64_impl(const float* x, BnParams xbn, const float* W, const float* bias, int rows, float* z, float* mean,
                               float* var, void* workspace, size_t workspace_bytes, void* stream) {
    EPC_CHECK_ARG(x && W && z && mean && var, "null pointer");
    EPC_CHECK_ARG(((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(z)) & 15) == 0, "x and z must be 16-byte aligned");
    if (int rc = colreduce_check("epc_linear_stats64: workspace too small", rows, 64, workspace, workspace_bytes)) return rc;
    unsigned int* counters = (unsigned int*)workspace;
    float* part = (float*)(counters + CR_COUNTERS);
    const int wgs = (rows + LS_ROWS_PER_WG - 1) / LS_ROWS_PER_WG;   // <= the 256-row panels the workspace is sized for
    hipLaunchKernelGGL(linear_stats64_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, x, xbn, W, bias, rows, z, part);
    launch_moments_finalize(part, wgs, 64, rows, LS_ROWS_PER_WG, bias, mean, var, (hipStream_t)stream);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// mean[c], var[c] (population) over `rows` rows of x (rows, C).  tf.nn.moments (utils/tf_util.py:472).
extern "C" int epc_col_moments(const float* x, int rows, int C, float* mean, float* var, void* workspace,
                               size_t workspace_bytes, void* stream) {
    EPC_CHECK_ARG(x && mean && var, "null pointer");
    if (int rc = colreduce_check("epc_col_moments: workspace too small", rows, C, workspace, workspace_bytes)) return rc;
    const int nb = (rows + CR_ROWS - 1) / CR_ROWS;
    unsigned int* counters = (unsigned int*)workspace;
    float* part = (float*)(counters + CR_COUNTERS);
    hipLaunchKernelGGL(colreduce_kernel<1>, dim3((C + 63) / 64, nb), dim3(256), 0, (hipStream_t)stream, x, nullptr, nullptr,
                       nullptr, nullptr, nullptr, 0.f, 0, rows, C, 1.0f / rows, part);
    hipLaunchKernelGGL(colreduce_finish_kernel<1>, dim3((C + CF_COLS - 1) / CF_COLS), dim3(256), 0, (hipStream_t)stream, part, x, nb, C,
                       1.0f / rows, mean, var);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// out[c] = sum over rows of x[:, c]  (bias gradients)
extern "C" int epc_col_sum(const float* x, int rows, int C, float* out, void* workspace, size_t workspace_bytes,
                           void* stream) {
    EPC_CHECK_ARG(x && out, "null pointer");
    if (int rc = colreduce_check("epc_col_sum: workspace too small", rows, C, workspace, workspace_bytes)) return rc;
    const int nb = (rows + CR_ROWS - 1) / CR_ROWS;
    unsigned int* counters = (unsigned int*)workspace;
    float* part = (float*)(counters + CR_COUNTERS);
    hipLaunchKernelGGL(colreduce_kernel<0>, dim3((C + 63) / 64, nb), dim3(256), 0, (hipStream_t)stream, x, nullptr, nullptr,
                       nullptr, nullptr, nullptr, 0.f, 0, rows, C, 1.0f, part);
    hipLaunchKernelGGL(colreduce_finish_kernel<0>, dim3((C + CF_COLS - 1) / CF_COLS), dim3(256), 0, (hipStream_t)stream, part, x, nb, C, 1.0f,
                       out, (float*)nullptr);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// y = act(z*s + t), s = gamma*rsqrt(var+eps), t = beta - mean*s  (tf.nn.batch_normalization, utils/tf_util.py:490).
// Panel kernels like the reductions above (256 rows x 64 columns per workgroup, float4 per lane): the per-column
// coefficients -- a square root and a division each -- are computed by 64 threads once per panel and shared through
// LDS.  (One thread per element with the coefficients recomputed in place made the backward VALU-bound: 445 us on the
// conv5 activations instead of 300.)
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                           const float* __restrict__ var, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, int relu, int rows,
                                                           int C, const float* __restrict__ addend, float* __restrict__ y) {
    // addend (optional, (rows, C)): y = act(bn(z)) + addend -- the block's residual, out = t + x1 (models/epc-net.py:86)
    __shared__ __attribute__((aligned(16))) float coef[2][64];
    const int tid = threadIdx.x, l16 = tid & 15, rg = tid >> 4;
    if (tid < 64 && blockIdx.x * 64 + tid < C) {
        const int cc = blockIdx.x * 64 + tid;
        const BnAffine a = bn_affine(mean[cc], var[cc], gamma[cc], beta[cc], eps);
        coef[0][tid] = a.s, coef[1][tid] = a.t;
    }
    __syncthreads();
    const int c = blockIdx.x * 64 + 4 * l16;
    if (c >= C) return;
    BnAffine af[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) af[q].s = coef[0][4 * l16 + q], af[q].t = coef[1][4 * l16 + q];
    const int r0 = blockIdx.y * CR_ROWS, r1 = min(rows, r0 + CR_ROWS);
#pragma unroll 4
    for (int r = r0 + rg; r < r1; r += 16) {
        const size_t o = (size_t)r * C + c;
        const float4 v = *reinterpret_cast<const float4*>(z + o);
        const float in[4] = {v.x, v.y, v.z, v.w};
        float out[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float t = bn_value(in[q], af[q]);
            out[q] = relu ? fmaxf(t, 0.f) : t;
        }
        if (addend) {
            const float4 a = *reinterpret_cast<const float4*>(addend + o);
            out[0] += a.x, out[1] += a.y, out[2] += a.z, out[3] += a.w;
        }
        *reinterpret_cast<float4*>(y + o) = make_float4(out[0], out[1], out[2], out[3]);
    }
}

extern "C" int epc_bn_apply_fwd(const float* z, const float* mean, const float* var, const float* gamma,
                                const float* beta, float eps, int relu, int rows, int C, float* y, void* stream) {
    EPC_CHECK_ARG(z && mean && var && gamma && beta && y, "null pointer");
    EPC_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0, "C must be a multiple of 4");
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3((C + 63) / 64, (rows + CR_ROWS - 1) / CR_ROWS), dim3(256), 0,
                       (hipStream_t)stream, z, mean, var, gamma, beta, eps, relu, rows, C, (const float*)nullptr, y);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_bn_apply_add_fwd(const float* z, const float* mean, const float* var, const float* gamma,
                                    const float* beta, float eps, int relu, int rows, int C, const float* addend, float* y,
                                    void* stream) {
    EPC_CHECK_ARG(z && mean && var && gamma && beta && addend && y, "null pointer");
    EPC_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0, "C must be a multiple of 4");
    hipLaunchKernelGGL(bn_apply_fwd_kernel, dim3((C + 63) / 64, (rows + CR_ROWS - 1) / CR_ROWS), dim3(256), 0,
                       (hipStream_t)stream, z, mean, var, gamma, beta, eps, relu, rows, C, addend, y);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// conv5's tail in one pass (models/epc-net.py:136-148): f = l2_normalize(relu(bn(z)), over the channels) and rn, the
// reciprocal norm, for C == 1024.  The BatchNorm output itself is never materialised: its backward recomputes the mask
// from z, and everything downstream (assignment product, aggregation, row-norm backward) takes f.  16 rows per workgroup
// (4 per wave), the 1024 (s, t) pairs computed once per workgroup into LDS.
#define BRN_C 1024
#define BRN_ROWS 16
__global__ __launch_bounds__(256) void bn_relu_rownorm_fwd_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                                  const float* __restrict__ var,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, float eps, int rows,
                                                                  float* __restrict__ f, float* __restrict__ rn_out) {
    __shared__ __attribute__((aligned(16))) float cs[BRN_C], ct[BRN_C];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < BRN_C; c += 256) {
        const BnAffine a = bn_affine(mean[c], var[c], gamma[c], beta[c], eps);
        cs[c] = a.s, ct[c] = a.t;
    }
    __syncthreads();
    for (int q = 0; q < BRN_ROWS / 4; ++q) {
        const int row = blockIdx.x * BRN_ROWS + wave * (BRN_ROWS / 4) + q;
        if (row >= rows) return;
        const float* pz = z + (size_t)row * BRN_C;
        float y[16];
        float ss = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = 256 * u + 4 * lane;
            const float4 v = *reinterpret_cast<const float4*>(pz + c);
            const float4 s4 = *reinterpret_cast<const float4*>(cs + c), t4 = *reinterpret_cast<const float4*>(ct + c);
            BnAffine a;
            a.s = s4.x, a.t = t4.x, y[4 * u + 0] = fmaxf(bn_value(v.x, a), 0.f);
            a.s = s4.y, a.t = t4.y, y[4 * u + 1] = fmaxf(bn_value(v.y, a), 0.f);
            a.s = s4.z, a.t = t4.z, y[4 * u + 2] = fmaxf(bn_value(v.z, a), 0.f);
            a.s = s4.w, a.t = t4.w, y[4 * u + 3] = fmaxf(bn_value(v.w, a), 0.f);
#pragma unroll
            for (int e = 0; e < 4; ++e) ss += y[4 * u + e] * y[4 * u + e];
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) ss += __shfl_xor(ss, off);
        const float rn = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        float* pf = f + (size_t)row * BRN_C;
#pragma unroll
        for (int u = 0; u < 4; ++u)
            *reinterpret_cast<float4*>(pf + 256 * u + 4 * lane) =
                make_float4(y[4 * u] * rn, y[4 * u + 1] * rn, y[4 * u + 2] * rn, y[4 * u + 3] * rn);
        if (lane == 0) rn_out[row] = rn;
    }
}

extern "C" int epc_bn_relu_rownorm_fwd(const float* z, const float* mean, const float* var, const float* gamma,
                                       const float* beta, float eps, int rows, int C, float* f, float* rn, void* stream) {
    EPC_CHECK_ARG(z && mean && var && gamma && beta && f && rn, "null pointer");
    EPC_CHECK_ARG(rows > 0 && C == BRN_C, "implemented for the 1024 channels of conv5");
    hipLaunchKernelGGL(bn_relu_rownorm_fwd_kernel, dim3((rows + BRN_ROWS - 1) / BRN_ROWS), dim3(256), 0, (hipStream_t)stream,
                       z, mean, var, gamma, beta, eps, rows, f, rn);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// Backward of conv5's tail, f = l2_normalize(relu(bn(z))) (models/epc-net.py:136-148), in TWO passes over (df, z) -- the
// separate operators (row-norm backward, BatchNorm sums, BatchNorm apply) made three, with the (rows, 1024) intermediate
// written and read twice and the saved f read once: 2.4 GB per step against 1.5 GB here.  Neither f nor the row-norm's input
// gradient exists in memory: both passes recompute u = relu(bn(z)) and f = u * rn from z with the forward's own expressions.
//   pass 1: per row  t = sum_c df f  (rowdot, kept for pass 2);  d = [u > 0] rn (df - f t)   (rn df where the norm was clamped)
//           per column  dbeta = sum_r d,  dgamma = sum_r d zhat -- one partial per workgroup, added in ascending order
//   pass 2: dz = gamma rstd (d - dbeta / rows - zhat dgamma / rows)
// A wave owns a row at a time: lane l holds columns 256 u + 4 l .. + 3 (four float4 per operand: 1 KB per wave-instruction), so
// the row's dot product is in-wave and the column sums are in-lane over the wave's rows.
// ----------------------------------------------------------------------------------------------------------------
#define BRB_ROWS 64   // rows per workgroup (16 per wave)

__global__ __launch_bounds__(256) void bn_relu_rownorm_bwd_sums_kernel(
    const float* __restrict__ df, const float* __restrict__ z, const float* __restrict__ rn_in, const float* __restrict__ mean,
    const float* __restrict__ var, const float* __restrict__ gamma, const float* __restrict__ beta, float eps, int rows,
    float* __restrict__ rowdot, float* __restrict__ partial) {
    __shared__ __attribute__((aligned(16))) float lds[4][BRN_C];   // coefficients s, t, mean, rstd; then the waves' sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < BRN_C; c += 256) {
        const BnAffine a = bn_affine(mean[c], var[c], gamma[c], beta[c], eps);
        lds[0][c] = a.s, lds[1][c] = a.t, lds[2][c] = mean[c], lds[3][c] = 1.0f / sqrtf(var[c] + eps);
    }
    __syncthreads();
    float cs[16], ct[16], mu[16], rs[16];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 256 * u + 4 * lane + e;
            cs[4 * u + e] = lds[0][c], ct[4 * u + e] = lds[1][c], mu[4 * u + e] = lds[2][c], rs[4 * u + e] = lds[3][c];
        }
    float s0[16], s1[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) s0[k] = s1[k] = 0.f;
    const int row0 = blockIdx.x * BRB_ROWS + wave;
    for (int q = 0; q < BRB_ROWS / 4; ++q) {
        const int row = row0 + 4 * q;          // wave-uniform
        if (row >= rows) break;
        const float4* g4 = reinterpret_cast<const float4*>(df + (size_t)row * BRN_C);
        const float4* z4 = reinterpret_cast<const float4*>(z + (size_t)row * BRN_C);
        float4 gv[4], zv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) gv[u] = g4[lane + 64 * u], zv[u] = z4[lane + 64 * u];
        const float rn = rn_in[row];
        float g[16], zz[16], f[16];
        float t = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            g[4 * u] = gv[u].x, g[4 * u + 1] = gv[u].y, g[4 * u + 2] = gv[u].z, g[4 * u + 3] = gv[u].w;
            zz[4 * u] = zv[u].x, zz[4 * u + 1] = zv[u].y, zz[4 * u + 2] = zv[u].z, zz[4 * u + 3] = zv[u].w;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * u + e;
                const float uu = fmaxf(zz[k] * cs[k] + ct[k], 0.f);   // bn_value: the forward's expression
                f[k] = uu * rn;
                t += g[k] * f[k];
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off);
        if (lane == 0) rowdot[row] = t;
        const bool clamped = rn >= 0.99e6f;   // sum u^2 <= 1e-12: f = u * 1e6, no projection term (rownorm_kernel)
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float du = clamped ? g[k] * rn : rn * (g[k] - f[k] * t);
            const float d = f[k] > 0.f ? du : 0.f;       // f > 0 exactly where bn(z) > 0 (rn > 0)
            s0[k] += d;
            s1[k] += d * ((zz[k] - mu[k]) * rs[k]);
        }
    }
    __syncthreads();   // the coefficients are in registers: the LDS rows now carry the waves' sums
    float* red = &lds[0][0];   // [wave][1024], one quantity per round
#pragma unroll
    for (int qn = 0; qn < 2; ++qn) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
            *reinterpret_cast<float4*>(red + wave * BRN_C + 256 * u + 4 * lane) =
                qn ? make_float4(s1[4 * u], s1[4 * u + 1], s1[4 * u + 2], s1[4 * u + 3])
                   : make_float4(s0[4 * u], s0[4 * u + 1], s0[4 * u + 2], s0[4 * u + 3]);
        __syncthreads();
        float* out = partial + ((size_t)blockIdx.x * 2 + qn) * BRN_C;
        for (int c = tid; c < BRN_C; c += 256)
            out[c] = (red[c] + red[BRN_C + c]) + (red[2 * BRN_C + c] + red[3 * BRN_C + c]);   // waves in a fixed order
        __syncthreads();
    }
}

// out[e] = sum over p < P of part[p][e] in a fixed order (64 groups take every 64th partial each and meet in group order):
// partial_sum_kernel for MANY partials (conv5's 1152 x 2048 floats), 1024 threads = 16 float4 columns x 64 groups.
__global__ __launch_bounds__(1024) void partial_sum_wide_kernel(const float* __restrict__ part, int P, int E,
                                                                float* __restrict__ out) {
    __shared__ float4 acc[64][16];
    const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int e4 = blockIdx.x * 16 + col;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e4 * 4 < E) {
        const float4* src = reinterpret_cast<const float4*>(part) + e4;
        const size_t stride = (size_t)E / 4;
        int p = grp;
        for (; p + 192 < P; p += 256) {
            const float4 a = src[(size_t)p * stride], b = src[(size_t)(p + 64) * stride], c = src[(size_t)(p + 128) * stride],
                         d = src[(size_t)(p + 192) * stride];
            s.x += a.x, s.y += a.y, s.z += a.z, s.w += a.w;
            s.x += b.x, s.y += b.y, s.z += b.z, s.w += b.w;
            s.x += c.x, s.y += c.y, s.z += c.z, s.w += c.w;
            s.x += d.x, s.y += d.y, s.z += d.z, s.w += d.w;
        }
        for (; p < P; p += 64) {
            const float4 a = src[(size_t)p * stride];
            s.x += a.x, s.y += a.y, s.z += a.z, s.w += a.w;
        }
    }
    acc[grp][col] = s;
    __syncthreads();
    if (grp == 0 && e4 * 4 < E) {
        float4 t = acc[0][col];
#pragma unroll 8
        for (int g = 1; g < 64; ++g) t.x += acc[g][col].x, t.y += acc[g][col].y, t.z += acc[g][col].z, t.w += acc[g][col].w;
        reinterpret_cast<float4*>(out)[e4] = t;
    }
}

__global__ __launch_bounds__(256) void bn_relu_rownorm_bwd_apply_kernel(
    const float* __restrict__ df, const float* __restrict__ z, const float* __restrict__ rn_in,
    const float* __restrict__ rowdot, const float* __restrict__ mean, const float* __restrict__ var,
    const float* __restrict__ gamma, const float* __restrict__ beta, const float* __restrict__ dbeta,
    const float* __restrict__ dgamma, float eps, float inv_rows, int rows, float* __restrict__ dz) {
    __shared__ __attribute__((aligned(16))) float lds[6][BRN_C];   // s, t, mean, k1, dbeta / rows, rstd dgamma / rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int c = tid; c < BRN_C; c += 256) {
        const float m = mean[c], r = 1.0f / sqrtf(var[c] + eps), ga = gamma[c];
        const BnAffine a = bn_affine(m, var[c], ga, beta[c], eps);
        lds[0][c] = a.s, lds[1][c] = a.t, lds[2][c] = m;
        lds[3][c] = ga * r, lds[4][c] = dbeta[c] * inv_rows, lds[5][c] = r * (dgamma[c] * inv_rows);
    }
    __syncthreads();
    float cs[16], ct[16], mu[16], k1[16], bb[16], gg[16];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 256 * u + 4 * lane + e, k = 4 * u + e;
            cs[k] = lds[0][c], ct[k] = lds[1][c], mu[k] = lds[2][c], k1[k] = lds[3][c], bb[k] = lds[4][c], gg[k] = lds[5][c];
        }
    const int row0 = blockIdx.x * BRB_ROWS + wave;
    for (int q = 0; q < BRB_ROWS / 4; ++q) {
        const int row = row0 + 4 * q;
        if (row >= rows) break;
        const float4* g4 = reinterpret_cast<const float4*>(df + (size_t)row * BRN_C);
        const float4* z4 = reinterpret_cast<const float4*>(z + (size_t)row * BRN_C);
        float4 gv[4], zv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) gv[u] = g4[lane + 64 * u], zv[u] = z4[lane + 64 * u];
        const float rn = rn_in[row], t = rowdot[row];
        const bool clamped = rn >= 0.99e6f;
        float4* o4 = reinterpret_cast<float4*>(dz + (size_t)row * BRN_C);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float g[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w}, zz[4] = {zv[u].x, zv[u].y, zv[u].z, zv[u].w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * u + e;
                const float uu = fmaxf(zz[e] * cs[k] + ct[k], 0.f);
                const float f = uu * rn;
                const float du = clamped ? g[e] * rn : rn * (g[e] - f * t);
                const float d = f > 0.f ? du : 0.f;
                o[e] = k1[k] * (d - bb[k] - (zz[e] - mu[k]) * gg[k]);
            }
            o4[lane + 64 * u] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// library-internal: out[e] = sum over the P partials of part[p][e], e < E (E a multiple of 64), in partial_sum_wide_kernel's fixed order
int epc_partial_sum_wide_launch(const float* partials, int P, int E, float* out, void* stream) {
    hipLaunchKernelGGL(partial_sum_wide_kernel, dim3(E / 4 / 16), dim3(1024), 0, (hipStream_t)stream, partials, P, E, out);
    return EPC_OK;
}

extern "C" size_t epc_bn_relu_rownorm_bwd_partial_floats(int rows) {
    return rows > 0 ? (size_t)((rows + BRB_ROWS - 1) / BRB_ROWS) * 2 * BRN_C : 0;
}

extern "C" int epc_bn_relu_rownorm_bwd(const float* df, const float* z, const float* rn, const float* mean, const float* var,
                                       const float* gamma, const float* beta, float eps, int rows, int C, float* dz,
                                       float* dbeta_dgamma, float* rowdot, float* partials, size_t partial_floats,
                                       void* stream) {
    EPC_CHECK_ARG(df && z && rn && mean && var && gamma && beta && dz && dbeta_dgamma && rowdot && partials, "null pointer");
    EPC_CHECK_ARG(rows > 0 && C == BRN_C, "implemented for the 1024 channels of conv5");
    EPC_CHECK_ARG(partial_floats >= epc_bn_relu_rownorm_bwd_partial_floats(rows), "partials buffer too small");
    hipStream_t st = (hipStream_t)stream;
    const int wgs = (rows + BRB_ROWS - 1) / BRB_ROWS;
    hipLaunchKernelGGL(bn_relu_rownorm_bwd_sums_kernel, dim3(wgs), dim3(256), 0, st, df, z, rn, mean, var, gamma, beta, eps, rows,
                       rowdot, partials);
    hipLaunchKernelGGL(partial_sum_wide_kernel, dim3(2 * BRN_C / 4 / 16), dim3(1024), 0, st, partials, wgs, 2 * BRN_C,
                       dbeta_dgamma);
    hipLaunchKernelGGL(bn_relu_rownorm_bwd_apply_kernel, dim3(wgs), dim3(256), 0, st, df, z, rn, rowdot, mean, var, gamma, beta,
                       dbeta_dgamma, dbeta_dgamma + BRN_C, eps, 1.0f / rows, rows, dz);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// dz = gamma*rstd * (dyr - dbeta/rows - zhat * dgamma/rows); the ReLU mask is recomputed from z with the forward's
// own expression (bn_value), so the forward output is neither stored for it nor read here.
__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                           const float* __restrict__ mean, const float* __restrict__ var,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ dbeta,
                                                           const float* __restrict__ dgamma, float eps, float inv_rows,
                                                           int relu, int rows, int C, float* __restrict__ dz) {
    // per column: s, t (mask), mean, k1 = gamma*rstd, b = dbeta/rows, g = rstd*dgamma/rows:  dz = k1*(d - b - (z-mean)*g)
    __shared__ __attribute__((aligned(16))) float coef[6][64];
    const int tid = threadIdx.x, l16 = tid & 15, rg = tid >> 4;
    if (tid < 64 && blockIdx.x * 64 + tid < C) {
        const int cc = blockIdx.x * 64 + tid;
        const float mu = mean[cc], rs = 1.0f / sqrtf(var[cc] + eps), ga = gamma[cc];
        const BnAffine a = bn_affine(mu, var[cc], ga, beta[cc], eps);
        coef[0][tid] = a.s, coef[1][tid] = a.t, coef[2][tid] = mu;
        coef[3][tid] = ga * rs, coef[4][tid] = dbeta[cc] * inv_rows, coef[5][tid] = rs * (dgamma[cc] * inv_rows);
    }
    __syncthreads();
    const int c = blockIdx.x * 64 + 4 * l16;
    if (c >= C) return;
    BnAffine af[4];
    float mu[4], k1[4], bb[4], gg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        af[q].s = coef[0][4 * l16 + q], af[q].t = coef[1][4 * l16 + q], mu[q] = coef[2][4 * l16 + q];
        k1[q] = coef[3][4 * l16 + q], bb[q] = coef[4][4 * l16 + q], gg[q] = coef[5][4 * l16 + q];
    }
    const int r0 = blockIdx.y * CR_ROWS, r1 = min(rows, r0 + CR_ROWS);
#pragma unroll 4
    for (int r = r0 + rg; r < r1; r += 16) {
        const size_t o = (size_t)r * C + c;
        const float4 zv = *reinterpret_cast<const float4*>(z + o), gv = *reinterpret_cast<const float4*>(dy + o);
        const float zi[4] = {zv.x, zv.y, zv.z, zv.w}, gi[4] = {gv.x, gv.y, gv.z, gv.w};
        float out[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float d = (relu && !(bn_value(zi[q], af[q]) > 0.f)) ? 0.f : gi[q];
            out[q] = k1[q] * (d - bb[q] - (zi[q] - mu[q]) * gg[q]);
        }
        *reinterpret_cast<float4*>(dz + o) = make_float4(out[0], out[1], out[2], out[3]);
    }
}

// Backward of training-mode BN (+ReLU).  Outputs dz (rows,C), dgamma (C), dbeta (C).  Two launches: the column sums
// (dbeta, dgamma), then dz.
extern "C" int epc_bn_apply_bwd(const float* dy, const float* z, const float* mean, const float* var,
                                const float* gamma, const float* beta, float eps, int relu, int rows, int C, float* dz,
                                float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream) {
    EPC_CHECK_ARG(dy && z && mean && var && gamma && beta && dz && dgamma && dbeta, "null pointer");
    if (int rc = colreduce_check("epc_bn_apply_bwd: workspace too small", rows, C, workspace, workspace_bytes)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int nb = (rows + CR_ROWS - 1) / CR_ROWS;
    unsigned int* counters = (unsigned int*)workspace;
    float* part = (float*)(counters + CR_COUNTERS);
    const dim3 grid(nb, (C + 63) / 64);
    hipLaunchKernelGGL(colreduce_kernel<2>, dim3(grid.y, grid.x), dim3(256), 0, st, z, dy, mean, var, gamma, beta, eps, relu, rows, C, 1.0f,
                       part);
    hipLaunchKernelGGL(colreduce_finish_kernel<2>, dim3((C + CF_COLS - 1) / CF_COLS), dim3(256), 0, st, part, z, nb, C, 1.0f, dbeta, dgamma);
    hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3(grid.y, grid.x), dim3(256), 0, st, dy, z, mean, var, gamma, beta, dbeta, dgamma, eps,
                       1.0f / rows, relu, rows, C, dz);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// bn_apply_bwd_kernel without a mask for conv5's 1024 channels in the shape of bn_relu_rownorm_bwd_apply_kernel: a wave takes whole
// 4-KB rows (the 64-column panels of the general kernel read 256-byte row segments: 189 us against 165 at 18 x 4096 rows).
__global__ __launch_bounds__(256) void bn_apply_bwd_given_wide_kernel(const float* __restrict__ dy, const float* __restrict__ z,
                                                                      const float* __restrict__ mean, const float* __restrict__ var,
                                                                      const float* __restrict__ gamma, const float* __restrict__ dbeta,
                                                                      const float* __restrict__ dgamma, float eps, float inv_rows,
                                                                      int rows, float* __restrict__ dz) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float mu[16], k1[16], bb[16], gg[16];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 256 * u + 4 * lane + e, k = 4 * u + e;
            const float r = 1.0f / sqrtf(var[c] + eps);
            mu[k] = mean[c], k1[k] = gamma[c] * r, bb[k] = dbeta[c] * inv_rows, gg[k] = r * (dgamma[c] * inv_rows);
        }
    const int row0 = blockIdx.x * BRB_ROWS + wave;
    for (int q = 0; q < BRB_ROWS / 4; ++q) {
        const int row = row0 + 4 * q;
        if (row >= rows) break;
        const float4* g4 = reinterpret_cast<const float4*>(dy + (size_t)row * BRN_C);
        const float4* z4 = reinterpret_cast<const float4*>(z + (size_t)row * BRN_C);
        float4 gv[4], zv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) gv[u] = g4[lane + 64 * u], zv[u] = z4[lane + 64 * u];
        float4* o4 = reinterpret_cast<float4*>(dz + (size_t)row * BRN_C);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float g[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w}, zz[4] = {zv[u].x, zv[u].y, zv[u].z, zv[u].w};
            float o[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int k = 4 * u + e;
                o[e] = k1[k] * (g[e] - bb[k] - (zz[e] - mu[k]) * gg[k]);
            }
            o4[lane + 64 * u] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// The last step of a BatchNorm backward whose column sums are known already (epc_vlad_df_tail leaves them):
// dz = gamma rstd (dy - dbeta / rows - zhat dgamma / rows), no ReLU mask (dy is masked).  dz may be dy (in place).
extern "C" int epc_bn_apply_bwd_given(const float* dy, const float* z, const float* mean, const float* var, const float* gamma,
                                      const float* beta, const float* dbeta, const float* dgamma, float eps, int rows, int C,
                                      float* dz, void* stream) {
    EPC_CHECK_ARG(dy && z && mean && var && gamma && beta && dbeta && dgamma && dz, "null pointer");
    EPC_CHECK_ARG(rows > 0 && C > 0 && C % 4 == 0, "C must be a multiple of 4");
    const int nb = (rows + CR_ROWS - 1) / CR_ROWS;
    if (C == BRN_C)
        hipLaunchKernelGGL(bn_apply_bwd_given_wide_kernel, dim3((rows + BRB_ROWS - 1) / BRB_ROWS), dim3(256), 0, (hipStream_t)stream, dy,
                           z, mean, var, gamma, dbeta, dgamma, eps, 1.0f / rows, rows, dz);
    else
        hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3((C + 63) / 64, nb), dim3(256), 0, (hipStream_t)stream, dy, z, mean, var, gamma, beta,
                           dbeta, dgamma, eps, 1.0f / rows, 0, rows, C, dz);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// Backward of a 64 -> 64 layer + training-mode BatchNorm (+ReLU) in ONE pass over its rows (utils/tf_util.py:94-106 seen
// from the gradient side).  The thin layers of the backbone (conv*_a, conv*_b, conv2..4: 11 of them) each took four launches
// -- BatchNorm sums, dz, dX = dz W^T, dW = x^T dz -- that are bound by launch structure, not by bytes (a read + write of one
// (rows, 64) tensor takes 5 us on this device; the four launches took ~80).  After the column sums (dbeta, dgamma: they need
// every row) one kernel does the rest: a wave takes 32 rows at a time and
//   * forms dz = gamma rstd (dy [z-mask] - dbeta/rows - zhat dgamma/rows) in registers, in the row layout (lane = row, eight
//     consecutive channels per k-step) -- which IS the B operand of dx^T = W dz^T: dx leaves as whole float4s per lane;
//   * forms dz again in the column layout (lane = channel, eight consecutive ROWS per k-step: eight coalesced dword loads) next to
//     x in the same layout: the operands of dW = x^T dz with the rows as K.  dz is never written;
//   * keeps its 64 x 64 dW partial in registers over its rows; the four waves meet in LDS (fixed order) and the workgroup
//     stores ONE partial.  partial_sum_kernel adds the partials in ascending order: dW is the same bits on every run.
// GEMM arithmetic: two bf16 pieces per operand, three products (that of epc_gemm_f32_fast, the other backward GEMMs).
// ----------------------------------------------------------------------------------------------------------------
#ifndef LB_TILES_PER_WAVE
#define LB_TILES_PER_WAVE 2
#endif
#define LB_ROWS_PER_WG (4 * 32 * LB_TILES_PER_WAVE)

#ifndef LB_COLUMN_LOADS
// dW = x^T dz needs both operands with the ROWS as k: lane = channel, eight consecutive rows per k-step.  The first form of this
// kernel (kept under -DLB_COLUMN_LOADS) read x, dy and z a second time in that layout -- 96 lane-coalesced dword loads per
// 32-row tile next to the 16 float4 loads of the row layout -- and was bound by issuing them (29 us per layer).  Here every
// tensor is read ONCE, in the row layout; the bf16 pieces of x and of dz go through a per-wave LDS image [row][channel]
// (128-B rows, 16-B chunks XOR-swizzled) and come back transposed by ds_read_b64_tr_b16: lane 4q + p of a 16-lane group
// addresses row r0 + q, channels c0 + 4p .. + 3; lane i receives channel c0 + i of the four rows.  One image (8 KB: hi + lo)
// per wave serves x, then dz.
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
#define LB_IMG_BYTES 4096   // one piece: 32 rows x 128 B
__device__ __forceinline__ int lb_img_off(int row, int chunk) {   // byte offset of 16-byte chunk `chunk` (0..7) of row `row`
    return 128 * row + 16 * (chunk ^ (((row >> 1) & 1) << 2) ^ (((row >> 2) & 1) << 1));
}
// the A / B fragment (k = rows 16 s2 + 8 h .. + 7, m or n = channel 32 t + (lane & 31)) of one piece, read transposed
__device__ __forceinline__ bf16x8 lb_tr_frag(const char* img, int t, int s2, int lane) {
    const int g16 = lane >> 4, l16 = lane & 15, q = l16 >> 2, pp = l16 & 3;
    const int chunk = 4 * t + 2 * (g16 & 1) + (pp >> 1);
    const int r0 = 16 * s2 + 8 * (g16 >> 1);
    typedef __attribute__((address_space(3))) s16x4* lds_ptr;
    const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + lb_img_off(r0 + q, chunk) + 8 * (pp & 1)));
    const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_ptr)(img + lb_img_off(r0 + 4 + q, chunk) + 8 * (pp & 1)));
    const s16x8 v = {lo4[0], lo4[1], lo4[2], lo4[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(256, 2) void linear_bn_bwd64_kernel(
    const float* __restrict__ dy, const float* __restrict__ z, const float* __restrict__ x, const float* __restrict__ W,
    const float* __restrict__ mean, const float* __restrict__ var, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ dbeta, const float* __restrict__ dgamma, float eps,
    float inv_rows, int relu, int rows, float* __restrict__ dx, float* __restrict__ dWpart, BnParams xbn,
    const float* __restrict__ dx_addend) {
    // xbn: the layer's input was relu(bn(x)) of the previous layer's pre-activation x (epc_linear_stats64_bn): the dW operand is
    // formed the same way as it is loaded.  dx_addend (rows, 64): dx leaves as W dz + addend (the gradient that reaches the same
    // tensor by the block's residual path, so that the neighbour backward gathers ONE tensor).
    __shared__ __attribute__((aligned(16))) float coef[6][64];          // s, t (mask), mean, k1, dbeta/rows, rstd dgamma/rows
    __shared__ __attribute__((aligned(16))) float xcoef[2][64];
    __shared__ u32x4 Wf[2][4][2][64];                                    // W as A fragments: [in tile][k-step][hi, lo][lane]
    __shared__ __attribute__((aligned(16))) char img[4][2 * LB_IMG_BYTES];   // per wave: hi + lo image; at the end the parked dW partials
    static_assert(sizeof(img) >= 2 * 4 * 16 * 64 * sizeof(float), "the parked partials alias the images");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    if (tid < 64) {
        const float mu = mean[tid], rs = 1.0f / sqrtf(var[tid] + eps), ga = gamma[tid];
        const BnAffine a = bn_affine(mu, var[tid], ga, beta[tid], eps);
        coef[0][tid] = a.s, coef[1][tid] = a.t, coef[2][tid] = mu;
        coef[3][tid] = ga * rs, coef[4][tid] = dbeta[tid] * inv_rows, coef[5][tid] = rs * (dgamma[tid] * inv_rows);
        if (xbn.mean) {
            const BnAffine xa = bn_affine(xbn.mean[tid], xbn.var[tid], xbn.gamma[tid], xbn.beta[tid], xbn.eps);
            xcoef[0][tid] = xa.s, xcoef[1][tid] = xa.t;
        }
    }
    // W (in, out) row-major: A[m = in][k = out]; lane (m = 32 mt + i, k group h) of k-step s holds W[m][16 s + 8 h .. + 7]
    for (int f = tid; f < 2 * 4 * 64; f += 256) {
        const int l = f & 63, s4 = (f >> 6) & 3, mt = f >> 8;
        const float* src = W + (size_t)(32 * mt + (l & 31)) * 64 + 16 * s4 + 8 * (l >> 5);
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        bf16x8 ph, pl;
        split8(v, ph, pl);
        Wf[mt][s4][0][l] = __builtin_bit_cast(u32x4, ph);
        Wf[mt][s4][1][l] = __builtin_bit_cast(u32x4, pl);
    }
    __syncthreads();

    f32x16 accW[2][2];   // [in tile mt][out tile nt]: register 4g + e = in channel 32 mt + 8 g + 4 h + e, lane = out channel 32 nt + i
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) accW[mt][nt][r] = 0.f;

    char* my = img[wave];
    auto ld8 = [&](const float* p, float (&v)[8]) {
        const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
        v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
    };
    auto c8 = [&](const float* c, float (&v)[8]) {   // eight consecutive per-channel coefficients from LDS
        const float4 a = *reinterpret_cast<const float4*>(c), b = *reinterpret_cast<const float4*>(c + 4);
        v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w;
    };
    auto put = [&](int s4, bf16x8 ph, bf16x8 pl) {   // the lane's row i, channels 16 s4 + 8 h .. + 7 = chunk 2 s4 + h
        const int o = lb_img_off(i, 2 * s4 + h);
        *reinterpret_cast<u32x4*>(my + o) = __builtin_bit_cast(u32x4, ph);
        *reinterpret_cast<u32x4*>(my + LB_IMG_BYTES + o) = __builtin_bit_cast(u32x4, pl);
    };

#pragma unroll 1
    for (int t = 0; t < LB_TILES_PER_WAVE; ++t) {
        const int base = blockIdx.x * LB_ROWS_PER_WG + (wave * LB_TILES_PER_WAVE + t) * 32;   // wave-uniform
        if (base >= rows) break;
        const int row = base + i;
        const bool ok = row < rows;
        const size_t o = (size_t)(ok ? row : 0) * 64 + 8 * h;
        // ---- dz (row layout): B fragments of dx^T = W dz^T as they stand; its bf16 pieces also go to the image ----
        bf16x8 zh[4], zl[4];
        {
            float gv[4][8], zv[4][8];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) ld8(dy + o + 16 * s4, gv[s4]), ld8(z + o + 16 * s4, zv[s4]);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                asm volatile("" ::: "memory");   // (keeps a k-step's coefficient reads next to their use: hoisted, they took 190 registers)
                float cs[8], ct[8], mu[8], k1[8], bb[8], gg[8], dzv[8];
                const int c0 = 16 * s4 + 8 * h;
                c8(&coef[0][c0], cs), c8(&coef[1][c0], ct), c8(&coef[2][c0], mu), c8(&coef[3][c0], k1), c8(&coef[4][c0], bb),
                    c8(&coef[5][c0], gg);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float d = (relu && !(zv[s4][q] * cs[q] + ct[q] > 0.f)) ? 0.f : gv[s4][q];   // the forward's own expression
                    const float v = k1[q] * (d - bb[q] - (zv[s4][q] - mu[q]) * gg[q]);
                    dzv[q] = ok ? v : 0.f;
                }
                split8(dzv, zh[s4], zl[s4]);
                put(s4, zh[s4], zl[s4]);
            }
        }
        // x's rows are requested now: they land under the dx products
        float xv[4][8];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) ld8(x + o + 16 * s4, xv[s4]);
        if (dx) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const bf16x8 wh = __builtin_bit_cast(bf16x8, Wf[mt][s4][0][lane]), wl = __builtin_bit_cast(bf16x8, Wf[mt][s4][1][lane]);
                    acc = mfma_bf16(wl, zh[s4], acc);
                    acc = mfma_bf16(wh, zl[s4], acc);
                    acc = mfma_bf16(wh, zh[s4], acc);
                }
                if (ok) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float4 v = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
                        if (dx_addend) {
                            const float4 a = *reinterpret_cast<const float4*>(dx_addend + (size_t)row * 64 + 32 * mt + 8 * g + 4 * h);
                            v.x += a.x, v.y += a.y, v.z += a.z, v.w += a.w;
                        }
                        *reinterpret_cast<float4*>(dx + (size_t)row * 64 + 32 * mt + 8 * g + 4 * h) = v;
                    }
                }
            }
        }
        // ---- dz^T: the B fragments of dW (k = rows, n = out channel), read transposed from the image ----
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (one wave: its own image writes have landed before its reads)
        bf16x8 dh[2][2], dl[2][2];   // [out tile][k-step]
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) dh[nt][s2] = lb_tr_frag(my, nt, s2, lane), dl[nt][s2] = lb_tr_frag(my + LB_IMG_BYTES, nt, s2, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // they are in registers: the image may be overwritten
        // ---- x (row layout) -> bf16 pieces -> image -> A fragments of dW (k = rows, m = in channel) ----
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            asm volatile("" ::: "memory");
            if (xbn.mean) {
                float cs[8], ct[8];
                c8(&xcoef[0][16 * s4 + 8 * h], cs), c8(&xcoef[1][16 * s4 + 8 * h], ct);
#pragma unroll
                for (int q = 0; q < 8; ++q) xv[s4][q] = fmaxf(xv[s4][q] * cs[q] + ct[q], 0.f);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) xv[s4][q] = ok ? xv[s4][q] : 0.f;   // (a select, not a branch)
            bf16x8 ph, pl;
            split8(xv[s4], ph, pl);
            put(s4, ph, pl);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 xh = lb_tr_frag(my, mt, s2, lane), xl = lb_tr_frag(my + LB_IMG_BYTES, mt, s2, lane);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    accW[mt][nt] = mfma_bf16(xl, dh[nt][s2], accW[mt][nt]);
                    accW[mt][nt] = mfma_bf16(xh, dl[nt][s2], accW[mt][nt]);
                    accW[mt][nt] = mfma_bf16(xh, dh[nt][s2], accW[mt][nt]);
                }
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // before the next tile overwrites the image
    }
    // ---- the four waves' partials meet pairwise, ((w0 + w1) + (w2 + w3)): a fixed order; wave 0 stores the workgroup's ----
    __syncthreads();   // every wave is done with its image: the parked partials alias them
    float (*red)[4][16][64] = reinterpret_cast<float (*)[4][16][64]>(&img[0][0]);   // [slot][tile][register][lane]
    auto park = [&](int slot) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[slot][mt * 2 + nt][r][lane] = accW[mt][nt][r];
    };
    auto take = [&](int slot) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) accW[mt][nt][r] += red[slot][mt * 2 + nt][r][lane];
    };
    if (wave & 1) park(wave >> 1);
    __syncthreads();
    if (!(wave & 1)) take(wave >> 1);
    __syncthreads();
    if (wave == 2) park(0);
    __syncthreads();
    if (wave == 0) {
        take(0);
        float* out = dWpart + (size_t)blockIdx.x * 4096;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) out[(32 * mt + mfma_row(r, h)) * 64 + 32 * nt + i] = accW[mt][nt][r];
    }
}
#else
__global__ __launch_bounds__(256) void linear_bn_bwd64_kernel(
    const float* __restrict__ dy, const float* __restrict__ z, const float* __restrict__ x, const float* __restrict__ W,
    const float* __restrict__ mean, const float* __restrict__ var, const float* __restrict__ gamma,
    const float* __restrict__ beta, const float* __restrict__ dbeta, const float* __restrict__ dgamma, float eps,
    float inv_rows, int relu, int rows, float* __restrict__ dx, float* __restrict__ dWpart, BnParams xbn,
    const float* __restrict__ dx_addend) {
    // xbn: the layer's input was relu(bn(x)) of the previous layer's pre-activation x (epc_linear_stats64_bn): the dW operand is
    // formed the same way as it is loaded.  dx_addend (rows, 64): dx leaves as W dz + addend (the gradient that reaches the same
    // tensor by the block's residual path, so that the neighbour backward gathers ONE tensor).
    __shared__ __attribute__((aligned(16))) float coef[6][64];          // s, t (mask), mean, k1, dbeta/rows, rstd dgamma/rows
    __shared__ float xcoef[2][64];
    __shared__ u32x4 Wf[2][4][2][64];                                    // W as A fragments: [in tile][k-step][hi, lo][lane]
    __shared__ __attribute__((aligned(16))) float red[2][4][16][64];     // parked dW partials: [slot][tile][register][lane]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    if (tid < 64) {
        const float mu = mean[tid], rs = 1.0f / sqrtf(var[tid] + eps), ga = gamma[tid];
        const BnAffine a = bn_affine(mu, var[tid], ga, beta[tid], eps);
        coef[0][tid] = a.s, coef[1][tid] = a.t, coef[2][tid] = mu;
        coef[3][tid] = ga * rs, coef[4][tid] = dbeta[tid] * inv_rows, coef[5][tid] = rs * (dgamma[tid] * inv_rows);
        if (xbn.mean) {
            const BnAffine xa = bn_affine(xbn.mean[tid], xbn.var[tid], xbn.gamma[tid], xbn.beta[tid], xbn.eps);
            xcoef[0][tid] = xa.s, xcoef[1][tid] = xa.t;
        }
    }
    // W (in, out) row-major: A[m = in][k = out]; lane (m = 32 mt + i, k group h) of k-step s holds W[m][16 s + 8 h .. + 7]
    for (int f = tid; f < 2 * 4 * 64; f += 256) {
        const int l = f & 63, s4 = (f >> 6) & 3, mt = f >> 8;
        const float* src = W + (size_t)(32 * mt + (l & 31)) * 64 + 16 * s4 + 8 * (l >> 5);
        const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
        bf16x8 ph, pl;
        split8(v, ph, pl);
        Wf[mt][s4][0][l] = __builtin_bit_cast(u32x4, ph);
        Wf[mt][s4][1][l] = __builtin_bit_cast(u32x4, pl);
    }
    __syncthreads();

    f32x16 accW[2][2];   // [in tile mt][out tile nt]: register 4g + e = in channel 32 mt + 8 g + 4 h + e, lane = out channel 32 nt + i
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) accW[mt][nt][r] = 0.f;

    auto dz_of = [&](float dyv, float zv, float cs, float ct, float mu, float k1, float bb, float gg) {
        const float d = (relu && !(zv * cs + ct > 0.f)) ? 0.f : dyv;   // the forward's own expression (bn_value)
        return k1 * (d - bb - (zv - mu) * gg);
    };

    for (int t = 0; t < LB_TILES_PER_WAVE; ++t) {
        const int base = blockIdx.x * LB_ROWS_PER_WG + (wave * LB_TILES_PER_WAVE + t) * 32;   // wave-uniform
        if (base >= rows) break;
        // ---- row layout: dz as B fragments (n = row base + i, k = out channel), dx^T = W dz^T ----
        if (dx) {
            const int row = base + i;
            const bool ok = row < rows;
            const size_t o = (size_t)(ok ? row : 0) * 64 + 8 * h;
            bf16x8 zh[4], zl[4];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const float4 g0 = *reinterpret_cast<const float4*>(dy + o + 16 * s4), g1 = *reinterpret_cast<const float4*>(dy + o + 16 * s4 + 4);
                const float4 z0 = *reinterpret_cast<const float4*>(z + o + 16 * s4), z1 = *reinterpret_cast<const float4*>(z + o + 16 * s4 + 4);
                const float gv[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
                const float zv[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
                float dzv[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int c = 16 * s4 + 8 * h + q;
                    const float v = dz_of(gv[q], zv[q], coef[0][c], coef[1][c], coef[2][c], coef[3][c], coef[4][c], coef[5][c]);
                    dzv[q] = ok ? v : 0.f;
                }
                split8(dzv, zh[s4], zl[s4]);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const bf16x8 wh = __builtin_bit_cast(bf16x8, Wf[mt][s4][0][lane]), wl = __builtin_bit_cast(bf16x8, Wf[mt][s4][1][lane]);
                    acc = mfma_bf16(wl, zh[s4], acc);
                    acc = mfma_bf16(wh, zl[s4], acc);
                    acc = mfma_bf16(wh, zh[s4], acc);
                }
                if (ok) {
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        float4 v = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
                        if (dx_addend) {
                            const float4 a = *reinterpret_cast<const float4*>(dx_addend + (size_t)row * 64 + 32 * mt + 8 * g + 4 * h);
                            v.x += a.x, v.y += a.y, v.z += a.z, v.w += a.w;
                        }
                        *reinterpret_cast<float4*>(dx + (size_t)row * 64 + 32 * mt + 8 * g + 4 * h) = v;
                    }
                }
            }
        }
        // ---- column layout: lane = channel, k = rows base + 16 s + 8 h + q;  dW += x^T dz ----
        bf16x8 xh[2][2], xl[2][2];   // [in tile][k-step]
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                // (loads from clamped rows, zeroed afterwards: a guarded load is an exec-masked branch with a full wait at its
                // join -- the 96 loads of this phase would go out one round trip at a time)
                float v[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = x[(size_t)min(base + 16 * s2 + 8 * h + q, rows - 1) * 64 + 32 * mt + i];
                if (xbn.mean) {
                    const float xs = xcoef[0][32 * mt + i], xt = xcoef[1][32 * mt + i];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q] * xs + xt, 0.f);
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (base + 16 * s2 + 8 * h + q < rows) ? v[q] : 0.f;
                split8(v, xh[mt][s2], xl[mt][s2]);
            }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int c = 32 * nt + i;
            const float cs = coef[0][c], ct = coef[1][c], mu = coef[2][c], k1 = coef[3][c], bb = coef[4][c], gg = coef[5][c];
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                float v[8], gy[8], gz[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const size_t o = (size_t)min(base + 16 * s2 + 8 * h + q, rows - 1) * 64 + c;
                    gy[q] = dy[o], gz[q] = z[o];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float d = dz_of(gy[q], gz[q], cs, ct, mu, k1, bb, gg);
                    v[q] = (base + 16 * s2 + 8 * h + q < rows) ? d : 0.f;
                }
                bf16x8 dh, dl;
                split8(v, dh, dl);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    accW[mt][nt] = mfma_bf16(xl[mt][s2], dh, accW[mt][nt]);
                    accW[mt][nt] = mfma_bf16(xh[mt][s2], dl, accW[mt][nt]);
                    accW[mt][nt] = mfma_bf16(xh[mt][s2], dh, accW[mt][nt]);
                }
            }
        }
    }
    // ---- the four waves' partials meet pairwise, ((w0 + w1) + (w2 + w3)): a fixed order; wave 0 stores the workgroup's ----
    auto park = [&](int slot) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) red[slot][mt * 2 + nt][r][lane] = accW[mt][nt][r];
    };
    auto take = [&](int slot) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) accW[mt][nt][r] += red[slot][mt * 2 + nt][r][lane];
    };
    if (wave & 1) park(wave >> 1);
    __syncthreads();
    if (!(wave & 1)) take(wave >> 1);
    __syncthreads();
    if (wave == 2) park(0);
    __syncthreads();
    if (wave == 0) {
        take(0);
        float* out = dWpart + (size_t)blockIdx.x * 4096;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) out[(32 * mt + mfma_row(r, h)) * 64 + 32 * nt + i] = accW[mt][nt][r];
    }
}

#endif   // LB_COLUMN_LOADS

// out[e] = sum over p < P of part[p][e], added in ascending p whatever the launch geometry (16 groups of a workgroup take
// every 16th partial each, their sums meet in LDS in group order: a fixed tree): the ordered counterpart of an atomic
// reduction.  E is a multiple of 4; one float4 column per (lane, group).
__global__ __launch_bounds__(256) void partial_sum_kernel(const float* __restrict__ part, int P, int E,
                                                          float* __restrict__ out) {
    __shared__ float4 acc[16][16];
    const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int e4 = blockIdx.x * 16 + col;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e4 * 4 < E) {
        const float4* src = reinterpret_cast<const float4*>(part) + e4;
        const size_t stride = (size_t)E / 4;
        int p = grp;
        for (; p + 48 < P; p += 64) {
            const float4 a = src[(size_t)p * stride], b = src[(size_t)(p + 16) * stride], c = src[(size_t)(p + 32) * stride],
                         d = src[(size_t)(p + 48) * stride];
            s.x += a.x, s.y += a.y, s.z += a.z, s.w += a.w;
            s.x += b.x, s.y += b.y, s.z += b.z, s.w += b.w;
            s.x += c.x, s.y += c.y, s.z += c.z, s.w += c.w;
            s.x += d.x, s.y += d.y, s.z += d.z, s.w += d.w;
        }
        for (; p < P; p += 16) {
            const float4 a = src[(size_t)p * stride];
            s.x += a.x, s.y += a.y, s.z += a.z, s.w += a.w;
        }
    }
    acc[grp][col] = s;
    __syncthreads();
    if (grp == 0 && e4 * 4 < E) {
        float4 t = acc[0][col];
#pragma unroll
        for (int g = 1; g < 16; ++g) t.x += acc[g][col].x, t.y += acc[g][col].y, t.z += acc[g][col].z, t.w += acc[g][col].w;
        reinterpret_cast<float4*>(out)[e4] = t;
    }
}

extern "C" size_t epc_linear_bn_bwd64_partial_floats(int rows) {
    return rows > 0 ? (size_t)((rows + LB_ROWS_PER_WG - 1) / LB_ROWS_PER_WG) * 4096 : 0;
}

static int linear_bn_bwd64_impl(const float* dy, const float* z, const float* x, BnParams xbn, const float* W, const float* mean,
                                const float* var, const float* gamma, const float* beta, float eps, int relu, int rows,
                                float* dx, const float* dx_addend, float* dW, float* dgamma, float* dbeta, void* workspace,
                                size_t workspace_bytes, float* dw_partials, size_t dw_partial_floats, void* stream) {
    EPC_CHECK_ARG(dy && z && x && W && mean && var && gamma && beta && dW && dgamma && dbeta && dw_partials, "null pointer");
    EPC_CHECK_ARG(dx || !dx_addend, "dx_addend without dx");
    if (int rc = colreduce_check("epc_linear_bn_bwd64: workspace too small", rows, 64, workspace, workspace_bytes)) return rc;
    EPC_CHECK_ARG(dw_partial_floats >= epc_linear_bn_bwd64_partial_floats(rows), "dW partial buffer too small (epc_linear_bn_bwd64_partial_floats)");
    hipStream_t st = (hipStream_t)stream;
    const int nb = (rows + CR_ROWS - 1) / CR_ROWS;
    unsigned int* counters = (unsigned int*)workspace;
    float* part = (float*)(counters + CR_COUNTERS);
    hipLaunchKernelGGL(colreduce_kernel<2>, dim3(1, nb), dim3(256), 0, st, z, dy, mean, var, gamma, beta, eps, relu, rows, 64, 1.0f,
                       part);
    hipLaunchKernelGGL(colreduce_finish_kernel<2>, dim3(64 / CF_COLS), dim3(256), 0, st, part, z, nb, 64, 1.0f, dbeta, dgamma);
    const int wgs = (rows + LB_ROWS_PER_WG - 1) / LB_ROWS_PER_WG;
    hipLaunchKernelGGL(linear_bn_bwd64_kernel, dim3(wgs), dim3(256), 0, st, dy, z, x, W, mean, var, gamma, beta, dbeta, dgamma, eps,
                       1.0f / rows, relu, rows, dx, dw_partials, xbn, dx_addend);
    hipLaunchKernelGGL(partial_sum_kernel, dim3(4096 / 4 / 16), dim3(256), 0, st, dw_partials, wgs, 4096, dW);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_linear_bn_bwd64(const float* dy, const float* z, const float* x, const float* W, const float* mean,
                                   const float* var, const float* gamma, const float* beta, float eps, int relu, int rows,
                                   float* dx, float* dW, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes,
                                   float* dw_partials, size_t dw_partial_floats, void* stream) {
    return linear_bn_bwd64_impl(dy, z, x, BnParams{nullptr, nullptr, nullptr, nullptr, 0.f}, W, mean, var, gamma, beta, eps, relu,
                                rows, dx, nullptr, dW, dgamma, dbeta, workspace, workspace_bytes, dw_partials, dw_partial_floats,
                                stream);
}

// The same for a layer inside a fused block: in_* (all four or none): the layer's input was relu(batch_norm(x)) of the previous
// layer's pre-activation x (epc_linear_stats64_bn) and is re-formed as x is loaded; dx_addend (optional, (rows, 64)): dx leaves
// as dz W^T + dx_addend.
extern "C" int epc_linear_bn_bwd64_ex(const float* dy, const float* z, const float* x, const float* in_mean, const float* in_var,
                                      const float* in_gamma, const float* in_beta, const float* W, const float* mean,
                                      const float* var, const float* gamma, const float* beta, float eps, int relu, int rows,
                                      float* dx, const float* dx_addend, float* dW, float* dgamma, float* dbeta, void* workspace,
                                      size_t workspace_bytes, float* dw_partials, size_t dw_partial_floats, void* stream) {
    const bool any = in_mean || in_var || in_gamma || in_beta, all = in_mean && in_var && in_gamma && in_beta;
    EPC_CHECK_ARG(any == all, "in_mean, in_var, in_gamma, in_beta: all four or none");
    return linear_bn_bwd64_impl(dy, z, x, BnParams{in_mean, in_var, in_gamma, in_beta, eps}, W, mean, var, gamma, beta, eps, relu,
                                rows, dx, dx_addend, dW, dgamma, dbeta, workspace, workspace_bytes, dw_partials, dw_partial_floats,
                                stream);
}

// ----------------------------------------------------------------------------------------------------------------
// conv1 of the training step (models/epc-net.py:66-69: 3 -> 64 on the coordinates): a K = 3 product is three FMAs per
// output, not a matrix-pipe job -- the general kernel pads K to a 32-deep tile (45 us forward, 40 us for dW over
// K = 73 728 rows with 256-way atomic split-K).  Forward: 16 lanes x 4 channels per row, one float4 store each.  dW (3, 64):
// a workgroup walks 1024 rows, every thread keeps its channel quad's 3 x 4 sums, the 16 row groups meet in LDS in order,
// one partial per workgroup, partial_sum_kernel adds the partials in ascending order (deterministic).  cin <= 4.
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void linear_smallk_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                                const float* __restrict__ bias, int rows, int cin, int cout,
                                                                float* __restrict__ z) {
    const int q4 = cout / 4;                       // channel quads per row
    const long t = (long)blockIdx.x * 256 + threadIdx.x;
    const long row = t / q4;
    const int c = (int)(t % q4) * 4;
    if (row >= rows) return;
    float4 acc = bias ? *reinterpret_cast<const float4*>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < cin; ++k) {
        const float xv = x[row * cin + k];
        const float4 w = *reinterpret_cast<const float4*>(W + (size_t)k * cout + c);
        acc.x = __builtin_fmaf(xv, w.x, acc.x), acc.y = __builtin_fmaf(xv, w.y, acc.y);
        acc.z = __builtin_fmaf(xv, w.z, acc.z), acc.w = __builtin_fmaf(xv, w.w, acc.w);
    }
    *reinterpret_cast<float4*>(z + row * cout + c) = acc;
}

#define SK_ROWS_PER_WG 256     // (round 6: 288 workgroups at 18 x 4096 rows instead of 72 -- 1024 rows per workgroup left two thirds of the CUs idle: 15.5 -> 6 us)
__global__ __launch_bounds__(256) void linear_smallk_dw_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                               int rows, int cin, float* __restrict__ part) {
    // cout == 64: thread = (row group rg of 16, channel quad l16 of 16)
    __shared__ float red[16][4][64];
    const int tid = threadIdx.x, l16 = tid & 15, rg = tid >> 4;
    const int r0 = blockIdx.x * SK_ROWS_PER_WG, r1 = min(rows, r0 + SK_ROWS_PER_WG);
    float4 s[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 4
    for (int r = r0 + rg; r < r1; r += 16) {
        const float4 g = *reinterpret_cast<const float4*>(dy + (size_t)r * 64 + 4 * l16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float xv = k < cin ? x[(size_t)r * cin + k] : 0.f;
            s[k].x = __builtin_fmaf(xv, g.x, s[k].x), s[k].y = __builtin_fmaf(xv, g.y, s[k].y);
            s[k].z = __builtin_fmaf(xv, g.z, s[k].z), s[k].w = __builtin_fmaf(xv, g.w, s[k].w);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<float4*>(&red[rg][k][4 * l16]) = s[k];
    __syncthreads();
    // 256 threads = 4 k x 64 channels: ordered sum over the 16 row groups
    const int k = tid >> 6, c = tid & 63;
    float t = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) t += red[g][k][c];
    if (k < cin) part[(size_t)blockIdx.x * (cin * 64) + k * 64 + c] = t;
}

extern "C" int epc_linear_smallk_fwd(const float* x, const float* W, const float* bias, int rows, int cin, int cout, float* z,
                                     void* stream) {
    EPC_CHECK_ARG(x && W && z, "null pointer");
    EPC_CHECK_ARG(rows > 0 && cin >= 1 && cin <= 4 && cout > 0 && cout % 4 == 0, "rows > 0, 1 <= cin <= 4, cout a multiple of 4");
    const long threads = (long)rows * (cout / 4);
    hipLaunchKernelGGL(linear_smallk_fwd_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, W,
                       bias, rows, cin, cout, z);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" size_t epc_linear_smallk_dw_partial_floats(int rows, int cin) {
    return rows > 0 ? (size_t)((rows + SK_ROWS_PER_WG - 1) / SK_ROWS_PER_WG) * cin * 64 : 0;
}

extern "C" int epc_linear_smallk_dw(const float* x, const float* dy, int rows, int cin, int cout, float* dW, float* partials,
                                    size_t partial_floats, void* stream) {
    EPC_CHECK_ARG(x && dy && dW && partials, "null pointer");
    EPC_CHECK_ARG(rows > 0 && cin >= 1 && cin <= 4 && cout == 64, "rows > 0, 1 <= cin <= 4, cout == 64");
    EPC_CHECK_ARG(partial_floats >= epc_linear_smallk_dw_partial_floats(rows, cin), "partial buffer too small");
    const int wgs = (rows + SK_ROWS_PER_WG - 1) / SK_ROWS_PER_WG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(linear_smallk_dw_kernel, dim3(wgs), dim3(256), 0, st, x, dy, rows, cin, partials);
    const int E = cin * 64;
    hipLaunchKernelGGL(partial_sum_kernel, dim3((E / 4 + 15) / 16), dim3(256), 0, st, partials, wgs, E, dW);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// Neighbour mean over the kNN index lists (64 channels): forward gather, backward scatter (f32 atomics).
// Rows with more than `cap` selected entries take the exact scan (same rule as the fused block kernel).
// ----------------------------------------------------------------------------------------------------------------
template <bool BWD>
__global__ __launch_bounds__(256) void neighbour_mean_kernel(const float* __restrict__ src, const float* __restrict__ xyz,
                                                             const int32_t* __restrict__ idx,
                                                             const int32_t* __restrict__ cnt,
                                                             const float* __restrict__ kth, int cap, int total_points,
                                                             int n, float kdiv, float* __restrict__ dst,
                                                             float* __restrict__ diff) {
    // (XCD-aware: consecutive workgroups -- consecutive points of a cloud, gathering from the same rows -- behind ONE L2: common.h)
    const int t = xcd_contiguous_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    const int g = t >> 4, q = t & 15;
    if (g >= total_points) return;
    const int cloud_base = (g / n) * n;
    const float4* s4 = reinterpret_cast<const float4*>(src);
    const int c = cnt[g];
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    float4 mine;
    if (BWD) {
        mine = s4[(size_t)g * 16 + q];
        mine.x /= kdiv, mine.y /= kdiv, mine.z /= kdiv, mine.w /= kdiv;
    }
    auto visit = [&](int j) {
        if (BWD) {
            float* d = dst + ((size_t)(cloud_base + j) * 16 + q) * 4;
            atomicAdd(d + 0, mine.x);
            atomicAdd(d + 1, mine.y);
            atomicAdd(d + 2, mine.z);
            atomicAdd(d + 3, mine.w);
        } else {
            const float4 v = s4[(size_t)(cloud_base + j) * 16 + q];
            acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
        }
    };
    if (c <= cap) {
        int m = 0;
        if (!BWD && c >= 20 && cap % 4 == 0) {
            // the first 20 entries (every ordinary row has at least 20): five int4 index loads, then all 20 row loads in flight
            // at once, added in ascending order -- the one-at-a-time loop below chains 20 dependent (index, row) round trips
            const int4* il = reinterpret_cast<const int4*>(idx + (size_t)g * cap);
            int nb[20];
#pragma unroll
            for (int m4 = 0; m4 < 5; ++m4) {
                const int4 tq = il[m4];
                nb[4 * m4] = tq.x, nb[4 * m4 + 1] = tq.y, nb[4 * m4 + 2] = tq.z, nb[4 * m4 + 3] = tq.w;
            }
            float4 v[20];
#pragma unroll
            for (int u = 0; u < 20; ++u) v[u] = s4[(size_t)(cloud_base + nb[u]) * 16 + q];
#pragma unroll
            for (int u = 0; u < 20; ++u) acc.x += v[u].x, acc.y += v[u].y, acc.z += v[u].z, acc.w += v[u].w;
            m = 20;
        }
        for (; m < c; ++m) visit(idx[(size_t)g * cap + m]);
    } else {
        const float* pc = xyz + (size_t)cloud_base * 3;
        const int i = g - cloud_base;
        const float xi = pc[3 * i], yi = pc[3 * i + 1], zi = pc[3 * i + 2];
        const float sqi = sq3(xi, yi, zi), kv = kth[g];
        for (int j = 0; j < n; ++j) {
            const float xj = pc[3 * j], yj = pc[3 * j + 1], zj = pc[3 * j + 2];
            if (neg_sq_dist(sqi, xi, yi, zi, xj, yj, zj, sq3(xj, yj, zj)) >= kv) visit(j);
        }
    }
    if (!BWD) {
        acc.x /= kdiv, acc.y /= kdiv, acc.z /= kdiv, acc.w /= kdiv;
        reinterpret_cast<float4*>(dst)[(size_t)g * 16 + q] = acc;
        if (diff) {   // xm - x (models/epc-net.py:72), written by the same launch
            const float4 own = s4[(size_t)g * 16 + q];
            reinterpret_cast<float4*>(diff)[(size_t)g * 16 + q] =
                make_float4(acc.x - own.x, acc.y - own.y, acc.z - own.z, acc.w - own.w);
        }
    }
}

extern "C" int epc_neighbour_mean_fwd(const float* x, const float* xyz, const int32_t* idx, const int32_t* cnt,
                                      const float* kth, int cap, int num_clouds, int n, int knn, float* xm,
                                      void* stream) {
    EPC_CHECK_ARG(x && xyz && idx && cnt && kth && xm, "null pointer");
    EPC_CHECK_ARG(num_clouds > 0 && n > 0 && knn > 0 && cap >= EPC_KNN_SELECT, "bad shape");
    const long total = (long)num_clouds * n;
    hipLaunchKernelGGL(neighbour_mean_kernel<false>, dim3((unsigned)((total * 16 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, x, xyz, idx, cnt, kth, cap, (int)total, n, (float)knn, xm, nullptr);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// xm as above and diff = xm - x (models/epc-net.py:70-72) in one launch
extern "C" int epc_neighbour_mean_diff_fwd(const float* x, const float* xyz, const int32_t* idx, const int32_t* cnt,
                                           const float* kth, int cap, int num_clouds, int n, int knn, float* xm,
                                           float* diff, void* stream) {
    EPC_CHECK_ARG(x && xyz && idx && cnt && kth && xm && diff, "null pointer");
    EPC_CHECK_ARG(num_clouds > 0 && n > 0 && knn > 0 && cap >= EPC_KNN_SELECT, "bad shape");
    const long total = (long)num_clouds * n;
    hipLaunchKernelGGL(neighbour_mean_kernel<false>, dim3((unsigned)((total * 16 + 255) / 256)), dim3(256), 0,
                       (hipStream_t)stream, x, xyz, idx, cnt, kth, cap, (int)total, n, (float)knn, xm, diff);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}


// ---- transposed graph: for every point j the list of points i whose neighbour list holds j ----------------------------
// The scatter above is bound by the f32 atomic rate (20 x 256-B atomics per point, 4 times per step); the graph is the
// same for every block of a step, so it is transposed ONCE (count -> per-cloud exclusive scan -> fill) and the backward
// becomes a gather like the forward.  Rows with more than `cap` entries (exact ties: duplicated / zero-padded clouds)
// are not listed; the scatter kernel adds their contributions afterwards.
// Layout: cloud c owns rlist[c*n*cap .. +n*cap); roff[j] = start of j's list (absolute), rdeg[j] = its length.
// (two points per wave -- 32 lanes each when the lists have at most 32 slots, half the waves -- was measured in round 6: count 20.8 -> 22.5 us,
// fill 26.7 -> 35.5 us at 18 x 4096; an LDS-segment rewrite, count + fill + sort per (cloud, 1024 targets) workgroup: 118 us against 89)
__global__ __launch_bounds__(256) void transpose_count_kernel(const int32_t* __restrict__ idx,
                                                              const int32_t* __restrict__ cnt, int cap, int total_points,
                                                              int n, int32_t* __restrict__ rdeg) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= total_points) return;
    const int c = cnt[g];
    if (c > cap) return;
    if (lane < c) atomicAdd(rdeg + (size_t)(g / n) * n + idx[(size_t)g * cap + lane], 1);
}

// one workgroup per cloud: roff = cloud base + exclusive scan of rdeg; cursor (fill positions) reset to 0
__global__ __launch_bounds__(1024) void transpose_scan_kernel(const int32_t* __restrict__ rdeg, int n, int cap,
                                                              int32_t* __restrict__ roff, int32_t* __restrict__ cursor) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int cloud = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += 1024) {
        const int j = j0 + tid;
        const int v = j < n ? rdeg[(size_t)cloud * n + j] : 0;
        int incl = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(incl, off);
            if (lane >= off) incl += t;
        }
        if (lane == 63) wsum[wave] = incl;
        __syncthreads();
        int before = carry;
        for (int w = 0; w < wave; ++w) before += wsum[w];
        if (j < n) {
            roff[(size_t)cloud * n + j] = cloud * n * cap + before + incl - v;
            cursor[(size_t)cloud * n + j] = 0;
        }
        __syncthreads();
        if (tid == 1023) carry = before + incl;
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void transpose_fill_kernel(const int32_t* __restrict__ idx, const int32_t* __restrict__ cnt,
                                                             int cap, int total_points, int n,
                                                             const int32_t* __restrict__ roff, int32_t* __restrict__ cursor,
                                                             int32_t* __restrict__ rlist) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= total_points) return;
    const int c = cnt[g];
    if (c > cap) return;
    if (lane < c) {
        const size_t j = (size_t)(g / n) * n + idx[(size_t)g * cap + lane];
        const int pos = atomicAdd(cursor + j, 1);
        rlist[roff[j] + pos] = g;   // absolute row of the contributing point
    }
}

// The fill's positions come from an atomic counter, so the ORDER of a list's entries varies from run to run -- and with it the
// order in which neighbour_gather_bwd_kernel adds the rows, i.e. the last bits of every gradient upstream.  Each list is
// therefore sorted ascending afterwards (entries are distinct rows): one wave per list, an entry per lane, rank = the number
// of smaller entries (64 readlanes); lists longer than 64 (hubs of clumped clouds) are insertion-sorted by one lane.
__global__ __launch_bounds__(256) void transpose_sort_kernel(const int32_t* __restrict__ rdeg, const int32_t* __restrict__ roff,
                                                             int total_points, int32_t* __restrict__ rlist) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= total_points) return;
    const int deg = rdeg[j];
    int32_t* lst = rlist + roff[j];
    if (deg <= 64) {
        const int mine = lane < deg ? lst[lane] : 0x7fffffff;
        int rank = 0;
        for (int u = 0; u < deg; ++u) rank += __builtin_amdgcn_readlane(mine, u) < mine ? 1 : 0;
        if (lane < deg) lst[rank] = mine;     // (every lane has read its entry before any lane writes: one wave, in order)
    } else if (lane == 0) {
        for (int a = 1; a < deg; ++a) {
            const int v = lst[a];
            int b = a - 1;
            while (b >= 0 && lst[b] > v) {
                lst[b + 1] = lst[b];
                --b;
            }
            lst[b + 1] = v;
        }
    }
}

extern "C" int epc_knn_transpose(const int32_t* idx, const int32_t* cnt, int cap, int num_clouds, int n, int32_t* rdeg,
                                 int32_t* roff, int32_t* cursor, int32_t* rlist, void* stream) {
    EPC_CHECK_ARG(idx && cnt && rdeg && roff && cursor && rlist, "null pointer");
    EPC_CHECK_ARG(num_clouds > 0 && n > 0 && cap >= EPC_KNN_SELECT && cap <= 64, "bad shape");
    const long total = (long)num_clouds * n;
    EPC_CHECK_ARG(total * cap < (1L << 31), "too many edges for 32-bit offsets");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(rdeg, 0, (size_t)total * sizeof(int32_t), st) != hipSuccess) {
        epc_set_error("epc_knn_transpose: hipMemsetAsync failed");
        return EPC_EHIP;
    }
    const unsigned blocks = (unsigned)((total + 3) / 4);
    hipLaunchKernelGGL(transpose_count_kernel, dim3(blocks), dim3(256), 0, st, idx, cnt, cap, (int)total, n, rdeg);
    hipLaunchKernelGGL(transpose_scan_kernel, dim3(num_clouds), dim3(1024), 0, st, rdeg, n, cap, roff, cursor);
    hipLaunchKernelGGL(transpose_fill_kernel, dim3(blocks), dim3(256), 0, st, idx, cnt, cap, (int)total, n, roff, cursor, rlist);
    hipLaunchKernelGGL(transpose_sort_kernel, dim3(blocks), dim3(256), 0, st, rdeg, roff, (int)total, rlist);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// gather form of the backward: one wave per point j, lane = channel; dx[j] = (sum over j's list of dxm[i]) / k
__global__ __launch_bounds__(256) void neighbour_gather_bwd_kernel(const float* __restrict__ dxm,
                                                                   const int32_t* __restrict__ rdeg,
                                                                   const int32_t* __restrict__ roff,
                                                                   const int32_t* __restrict__ rlist, int total_points,
                                                                   float kdiv, const float* __restrict__ ddiff,
                                                                   const float* __restrict__ own_minus,
                                                                   float* __restrict__ dx) {
    // ddiff (optional): the gradient of diff = xm - x of the fused forward.  Then the rows gathered are dxm + ddiff and
    // the point's own -ddiff is added:  dx[j] = (sum_i (dxm[i] + ddiff[i])) / k - ddiff[j]
    // 16 lanes x float4 per point (four points per wave, like the forward gather): a quarter of the load instructions of the
    // lane = channel form; eight list entries and their eight (sixteen) rows in flight per round, added in list order.
    const int t = xcd_contiguous_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;   // (XCD-aware, like the forward gather)
    const int j = t >> 4, q = t & 15;
    if (j >= total_points) return;
    const int deg = rdeg[j];
    const int32_t* lst = rlist + roff[j];
    const float4* s4 = reinterpret_cast<const float4*>(dxm);
    const float4* d4 = reinterpret_cast<const float4*>(ddiff);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    int m = 0;
    for (; m + 8 <= deg; m += 8) {
        int ii[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) ii[u] = lst[m + u];
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = s4[(size_t)ii[u] * 16 + q];
        if (ddiff) {
            float4 w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = d4[(size_t)ii[u] * 16 + q];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u].x += w[u].x, v[u].y += w[u].y, v[u].z += w[u].z, v[u].w += w[u].w;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) acc.x += v[u].x, acc.y += v[u].y, acc.z += v[u].z, acc.w += v[u].w;
    }
    for (; m < deg; ++m) {
        const size_t o = (size_t)lst[m] * 16 + q;
        float4 v = s4[o];
        if (ddiff) {
            const float4 w = d4[o];
            v.x += w.x, v.y += w.y, v.z += w.z, v.w += w.w;
        }
        acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
    float4 own = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ddiff) own = d4[(size_t)j * 16 + q];
    if (own_minus) {   // the gathered tensor is the SUM dxm + ddiff: the point's own ddiff = sum - dxm
        const float4 a = s4[(size_t)j * 16 + q], b = reinterpret_cast<const float4*>(own_minus)[(size_t)j * 16 + q];
        own = make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w);
    }
    reinterpret_cast<float4*>(dx)[(size_t)j * 16 + q] =
        make_float4(acc.x / kdiv - own.x, acc.y / kdiv - own.y, acc.z / kdiv - own.z, acc.w / kdiv - own.w);
}

// scatter restricted to the rows the transposed graph does not list (cnt > cap)
__global__ __launch_bounds__(256) void neighbour_scatter_overflow_kernel(const float* __restrict__ dxm,
                                                                         const float* __restrict__ xyz,
                                                                         const int32_t* __restrict__ cnt,
                                                                         const float* __restrict__ kth, int cap,
                                                                         int total_points, int n, float kdiv,
                                                                         const float* __restrict__ ddiff,
                                                                         float* __restrict__ dx) {
    const int g = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (g >= total_points || cnt[g] <= cap) return;
    const int cloud_base = (g / n) * n;
    const float v = (dxm[(size_t)g * 64 + lane] + (ddiff ? ddiff[(size_t)g * 64 + lane] : 0.f)) / kdiv;
    const float* pc = xyz + (size_t)cloud_base * 3;
    const int i = g - cloud_base;
    const float xi = pc[3 * i], yi = pc[3 * i + 1], zi = pc[3 * i + 2];
    const float sqi = sq3(xi, yi, zi), kv = kth[g];
    for (int j = 0; j < n; ++j) {
        const float xj = pc[3 * j], yj = pc[3 * j + 1], zj = pc[3 * j + 2];
        if (neg_sq_dist(sqi, xi, yi, zi, xj, yj, zj, sq3(xj, yj, zj)) >= kv)
            atomicAdd(dx + (size_t)(cloud_base + j) * 64 + lane, v);
    }
}

// dx[j] = sum_{i : j in nbr(i)} dxm[i] / k from the transposed graph (epc_knn_transpose); dx is overwritten.
extern "C" int epc_neighbour_mean_bwd_gather(const float* dxm, const float* xyz, const int32_t* cnt, const float* kth,
                                             int cap, const int32_t* rdeg, const int32_t* roff, const int32_t* rlist,
                                             int num_clouds, int n, int knn, float* dx, void* stream) {
    EPC_CHECK_ARG(dxm && xyz && cnt && kth && rdeg && roff && rlist && dx, "null pointer");
    EPC_CHECK_ARG(num_clouds > 0 && n > 0 && knn > 0 && cap >= EPC_KNN_SELECT && cap <= 64, "bad shape");
    const long total = (long)num_clouds * n;
    const unsigned blocks = (unsigned)((total + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(neighbour_gather_bwd_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, st, dxm, rdeg, roff, rlist, (int)total,
                       (float)knn, (const float*)nullptr, (const float*)nullptr, dx);
    hipLaunchKernelGGL(neighbour_scatter_overflow_kernel, dim3(blocks), dim3(256), 0, st, dxm, xyz, cnt, kth, cap,
                       (int)total, n, (float)knn, (const float*)nullptr, dx);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// Backward of epc_neighbour_mean_diff_fwd: dx = mask^T (dxm + ddiff) / k - ddiff, dx overwritten.
extern "C" int epc_neighbour_mean_diff_bwd_gather(const float* dxm, const float* ddiff, const float* xyz,
                                                  const int32_t* cnt, const float* kth, int cap, const int32_t* rdeg,
                                                  const int32_t* roff, const int32_t* rlist, int num_clouds, int n,
                                                  int knn, float* dx, void* stream) {
    EPC_CHECK_ARG(dxm && ddiff && xyz && cnt && kth && rdeg && roff && rlist && dx, "null pointer");
    EPC_CHECK_ARG(num_clouds > 0 && n > 0 && knn > 0 && cap >= EPC_KNN_SELECT && cap <= 64, "bad shape");
    const long total = (long)num_clouds * n;
    const unsigned blocks = (unsigned)((total + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(neighbour_gather_bwd_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, st, dxm, rdeg, roff, rlist, (int)total,
                       (float)knn, ddiff, (const float*)nullptr, dx);
    hipLaunchKernelGGL(neighbour_scatter_overflow_kernel, dim3(blocks), dim3(256), 0, st, dxm, xyz, cnt, kth, cap,
                       (int)total, n, (float)knn, ddiff, dx);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// The same from the SUM s = dxm + ddiff (written by epc_linear_bn_bwd64_ex with dx_addend = dxm): half the gathered bytes;
// dx = mask^T s / k - (s - dxm).
extern "C" int epc_neighbour_mean_diff_bwd_gather_sum(const float* s, const float* dxm, const float* xyz, const int32_t* cnt,
                                                      const float* kth, int cap, const int32_t* rdeg, const int32_t* roff,
                                                      const int32_t* rlist, int num_clouds, int n, int knn, float* dx,
                                                      void* stream) {
    EPC_CHECK_ARG(s && dxm && xyz && cnt && kth && rdeg && roff && rlist && dx, "null pointer");
    EPC_CHECK_ARG(num_clouds > 0 && n > 0 && knn > 0 && cap >= EPC_KNN_SELECT && cap <= 64, "bad shape");
    const long total = (long)num_clouds * n;
    const unsigned blocks = (unsigned)((total + 3) / 4);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(neighbour_gather_bwd_kernel, dim3((unsigned)((total + 15) / 16)), dim3(256), 0, st, s, rdeg, roff, rlist,
                       (int)total, (float)knn, (const float*)nullptr, dxm, dx);
    hipLaunchKernelGGL(neighbour_scatter_overflow_kernel, dim3(blocks), dim3(256), 0, st, s, xyz, cnt, kth, cap, (int)total, n,
                       (float)knn, (const float*)nullptr, dx);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// Row L2 normalisation (tf.nn.l2_normalize, eps 1e-12) over C channels: one wave per row.
//   fwd: y = x * rn, rn = rsqrt(max(sum x^2, eps));  bwd: dx = rn * (dy - y * sum(dy*y))   (0 where the clamp is active)
// ----------------------------------------------------------------------------------------------------------------
template <bool BWD>
__global__ __launch_bounds__(256) void rownorm_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                      const float* __restrict__ rn_in, int rows, int C,
                                                      float* __restrict__ out, float* __restrict__ rn_out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const float* pa = a + (size_t)row * C;
    float* po = out + (size_t)row * C;
    if (C == 1024) {
        // conv5's rows (models/epc-net.py:148 on 1024 channels): the lane's 16 values as four float4 loads per operand, all in
        // flight before the dot product, and kept for the second half -- one pass over the row instead of two, a quarter of
        // the load / store instructions.  (Its own fixed summation order: float4 groups instead of strided scalars.)
        const float4* a4 = reinterpret_cast<const float4*>(pa);
        const float4* b4 = BWD ? reinterpret_cast<const float4*>(b + (size_t)row * C) : nullptr;
        float4 va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            va[u] = a4[lane + 64 * u];
            if (BWD) vb[u] = b4[lane + 64 * u];
        }
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float4 o = BWD ? vb[u] : va[u];
            s += va[u].x * o.x, s += va[u].y * o.y, s += va[u].z * o.z, s += va[u].w * o.w;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        float4* o4 = reinterpret_cast<float4*>(po);
        if (!BWD) {
            const float rn = 1.0f / sqrtf(fmaxf(s, 1e-12f));
#pragma unroll
            for (int u = 0; u < 4; ++u) o4[lane + 64 * u] = make_float4(va[u].x * rn, va[u].y * rn, va[u].z * rn, va[u].w * rn);
            if (lane == 0) rn_out[row] = rn;
        } else {
            const float rn = rn_in[row];
            const bool clamped = rn >= 0.99e6f;  // sum x^2 <= 1e-12: y = x * 1e6, d/dx = 1e6 (no projection term)
#pragma unroll
            for (int u = 0; u < 4; ++u)
                o4[lane + 64 * u] = clamped ? make_float4(va[u].x * rn, va[u].y * rn, va[u].z * rn, va[u].w * rn)
                                            : make_float4(rn * (va[u].x - vb[u].x * s), rn * (va[u].y - vb[u].y * s),
                                                          rn * (va[u].z - vb[u].z * s), rn * (va[u].w - vb[u].w * s));
        }
        return;
    }
    float s = 0.f;
    if (!BWD) {
        for (int c = lane; c < C; c += 64) s += pa[c] * pa[c];
    } else {
        const float* pb = b + (size_t)row * C;
        for (int c = lane; c < C; c += 64) s += pa[c] * pb[c];  // a = dy, b = y
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (!BWD) {
        const float rn = 1.0f / sqrtf(fmaxf(s, 1e-12f));
        for (int c = lane; c < C; c += 64) po[c] = pa[c] * rn;
        if (lane == 0) rn_out[row] = rn;
    } else {
        const float* pb = b + (size_t)row * C;
        const float rn = rn_in[row];
        const bool clamped = rn >= 0.99e6f;  // sum x^2 <= 1e-12: y = x * 1e6, d/dx = 1e6 (no projection term)
        for (int c = lane; c < C; c += 64) po[c] = clamped ? pa[c] * rn : rn * (pa[c] - pb[c] * s);
    }
}

extern "C" int epc_rownorm_fwd(const float* x, int rows, int C, float* y, float* rn, void* stream) {
    EPC_CHECK_ARG(x && y && rn && rows > 0 && C > 0, "bad argument");
    hipLaunchKernelGGL(rownorm_kernel<false>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, nullptr,
                       nullptr, rows, C, y, rn);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_rownorm_bwd(const float* dy, const float* y, const float* rn, int rows, int C, float* dx,
                               void* stream) {
    EPC_CHECK_ARG(dy && y && rn && dx && rows > 0 && C > 0, "bad argument");
    hipLaunchKernelGGL(rownorm_kernel<true>, dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, y, rn, rows, C,
                       dx, nullptr);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// Softmax over 64 columns, one wave per row.  bwd: dx = y * (dy - sum(dy*y)); with BCAST the incoming gradient is
// dy + dsum[row / n_points] -- the gradient of a_sum = sum over the cloud's points (loupe.py:276) reaches every point
// of the cloud unchanged, so it is added here instead of being expanded to (rows, 64) in memory.
// ----------------------------------------------------------------------------------------------------------------
template <bool BWD, bool BCAST>
__global__ __launch_bounds__(256) void softmax64_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ dsum, int n_points, int rows,
                                                        float* __restrict__ out) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const size_t o = (size_t)row * 64 + lane;
    if (!BWD) {
        const float v = a[o];
        float m = v;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        const float e = expf(v - m);
        float s = e;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        out[o] = e / s;
    } else {
        float dy = a[o];
        if (BCAST) dy += dsum[(size_t)(row / n_points) * 64 + lane];
        const float y = b[o];
        float s = dy * y;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        out[o] = y * (dy - s);
    }
}

extern "C" int epc_softmax64_fwd(const float* x, int rows, float* y, void* stream) {
    EPC_CHECK_ARG(x && y && rows > 0, "bad argument");
    hipLaunchKernelGGL((softmax64_kernel<false, false>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, nullptr,
                       nullptr, 1, rows, y);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_softmax64_bwd(const float* dy, const float* y, int rows, float* dx, void* stream) {
    EPC_CHECK_ARG(dy && y && dx && rows > 0, "bad argument");
    hipLaunchKernelGGL((softmax64_kernel<true, false>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, y, nullptr,
                       1, rows, dx);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_softmax64_bwd_bcast(const float* dy, const float* dsum, int n_points, const float* y, int rows, float* dx,
                                       void* stream) {
    EPC_CHECK_ARG(dy && dsum && y && dx && rows > 0 && n_points > 0 && rows % n_points == 0, "bad argument");
    hipLaunchKernelGGL((softmax64_kernel<true, true>), dim3((rows + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, y, dsum,
                       n_points, rows, dx);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// a_sum[b][c] = sum over the cloud's points of a[b][n][c] (loupe.py:276), 64 columns: CS_SEG segments per cloud leave partial sums (the
// fused soft-assignment forward below), the finish kernel adds them in order -- the same bits every run.
// ----------------------------------------------------------------------------------------------------------------
constexpr int CS_SEG = 64;   // (round 6: 64 segments per cloud -- 18 x 64 workgroups of four waves -- instead of 16: the forward 17 -> 11 us at 18 clouds)

__global__ void cloud_colsum64_finish_kernel(const float* __restrict__ part, int total, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const float* p = part + (size_t)(i >> 6) * CS_SEG * 64 + (i & 63);
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < CS_SEG; ++k) s += p[k * 64];
    out[i] = s;
}

extern "C" size_t epc_cloud_colsum64_partial_floats(int num_clouds) { return (size_t)max(num_clouds, 0) * CS_SEG * 64; }

// ----------------------------------------------------------------------------------------------------------------
// The soft assignment behind its product (loupe.py:255-276) in one pass each way.
//   forward : a = softmax(batch_norm(z)) over the 64 clusters and a_sum = sum of a over the cloud's points.  z is read once
//             and a written once (batch-norm output, softmax and the column sums were three passes: 5 x 19 MB at 18 x 4096
//             rows instead of 2 x).  A row is 16 lanes x float4; the maximum and the sum cross the 16 lanes by four
//             exchanges; thread (row group, 4 columns) adds its rows in order, the 16 row groups are added in order, the
//             CS_SEG segments of a cloud by cloud_colsum64_finish_kernel: the same bits every run.
//   backward: dpre = a (dy - sum(dy a)), dy = da + dsum[cloud]  (softmax64_kernel<true, true>) with BatchNorm's two column
//             sums (sum dpre, sum dpre zhat: colreduce_kernel<2>'s partials, same layout) from the registers that hold dpre;
//             dpre is written into dz, colreduce_finish_kernel<2> adds the partials, bn_apply_bwd_kernel turns dz into the
//             BatchNorm input gradient in place.
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void assign_softmax_fwd_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                                 const float* __restrict__ var, const float* __restrict__ gamma,
                                                                 const float* __restrict__ beta, float eps, int n_points,
                                                                 float* __restrict__ a, float* __restrict__ part) {
    __shared__ __attribute__((aligned(16))) float coef[2][64];
    __shared__ float red[16][64];
    const int tid = threadIdx.x, l16 = tid & 15, rg = tid >> 4;
    const int seg = blockIdx.x, b = blockIdx.y;
    if (tid < 64) {
        const BnAffine c = bn_affine(mean[tid], var[tid], gamma[tid], beta[tid], eps);
        coef[0][tid] = c.s, coef[1][tid] = c.t;
    }
    __syncthreads();
    BnAffine af[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) af[q].s = coef[0][4 * l16 + q], af[q].t = coef[1][4 * l16 + q];
    const int len = (n_points + CS_SEG - 1) / CS_SEG;
    const int r0 = seg * len, r1 = min(n_points, r0 + len);
    const size_t base = (size_t)b * n_points * 64 + 4 * l16;
    float cs[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 4
    for (int r = r0 + rg; r < r1; r += 16) {
        const size_t o = base + (size_t)r * 64;
        const float4 v = *reinterpret_cast<const float4*>(z + o);
        float p[4] = {bn_value(v.x, af[0]), bn_value(v.y, af[1]), bn_value(v.z, af[2]), bn_value(v.w, af[3])};
        float m = fmaxf(fmaxf(p[0], p[1]), fmaxf(p[2], p[3]));
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
#pragma unroll
        for (int q = 0; q < 4; ++q) p[q] = expf(p[q] - m);
        float s = (p[0] + p[1]) + (p[2] + p[3]);
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            p[q] = p[q] / s;
            cs[q] += p[q];
        }
        *reinterpret_cast<float4*>(a + o) = make_float4(p[0], p[1], p[2], p[3]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[rg][4 * l16 + q] = cs[q];
    __syncthreads();
    if (tid < 64) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][tid];
        part[((size_t)b * CS_SEG + seg) * 64 + tid] = t;
    }
}

extern "C" int epc_assign_softmax_fwd(const float* z, const float* mean, const float* var, const float* gamma, const float* beta,
                                      float eps, int num_clouds, int n_points, float* a, float* a_sum, float* partials,
                                      size_t partial_floats, void* stream) {
    EPC_CHECK_ARG(z && mean && var && gamma && beta && a && a_sum && partials && num_clouds > 0 && n_points > 0, "bad argument");
    EPC_CHECK_ARG(partial_floats >= epc_cloud_colsum64_partial_floats(num_clouds), "partials buffer too small");
    hipLaunchKernelGGL(assign_softmax_fwd_kernel, dim3(CS_SEG, num_clouds), dim3(256), 0, (hipStream_t)stream, z, mean, var,
                       gamma, beta, eps, n_points, a, partials);
    const int total = num_clouds * 64;
    hipLaunchKernelGGL(cloud_colsum64_finish_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, partials,
                       total, a_sum);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

__global__ __launch_bounds__(256) void assign_softmax_bwd_kernel(const float* __restrict__ da, const float* __restrict__ dsum,
                                                                 const float* __restrict__ a, const float* __restrict__ z,
                                                                 const float* __restrict__ mean, const float* __restrict__ var,
                                                                 float eps, int n_points, int rows, float* __restrict__ dpre,
                                                                 float* __restrict__ partial, float* __restrict__ rowdot) {
    __shared__ float red[2][16][64];
    const int tid = threadIdx.x, l16 = tid & 15, rg = tid >> 4;
    const int bx = blockIdx.x, nb = gridDim.x;
    const int r0 = bx * CR_ROWS, r1 = min(rows, r0 + CR_ROWS);
    float mu[4], rs[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) mu[q] = mean[4 * l16 + q], rs[q] = 1.0f / sqrtf(var[4 * l16 + q] + eps);
    float s0[4] = {0.f, 0.f, 0.f, 0.f}, s1[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int r = r0 + rg; r < r1; r += 16) {
        const size_t o = (size_t)r * 64 + 4 * l16;
        const float4 g = *reinterpret_cast<const float4*>(da + o), y = *reinterpret_cast<const float4*>(a + o);
        const float4 zv = *reinterpret_cast<const float4*>(z + o);
        float dy[4] = {g.x, g.y, g.z, g.w};
        if (rowdot) {   // sum_k a_k da_k with da as the product left it (before the a_sum gradient): the first half of t_row
            float t1 = (g.x * y.x + g.y * y.y) + (g.z * y.z + g.w * y.w);
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) t1 += __shfl_xor(t1, off);
            if (l16 == 0) rowdot[r] = t1;
        }
        if (dsum) {
            const float4 e = *reinterpret_cast<const float4*>(dsum + (size_t)(r / n_points) * 64 + 4 * l16);
            dy[0] += e.x, dy[1] += e.y, dy[2] += e.z, dy[3] += e.w;
        }
        const float yy[4] = {y.x, y.y, y.z, y.w}, zz[4] = {zv.x, zv.y, zv.z, zv.w};
        float s = (dy[0] * yy[0] + dy[1] * yy[1]) + (dy[2] * yy[2] + dy[3] * yy[3]);
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) s += __shfl_xor(s, off);
        float d[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            d[q] = yy[q] * (dy[q] - s);
            s0[q] += d[q];
            s1[q] += d[q] * ((zz[q] - mu[q]) * rs[q]);
        }
        *reinterpret_cast<float4*>(dpre + o) = make_float4(d[0], d[1], d[2], d[3]);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) red[0][rg][4 * l16 + q] = s0[q], red[1][rg][4 * l16 + q] = s1[q];
    __syncthreads();
    if (tid < 128) {
        const int q = tid >> 6, l = tid & 63;
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[q][g][l];
        partial[((size_t)q * nb + bx) * 64 + l] = t;
    }
}

// bn_apply_bwd_kernel for the assignment's 64 columns without a ReLU, dz in place of dpre, and the second half of t_row:
// rowdot[r] += sum_k dz[r][k] z[r][k]  (z = f Wc, the assignment's pre-BatchNorm product)
__global__ __launch_bounds__(256) void assign_dz_rowdot_kernel(const float* __restrict__ z, const float* __restrict__ mean,
                                                               const float* __restrict__ var, const float* __restrict__ gamma,
                                                               const float* __restrict__ dbeta, const float* __restrict__ dgamma,
                                                               float eps, float inv_rows, int rows, float* __restrict__ dz,
                                                               float* __restrict__ rowdot) {
    const int tid = threadIdx.x, l16 = tid & 15, rg = tid >> 4;
    float mu[4], k1[4], bb[4], gg[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = 4 * l16 + q;
        const float rs = 1.0f / sqrtf(var[c] + eps);
        mu[q] = mean[c], k1[q] = gamma[c] * rs, bb[q] = dbeta[c] * inv_rows, gg[q] = rs * (dgamma[c] * inv_rows);
    }
    const int r0 = blockIdx.x * CR_ROWS, r1 = min(rows, r0 + CR_ROWS);
#pragma unroll 2
    for (int r = r0 + rg; r < r1; r += 16) {
        const size_t o = (size_t)r * 64 + 4 * l16;
        const float4 zv = *reinterpret_cast<const float4*>(z + o), gv = *reinterpret_cast<const float4*>(dz + o);
        const float zi[4] = {zv.x, zv.y, zv.z, zv.w}, gi[4] = {gv.x, gv.y, gv.z, gv.w};
        float out[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) out[q] = k1[q] * (gi[q] - bb[q] - (zi[q] - mu[q]) * gg[q]);
        *reinterpret_cast<float4*>(dz + o) = make_float4(out[0], out[1], out[2], out[3]);
        float t2 = (out[0] * zi[0] + out[1] * zi[1]) + (out[2] * zi[2] + out[3] * zi[3]);
#pragma unroll
        for (int off = 8; off >= 1; off >>= 1) t2 += __shfl_xor(t2, off);
        if (l16 == 0) rowdot[r] += t2;
    }
}

extern "C" int epc_assign_softmax_bwd(const float* da, const float* dsum, const float* a, const float* z, const float* mean,
                                      const float* var, const float* gamma, const float* beta, float eps, int num_clouds,
                                      int n_points, float* dz, float* dgamma, float* dbeta, float* rowdot, void* workspace,
                                      size_t workspace_bytes, void* stream) {
    EPC_CHECK_ARG(da && a && z && mean && var && gamma && beta && dz && dgamma && dbeta && num_clouds > 0 && n_points > 0,
                  "bad argument");
    const long rows_l = (long)num_clouds * n_points;
    EPC_CHECK_ARG(rows_l < (1l << 31), "too many rows");
    const int rows = (int)rows_l;
    if (int rc = colreduce_check("epc_assign_softmax_bwd: workspace too small", rows, 64, workspace, workspace_bytes)) return rc;
    hipStream_t st = (hipStream_t)stream;
    const int nb = (rows + CR_ROWS - 1) / CR_ROWS;
    float* part = (float*)((unsigned int*)workspace + CR_COUNTERS);
    hipLaunchKernelGGL(assign_softmax_bwd_kernel, dim3(nb), dim3(256), 0, st, da, dsum, a, z, mean, var, eps, n_points, rows, dz,
                       part, rowdot);
    hipLaunchKernelGGL(colreduce_finish_kernel<2>, dim3(64 / CF_COLS), dim3(256), 0, st, part, z, nb, 64, 1.0f, dbeta, dgamma);
    // dz holds dpre: every element is read and then overwritten by the thread that owns it
    if (rowdot)
        hipLaunchKernelGGL(assign_dz_rowdot_kernel, dim3(nb), dim3(256), 0, st, z, mean, var, gamma, dbeta, dgamma, eps, 1.0f / rows,
                           rows, dz, rowdot);
    else
        hipLaunchKernelGGL(bn_apply_bwd_kernel, dim3(1, nb), dim3(256), 0, st, (const float*)dz, z, mean, var, gamma, beta, dbeta,
                           dgamma, eps, 1.0f / rows, 0, rows, 64, dz);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// Context gating's product (loupe.py:99-100): out = y * sigmoid(g); bwd: dy = dout*s, dg = dout*y*s*(1-s).
// ----------------------------------------------------------------------------------------------------------------
__global__ void gate_fwd_kernel(const float* __restrict__ y, const float* __restrict__ g, long n, float* __restrict__ out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = y[i] * (1.f / (1.f + expf(-g[i])));
}

__global__ void gate_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ y, const float* __restrict__ g,
                                long n, float* __restrict__ dy, float* __restrict__ dg) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float s = 1.f / (1.f + expf(-g[i])), d = dout[i];
    dy[i] = d * s;
    dg[i] = d * y[i] * (s * (1.f - s));
}

extern "C" int epc_gate_fwd(const float* y, const float* g, long n, float* out, void* stream) {
    EPC_CHECK_ARG(y && g && out && n > 0, "bad argument");
    hipLaunchKernelGGL(gate_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, g, n, out);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_gate_bwd(const float* dout, const float* y, const float* g, long n, float* dy, float* dg, void* stream) {
    EPC_CHECK_ARG(dout && y && g && dy && dg && n > 0, "bad argument");
    hipLaunchKernelGGL(gate_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dout, y, g, n,
                       dy, dg);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// tf.train.AdamOptimizer (train.py:273; beta1 .9, beta2 .999, eps 1e-8): lr_t = lr*sqrt(1-b2^t)/(1-b1^t);
// m = b1 m + (1-b1) g ; v = b2 v + (1-b2) g^2 ; w -= lr_t * m / (sqrt(v) + eps)
// ----------------------------------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                            const float* __restrict__ g, long n, float lr_t, float b1, float b2, float eps) {
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= n) return;
    const float gi = g[o];
    const float mi = b1 * m[o] + (1.0f - b1) * gi;
    const float vi = b2 * v[o] + (1.0f - b2) * gi * gi;
    m[o] = mi;
    v[o] = vi;
    w[o] = w[o] - lr_t * mi / (sqrtf(vi) + eps);
}

// the same update with lr_t read from device memory: a captured HIP graph of the training step is replayed with a new
// learning rate / bias correction every step without re-recording
__global__ void adam_dev_kernel(float* __restrict__ w, float* __restrict__ m, float* __restrict__ v,
                                const float* __restrict__ g, long n, const float* __restrict__ lr_t_dev, float b1,
                                float b2, float eps) {
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= n) return;
    const float lr_t = *lr_t_dev;
    const float gi = g[o];
    const float mi = b1 * m[o] + (1.0f - b1) * gi;
    const float vi = b2 * v[o] + (1.0f - b2) * gi * gi;
    m[o] = mi;
    v[o] = vi;
    w[o] = w[o] - lr_t * mi / (sqrtf(vi) + eps);
}

extern "C" int epc_adam_step_dev(float* w, float* m, float* v, const float* g, long n, const float* lr_t_dev,
                                 float beta1, float beta2, float eps, void* stream) {
    EPC_CHECK_ARG(w && m && v && g && lr_t_dev && n > 0, "bad argument");
    hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, m, v, g,
                       n, lr_t_dev, beta1, beta2, eps);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_adam_step(float* w, float* m, float* v, const float* g, long n, float lr, float beta1, float beta2,
                             float eps, int t, void* stream) {
    EPC_CHECK_ARG(w && m && v && g && n > 0 && t >= 1, "bad argument");
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t));
    hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, m, v, g, n,
                       (float)lr_t, beta1, beta2, eps);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// Moving-average update of the BatchNorm statistics (tf.train.ExponentialMovingAverage.apply, utils/tf_util.py:474-487;
// slim's assign_moving_average): shadow -= (1 - decay) * (shadow - value).  One launch per statistic instead of three
// elementwise torch kernels (34 statistics per EPC-Net step); decay comes from device memory when decay_dev != NULL
// (HIP-graph replay with a changing bn_decay).
// ----------------------------------------------------------------------------------------------------------------
__global__ void ema_update_kernel(float* __restrict__ shadow, const float* __restrict__ value, long n, float decay,
                                  const float* __restrict__ decay_dev) {
    const long o = (long)blockIdx.x * 256 + threadIdx.x;
    if (o >= n) return;
    const float d = decay_dev ? *decay_dev : decay;
    const float sh = shadow[o];
    shadow[o] = sh - (sh - value[o]) * (1.0f - d);
}

extern "C" int epc_ema_update(float* shadow, const float* value, long n, float decay, const float* decay_dev,
                              void* stream) {
    EPC_CHECK_ARG(shadow && value && n > 0, "bad argument");
    hipLaunchKernelGGL(ema_update_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, shadow,
                       value, n, decay, decay_dev);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// Many small tensors, one launch: the 62 Adam updates and the 34 moving-average updates of a step are each ~5 us of
// fixed launch cost for a few KB of work.  The pointer table travels BY VALUE in the kernel arguments (up to
// MT_MAX tensors per launch, 40 B each: inside the 4-KB kernarg segment), so nothing is staged through memory and the
// launch can be captured in a HIP graph.  blk0[t] = first block of tensor t (256 elements per block).
// ----------------------------------------------------------------------------------------------------------------
#define MT_MAX 64
struct MultiTensorArgs {
    float* a[MT_MAX];        // adam: w      ema: shadow
    float* b[MT_MAX];        // adam: m      ema: (unused)
    float* c[MT_MAX];        // adam: v      ema: (unused)
    const float* d[MT_MAX];  // adam: g      ema: value
    int blk0[MT_MAX + 1];
    int count;
};

struct MultiTensorLens {
    long n[MT_MAX];
};

__device__ __forceinline__ int mt_find(const MultiTensorArgs& t, int block) {
    int lo = 0, hi = t.count - 1;  // last tensor whose first block is <= block
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (t.blk0[mid] <= block) lo = mid; else hi = mid - 1;
    }
    return lo;
}

__global__ void adam_multi_kernel(MultiTensorArgs t, MultiTensorLens n, float lr_t, const float* __restrict__ lr_t_dev,
                                  float b1, float b2, float eps) {
    const int k = mt_find(t, blockIdx.x);
    const long o = (long)(blockIdx.x - t.blk0[k]) * 256 + threadIdx.x;
    if (o >= n.n[k]) return;
    const float lr = lr_t_dev ? *lr_t_dev : lr_t;
    const float gi = t.d[k][o];
    const float mi = b1 * t.b[k][o] + (1.0f - b1) * gi;
    const float vi = b2 * t.c[k][o] + (1.0f - b2) * gi * gi;
    t.b[k][o] = mi;
    t.c[k][o] = vi;
    t.a[k][o] = t.a[k][o] - lr * mi / (sqrtf(vi) + eps);
}

// t.c[k] != NULL marks a statistic that follows the scheduled decay (tf_util BN); the others use the fixed one (slim BN)
__global__ void ema_multi_kernel(MultiTensorArgs t, MultiTensorLens n, float fixed_decay, float sched_decay,
                                 const float* __restrict__ sched_decay_dev) {
    const int k = mt_find(t, blockIdx.x);
    const long o = (long)(blockIdx.x - t.blk0[k]) * 256 + threadIdx.x;
    if (o >= n.n[k]) return;
    const float d = t.c[k] ? (sched_decay_dev ? *sched_decay_dev : sched_decay) : fixed_decay;
    const float sh = t.a[k][o];
    t.a[k][o] = sh - (sh - t.d[k][o]) * (1.0f - d);
}

static int mt_fill(MultiTensorArgs& t, MultiTensorLens& ln, int count, float* const* a, float* const* b, float* const* c,
                   const float* const* d, const long* n) {
    int blocks = 0;
    t.count = count;
    for (int k = 0; k < count; ++k) {
        t.a[k] = a[k];
        t.b[k] = b ? b[k] : nullptr;
        t.c[k] = c ? c[k] : nullptr;
        t.d[k] = d[k];
        ln.n[k] = n[k];
        t.blk0[k] = blocks;
        blocks += (int)((n[k] + 255) / 256);
    }
    t.blk0[count] = blocks;
    return blocks;
}

// epc_adam_step / epc_adam_step_dev over `count` tensors in ceil(count / 64) launches (host arrays of device pointers).
extern "C" int epc_adam_multi(int count, float* const* w, float* const* m, float* const* v, const float* const* g,
                              const long* n, float lr, float beta1, float beta2, float eps, int t,
                              const float* lr_t_dev, void* stream) {
    EPC_CHECK_ARG(count > 0 && w && m && v && g && n && (lr_t_dev || t >= 1), "bad argument");
    const double lr_t = lr_t_dev ? 0.0 : (double)lr * sqrt(1.0 - pow((double)beta2, t)) / (1.0 - pow((double)beta1, t));
    for (int k0 = 0; k0 < count; k0 += MT_MAX) {
        MultiTensorArgs a;
        MultiTensorLens ln;
        const int c = count - k0 < MT_MAX ? count - k0 : MT_MAX;
        for (int k = 0; k < c; ++k) EPC_CHECK_ARG(w[k0 + k] && m[k0 + k] && v[k0 + k] && g[k0 + k] && n[k0 + k] > 0, "null tensor");
        const int blocks = mt_fill(a, ln, c, w + k0, m + k0, v + k0, g + k0, n + k0);
        hipLaunchKernelGGL(adam_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, ln, (float)lr_t, lr_t_dev,
                           beta1, beta2, eps);
    }
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// epc_ema_update over `count` statistics in one launch.  scheduled[k] != 0: the statistic uses sched_decay (read from
// sched_decay_dev when that is not NULL); otherwise fixed_decay.
extern "C" int epc_ema_multi(int count, float* const* shadow, const float* const* value, const long* n,
                             const int* scheduled, float fixed_decay, float sched_decay, const float* sched_decay_dev,
                             void* stream) {
    EPC_CHECK_ARG(count > 0 && shadow && value && n && scheduled, "bad argument");
    for (int k0 = 0; k0 < count; k0 += MT_MAX) {
        MultiTensorArgs a;
        MultiTensorLens ln;
        const int c = count - k0 < MT_MAX ? count - k0 : MT_MAX;
        for (int k = 0; k < c; ++k) EPC_CHECK_ARG(shadow[k0 + k] && value[k0 + k] && n[k0 + k] > 0, "null tensor");
        const int blocks = mt_fill(a, ln, c, shadow + k0, nullptr, nullptr, value + k0, n + k0);
        for (int k = 0; k < c; ++k) a.c[k] = scheduled[k0 + k] ? shadow[k0 + k] : nullptr;
        hipLaunchKernelGGL(ema_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, a, ln, fixed_decay, sched_decay,
                           sched_decay_dev);
    }
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}
