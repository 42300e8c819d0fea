// Shared by train_head16.hip (the head of the training step on bf16-stored tensors) and train_head32.hip (the same streaming kernels on
// f32 tensors in the f32-accurate split arithmetic): bf16 helpers, the operand packs, the row-streamed product (hx_rowgemm_kernel), the
// column product with the rows as the contraction (hx_colgemm_kernel) and the ordered reduction of row-slice partials.  The kernels are
// templates over the storage type of the (rows, 1024) tensor (u16 = bf16, float) and the number of bf16 pieces per operand.
#pragma once
#include "common.h"

typedef unsigned short u16;

__device__ __forceinline__ float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }
__device__ __forceinline__ unsigned pack2bf(float lo, float hi) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 p;
    p[0] = (__bf16)lo;
    p[1] = (__bf16)hi;
    return __builtin_bit_cast(unsigned, p);
}
__device__ __forceinline__ bf16x8 cvt8(const float (&v)[8]) {
    bf16x8 p;
#pragma unroll
    for (int j = 0; j < 8; ++j) p[j] = (__bf16)v[j];
    return p;
}
// P bf16 pieces of 8 values: p[0] = bf16(v), p[1] = bf16(v - p[0]), p[2] = bf16(v - p[0] - p[1])  (8 significant bits each)
template <int P>
__device__ __forceinline__ void hx_split(const float (&v)[8], bf16x8 (&p)[P]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float r = v[j];
#pragma unroll
        for (int q = 0; q < P; ++q) {
            p[q][j] = (__bf16)r;
            if (q + 1 < P) r -= (float)p[q][j];
        }
    }
}
// acc += the products of two P-piece operands whose piece indices sum to less than P, smallest terms first: P = 1 one product,
// P = 2 three (2^-16 per product: the backward arithmetic of epc_gemm_f32_fast), P = 3 six (f32-accurate: epc_gemm_f32's)
template <int P>
__device__ __forceinline__ f32x16 hx_prod(const bf16x8 (&a)[P], const bf16x8 (&b)[P], f32x16 c) {
#pragma unroll
    for (int sum = P - 1; sum >= 0; --sum)
#pragma unroll
        for (int pa = sum; pa >= 0; --pa) c = mfma_bf16(a[pa], b[sum - pa], c);
    return c;
}

struct H16Affine {   // y = z s + t: the expression train_ops.hip's bn_value evaluates (same association, contracted to one FMA)
    float s, t;
};
__device__ __forceinline__ H16Affine h16_affine(float mean, float var, float gamma, float beta, float eps) {
    H16Affine a;
    a.s = (1.0f / sqrtf(var + eps)) * gamma;
    a.t = beta - mean * a.s;
    return a;
}
struct H16Bn {
    const float *mean, *var, *gamma, *beta;
    float eps;
};

// ----------------------------------------------------------------------------------------------------------------
// Operand packs: a (K, N) f32 matrix (any strides) as P-piece bf16 MFMA B fragments of v_mfma_f32_32x32x16_bf16 -- lane
// (i = l & 31, h = l >> 5) holds B[k = 16 s + 8 h + j][column], j < 8.
//   layout 0 (column chunks, conv5's forward): [chunk c < N / 64][k-step s < K / 16][nt < 2][piece][lane]; column = 64 c + 2 i + nt --
//            the two accumulators of a lane are ADJACENT columns, so a 16-bit result row leaves as one dword per lane (128-byte runs).
//   layout 1 (k chunks, the streamed products): [k chunk kc < K / (16 KSC)][s' < KSC][tile < N / 32][piece][lane];
//            k = 16 (KSC kc + s') + 8 h + j, column = 32 tile + i.
// Packed once per step from the f32 master weights (a few MB at most): the rounding / splitting happens here.
// ----------------------------------------------------------------------------------------------------------------
template <int P>
__global__ __launch_bounds__(256) void h16_pack_kernel(const float* __restrict__ W, long sk, long sn, long sbatch, int K, int N,
                                                       int layout, int ksc, u32x4* __restrict__ out) {
    const long per = (long)K * N / 8;        // fragment entries per piece and batch
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= per) return;
    const int l = (int)(e & 63), i = l & 31, h = l >> 5;
    long rest = e >> 6;
    int k0, col;
    if (layout == 0) {
        const int nt = (int)(rest & 1);
        rest >>= 1;
        const int ks = K / 16;
        const int s = (int)(rest % ks), c = (int)(rest / ks);
        k0 = 16 * s + 8 * h, col = 64 * c + 2 * i + nt;
    } else {
        const int tiles = N / 32;
        const int tile = (int)(rest % tiles);
        rest /= tiles;
        const int sp = (int)(rest % ksc), kc = (int)(rest / ksc);
        k0 = 16 * (ksc * kc + sp) + 8 * h, col = 32 * tile + i;
    }
    const float* src = W + (size_t)blockIdx.y * sbatch + (size_t)k0 * sk + (size_t)col * sn;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = src[(size_t)j * sk];
    bf16x8 p[P];
    hx_split<P>(v, p);
    // entry e of the one-piece order becomes P consecutive 64-lane groups: [.. (e >> 6)][piece][lane]
    u32x4* dst = out + (size_t)blockIdx.y * per * P + (size_t)(e >> 6) * P * 64 + l;
#pragma unroll
    for (int q = 0; q < P; ++q) dst[q * 64] = __builtin_bit_cast(u32x4, p[q]);
}

template <int P>
static int h16_pack(const float* W, long sk, long sn, long sbatch, int batch, int K, int N, int layout, int ksc, void* out,
                    hipStream_t st) {
    const long per = (long)K * N / 8;
    hipLaunchKernelGGL(h16_pack_kernel<P>, dim3((unsigned)((per + 255) / 256), batch), dim3(256), 0, st, W, sk, sn, sbatch, K, N, layout,
                       ksc, (u32x4*)out);
    return EPC_OK;
}

// Pivot-shifted column statistics of a workgroup's four 32-row waves, merged to the first wave's pivot (exact algebra on the shifted
// sums): S1' = S1 + n d, S2' = S2 + 2 d S1 + n d^2 with d = p_w - p_0.  Output in moments_finalize_kernel's (S1, S2, pivot) form.
template <int NW = 4>
__device__ __forceinline__ void h16_merge_stats(const float (*w)[3][64], const bool* live, int c, float& S1, float& S2, float& P) {
    P = w[0][2][c];
    S1 = w[0][0][c], S2 = w[0][1][c];
#pragma unroll
    for (int q = 1; q < NW; ++q)
        if (live[q]) {
            const float d = w[q][2][c] - P, s1 = w[q][0][c];
            S1 += s1 + 32.f * d;
            S2 += w[q][1][c] + 2.f * d * s1 + 32.f * d * d;
        }
}

// ----------------------------------------------------------------------------------------------------------------
// Row-streamed product: out (rows, 32 NT) f32 = T(A) B with A (rows, 1024) -- bf16 (TA = u16) or f32 (TA = float) -- streamed ONCE
// (the next k chunk requested under this chunk's products), B (1024, 32 NT) packed in layout 1 with P pieces -- one matrix, or one per
// cloud -- and streamed through a double-buffered LDS chunk of 16 KSC k shared by the workgroup's four 32-row waves.
//   XFORM: T = relu(batch_norm(.)) per channel (conv5's BatchNorm from its batch moments; coefficients in LDS) and the row factor
//          rn = rsqrt(max(sum_c u^2, 1e-12)) from the f32 values of u, applied to the accumulators: out = rn (u B).
//          rn_out != null: rn is written.  stats != null: column statistics of out ([workgroups][3][32 NT], tile_rows = 128).
//   P: bf16 pieces per operand (hx_prod): 1 = the bf16 arithmetic; 2 / 3 = three / six products on f32 operands.
//   NT = 2: the assignment's product and its gradient (B = Wc / dvlad[cloud]);  NT = 8, no XFORM: dcat = dz5 W5^T.
//   BNB (with NT = 8, no XFORM): A is not read but FORMED -- the last step of conv5's BatchNorm backward,
//          dz5 = gamma rstd (du - dbeta / R - zhat dgamma / R)   (bn_apply_bwd_given_wide_kernel's / h16_bn_bwd_apply_kernel's expression),
//          from du (the A pointer) and z5 as the two are streamed, written back once (bnb.dz_out, may be du: every lane rewrites the
//          16 or 32 bytes it read) for dW5's product and used as this product's operand in registers: the separate apply pass -- a
//          read of du, a read of z5 and a write of dz5 -- and this kernel's own read of dz5 become two reads and one write.
// Workgroups never straddle clouds: grid = (ceil(n_points / 128), clouds); n_points a multiple of 32.
// ----------------------------------------------------------------------------------------------------------------
template <typename TA>
struct HxBnb {           // the BatchNorm backward fused into dcat's product (BNB)
    const TA* z;         // (rows, 1024) conv5's pre-activation
    const float* dbeta;  // (1024) sum du
    const float* dgamma; // (1024) sum du zhat
    float inv_rows;
    TA* dz_out;          // (rows, 1024)
};

// The streamed operand's pointer: restrict-qualified unless the kernel writes THROUGH an alias of it -- BNB's dz_out may be du itself
// (every lane rewrites exactly the bytes it read, one chunk after it has consumed them: the in-place contract of epc_h16/h32_conv5_dx_bn),
// and a store through an alias of a restrict pointer would let the compiler move a later load of A above it.
template <typename TA, bool RESTRICT> struct HxAPtr { typedef const TA* __restrict__ type; };
template <typename TA> struct HxAPtr<TA, false> { typedef const TA* type; };

template <int NT, bool XFORM, typename TA, int P, int KSC, bool BNB = false, int NW = 4>
__global__ __launch_bounds__(64 * NW, 2) void hx_rowgemm_kernel(typename HxAPtr<TA, !BNB>::type A, int n_points, const u32x4* __restrict__ Bp,
                                                            long b_cloud_stride_u4, H16Bn bn, float* __restrict__ out,
                                                            float* __restrict__ rn_out, float* __restrict__ stats, HxBnb<TA> bnb) {
    static_assert(!(BNB && XFORM), "one transformation of the streamed operand at a time");
    constexpr bool A32 = sizeof(TA) == 4;
    constexpr int CHUNK_U4 = KSC * NT * P * 64;
    constexpr int CHUNKS = 64 / KSC;             // K = 1024 = 64 k-steps
    constexpr int AV = A32 ? 2 : 1;              // 16-byte loads per 8-value fragment
    __shared__ u32x4 Bs[2][CHUNK_U4];
    // BNB: k1 = gamma rstd, b2 = dbeta / R - mean gg, gg = rstd dgamma / R:  dz5 = k1 (du - b2 - z5 gg)  (three tables: two workgroups per CU)
    __shared__ __attribute__((aligned(16))) float coef[XFORM ? 2 : (BNB ? 3 : 1)][XFORM || BNB ? 1024 : 4];
    __shared__ float rowc[NW][32];
    __shared__ float wst[NW][3][XFORM ? 32 * NT : 1];   // (column statistics: the assignment's launches only)
    __shared__ bool wlive[NW];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int cloud = blockIdx.y;
    const int r0 = blockIdx.x * (32 * NW) + wave * 32;         // within the cloud
    const bool live = r0 < n_points;
    if (lane == 0) wlive[wave] = live;
    const u32x4* src = Bp + (size_t)cloud * b_cloud_stride_u4;
    // B's chunks, global -> LDS.  Without BNB: by LDS-DMA (round 6: no staging registers, no ds_writes; 1-KB pieces, piece wave + u NW of the
    // chunk's KSC NT P) -- a request is issued BEFORE the streamed operand's loads of the same chunk and the counter retires in order, so once
    // those loads have landed (`deposit` pins them in front of the barrier that publishes the buffer: the compiler's own wait) the pieces
    // have too; 1 - 8 % per launch (f32: assignment 94.5 / 96.5 -> 92 / 93 us, dcat's product 154 -> 143; bf16: 42.9 / 40.2 -> 41.8 / 38.1, 64 -> 59).
    // With BNB: through staging registers as before -- the DMA form measured the same or 3 % slower there (the fused form's chunk is short and
    // it streams two tensors; its loads sit behind dz5's stores).
    constexpr bool DMA = !BNB;
    constexpr int WGT = 64 * NW, PER = DMA ? 1 : (CHUNK_U4 + WGT - 1) / WGT;   // (three-wave workgroups: the last piece is predicated)
    constexpr int PIECES = CHUNK_U4 / 64, DPER = (PIECES + NW - 1) / NW;
    const unsigned bs_base = (unsigned)(size_t)(const __attribute__((address_space(3))) u32x4*)&Bs[0][0];
    u32x4 pre[PER];
    auto request = [&](int kc, int buf) {
        if constexpr (DMA) {
            const float* from = reinterpret_cast<const float*>(src + (size_t)kc * CHUNK_U4);
#pragma unroll
            for (int u = 0; u < DPER; ++u)
                if (PIECES % NW == 0 || wave + u * NW < PIECES)
                    glds16(from, 16u * ((wave + u * NW) * 64 + lane), bs_base + 16u * (buf * CHUNK_U4 + (wave + u * NW) * 64));
        } else {
#pragma unroll
            for (int u = 0; u < PER; ++u)
                if (CHUNK_U4 % WGT == 0 || tid + u * WGT < CHUNK_U4) pre[u] = src[(size_t)kc * CHUNK_U4 + tid + u * WGT];
        }
    };
    request(0, 0);
    const size_t grow = (size_t)cloud * n_points + min(r0 + i, n_points - 1);
    const TA* arow = A + grow * 1024 + 8 * h;
    const TA* zrow = BNB ? bnb.z + grow * 1024 + 8 * h : nullptr;
    // (the streamed operand runs ONE chunk ahead of the products; two chunks ahead measured the same: 64.7 / 151.9 us against 65.7 / 153)
    u32x4 an[KSC][AV], zn[BNB ? KSC : 1][AV];
    auto aload = [&](int kc) {
#pragma unroll
        for (int s = 0; s < KSC; ++s)
#pragma unroll
            for (int w = 0; w < AV; ++w) {
                an[s][w] = *reinterpret_cast<const u32x4*>(arow + 16 * (KSC * kc + s) + (A32 ? 4 * w : 0));
                if constexpr (BNB) zn[s][w] = *reinterpret_cast<const u32x4*>(zrow + 16 * (KSC * kc + s) + (A32 ? 4 * w : 0));
            }
    };
    aload(0);
    if constexpr (BNB) {
        for (int c = tid; c < 1024; c += 64 * NW) {
            const float r = 1.0f / sqrtf(bn.var[c] + bn.eps), gg = r * (bnb.dgamma[c] * bnb.inv_rows);
            coef[0][c] = bn.gamma[c] * r, coef[1][c] = bnb.dbeta[c] * bnb.inv_rows - bn.mean[c] * gg, coef[2][c] = gg;
        }
    }
    if constexpr (XFORM) {
        for (int c = tid; c < 1024; c += 64 * NW) {
            const H16Affine a = h16_affine(bn.mean[c], bn.var[c], bn.gamma[c], bn.beta[c], bn.eps);
            coef[0][c] = a.s, coef[1][c] = a.t;
        }
    }
    auto deposit = [&](int buf) {
        if constexpr (DMA) {
#pragma unroll
            for (int s = 0; s < KSC; ++s)
#pragma unroll
                for (int w = 0; w < AV; ++w) asm volatile("" : "+v"(an[s][w]));
        } else {
#pragma unroll
            for (int u = 0; u < PER; ++u)
                if (CHUNK_U4 % WGT == 0 || tid + u * WGT < CHUNK_U4) Bs[buf][tid + u * WGT] = pre[u];
        }
    };
    deposit(0);
    __syncthreads();
    f32x16 acc[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
    float ss = 0.f;
    for (int kc = 0; kc < CHUNKS; ++kc) {
        const int buf = kc & 1;
        u32x4 av[BNB ? 1 : KSC][AV];
        bf16x8 afr[BNB ? KSC : 1][P];   // BNB: the chunk's fragments, formed (and dz5 written back) before the next chunk is requested
        if constexpr (BNB) {
            if (live) {
#pragma unroll
                for (int s = 0; s < KSC; ++s) {
                    const int c0 = 16 * (KSC * kc + s) + 8 * h;
                    float g[8], z[8], k1[8], bb[8], gg[8], dzv[8];
                    if constexpr (A32) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            g[w] = __uint_as_float(an[s][0][w]), g[4 + w] = __uint_as_float(an[s][AV - 1][w]);
                            z[w] = __uint_as_float(zn[s][0][w]), z[4 + w] = __uint_as_float(zn[s][AV - 1][w]);
                        }
                    } else {
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            g[2 * w] = bf_lo(an[s][0][w]), g[2 * w + 1] = bf_hi(an[s][0][w]);
                            z[2 * w] = bf_lo(zn[s][0][w]), z[2 * w + 1] = bf_hi(zn[s][0][w]);
                        }
                    }
#pragma unroll
                    for (int q4 = 0; q4 < 2; ++q4) {
                        const float4 k4 = *reinterpret_cast<const float4*>(&coef[0][c0 + 4 * q4]), b4 = *reinterpret_cast<const float4*>(&coef[1][c0 + 4 * q4]);
                        const float4 g4 = *reinterpret_cast<const float4*>(&coef[2][c0 + 4 * q4]);
                        k1[4 * q4] = k4.x, k1[4 * q4 + 1] = k4.y, k1[4 * q4 + 2] = k4.z, k1[4 * q4 + 3] = k4.w;
                        bb[4 * q4] = b4.x, bb[4 * q4 + 1] = b4.y, bb[4 * q4 + 2] = b4.z, bb[4 * q4 + 3] = b4.w;
                        gg[4 * q4] = g4.x, gg[4 * q4 + 1] = g4.y, gg[4 * q4 + 2] = g4.z, gg[4 * q4 + 3] = g4.w;
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) dzv[j] = k1[j] * (g[j] - bb[j] - z[j] * gg[j]);   // (the apply kernels' expression, mean folded into bb)
                    TA* orow = bnb.dz_out + grow * 1024 + 8 * h + 16 * (KSC * kc + s);
                    if constexpr (A32) {
                        *reinterpret_cast<float4*>(orow) = make_float4(dzv[0], dzv[1], dzv[2], dzv[3]);
                        *reinterpret_cast<float4*>(orow + 4) = make_float4(dzv[4], dzv[5], dzv[6], dzv[7]);
                        hx_split<P>(dzv, afr[s]);
                    } else {
                        u32x4 pk;
#pragma unroll
                        for (int w = 0; w < 4; ++w) pk[w] = pack2bf(dzv[2 * w], dzv[2 * w + 1]);
                        *reinterpret_cast<u32x4*>(orow) = pk;
                        afr[s][0] = __builtin_bit_cast(bf16x8, pk);   // (the stored, rounded values ARE the operand)
                    }
                }
            }
        } else {
#pragma unroll
            for (int s = 0; s < KSC; ++s)
#pragma unroll
                for (int w = 0; w < AV; ++w) av[s][w] = an[s][w];
        }
        if (kc + 1 < CHUNKS) {
            request(kc + 1, buf ^ 1);      // (DMA: the buffer's last readers passed the barrier at the end of the previous chunk)
            aload(kc + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (live) {
#pragma unroll
            for (int s = 0; s < KSC; ++s) {
                bf16x8 a[P];
                if constexpr (BNB) {
#pragma unroll
                    for (int q = 0; q < P; ++q) a[q] = afr[s][q];
                } else if constexpr (!XFORM && !A32) {
                    a[0] = __builtin_bit_cast(bf16x8, av[s][0]);
                } else {
                    float u[8];
                    if constexpr (A32) {
#pragma unroll
                        for (int w = 0; w < 4; ++w) u[w] = __uint_as_float(av[s][0][w]), u[4 + w] = __uint_as_float(av[s][AV - 1][w]);
                    } else {
#pragma unroll
                        for (int w = 0; w < 4; ++w) u[2 * w] = bf_lo(av[s][0][w]), u[2 * w + 1] = bf_hi(av[s][0][w]);
                    }
                    if constexpr (XFORM) {
                        const int c0 = 16 * (KSC * kc + s) + 8 * h;
                        const float4 s0 = *reinterpret_cast<const float4*>(&coef[0][c0]), s1 = *reinterpret_cast<const float4*>(&coef[0][c0 + 4]);
                        const float4 t0 = *reinterpret_cast<const float4*>(&coef[1][c0]), t1 = *reinterpret_cast<const float4*>(&coef[1][c0 + 4]);
                        const float cs[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
                        const float ct[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            u[j] = fmaxf(u[j] * cs[j] + ct[j], 0.f);
                            ss += u[j] * u[j];
                        }
                    }
                    hx_split<P>(u, a);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    bf16x8 b[P];
#pragma unroll
                    for (int q = 0; q < P; ++q) b[q] = __builtin_bit_cast(bf16x8, Bs[buf][((s * NT + nt) * P + q) * 64 + lane]);
                    acc[nt] = hx_prod<P>(a, b, acc[nt]);
                }
            }
        }
        if (kc + 1 < CHUNKS) deposit(buf ^ 1);
        __syncthreads();
    }
    if constexpr (XFORM) {
        ss += __shfl_xor(ss, 32);
        const float rnv = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        if (h == 0) {
            rowc[wave][i] = rnv;
            if (rn_out && live) rn_out[grow] = rnv;
        }
        __syncthreads();
        if (live) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float f = rowc[wave][mfma_row(r, h)];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt][r] *= f;
            }
        }
    }
    constexpr int N = 32 * NT;
    if (live) {
        const size_t obase = ((size_t)cloud * n_points + r0) * N;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float* ob = out + obase + (size_t)mfma_row(r, h) * N + i;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) ob[32 * nt] = acc[nt][r];
        }
    }
    if constexpr (XFORM) if (stats) {   // (workgroup-uniform; the assignment's launches only)
        if (live) {
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float p = __shfl(acc[nt][0], i);
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = acc[nt][r] - p;
                    s1 += d, s2 += d * d;
                }
                s1 += __shfl_xor(s1, 32), s2 += __shfl_xor(s2, 32);
                if (h == 0) wst[wave][0][32 * nt + i] = s1, wst[wave][1][32 * nt + i] = s2, wst[wave][2][32 * nt + i] = p;
            }
        }
        __syncthreads();
        if (tid < N) {
            float Pv = wst[0][2][tid], S1 = wst[0][0][tid], S2 = wst[0][1][tid];
#pragma unroll
            for (int q = 1; q < NW; ++q)
                if (wlive[q]) {
                    const float d = wst[q][2][tid] - Pv, s1 = wst[q][0][tid];
                    S1 += s1 + 32.f * d;
                    S2 += wst[q][1][tid] + 2.f * d * s1 + 32.f * d * d;
                }
            const size_t wg = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
            float* o = stats + wg * 3 * N + tid;
            o[0] = S1, o[N] = S2, o[2 * N] = Pv;
        }
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Column product with the rows as the contraction: Pout[split][c][n] = sum over the workgroup's rows r of u[r][c] (rn[r] C[r][n]),
// u = relu(batch_norm(z5)) -- the VLAD aggregation vlad[b] = f[b]^T a[b] (loupe.py:286-291; C = a) and the cluster weights' gradient
// dWc = f^T dz (C = dz) with f = u rn never materialised.  z5 (rows, 1024), bf16 or f32, is read once chip-wide: a workgroup owns 128
// channels (4 per lane and row: 256- or 512-byte runs) and a range of rows; the MFMA's A operand wants 8 consecutive ROWS of one channel
// per lane, which is a register transposition of the 8 x 4 values a lane loads (free: the conversions write the fragments' elements
// directly), with the BatchNorm coefficients of the lane's four channels in registers for the whole kernel.  C (rows, 64) f32 is read
// lane-coalesced (one column per lane), scaled by rn and rounded / split.  P pieces per operand (hx_prod).  A k-step's values are
// turned into fragments BEFORE the next step's loads are issued, so the two never hold registers together.  The four waves take a
// quarter of the rows each and meet in LDS in a fixed order; the splits of a cloud (or of everything) are added by
// h16_partial_reduce_kernel in ascending order: same bits every run.
// ----------------------------------------------------------------------------------------------------------------
template <typename TA, int P>
__global__ __launch_bounds__(256, 2) void hx_colgemm_kernel(const TA* __restrict__ Z, H16Bn bn, const float* __restrict__ C,
                                                            const float* __restrict__ rn, int rows_per_wg, int rows_per_batch,
                                                            int splits, float* __restrict__ Pout) {
    // a lane loads 8 bytes per row: four bf16 channels (a workgroup then owns 128 channels) or two f32 ones (64 channels: with f32
    // rows and two pieces per operand the 128-channel form needs more than the 256 registers two waves per SIMD leave)
    constexpr bool A32 = sizeof(TA) == 4;
    constexpr int CPL = A32 ? 2 : 4, MT = 32 * CPL;
    __shared__ float red[MT * 64];    // one wave's accumulators
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * MT;
    const int batch = blockIdx.y / splits, split = blockIdx.y % splits;
    const int rbeg = batch * rows_per_batch + split * rows_per_wg;
    const int rend = min(rbeg + rows_per_wg, (batch + 1) * rows_per_batch);
    const int per_wave = ((rows_per_wg + 63) / 64) * 16;          // rows per wave, a multiple of 16
    const int wbeg = rbeg + wave * per_wave, wend = min(wbeg + per_wave, rend);
    float cs[CPL], ct[CPL];
#pragma unroll
    for (int q = 0; q < CPL; ++q) {
        const int c = m0 + CPL * i + q;
        const H16Affine a = h16_affine(bn.mean[c], bn.var[c], bn.gamma[c], bn.beta[c], bn.eps);
        cs[q] = a.s, ct[q] = a.t;
    }
    f32x16 acc[CPL][2];
#pragma unroll
    for (int q = 0; q < CPL; ++q)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][nt][r] = 0.f;
    const int last = max(rend - 1, rbeg);
    uint2 zn[8];
    float cn[2][8], rnn[8];
    auto load = [&](int rb) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int row = rb + 8 * h + j;
            const bool ok = row < wend;
            const size_t rr = (size_t)min(row, last);
            zn[j] = *reinterpret_cast<const uint2*>(Z + rr * 1024 + m0 + CPL * i);
            rnn[j] = ok ? rn[rr] : 0.f;                        // (a row past the range contributes rn = 0)
            cn[0][j] = C[rr * 64 + i];
            cn[1][j] = C[rr * 64 + 32 + i];
        }
    };
    if (wbeg < wend) load(wbeg);
    for (int rb = wbeg; rb < wend; rb += 16) {
        uint2 zv[8];
        float cv[2][8], rv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) zv[j] = zn[j], rv[j] = rnn[j], cv[0][j] = cn[0][j], cv[1][j] = cn[1][j];
        if (rb + 16 < wend) load(rb + 16);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 a[CPL][P], b[2][P];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float t[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) t[j] = cv[nt][j] * rv[j];
            hx_split<P>(t, b[nt]);
        }
#pragma unroll
        for (int q = 0; q < CPL; ++q) {
            float u[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float z;
                if constexpr (A32) z = __uint_as_float(q ? zv[j].y : zv[j].x);
                else z = (q & 1) ? bf_hi(q >> 1 ? zv[j].y : zv[j].x) : bf_lo(q >> 1 ? zv[j].y : zv[j].x);
                u[j] = fmaxf(z * cs[q] + ct[q], 0.f);
            }
            hx_split<P>(u, a[q]);
        }
#pragma unroll
        for (int q = 0; q < CPL; ++q)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) acc[q][nt] = hx_prod<P>(a[q], b[nt], acc[q][nt]);
    }
    // waves 1, 2, 3 hand their accumulators down in turn: ((w0 + w1) + w2) + w3 -- a fixed order
    for (int w = 1; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int q = 0; q < CPL; ++q)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) red[((q * 2 + nt) * 16 + r) * 64 + lane] = acc[q][nt][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int q = 0; q < CPL; ++q)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[q][nt][r] += red[((q * 2 + nt) * 16 + r) * 64 + lane];
        }
        __syncthreads();
    }
    if (wave == 0) {
        // D: lane (i', h'), register r of (q, nt) = out[channel m0 + CPL mfma_row(r, h') + q][column 32 nt + i']
        float* o = Pout + (size_t)blockIdx.y * 1024 * 64;
#pragma unroll
        for (int q = 0; q < CPL; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ch = m0 + CPL * mfma_row(r, h) + q;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) o[(size_t)ch * 64 + 32 * nt + i] = acc[q][nt][r];
            }
    }
}

// out[b][e] = sum over s < splits of P[b splits + s][e], ascending s; one float4 per thread, eight slices' loads in flight at a time
// (the 72 slices of dWc were 72 dependent round trips for 64 workgroups: 20 us for 19 MB)
static __global__ __launch_bounds__(256) void h16_partial_reduce_kernel(const float* __restrict__ P, int splits, long per,
                                                                        float* __restrict__ out) {
    const long e = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (e >= per) return;
    const float* p = P + (size_t)blockIdx.y * splits * per + e;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int k = 0;
    for (; k + 8 <= splits; k += 8) {
        float4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(p + (size_t)(k + u) * per);
#pragma unroll
        for (int u = 0; u < 8; ++u) s.x += v[u].x, s.y += v[u].y, s.z += v[u].z, s.w += v[u].w;
    }
    for (; k < splits; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(p + (size_t)k * per);
        s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
    }
    *reinterpret_cast<float4*>(out + (size_t)blockIdx.y * per + e) = s;
}

// Row slices per cloud of the column product (hx_colgemm_kernel): `tiles` channel tiles x num_clouds x S workgroups of four waves, two
// to a CU.  S is chosen so that the grid fills whole rounds of the chip's 2 x CUs workgroup slots: 8 x 18 x 4 = 576 workgroups ran as a
// full round and a ninth of one; 8 x 18 x 3 = 432 run in one (0.84 of the slots).  At least 256 rows per slice (the four waves' partial
// sums meet in LDS and leave as a 32-KB partial per slice); the 0.02 per slice prices those partials.
static inline int h16_splits(int num_clouds, int n_points, int tiles) {
    const long slots = 2L * epc_device_cu_count();
    int best = 1;
    double best_score = -1.0;
    for (int s = 1; s <= 8 && (s == 1 || n_points / s >= 256); ++s) {
        const long wgs = (long)tiles * num_clouds * s;
        const long rounds = (wgs + slots - 1) / slots;
        const double score = (double)wgs / (double)(rounds * slots) - 0.02 * s;
        if (score > best_score) best_score = score, best = s;
    }
    return best;
}
static inline bool h16_aligned16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }
