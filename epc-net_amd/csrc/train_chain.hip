// The 64-channel backbone of the TRAINING step as a chain of fused launches (models/epc-net.py:66-134, train.py:251-277;
// utils/tf_util.py:52-107, 454-519 in training mode).
//
// A ProxyConv block is  x = relu(bn0(u W0 + b0));  xm = mask x / k;  d = xm - x;  za = d Wa + ba;  zb = relu(bna(za)) Wb + bb;
// out = relu(bnb(zb)) + xm  -- three 64 -> 64 layers, each followed by a training-mode BatchNorm whose batch statistics need
// EVERY row before the first output can be normalised, and one neighbour mean that needs every row of x.  Until round 3 each
// of these grid-wide dependencies cost a chain of small launches (product + statistics, finalize, apply; column sums, finish,
// backward product, partial sum: >= 75 launches and ~0.9 ms per step for eleven layers whose tensors are 19 MB each).  Here a
// dependency costs ONE kernel boundary:
//
//   * a producer leaves per-workgroup PARTIALS (pivot-shifted moments of its pre-activation; or the two column sums of a
//     BatchNorm backward) and the CONSUMER's prologue pools them -- every workgroup redundantly, in double precision, in a fixed
//     order (a few hundred KB that sit in L2 / MALL): no finalize launch, bit-reproducible;
//   * a BatchNorm + ReLU is applied to an operand AS IT IS LOADED (the gather of the neighbour mean included: x is never
//     written), so a layer's kernel is "normalise the previous pre-activation, multiply, leave the next pre-activation + its
//     partial moments";
//   * in the backward every kernel also leaves the column sums the NEXT BatchNorm backward needs (it holds the gradient it
//     just formed and reads that layer's pre-activation for the mask), so no kernel exists only to reduce.
//
// Forward per block:  [gather: xm, d, za] -> [mid: zb] -> [head: out = relu(bnb(zb)) + xm -> cat slice; next block's z0]
// Backward per block: [conv_b] -> [conv_a: s = dd + dout] -> [gather^T: dx] -> [conv0: du + dcat slice]       (+ one launch that
// adds every layer's dW partials).  GEMM arithmetic as before: forward six products on three bf16 pieces per operand
// (f32-accurate), backward three products on two pieces; template parameter 1 = one bf16 value per operand (BASELINE.json
// configs[2]'s "bf16").  Layouts (rows x 64 f32, dense unless a stride is given) and the MFMA tile algebra are those of
// linear_stats64_kernel / linear_bn_bwd64_kernel (train_ops.hip), which these kernels replace inside the step.
#include "train_chain_common.h"


// ----------------------------------------------------------------------------------------------------------------
// FORWARD, row-streaming layer:  a = relu(bn(zin)) (+ resid);  [a -> a_out];  [z_out = a W + bias, moment partials]
//   mid  (conv_b):        zin = za, W = Wb                                   (models/epc-net.py:78-79)
//   head (block b + 1):   zin = zb, resid = xm, a_out = cat slice b, W = the next block's leading conv (:81-83)
//   tail (last block):    the same without W                                                        (:132-134)
// ----------------------------------------------------------------------------------------------------------------
struct ChFwdLinearArgs {
    const float* zin;
    ChBn bn;
    const float* resid;
    float* a_out;
    int a_stride;
    unsigned short* a_out16;   // optional: the same activation as bf16, same row stride in ELEMENTS (the bf16 head's operand: train_head16.hip)
    const float* W;
    const float* bias;
    float* z_out;
    float* stats_out;
    int rows, wg_rows;
    float eps;
};


template <int PF>
__global__ __launch_bounds__(64 * CH_FWD_MAX_WAVES) void chain_fwd_linear_kernel(ChFwdLinearArgs g) {
    extern __shared__ __attribute__((aligned(16))) double scratch[];   // 4 * 16 * 64 doubles: the pooling slices
    __shared__ u32x4 Wf[2][4][PF][64];
    __shared__ __attribute__((aligned(16))) float coef[2][64];
    __shared__ __attribute__((aligned(16))) float s_mean[64], s_var[64];
    __shared__ __attribute__((aligned(16))) float sred[CH_FWD_MAX_WAVES][3][64];
    __shared__ int snrows[CH_FWD_MAX_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int rows = g.rows;
    const int wg0 = blockIdx.x * g.wg_rows, tiles = min(g.wg_rows, rows - wg0 + 31) / 32;   // tiles of this workgroup (>= 1)
    // Prologue order: the partials (folded in registers) -> the wave's first ROWS are requested -> the slices merge and the weights
    // are staged while the rows travel.
    const ChBnRegs bnr = ch_bn_begin(g.bn, rows, scratch);
    float zr[4][8];
    {
        const int row = min(wg0 + wave * 32 + i, rows - 1);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) ch_ld8(g.zin + (size_t)row * 64 + 16 * s4 + 8 * h, zr[s4]);
    }
    if (g.W) {   // B[k = in][n = out]: lane (n = 32 nt + i, k group h) of k-step s holds W[16 s + 8 h .. + 7][n]
        for (int f = tid; f < 2 * 4 * 64; f += blockDim.x) {
            const int l = f & 63, s4 = (f >> 6) & 3, nt = f >> 8;
            const float* src = g.W + (size_t)(16 * s4 + 8 * (l >> 5)) * 64 + 32 * nt + (l & 31);
            float v[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = src[(size_t)q * 64];
            bf16x8 p[PF];
            ch_split<PF>(v, p);
#pragma unroll
            for (int pc = 0; pc < PF; ++pc) Wf[nt][s4][pc][l] = __builtin_bit_cast(u32x4, p[pc]);
        }
    }
    ch_bn_finish(g.bn, bnr, rows, g.eps, scratch, s_mean, s_var, coef);   // (its barriers also cover Wf)

    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, piv[2] = {0.f, 0.f};
    int my_rows = 0;
    const float b0 = g.bias ? g.bias[i] : 0.f, b1 = g.bias ? g.bias[32 + i] : 0.f;
#pragma unroll 1
    for (int tl = wave; tl < tiles; tl += nw) {
        const int base = wg0 + tl * 32;   // wave-uniform, < rows
        const int row = base + i;
        const bool ok = row < rows;
        if (tl != wave) {                 // (the first tile's rows were requested in the prologue)
            const int r = min(row, rows - 1);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) ch_ld8(g.zin + (size_t)r * 64 + 16 * s4 + 8 * h, zr[s4]);
        }
        bf16x8 a[4][PF];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int c0 = 16 * s4 + 8 * h;
            float cs[8], ct[8], v[8];
            ch_ld8(&coef[0][c0], cs), ch_ld8(&coef[1][c0], ct);
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = fmaxf(zr[s4][q] * cs[q] + ct[q], 0.f);
            if (g.resid) {
                float r[8];
                ch_ld8(g.resid + (size_t)(ok ? row : 0) * 64 + c0, r);
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] += r[q];
            }
            if (g.a_out && ok) ch_st8(g.a_out + (size_t)row * g.a_stride + c0, v);
            if (g.a_out16 && ok) {
                bf16x8 pk;
#pragma unroll
                for (int q = 0; q < 8; ++q) pk[q] = (__bf16)v[q];
                *reinterpret_cast<u32x4*>(g.a_out16 + (size_t)row * g.a_stride + c0) = __builtin_bit_cast(u32x4, pk);
            }
            if (!ok) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = 0.f;
            }
            ch_split<PF>(v, a[s4]);
        }
        if (!g.W) continue;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                bf16x8 w[PF];
#pragma unroll
                for (int pc = 0; pc < PF; ++pc) w[pc] = __builtin_bit_cast(bf16x8, Wf[nt][s4][pc][lane]);
                acc = ch_prod<PF>(a[s4], w, acc);
            }
            const float bv = nt ? b1 : b0;
            if (tl == wave) piv[nt] = __shfl(acc[0], i);   // row `base` of the wave's first tile
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = base + mfma_row(r, h);
                if (rr < rows) {
                    const float v = acc[r];   // (statistics of the product WITHOUT the bias: the pooling adds it to the mean)
                    const float d = v - piv[nt];
                    s1[nt] += d;
                    s2[nt] += d * d;
                    g.z_out[(size_t)rr * 64 + 32 * nt + i] = v + bv;
                }
            }
        }
        my_rows += min(32, rows - base);
    }
    if (!g.W) return;
    ch_store_stats(s1, s2, piv, my_rows, sred, snrows, g.stats_out + (size_t)blockIdx.x * 192);
}

// ----------------------------------------------------------------------------------------------------------------
// FORWARD, gather layer (models/epc-net.py:70-76):  x = relu(bn0(z0)) formed as the rows are GATHERED (x is never written);
//   xm = (sum over the point's selected neighbours of x) / k;  d = xm - x;  za = d Wa + ba  (+ moment partials of za).
// 16 lanes x float4 per row, four points per wave-instruction, the first 20 rows of a list in flight at once, summed in list order
// (neighbour_mean_kernel's order); rows with more than `cap` selected entries (exact ties: duplicated / zero-padded clouds) take
// the exact scan.  d goes through a per-wave LDS tile into the MFMA's row layout.
// ----------------------------------------------------------------------------------------------------------------
struct ChFwdGatherArgs {
    const float* z0;
    ChBn bn;
    const float* xyz;
    const int32_t* idx;
    const int32_t* cnt;
    const float* kth;
    int cap, n;
    float kdiv;
    const float* W;
    const float* bias;
    float* xm;
    float* d;
    float* z_out;
    float* stats_out;
    int rows, wg_rows;
    float eps;
};

template <int PF>
__global__ __launch_bounds__(64 * CH_GATHER_MAX_WAVES) void chain_fwd_gather_kernel(ChFwdGatherArgs g) {
    extern __shared__ __attribute__((aligned(16))) double scratch[];   // pooling slices (4 * 16 * 64 doubles); then one staging tile per wave
    __shared__ u32x4 Wf[2][4][PF][64];
    __shared__ __attribute__((aligned(16))) float coef[2][64];
    __shared__ __attribute__((aligned(16))) float s_mean[64], s_var[64];
    __shared__ __attribute__((aligned(16))) float sred[CH_GATHER_MAX_WAVES][3][64];
    __shared__ int snrows[CH_GATHER_MAX_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int p4 = lane >> 4, q = lane & 15;
    const int rows = g.rows;
    const int lb = xcd_contiguous_block(blockIdx.x, gridDim.x);   // a cloud's tiles behind ONE L2 (speed only)
    const int wg0 = lb * g.wg_rows, tiles = min(g.wg_rows, rows - wg0 + 31) / 32;
    const ChBnRegs bnr = ch_bn_begin(g.bn, rows, scratch);
    for (int f = tid; f < 2 * 4 * 64; f += blockDim.x) {
        const int l = f & 63, s4 = (f >> 6) & 3, nt = f >> 8;
        const float* src = g.W + (size_t)(16 * s4 + 8 * (l >> 5)) * 64 + 32 * nt + (l & 31);
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)u * 64];
        bf16x8 p[PF];
        ch_split<PF>(v, p);
#pragma unroll
        for (int pc = 0; pc < PF; ++pc) Wf[nt][s4][pc][l] = __builtin_bit_cast(u32x4, p[pc]);
    }
    // (pooling writes mean / var from workgroup 0 of the DISPATCH order: any one workgroup will do)
    ch_bn_finish(g.bn, bnr, rows, g.eps, scratch, s_mean, s_var, coef);
    const float4 cs = *reinterpret_cast<const float4*>(&coef[0][4 * q]), ct = *reinterpret_cast<const float4*>(&coef[1][4 * q]);
    auto act = [&](const float4& v) {   // relu(bn0(.)) of the lane's four channels: the forward's own expression
        return make_float4(fmaxf(v.x * cs.x + ct.x, 0.f), fmaxf(v.y * cs.y + ct.y, 0.f), fmaxf(v.z * cs.z + ct.z, 0.f),
                           fmaxf(v.w * cs.w + ct.w, 0.f));
    };
    __syncthreads();   // every wave has its coefficients: the pooling slices may be overwritten by the staging tiles
    float* stg = reinterpret_cast<float*>(scratch) + wave * CH_STG_FLOATS;
    const float4* z4 = reinterpret_cast<const float4*>(g.z0);

    float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, piv[2] = {0.f, 0.f};
    int my_rows = 0;
    const float b0 = g.bias ? g.bias[i] : 0.f, b1 = g.bias ? g.bias[32 + i] : 0.f;
#pragma unroll 1
    for (int tl = wave; tl < tiles; tl += nw) {
        const int base = wg0 + tl * 32;   // wave-uniform, < rows
#pragma unroll 1
        for (int r8 = 0; r8 < 8; ++r8) {
            const int pt = base + 4 * r8 + p4;
            float4 dd = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pt < rows) {
                const int cloud_base = (pt / g.n) * g.n;
                const int c = g.cnt[pt];
                float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                auto add = [&](const float4& v) {
                    const float4 y = act(v);
                    acc.x += y.x, acc.y += y.y, acc.z += y.z, acc.w += y.w;
                };
                if (c <= g.cap) {
                    int m = 0;
                    if (c >= 20 && g.cap % 4 == 0) {
                        const int4* il = reinterpret_cast<const int4*>(g.idx + (size_t)pt * g.cap);
                        int nb[20];
#pragma unroll
                        for (int m4 = 0; m4 < 5; ++m4) {
                            const int4 tq = il[m4];
                            nb[4 * m4] = tq.x, nb[4 * m4 + 1] = tq.y, nb[4 * m4 + 2] = tq.z, nb[4 * m4 + 3] = tq.w;
                        }
                        float4 v[20];
#pragma unroll
                        for (int u = 0; u < 20; ++u) v[u] = z4[(size_t)(cloud_base + nb[u]) * 16 + q];
#pragma unroll
                        for (int u = 0; u < 20; ++u) add(v[u]);
                        m = 20;
                    }
                    for (; m < c; ++m) add(z4[(size_t)(cloud_base + g.idx[(size_t)pt * g.cap + m]) * 16 + q]);
                } else {
                    const float* pc = g.xyz + (size_t)cloud_base * 3;
                    const int ii = pt - cloud_base;
                    const float xi = pc[3 * ii], yi = pc[3 * ii + 1], zi = pc[3 * ii + 2];
                    const float sqi = sq3(xi, yi, zi), kv = g.kth[pt];
                    for (int j = 0; j < g.n; ++j) {
                        const float xj = pc[3 * j], yj = pc[3 * j + 1], zj = pc[3 * j + 2];
                        if (neg_sq_dist(sqi, xi, yi, zi, xj, yj, zj, sq3(xj, yj, zj)) >= kv) add(z4[(size_t)(cloud_base + j) * 16 + q]);
                    }
                }
                acc.x /= g.kdiv, acc.y /= g.kdiv, acc.z /= g.kdiv, acc.w /= g.kdiv;
                const float4 own = act(z4[(size_t)pt * 16 + q]);
                dd = make_float4(acc.x - own.x, acc.y - own.y, acc.z - own.z, acc.w - own.w);
                reinterpret_cast<float4*>(g.xm)[(size_t)pt * 16 + q] = acc;
                reinterpret_cast<float4*>(g.d)[(size_t)pt * 16 + q] = dd;
            }
            *reinterpret_cast<float4*>(stg + (4 * r8 + p4) * CH_STG_STRIDE + 4 * q) = dd;   // (rows past the end: zeros)
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave: its own tile writes have landed before its reads
        bf16x8 a[4][PF];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            float v[8];
            ch_ld8(stg + i * CH_STG_STRIDE + 16 * s4 + 8 * h, v);
            ch_split<PF>(v, a[s4]);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // in registers: the tile may be overwritten by the next round
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                bf16x8 w[PF];
#pragma unroll
                for (int pc = 0; pc < PF; ++pc) w[pc] = __builtin_bit_cast(bf16x8, Wf[nt][s4][pc][lane]);
                acc = ch_prod<PF>(a[s4], w, acc);
            }
            const float bv = nt ? b1 : b0;
            if (tl == wave) piv[nt] = __shfl(acc[0], i);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int rr = base + mfma_row(r, h);
                if (rr < rows) {
                    const float v = acc[r];
                    const float dlt = v - piv[nt];
                    s1[nt] += dlt;
                    s2[nt] += dlt * dlt;
                    g.z_out[(size_t)rr * 64 + 32 * nt + i] = v + bv;
                }
            }
        }
        my_rows += min(32, rows - base);
    }
    ch_store_stats(s1, s2, piv, my_rows, sred, snrows, g.stats_out + (size_t)lb * 192);
}

// ----------------------------------------------------------------------------------------------------------------
// BACKWARD of a 64 -> 64 layer + training-mode BatchNorm + ReLU in one pass over its rows (linear_bn_bwd64_kernel of
// train_ops.hip with the launches around it folded in):
//   prologue  the layer's two BatchNorm sums pooled from the PRODUCER's partials (-> dbeta, dgamma);
//   per tile  dz = gamma rstd (dy [z-mask] - dbeta / rows - zhat dgamma / rows);  dx = dz W^T (+ addend) -> dx_out;
//             dW partial += in^T dz with in = x, or relu(bn_x(x)) formed as x is loaded (conv_b: the activation between conv_a and
//             conv_b was never written);
//             the column sums the NEXT BatchNorm backward needs, from the gradient just formed (dx + addend) and that layer's
//             pre-activation zp: sum g [zp-mask], sum g [zp-mask] zhat_p  -> one partial per workgroup.
// Strides: dy, x and the addend may be 64-channel slices of wider tensors (the concat buffer of models/epc-net.py:134 and its
// gradient).
// ----------------------------------------------------------------------------------------------------------------
struct ChBwdLinearArgs {
    const float* dy;
    int dy_stride;
    const float* z;
    ChBnGiven bn;
    const float* sums;   // partials [parts][2][64] of this layer's BatchNorm sums
    int parts;
    float* dgamma;
    float* dbeta;
    const float* W;
    const float* x;
    int x_stride;
    ChBnGiven xbn;
    float* dx;
    const float* dx_addend;
    int addend_stride;
    float* dWpart;
    const float* zp;
    ChBnGiven pbn;
    float* psums;
    int rows, wg_rows;
    float eps;
};

template <int PB>
__global__ __launch_bounds__(64 * CH_BWD_MAX_WAVES) void chain_bwd_linear_kernel(ChBwdLinearArgs g) {
    extern __shared__ __attribute__((aligned(16))) char img_all[];   // per wave: hi + lo image (2 * CH_IMG_BYTES); also the pooling scratch
                                                                      // (16 KB) and, at the end, the parked partials (16 KB per parked wave)
    __shared__ __attribute__((aligned(16))) float coef[6][64];    // s, t (mask), mean, k1, dbeta / rows, rstd dgamma / rows
    __shared__ __attribute__((aligned(16))) float xcoef[2][64];   // the input's BatchNorm: s, t
    __shared__ __attribute__((aligned(16))) float pcoef[4][64];   // the producer's BatchNorm: s, t (mask), mean, rstd
    __shared__ __attribute__((aligned(16))) float s_sum[2][64];
    __shared__ u32x4 Wf[2][4][PB][64];                             // W as A fragments: [in tile][k-step][piece][lane]
    __shared__ __attribute__((aligned(16))) float sumt[CH_BWD_MAX_WAVES][CH_SUMT_WORDS];   // per wave: [2 quantities][32 rows][8 channels] of the producer's sums
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int rows = g.rows;
    const int wg0 = blockIdx.x * g.wg_rows, tiles = min(g.wg_rows, rows - wg0 + 31) / 32;
    const float inv_rows = 1.0f / (float)rows;
    // W (in, out) row-major: A[m = in][k = out]; lane (m = 32 mt + i, k group h) of k-step s holds W[m][16 s + 8 h .. + 7]
    for (int f = tid; f < 2 * 4 * 64; f += blockDim.x) {
        const int l = f & 63, s4 = (f >> 6) & 3, mt = f >> 8;
        float v[8];
        ch_ld8(g.W + (size_t)(32 * mt + (l & 31)) * 64 + 16 * s4 + 8 * (l >> 5), v);
        bf16x8 p[PB];
        ch_split<PB>(v, p);
#pragma unroll
        for (int pc = 0; pc < PB; ++pc) Wf[mt][s4][pc][l] = __builtin_bit_cast(u32x4, p[pc]);
    }
    // the wave's first tile of dy and z is requested BEFORE the prologue's dependent round trips (pooling the sum partials, the
    // coefficients): in flight under all of it (later tiles' at the head of their body; clamped rows)
    float gv0[4][8], zv0[4][8];
    auto request_tile = [&](int tl, float (&gv)[4][8], float (&zv)[4][8]) {
        const int r = min(wg0 + tl * 32 + i, rows - 1);
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4)
            ch_ld8(g.dy + (size_t)r * g.dy_stride + 16 * s4 + 8 * h, gv[s4]), ch_ld8(g.z + (size_t)r * 64 + 16 * s4 + 8 * h, zv[s4]);
    };
    request_tile(wave, gv0, zv0);
    // the three BatchNorms' vectors are requested BEFORE the pooling, unconditionally (an absent one reads this layer's instead): one round
    // trip under the pooling -- guarded by their pointers and placed behind it they were three dependent round trips of every launch's
    // prologue (found in the ISA, round 6)
    float bv[3][4] = {};
    if (tid < 64) {
        const bool hx = g.xbn.mean != nullptr, hp = g.zp != nullptr;
        bv[0][0] = g.bn.mean[tid], bv[0][1] = g.bn.var[tid], bv[0][2] = g.bn.gamma[tid], bv[0][3] = g.bn.beta[tid];
        bv[1][0] = (hx ? g.xbn.mean : g.bn.mean)[tid], bv[1][1] = (hx ? g.xbn.var : g.bn.var)[tid];
        bv[1][2] = (hx ? g.xbn.gamma : g.bn.gamma)[tid], bv[1][3] = (hx ? g.xbn.beta : g.bn.beta)[tid];
        bv[2][0] = (hp ? g.pbn.mean : g.bn.mean)[tid], bv[2][1] = (hp ? g.pbn.var : g.bn.var)[tid];
        bv[2][2] = (hp ? g.pbn.gamma : g.bn.gamma)[tid], bv[2][3] = (hp ? g.pbn.beta : g.bn.beta)[tid];
    }
#ifdef CH_ABL_NOPOOL   // timing only
    if (tid < 128) s_sum[tid >> 6][tid & 63] = 0.f;
    __syncthreads();
#else
    ch_pool_sums(g.sums, g.parts, reinterpret_cast<double*>(img_all), s_sum, g.dbeta, g.dgamma);
#endif
    if (tid < 64) {
        const float mu = bv[0][0], rs = 1.0f / sqrtf(bv[0][1] + g.eps), ga = bv[0][2];
        const ChBnAffine a = ch_bn_affine(mu, bv[0][1], ga, bv[0][3], g.eps);
        coef[0][tid] = a.s, coef[1][tid] = a.t, coef[2][tid] = mu;
        coef[3][tid] = ga * rs, coef[4][tid] = s_sum[0][tid] * inv_rows, coef[5][tid] = rs * (s_sum[1][tid] * inv_rows);
        if (g.xbn.mean) {
            const ChBnAffine xa = ch_bn_affine(bv[1][0], bv[1][1], bv[1][2], bv[1][3], g.eps);
            xcoef[0][tid] = xa.s, xcoef[1][tid] = xa.t;
        }
        if (g.zp) {
            const float pm = bv[2][0];
            const ChBnAffine pa = ch_bn_affine(pm, bv[2][1], bv[2][2], bv[2][3], g.eps);
            pcoef[0][tid] = pa.s, pcoef[1][tid] = pa.t, pcoef[2][tid] = pm, pcoef[3][tid] = 1.0f / sqrtf(bv[2][1] + g.eps);
        }
    }
    __syncthreads();

    f32x16 accW[2][2];   // [in tile mt][out tile nt]: register 4g + e = in channel 32 mt + 8 g + 4 h + e, lane = out channel 32 nt + i
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) accW[mt][nt][r] = 0.f;
    // the producer's sums: lane l keeps, for each of the 8 channel octets (mt, gq), the column (quantity (l >> 3) & 1, channel
    // 32 mt + 8 gq + (l & 7)) summed over the rows of its tiles
    float psum[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) psum[o] = 0.f;
    float* mysum = sumt[wave];

    char* my = img_all + (size_t)wave * 2 * CH_IMG_BYTES;
    auto put = [&](int s4, const bf16x8 (&p)[PB]) {   // the lane's row i, channels 16 s4 + 8 h .. + 7 = chunk 2 s4 + h
        const int o = ch_img_off(i, 2 * s4 + h);
#pragma unroll
        for (int pc = 0; pc < PB; ++pc) *reinterpret_cast<u32x4*>(my + pc * CH_IMG_BYTES + o) = __builtin_bit_cast(u32x4, p[pc]);
    };

    // one tile of 32 rows; gv / zv: its dy and z rows, already requested
    auto do_tile = [&](int tl, float (&gv)[4][8], float (&zv)[4][8]) {
        const int base = wg0 + tl * 32;   // wave-uniform, < rows
        const int row = base + i;
        const bool ok = row < rows;
        const size_t rsafe = (size_t)(ok ? row : 0);
        // ---- dz (row layout): B fragments of dx^T = W dz^T as they stand; its pieces also go to the image ----
        bf16x8 zf[4][PB];
        {
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                asm volatile("" ::: "memory");   // (keeps a k-step's coefficient reads next to their use)
                float cs[8], ct[8], mu[8], k1[8], bb[8], gg[8], dzv[8];
                const int c0 = 16 * s4 + 8 * h;
                ch_ld8(&coef[0][c0], cs), ch_ld8(&coef[1][c0], ct), ch_ld8(&coef[2][c0], mu), ch_ld8(&coef[3][c0], k1),
                    ch_ld8(&coef[4][c0], bb), ch_ld8(&coef[5][c0], gg);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float dd = !(zv[s4][u] * cs[u] + ct[u] > 0.f) ? 0.f : gv[s4][u];   // the forward's own expression
                    const float v = k1[u] * (dd - bb[u] - (zv[s4][u] - mu[u]) * gg[u]);
                    dzv[u] = ok ? v : 0.f;
                }
                ch_split<PB>(dzv, zf[s4]);
                put(s4, zf[s4]);
            }
        }
        // x's rows are requested now: they land under the dx products
        float xv[4][8];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) ch_ld8(g.x + rsafe * g.x_stride + 16 * s4 + 8 * h, xv[s4]);
#ifdef CH_ABL_NODX   // timing only
        if (false) {
#else
        if (g.dx) {
#endif
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    bf16x8 w[PB];
#pragma unroll
                    for (int pc = 0; pc < PB; ++pc) w[pc] = __builtin_bit_cast(bf16x8, Wf[mt][s4][pc][lane]);
                    acc = ch_prod<PB>(w, zf[s4], acc);
                }
                // the in-tile's addend and producer rows are requested together, ahead of the octet loop (whose LDS waits would
                // otherwise serialise eight global round trips)
                float4 av[4], zq4[4];
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int c0 = 32 * mt + 8 * gq + 4 * h;
                    av[gq] = g.dx_addend ? *reinterpret_cast<const float4*>(g.dx_addend + rsafe * g.addend_stride + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
                    zq4[gq] = g.zp ? *reinterpret_cast<const float4*>(g.zp + rsafe * 64 + c0) : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const int c0 = 32 * mt + 8 * gq + 4 * h;
                    float4 v = make_float4(acc[4 * gq], acc[4 * gq + 1], acc[4 * gq + 2], acc[4 * gq + 3]);
                    if (g.dx_addend) v.x += av[gq].x, v.y += av[gq].y, v.z += av[gq].z, v.w += av[gq].w;
                    if (ok) *reinterpret_cast<float4*>(g.dx + (size_t)row * 64 + c0) = v;
                    if (g.zp) {   // the producer's BatchNorm sums of this gradient: d1 = g [mask], d2 = g [mask] zhat -> the sum tile
                        const float4 zq = zq4[gq];
                        const float4 qs = *reinterpret_cast<const float4*>(&pcoef[0][c0]), qt = *reinterpret_cast<const float4*>(&pcoef[1][c0]);
                        const float4 qm = *reinterpret_cast<const float4*>(&pcoef[2][c0]), qr = *reinterpret_cast<const float4*>(&pcoef[3][c0]);
                        const float vv[4] = {v.x, v.y, v.z, v.w}, zz[4] = {zq.x, zq.y, zq.z, zq.w};
                        const float a_s[4] = {qs.x, qs.y, qs.z, qs.w}, a_t[4] = {qt.x, qt.y, qt.z, qt.w};
                        const float a_m[4] = {qm.x, qm.y, qm.z, qm.w}, a_r[4] = {qr.x, qr.y, qr.z, qr.w};
                        float d1[4], d2[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            d1[e] = (ok && (zz[e] * a_s[e] + a_t[e] > 0.f)) ? vv[e] : 0.f;
                            d2[e] = d1[e] * ((zz[e] - a_m[e]) * a_r[e]);
                        }
                        *reinterpret_cast<float4*>(mysum + ch_sumt_word(0, i, 4 * h)) = make_float4(d1[0], d1[1], d1[2], d1[3]);
                        *reinterpret_cast<float4*>(mysum + ch_sumt_word(1, i, 4 * h)) = make_float4(d2[0], d2[1], d2[2], d2[3]);
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave: its writes have landed
                        const int qn = (lane >> 3) & 1, chn = lane & 7, rgp = lane >> 4;
                        float part = 0.f;
#pragma unroll
                        for (int r = 0; r < 8; ++r) part += mysum[ch_sumt_word(qn, 8 * rgp + r, chn)];   // rows in ascending order
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read: the tile may be overwritten by the next octet
                        // the four row groups (lanes l, l ^ 16, l ^ 32, l ^ 48) in a fixed order: ((g0 + g1) + (g2 + g3))
                        part += __shfl_xor(part, 16);
                        part += __shfl_xor(part, 32);
                        psum[4 * mt + gq] += part;
                    }
                }
            }
        }
#ifdef CH_ABL_NODW   // timing only
        return;
#endif
        // ---- dz^T: the B fragments of dW (k = rows, n = out channel), read transposed from the image ----
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (one wave: its own image writes have landed before its reads)
        bf16x8 dzt[2][2][PB];   // [out tile][k-step][piece]
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int pc = 0; pc < PB; ++pc) dzt[nt][s2][pc] = ch_tr_frag(my + pc * CH_IMG_BYTES, nt, s2, lane);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // they are in registers: the image may be overwritten
        // ---- x (row layout) -> pieces -> image -> A fragments of dW (k = rows, m = in channel) ----
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            asm volatile("" ::: "memory");
            if (g.xbn.mean) {
                float cs[8], ct[8];
                ch_ld8(&xcoef[0][16 * s4 + 8 * h], cs), ch_ld8(&xcoef[1][16 * s4 + 8 * h], ct);
#pragma unroll
                for (int u = 0; u < 8; ++u) xv[s4][u] = fmaxf(xv[s4][u] * cs[u] + ct[u], 0.f);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) xv[s4][u] = ok ? xv[s4][u] : 0.f;   // (a select, not a branch)
            bf16x8 p[PB];
            ch_split<PB>(xv[s4], p);
            put(s4, p);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bf16x8 xt[PB];
#pragma unroll
                for (int pc = 0; pc < PB; ++pc) xt[pc] = ch_tr_frag(my + pc * CH_IMG_BYTES, mt, s2, lane);
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) accW[mt][nt] = ch_prod<PB>(xt, dzt[nt][s2], accW[mt][nt]);
            }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // before the next tile overwrites the image
    };
    if (wave < tiles) do_tile(wave, gv0, zv0);
#pragma unroll 1
    for (int tl = wave + nw; tl < tiles; tl += nw) {   // (their rows are requested at the head of the body: a prefetch under the previous
        float gv[4][8], zv[4][8];                      //  tile would keep 64 more registers live beside accW)
        request_tile(tl, gv, zv);
        do_tile(tl, gv, zv);
    }
    // ---- the producer's sums: the waves' column sums meet in wave order: one partial per workgroup ----
    __syncthreads();   // every wave is done with its image: what follows aliases them
    if (g.zp) {
        float (*pred)[2][64] = reinterpret_cast<float (*)[2][64]>(img_all);   // [wave][sum][channel]
        if (lane < 16) {
#pragma unroll
            for (int o = 0; o < 8; ++o) pred[wave][(lane >> 3) & 1][32 * (o >> 2) + 8 * (o & 3) + (lane & 7)] = psum[o];
        }
        __syncthreads();
        for (int o = tid; o < 128; o += blockDim.x) {   // (a one-wave workgroup of a short input takes both sums in turn)
            const int k = o >> 6, c = o & 63;
            float t = pred[0][k][c];
            for (int w = 1; w < nw; ++w) t += pred[w][k][c];
            g.psums[((size_t)blockIdx.x * 2 + k) * 64 + c] = t;
        }
        __syncthreads();
    }
    // ---- the waves' dW partials meet in a fixed binary tree, ((w0 + w1) + (w2 + w3)) + ((w4 + w5) + (w6 + w7)) with absent waves
    //      left out; wave 0 stores the workgroup's.  A parked partial is 16 KB: the images of two waves. ----
    float (*red)[4][16][64] = reinterpret_cast<float (*)[4][16][64]>(img_all);   // [slot][tile][register][lane]
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
#pragma unroll
    for (int step = 1; step < CH_BWD_MAX_WAVES; step <<= 1) {
        // waves that are odd multiples of `step` park (slot = wave / (2 step)); their even partners take -- if the parker exists
        if ((wave & (2 * step - 1)) == step) park(wave / (2 * step));
        __syncthreads();
        if ((wave & (2 * step - 1)) == 0 && wave + step < nw) take(wave / (2 * step));
        __syncthreads();
    }
    if (wave == 0) {
        float* out = g.dWpart + (size_t)blockIdx.x * 4096;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) out[(32 * mt + mfma_row(r, h)) * 64 + 32 * nt + i] = accW[mt][nt][r];
    }
}

// ----------------------------------------------------------------------------------------------------------------
// BACKWARD of the gather layer:  s = d(d) + d(out) arrives as ONE tensor (the conv_a kernel above wrote dx + addend), dout = d(out):
//   dx[j] = (sum over the points i that list j of s[i]) / k - (s[j] - dout[j])            (mask^T s / k - d(d), models/epc-net.py:70-72)
// over the transposed graph (epc_knn_transpose); the points whose own list overflowed (cnt > cap) are not in it and are visited
// through the cloud's overflow list with the exact test -- a gather as well, so no float atomics and no second launch.  Also leaves
// the column sums of x's BatchNorm backward (x = relu(bn0(z0))): sum dx [z0-mask], sum dx [z0-mask] zhat0.
// ----------------------------------------------------------------------------------------------------------------
struct ChBwdGatherArgs {
    const float* s;
    const float* dout;
    int dout_stride;
    const int32_t* rdeg;
    const int32_t* roff;
    const int32_t* rlist;
    const int32_t* ovf_cnt;    // per cloud: how many of its points have cnt > cap
    const int32_t* ovf_list;   // per cloud n slots: their in-cloud indices, ascending
    const float* xyz;
    const float* kth;
    int n, total, wg_rows;
    float kdiv;
    const float* z0;
    ChBnGiven bn;
    float* psums;
    float* dx;
    float eps;
};

// 1024 threads: 64 points in flight per workgroup, wg_rows / 64 sequential points per 16-lane slot (one partial per workgroup).
// (256 threads walking 16 points each took 107 us per launch at 18 x 4096: the chain degree -> offset -> list -> rows is four
// dependent round trips per point, and only waves in flight hide them.)
#define CH_GB_THREADS 1024
// entry u of a point's list, clamped to its last one; an EMPTY list (a point nobody lists) reads the word at its offset -- possibly past the
// written part of rlist -- and takes the point itself instead (a valid row; zero weight either way)
__device__ __forceinline__ int ch_gb_entry(const int32_t* __restrict__ lst, int u, int deg, int self) {
    const int e = lst[min(u, max(deg - 1, 0))];
    return deg > 0 ? e : self;
}
#define CH_GB_SLOTS (CH_GB_THREADS / 16)
__global__ __launch_bounds__(CH_GB_THREADS) void chain_bwd_gather_kernel(ChBwdGatherArgs g) {
    __shared__ __attribute__((aligned(16))) float pcoef[4][64];
    __shared__ __attribute__((aligned(16))) float red[CH_GB_SLOTS][2][64];
    const int tid = threadIdx.x, slot = tid >> 4, q = tid & 15;
    const int lb = xcd_contiguous_block(blockIdx.x, gridDim.x);
    if (tid < 64) {
        const float pm = g.bn.mean[tid];
        const ChBnAffine pa = ch_bn_affine(pm, g.bn.var[tid], g.bn.gamma[tid], g.bn.beta[tid], g.eps);
        pcoef[0][tid] = pa.s, pcoef[1][tid] = pa.t, pcoef[2][tid] = pm, pcoef[3][tid] = 1.0f / sqrtf(g.bn.var[tid] + g.eps);
    }
    __syncthreads();
    const float4* s4 = reinterpret_cast<const float4*>(g.s);
    float4 t1 = make_float4(0.f, 0.f, 0.f, 0.f), t2 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int wg_end = min(g.total, (lb + 1) * g.wg_rows);
    // A slot's points are a chain of dependent round trips -- degree / offset -> list entries -> rows, batch after batch -- and what hides
    // them is only the other slots in flight: the first form of this loop (everything of a point requested when the point came up, the
    // list in dependent batches of 8) took ~9 round trips per point, 37 us per launch at 18 x 4096.  Now the NEXT point's degree, offset
    // and first 8 entries are requested under this point's rows, the entries 8 .. 23 together with the rows of the first 8, and rows
    // travel in two buffers of 8: two round trips for a point of up to 24 listers (the mean is 20).  Every load is UNCONDITIONAL (entries
    // clamped to the list's last one, their rows given zero weight by a select; the row sums keep the list order: the same bits).
    int j = lb * g.wg_rows + slot;
    int deg = 0;
    const int32_t* lst = g.rlist;
    int e0[8];
    {
        const int jc = min(j, wg_end - 1);
        deg = g.rdeg[jc];
        lst = g.rlist + g.roff[jc];
#pragma unroll
        for (int u = 0; u < 8; ++u) e0[u] = ch_gb_entry(lst, u, deg, jc);
    }
#pragma unroll 1
    for (; j < wg_end; j += CH_GB_SLOTS) {
        const int jn = min(j + CH_GB_SLOTS, wg_end - 1);
        const int degn = g.rdeg[jn], offn = g.roff[jn];
        int e1[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) e1[u] = ch_gb_entry(lst, 8 + u, deg, j);
        float4 v0[8], v1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v0[u] = s4[(size_t)e0[u] * 16 + q];
#pragma unroll
        for (int u = 0; u < 8; ++u) v1[u] = s4[(size_t)e1[u] * 16 + q];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float w = u < deg ? 1.f : 0.f;
            acc.x += w * v0[u].x, acc.y += w * v0[u].y, acc.z += w * v0[u].z, acc.w += w * v0[u].w;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) v0[u] = s4[(size_t)e1[8 + u] * 16 + q];
        // the next point's first entries, and this point's own rows (its epilogue)
        const int32_t* lstn = g.rlist + offn;
        int e0n[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) e0n[u] = ch_gb_entry(lstn, u, degn, jn);
        const float4 a = s4[(size_t)j * 16 + q], b = *reinterpret_cast<const float4*>(g.dout + (size_t)j * g.dout_stride + 4 * q);
        const float4 zq = reinterpret_cast<const float4*>(g.z0)[(size_t)j * 16 + q];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float w = 8 + u < deg ? 1.f : 0.f;
            acc.x += w * v1[u].x, acc.y += w * v1[u].y, acc.z += w * v1[u].z, acc.w += w * v1[u].w;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const float w = 16 + u < deg ? 1.f : 0.f;
            acc.x += w * v0[u].x, acc.y += w * v0[u].y, acc.z += w * v0[u].z, acc.w += w * v0[u].w;
        }
        for (int m = 24; m < deg; m += 8) {   // (a hub: dependent batches of 8, the last one clamped)
            int ii[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) ii[u] = lst[min(m + u, deg - 1)];
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = s4[(size_t)ii[u] * 16 + q];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (m + u < deg) acc.x += v[u].x, acc.y += v[u].y, acc.z += v[u].z, acc.w += v[u].w;
        }
        const int cloud = j / g.n, cloud_base = cloud * g.n;
        const int novf = g.ovf_cnt[cloud];
        if (novf > 0) {   // the cloud's overflowed points: does i select j?  (a_ij >= kth_i, utils/tf_util.py:662-664)
            const float* pc = g.xyz + (size_t)cloud_base * 3;
            const int jj = j - cloud_base;
            const float xj = pc[3 * jj], yj = pc[3 * jj + 1], zj = pc[3 * jj + 2];
            const float sqj = sq3(xj, yj, zj);
            const int32_t* ol = g.ovf_list + (size_t)cloud * g.n;
            for (int u = 0; u < novf; ++u) {
                const int ii = ol[u];
                const float xi = pc[3 * ii], yi = pc[3 * ii + 1], zi = pc[3 * ii + 2];
                if (neg_sq_dist(sq3(xi, yi, zi), xi, yi, zi, xj, yj, zj, sqj) >= g.kth[cloud_base + ii]) {
                    const float4 v = s4[(size_t)(cloud_base + ii) * 16 + q];
                    acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
                }
            }
        }
        const float4 dxv = make_float4(acc.x / g.kdiv - (a.x - b.x), acc.y / g.kdiv - (a.y - b.y), acc.z / g.kdiv - (a.z - b.z),
                                       acc.w / g.kdiv - (a.w - b.w));
        reinterpret_cast<float4*>(g.dx)[(size_t)j * 16 + q] = dxv;
        // (the producer's coefficients from LDS here, not from sixteen registers held across the loop: the rows' buffers need them)
        const float4 qs = *reinterpret_cast<const float4*>(&pcoef[0][4 * q]), qt = *reinterpret_cast<const float4*>(&pcoef[1][4 * q]);
        const float4 qm = *reinterpret_cast<const float4*>(&pcoef[2][4 * q]), qr = *reinterpret_cast<const float4*>(&pcoef[3][4 * q]);
        const float vv[4] = {dxv.x, dxv.y, dxv.z, dxv.w}, zz[4] = {zq.x, zq.y, zq.z, zq.w};
        const float a_s[4] = {qs.x, qs.y, qs.z, qs.w}, a_t[4] = {qt.x, qt.y, qt.z, qt.w};
        const float a_m[4] = {qm.x, qm.y, qm.z, qm.w}, a_r[4] = {qr.x, qr.y, qr.z, qr.w};
        float o1[4], o2[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float dd = (zz[e] * a_s[e] + a_t[e] > 0.f) ? vv[e] : 0.f;
            o1[e] = dd, o2[e] = dd * ((zz[e] - a_m[e]) * a_r[e]);
        }
        t1.x += o1[0], t1.y += o1[1], t1.z += o1[2], t1.w += o1[3];
        t2.x += o2[0], t2.y += o2[1], t2.z += o2[2], t2.w += o2[3];
        deg = degn, lst = lstn;
#pragma unroll
        for (int u = 0; u < 8; ++u) e0[u] = e0n[u];
    }
    *reinterpret_cast<float4*>(&red[slot][0][4 * q]) = t1;
    *reinterpret_cast<float4*>(&red[slot][1][4 * q]) = t2;
    __syncthreads();
    if (tid < 128) {
        const int k = tid >> 6, c = tid & 63;
        float t = 0.f;
#pragma unroll 16
        for (int sl = 0; sl < CH_GB_SLOTS; ++sl) t += red[sl][k][c];
        g.psums[((size_t)lb * 2 + k) * 64 + c] = t;
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Small companions.
//   chain_stats_kernel    moment partials [parts][3][64] of a (rows, 64) tensor as it stands (the first block's z0 = conv1's output,
//                         bias included: pooled with a null bias)
//   chain_sums_kernel     BatchNorm-backward sum partials [parts][2][64] of (dy, z): the chain's LAST layer, whose gradient comes
//                         from outside the chain (conv5's dx)
//   chain_bn_bwd_kernel   dz = gamma rstd (dy [z-mask] - dbeta / rows - zhat dgamma / rows) with the sums pooled in the prologue:
//                         the first block's leading BatchNorm (its layer, conv1 with K = 3, keeps its own small kernels)
//   chain_dw_sum_kernel   dW[l] = sum over the workgroup partials of layer l, ascending, every layer of the chain in ONE launch
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void chain_stats_kernel(const float* __restrict__ z, int rows, int wg_rows, float* __restrict__ stats) {
    __shared__ __attribute__((aligned(16))) float red[2][16][64];
    const int tid = threadIdx.x, q = tid & 15, rg = tid >> 4;
    const int r0 = blockIdx.x * wg_rows, r1 = min(rows, r0 + wg_rows);
    const float4 pv = *reinterpret_cast<const float4*>(z + (size_t)r0 * 64 + 4 * q);   // the pivot: the workgroup's first row
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
#pragma unroll 4
    for (int r = r0 + rg; r < r1; r += 16) {
        const float4 v = *reinterpret_cast<const float4*>(z + (size_t)r * 64 + 4 * q);
        const float d0 = v.x - pv.x, d1 = v.y - pv.y, d2 = v.z - pv.z, d3 = v.w - pv.w;
        a.x += d0, a.y += d1, a.z += d2, a.w += d3;
        b.x += d0 * d0, b.y += d1 * d1, b.z += d2 * d2, b.w += d3 * d3;
    }
    *reinterpret_cast<float4*>(&red[0][rg][4 * q]) = a;
    *reinterpret_cast<float4*>(&red[1][rg][4 * q]) = b;
    __syncthreads();
    if (tid < 192) {
        const int k = tid >> 6, c = tid & 63;
        float t;
        if (k < 2) {
            t = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) t += red[k][s][c];
        } else {
            t = z[(size_t)r0 * 64 + c];
        }
        stats[((size_t)blockIdx.x * 3 + k) * 64 + c] = t;
    }
}

// (Round 6: 1024 threads -- 64 row groups, 4 - 5 rows a thread, every load in flight at once -- instead of 256 threads walking 18 rows each
// in five dependent round trips on a CU that holds nothing else: chain_sums 14.7 -> 6.5 us.  chain_bn_bwd the same way measured 20.3 us
// against 14.8 -- it writes what it reads, and its pooling prologue wants few threads: kept at 256.)
#define CH_ROWG 64
__global__ __launch_bounds__(16 * CH_ROWG) void chain_sums_kernel(const float* __restrict__ dy, int dy_stride, const float* __restrict__ z,
                                                         ChBnGiven bn, float eps, int rows, int wg_rows, float* __restrict__ psums) {
    __shared__ __attribute__((aligned(16))) float pcoef[4][64];
    __shared__ __attribute__((aligned(16))) float red[2][CH_ROWG][64];
    const int tid = threadIdx.x, q = tid & 15, rg = tid >> 4;
    if (tid < 64) {
        const float pm = bn.mean[tid];
        const ChBnAffine pa = ch_bn_affine(pm, bn.var[tid], bn.gamma[tid], bn.beta[tid], eps);
        pcoef[0][tid] = pa.s, pcoef[1][tid] = pa.t, pcoef[2][tid] = pm, pcoef[3][tid] = 1.0f / sqrtf(bn.var[tid] + eps);
    }
    __syncthreads();
    const float4 qs = *reinterpret_cast<const float4*>(&pcoef[0][4 * q]), qt = *reinterpret_cast<const float4*>(&pcoef[1][4 * q]);
    const float4 qm = *reinterpret_cast<const float4*>(&pcoef[2][4 * q]), qr = *reinterpret_cast<const float4*>(&pcoef[3][4 * q]);
    const int r0 = blockIdx.x * wg_rows, r1 = min(rows, r0 + wg_rows);
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
#pragma unroll 8
    for (int r = r0 + rg; r < r1; r += CH_ROWG) {
        const float4 zv = *reinterpret_cast<const float4*>(z + (size_t)r * 64 + 4 * q);
        const float4 gv = *reinterpret_cast<const float4*>(dy + (size_t)r * dy_stride + 4 * q);
        const float d0 = (zv.x * qs.x + qt.x > 0.f) ? gv.x : 0.f, d1 = (zv.y * qs.y + qt.y > 0.f) ? gv.y : 0.f;
        const float d2 = (zv.z * qs.z + qt.z > 0.f) ? gv.z : 0.f, d3 = (zv.w * qs.w + qt.w > 0.f) ? gv.w : 0.f;
        a.x += d0, a.y += d1, a.z += d2, a.w += d3;
        b.x += d0 * ((zv.x - qm.x) * qr.x), b.y += d1 * ((zv.y - qm.y) * qr.y), b.z += d2 * ((zv.z - qm.z) * qr.z),
            b.w += d3 * ((zv.w - qm.w) * qr.w);
    }
    *reinterpret_cast<float4*>(&red[0][rg][4 * q]) = a;
    *reinterpret_cast<float4*>(&red[1][rg][4 * q]) = b;
    __syncthreads();
    if (tid < 128) {
        const int k = tid >> 6, c = tid & 63;
        float t = 0.f;
#pragma unroll
        for (int s = 0; s < CH_ROWG; ++s) t += red[k][s][c];
        psums[((size_t)blockIdx.x * 2 + k) * 64 + c] = t;
    }
}

__global__ __launch_bounds__(256) void chain_bn_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ z, ChBnGiven bn,
                                                           const float* __restrict__ sums, int parts, float* dgamma, float* dbeta,
                                                           float eps, int rows, int wg_rows, float* __restrict__ dz) {
    __shared__ __attribute__((aligned(16))) double scratch[2 * 16 * 64];
    __shared__ __attribute__((aligned(16))) float s_sum[2][64];
    __shared__ __attribute__((aligned(16))) float coef[6][64];
    const int tid = threadIdx.x, q = tid & 15, rg = tid >> 4;
    ch_pool_sums(sums, parts, scratch, s_sum, dbeta, dgamma);
    const float inv_rows = 1.0f / (float)rows;
    if (tid < 64) {
        const float mu = bn.mean[tid], rs = 1.0f / sqrtf(bn.var[tid] + eps), ga = bn.gamma[tid];
        const ChBnAffine a = ch_bn_affine(mu, bn.var[tid], ga, bn.beta[tid], eps);
        coef[0][tid] = a.s, coef[1][tid] = a.t, coef[2][tid] = mu;
        coef[3][tid] = ga * rs, coef[4][tid] = s_sum[0][tid] * inv_rows, coef[5][tid] = rs * (s_sum[1][tid] * inv_rows);
    }
    __syncthreads();
    float cs[4], ct[4], mu[4], k1[4], bb[4], gg[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        cs[e] = coef[0][4 * q + e], ct[e] = coef[1][4 * q + e], mu[e] = coef[2][4 * q + e];
        k1[e] = coef[3][4 * q + e], bb[e] = coef[4][4 * q + e], gg[e] = coef[5][4 * q + e];
    }
    const int r0 = blockIdx.x * wg_rows, r1 = min(rows, r0 + wg_rows);
#pragma unroll 4
    for (int r = r0 + rg; r < r1; r += 16) {
        const size_t o = (size_t)r * 64 + 4 * q;
        const float4 zv = *reinterpret_cast<const float4*>(z + o), gv = *reinterpret_cast<const float4*>(dy + o);
        const float zi[4] = {zv.x, zv.y, zv.z, zv.w}, gi[4] = {gv.x, gv.y, gv.z, gv.w};
        float out[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float dd = !(zi[e] * cs[e] + ct[e] > 0.f) ? 0.f : gi[e];
            out[e] = k1[e] * (dd - bb[e] - (zi[e] - mu[e]) * gg[e]);
        }
        *reinterpret_cast<float4*>(dz + o) = make_float4(out[0], out[1], out[2], out[3]);
    }
}

#define CH_MAX_LAYERS 16
struct ChDwSumArgs {
    const float* part[CH_MAX_LAYERS];
    float* out[CH_MAX_LAYERS];
    int parts;
};
__global__ __launch_bounds__(256) void chain_dw_sum_kernel(ChDwSumArgs g) {
    __shared__ float4 acc[16][16];
    const int layer = blockIdx.y;
    const int col = threadIdx.x & 15, grp = threadIdx.x >> 4;
    const int e4 = blockIdx.x * 16 + col;   // 1024 float4 per layer: 64 workgroups
    const float4* src = reinterpret_cast<const float4*>(g.part[layer]) + e4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    int p = grp;
    for (; p + 48 < g.parts; p += 64) {
        const float4 a = src[(size_t)p * 1024], b = src[(size_t)(p + 16) * 1024], c = src[(size_t)(p + 32) * 1024],
                     d = src[(size_t)(p + 48) * 1024];
        s.x += a.x, s.y += a.y, s.z += a.z, s.w += a.w;
        s.x += b.x, s.y += b.y, s.z += b.z, s.w += b.w;
        s.x += c.x, s.y += c.y, s.z += c.z, s.w += c.w;
        s.x += d.x, s.y += d.y, s.z += d.z, s.w += d.w;
    }
    for (; p < g.parts; p += 16) {
        const float4 a = src[(size_t)p * 1024];
        s.x += a.x, s.y += a.y, s.z += a.z, s.w += a.w;
    }
    acc[grp][col] = s;
    __syncthreads();
    if (grp == 0) {
        float4 t = acc[0][col];
#pragma unroll
        for (int gq = 1; gq < 16; ++gq) t.x += acc[gq][col].x, t.y += acc[gq][col].y, t.z += acc[gq][col].z, t.w += acc[gq][col].w;
        reinterpret_cast<float4*>(g.out[layer])[e4] = t;
    }
}

// the points of each cloud whose neighbour list overflowed (cnt > cap), ascending: one workgroup per cloud (1024 threads: four rounds of
// three barriers for 4096 points instead of sixteen)
#define OVF_THREADS 1024
__global__ __launch_bounds__(OVF_THREADS) void knn_overflow_list_kernel(const int32_t* __restrict__ cnt, int cap, int n,
                                                                        int32_t* __restrict__ ovf_cnt, int32_t* __restrict__ ovf_list) {
    __shared__ int wtot[OVF_THREADS / 64];
    __shared__ int carry;
    const int cloud = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int j0 = 0; j0 < n; j0 += OVF_THREADS) {
        const int j = j0 + tid;
        const bool on = j < n && cnt[(size_t)cloud * n + j] > cap;
        const unsigned long long bal = __ballot(on);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wtot[wave] = __popcll(bal);
        __syncthreads();
        int off = carry;
        for (int w = 0; w < wave; ++w) off += wtot[w];
        if (on) ovf_list[(size_t)cloud * n + off + before] = j;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int w = 0; w < OVF_THREADS / 64; ++w) t += wtot[w];
            carry += t;
        }
        __syncthreads();
    }
    if (tid == 0) ovf_cnt[cloud] = carry;
}

// ---- C ABI ------------------------------------------------------------------------------------------------------
static bool ch_aligned16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }

// rows per workgroup (= per partial): the 32-row tiles spread evenly over the CUs, at most CH_MAX_TILES per workgroup
static int ch_wg_rows(int rows) {
    const int tiles = (rows + 31) / 32, cus = epc_device_cu_count();
    int t = (tiles + cus - 1) / cus;
    t = t < 1 ? 1 : (t > CH_MAX_TILES ? CH_MAX_TILES : t);
    return 32 * t;
}
extern "C" int epc_chain_parts(int rows) {
    if (rows <= 0) return 0;
    const int r = ch_wg_rows(rows);
    return (rows + r - 1) / r;
}
static int ch_waves(int rows, int cap) {
    const int t = ch_wg_rows(rows) / 32;
    return t < cap ? t : cap;
}
static int ch_set_lds(const void* fn, size_t bytes, const char* who) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) != hipSuccess) {
        epc_set_error("%s: hipFuncSetAttribute failed", who);
        return EPC_EHIP;
    }
    return EPC_OK;
}

static ChBn make_bn(const float* stats, int rows, const float* bias, float* mean, float* var, const float* gamma, const float* beta) {
    ChBn b;
    b.stats = stats, b.parts = epc_chain_parts(rows), b.part_rows = ch_wg_rows(rows), b.bias = bias, b.mean = mean, b.var = var;
    b.gamma = gamma, b.beta = beta;
    return b;
}

extern "C" int epc_chain_fwd_linear(const float* zin, const float* in_stats, const float* in_bias, float* in_mean, float* in_var,
                                    const float* in_gamma, const float* in_beta, float eps, const float* resid, float* a_out,
                                    int a_stride, void* a_out_bf16, const float* W, const float* bias, float* z_out, float* stats_out,
                                    int rows, int pieces, void* stream) {
    EPC_CHECK_ARG(zin && in_mean && in_var && in_gamma && in_beta, "null pointer");
    EPC_CHECK_ARG(!a_out_bf16 || (a_out && a_stride % 8 == 0 && ch_aligned16(a_out_bf16)), "the bf16 copy comes with a_out, 16-byte aligned, stride a multiple of 8");
    EPC_CHECK_ARG(rows > 0 && (pieces == 3 || pieces == 1), "bad shape (pieces: 3 or 1)");
    EPC_CHECK_ARG(!W || (z_out && stats_out), "a layer needs z_out and stats_out");
    EPC_CHECK_ARG(W || a_out, "nothing to do: neither a layer nor an activation output");
    EPC_CHECK_ARG(ch_aligned16(zin) && ch_aligned16(resid) && ch_aligned16(a_out) && ch_aligned16(z_out) && ch_aligned16(in_stats) &&
                      (!a_out || a_stride % 4 == 0),
                  "tensors must be 16-byte aligned, strides multiples of 4");
    ChFwdLinearArgs g;
    g.zin = zin, g.bn = make_bn(in_stats, rows, in_bias, in_mean, in_var, in_gamma, in_beta);
    g.resid = resid, g.a_out = a_out, g.a_stride = a_stride, g.a_out16 = (unsigned short*)a_out_bf16, g.W = W, g.bias = bias, g.z_out = z_out, g.stats_out = stats_out;
    g.rows = rows, g.wg_rows = ch_wg_rows(rows), g.eps = eps;
    const dim3 grid(epc_chain_parts(rows)), block(64 * ch_waves(rows, CH_FWD_MAX_WAVES));
    const size_t lds = 4 * 16 * 64 * sizeof(double);
    if (pieces == 3) hipLaunchKernelGGL(chain_fwd_linear_kernel<3>, grid, block, lds, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(chain_fwd_linear_kernel<1>, grid, block, lds, (hipStream_t)stream, g);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_chain_fwd_gather(const float* z0, const float* in_stats, const float* in_bias, float* in_mean, float* in_var,
                                    const float* in_gamma, const float* in_beta, float eps, const float* xyz, const int32_t* idx,
                                    const int32_t* cnt, const float* kth, int cap, int num_clouds, int n, int knn, const float* W,
                                    const float* bias, float* xm, float* d, float* z_out, float* stats_out, int pieces, void* stream) {
    EPC_CHECK_ARG(z0 && in_mean && in_var && in_gamma && in_beta && xyz && idx && cnt && kth && W && xm && d && z_out && stats_out,
                  "null pointer");
    EPC_CHECK_ARG(num_clouds > 0 && n > 0 && knn > 0 && cap >= EPC_KNN_SELECT && (pieces == 3 || pieces == 1), "bad shape");
    EPC_CHECK_ARG((long)num_clouds * n < (1L << 31) / 64, "too many rows");
    EPC_CHECK_ARG(ch_aligned16(z0) && ch_aligned16(xm) && ch_aligned16(d) && ch_aligned16(z_out) && ch_aligned16(in_stats) && ch_aligned16(idx),
                  "tensors must be 16-byte aligned");
    const int rows = num_clouds * n;
    ChFwdGatherArgs g;
    g.z0 = z0, g.bn = make_bn(in_stats, rows, in_bias, in_mean, in_var, in_gamma, in_beta);
    g.xyz = xyz, g.idx = idx, g.cnt = cnt, g.kth = kth, g.cap = cap, g.n = n, g.kdiv = (float)knn, g.W = W, g.bias = bias;
    g.xm = xm, g.d = d, g.z_out = z_out, g.stats_out = stats_out, g.rows = rows, g.wg_rows = ch_wg_rows(rows), g.eps = eps;
    const int nw = ch_waves(rows, CH_GATHER_MAX_WAVES);
    const dim3 grid(epc_chain_parts(rows)), block(64 * nw);
    size_t lds = (size_t)nw * CH_STG_FLOATS * sizeof(float);
    if (lds < 4 * 16 * 64 * sizeof(double)) lds = 4 * 16 * 64 * sizeof(double);
    const void* fn = pieces == 3 ? reinterpret_cast<const void*>(chain_fwd_gather_kernel<3>) : reinterpret_cast<const void*>(chain_fwd_gather_kernel<1>);
    if (int rc = ch_set_lds(fn, lds, __func__)) return rc;
    if (pieces == 3) hipLaunchKernelGGL(chain_fwd_gather_kernel<3>, grid, block, lds, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(chain_fwd_gather_kernel<1>, grid, block, lds, (hipStream_t)stream, g);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

static ChBnGiven make_given(const float* mean, const float* var, const float* gamma, const float* beta) {
    ChBnGiven b;
    b.mean = mean, b.var = var, b.gamma = gamma, b.beta = beta;
    return b;
}

extern "C" int epc_chain_bwd_linear(const float* dy, int dy_stride, const float* z, const float* mean, const float* var,
                                    const float* gamma, const float* beta, float eps, const float* sums, float* dgamma, float* dbeta,
                                    const float* W, const float* x, int x_stride, const float* x_mean, const float* x_var,
                                    const float* x_gamma, const float* x_beta, float* dx, const float* dx_addend, int addend_stride,
                                    float* dw_partials, const float* zp, const float* p_mean, const float* p_var, const float* p_gamma,
                                    const float* p_beta, float* psums, int rows, int pieces, void* stream) {
    EPC_CHECK_ARG(dy && z && mean && var && gamma && beta && sums && dgamma && dbeta && W && x && dw_partials, "null pointer");
    EPC_CHECK_ARG(rows > 0 && (pieces == 2 || pieces == 1), "bad shape (pieces: 2 or 1)");
    EPC_CHECK_ARG(dx || (!dx_addend && !zp), "dx_addend / zp without dx");
    EPC_CHECK_ARG(!zp || (p_mean && p_var && p_gamma && p_beta && psums), "zp needs its BatchNorm and psums");
    const bool any = x_mean || x_var || x_gamma || x_beta, all = x_mean && x_var && x_gamma && x_beta;
    EPC_CHECK_ARG(any == all, "x_mean, x_var, x_gamma, x_beta: all four or none");
    EPC_CHECK_ARG(ch_aligned16(dy) && ch_aligned16(z) && ch_aligned16(x) && ch_aligned16(dx) && ch_aligned16(dx_addend) && ch_aligned16(zp) &&
                      ch_aligned16(sums) && ch_aligned16(W) && dy_stride % 4 == 0 && x_stride % 4 == 0 && (!dx_addend || addend_stride % 4 == 0),
                  "tensors must be 16-byte aligned, strides multiples of 4");
    ChBwdLinearArgs g;
    g.dy = dy, g.dy_stride = dy_stride, g.z = z, g.bn = make_given(mean, var, gamma, beta), g.sums = sums, g.parts = epc_chain_parts(rows);
    g.dgamma = dgamma, g.dbeta = dbeta, g.W = W, g.x = x, g.x_stride = x_stride, g.xbn = make_given(x_mean, x_var, x_gamma, x_beta);
    g.dx = dx, g.dx_addend = dx_addend, g.addend_stride = addend_stride, g.dWpart = dw_partials;
    g.zp = zp, g.pbn = make_given(p_mean, p_var, p_gamma, p_beta), g.psums = psums, g.rows = rows, g.wg_rows = ch_wg_rows(rows), g.eps = eps;
    const int nw = ch_waves(rows, CH_BWD_MAX_WAVES);
    const dim3 grid(epc_chain_parts(rows)), block(64 * nw);
    size_t lds = (size_t)nw * 2 * CH_IMG_BYTES;
    if (lds < 2 * 16 * 64 * sizeof(double)) lds = 2 * 16 * 64 * sizeof(double);
    const void* fn = pieces == 2 ? reinterpret_cast<const void*>(chain_bwd_linear_kernel<2>) : reinterpret_cast<const void*>(chain_bwd_linear_kernel<1>);
    if (int rc = ch_set_lds(fn, lds, __func__)) return rc;
    if (pieces == 2) hipLaunchKernelGGL(chain_bwd_linear_kernel<2>, grid, block, lds, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(chain_bwd_linear_kernel<1>, grid, block, lds, (hipStream_t)stream, g);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_chain_bwd_gather(const float* s, const float* dout, int dout_stride, const int32_t* rdeg, const int32_t* roff, const int32_t* rlist,
                                    const int32_t* ovf_cnt, const int32_t* ovf_list, const float* xyz, const float* kth,
                                    int num_clouds, int n, int knn, const float* z0, const float* mean, const float* var,
                                    const float* gamma, const float* beta, float eps, float* psums, float* dx, void* stream) {
    EPC_CHECK_ARG(s && dout && rdeg && roff && rlist && ovf_cnt && ovf_list && xyz && kth && z0 && mean && var && gamma && beta && psums && dx,
                  "null pointer");
    EPC_CHECK_ARG(num_clouds > 0 && n > 0 && knn > 0 && (long)num_clouds * n < (1L << 31) / 64, "bad shape");
    EPC_CHECK_ARG(ch_aligned16(s) && ch_aligned16(dout) && ch_aligned16(z0) && ch_aligned16(dx) && dout_stride % 4 == 0,
                  "tensors must be 16-byte aligned, strides multiples of 4");
    ChBwdGatherArgs g;
    g.s = s, g.dout = dout, g.dout_stride = dout_stride, g.rdeg = rdeg, g.roff = roff, g.rlist = rlist, g.ovf_cnt = ovf_cnt, g.ovf_list = ovf_list, g.xyz = xyz;
    g.kth = kth, g.n = n, g.total = num_clouds * n, g.wg_rows = ch_wg_rows(g.total), g.kdiv = (float)knn, g.z0 = z0;
    g.bn = make_given(mean, var, gamma, beta), g.psums = psums, g.dx = dx, g.eps = eps;
    hipLaunchKernelGGL(chain_bwd_gather_kernel, dim3(epc_chain_parts(g.total)), dim3(CH_GB_THREADS), 0, (hipStream_t)stream, g);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_chain_stats(const float* z, int rows, float* stats, void* stream) {
    EPC_CHECK_ARG(z && stats && rows > 0 && ch_aligned16(z), "null pointer / bad shape / alignment");
    hipLaunchKernelGGL(chain_stats_kernel, dim3(epc_chain_parts(rows)), dim3(256), 0, (hipStream_t)stream, z, rows, ch_wg_rows(rows), stats);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_chain_sums(const float* dy, int dy_stride, const float* z, const float* mean, const float* var, const float* gamma,
                              const float* beta, float eps, int rows, float* psums, void* stream) {
    EPC_CHECK_ARG(dy && z && mean && var && gamma && beta && psums && rows > 0, "null pointer / bad shape");
    EPC_CHECK_ARG(ch_aligned16(dy) && ch_aligned16(z) && dy_stride % 4 == 0, "tensors must be 16-byte aligned, strides multiples of 4");
    hipLaunchKernelGGL(chain_sums_kernel, dim3(epc_chain_parts(rows)), dim3(16 * CH_ROWG), 0, (hipStream_t)stream, dy, dy_stride, z,
                       make_given(mean, var, gamma, beta), eps, rows, ch_wg_rows(rows), psums);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_chain_bn_bwd(const float* dy, const float* z, const float* mean, const float* var, const float* gamma,
                                const float* beta, float eps, const float* sums, float* dgamma, float* dbeta, int rows, float* dz,
                                void* stream) {
    EPC_CHECK_ARG(dy && z && mean && var && gamma && beta && sums && dgamma && dbeta && dz && rows > 0, "null pointer / bad shape");
    EPC_CHECK_ARG(ch_aligned16(dy) && ch_aligned16(z) && ch_aligned16(dz) && ch_aligned16(sums), "tensors must be 16-byte aligned");
    hipLaunchKernelGGL(chain_bn_bwd_kernel, dim3(epc_chain_parts(rows)), dim3(256), 0, (hipStream_t)stream, dy, z,
                       make_given(mean, var, gamma, beta), sums, epc_chain_parts(rows), dgamma, dbeta, eps, rows, ch_wg_rows(rows), dz);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_chain_dw_sum(int layers, const float* const* partials, float* const* dW, int rows, void* stream) {
    EPC_CHECK_ARG(layers > 0 && layers <= CH_MAX_LAYERS && partials && dW && rows > 0, "bad argument (at most 16 layers per call)");
    ChDwSumArgs g;
    for (int l = 0; l < layers; ++l) {
        EPC_CHECK_ARG(partials[l] && dW[l] && ch_aligned16(partials[l]) && ch_aligned16(dW[l]), "null / unaligned tensor");
        g.part[l] = partials[l], g.out[l] = dW[l];
    }
    g.parts = epc_chain_parts(rows);
    hipLaunchKernelGGL(chain_dw_sum_kernel, dim3(64, layers), dim3(256), 0, (hipStream_t)stream, g);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_knn_overflow_lists(const int32_t* cnt, int cap, int num_clouds, int n, int32_t* ovf_cnt, int32_t* ovf_list,
                                      void* stream) {
    EPC_CHECK_ARG(cnt && ovf_cnt && ovf_list && num_clouds > 0 && n > 0 && cap >= EPC_KNN_SELECT, "null pointer / bad shape");
    hipLaunchKernelGGL(knn_overflow_list_kernel, dim3(num_clouds), dim3(OVF_THREADS), 0, (hipStream_t)stream, cnt, cap, n, ovf_cnt, ovf_list);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}
