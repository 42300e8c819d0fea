// ProxyConv block (models/epc-net.py:66-83 and its three repeats) as ONE kernel per block:
//
//   xm  = (sum_{j in nbr(i)} x_j) / k         gather over the kNN index lists (the reference multiplies a dense
//                                             (N,N) 0/1 mask: 2.15 GFLOP per cloud and block; here 5.2 M adds)
//   t   = xm - x
//   t   = relu(bn(conv_a(t)))                 64x64 on the half-precision MFMA, BN folded into the weights
//   t   = relu(bn(conv_b(t)))                 B operand = the previous accumulators (no LDS round trip)
//   out = t + xm                  -> concat buffer slice (models/epc-net.py:134)
//   x'  = relu(bn(conv_{b+1}(out)))           the next block's leading conv, fused (grid-wide dependency sits
//                                             only at the gather, so one launch per block is the minimum)
//
// Two kernels: proxyconv_block_f16_kernel (EPC-Net: fp16 rows in HBM, one fp16 value per activation
// against fp16 hi+lo weights) and proxyconv_block_kernel (EPC-Net-L: f32 rows, split-bf16 x3 layers, f32-accurate at
// every stage boundary).  Each wave owns 32 consecutive points (one MFMA column tile: point = lane&31).
// Transposed orientation out^T[ch][pt] = W^T x^T so that a layer's accumulators (channel in the register index,
// point on the lane) are directly the next layer's B operands.  The gather runs whole rows (f32: 16 lanes x float4,
// 4 points per wave-instruction; fp16: 8 lanes x 16 B, 8 points), results go through a per-wave LDS staging tile to
// switch between the row layout and the MFMA layout.
#include "common.h"

#ifndef BLK_WAVES
#define BLK_WAVES 12  // f32 kernel: 12 x 8.7 KB staging tiles + the 49-KB weight pack = 154 KB (3 waves per SIMD)
#endif
#define BLK_THREADS (64 * BLK_WAVES)
#ifndef BLK_PERSIST_MAX_ROUNDS
#define BLK_PERSIST_MAX_ROUNDS 4096   // grids of up to this many rounds of short workgroups run as one persistent workgroup per CU (i.e. always)
#endif
#define ST_STRIDE 68  // floats per staged row: 64 + 4 pad -> conflict-free b128 reads in both layouts
#define BLK_PACK EPC_BLOCK_PACK_FLOATS
#define BLK_PACK_S EPC_BLOCK_PACK_FLOATS_S   // f32 kernel: the three layers' inverse column scales follow the layer packs
#define BLK_LDS_FLOATS (BLK_PACK_S + BLK_WAVES * 32 * ST_STRIDE)


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

// compensated summation step: sum += v with the rounding error carried in comp (no fast-math: not re-associated)
__device__ __forceinline__ void kahan_add(float& sum, float& comp, float v) {
    const float y = v - comp;
    const float t = sum + y;
    comp = (t - sum) - y;
    sum = t;
}

__device__ __forceinline__ void relu16(f32x16& a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.f);
}


// The three 64x64 layers of the f32 kernel run on the fp16 MFMA in SCALED split-fp16 arithmetic (common.h: every point's
// 64-channel row and every output channel's weight column scaled by a power of two into [2^14, 2^15), hi + lo fp16 parts,
// three products): 24 MFMAs of 32 cycles per layer instead of 64 f32 MFMAs of 64 cycles, 2^-21 per product.  Weight
// fragments: pack.hip fold_pack_block_kernel (f16 = 0); `ltinv` = the layer's 64 inverse column scales.
__device__ __forceinline__ f16x8 ldfrag_h(const float* p) {
    return __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(p));
}
// the three products of one layer on ready B fragments, then out = acc * (inv_row * inv_col[ch]) + bias[ch]
__device__ __forceinline__ void layer_s64(const float* lw, const float* lbias, const float* ltinv, const f16x8 (&bh)[4],
                                          const f16x8 (&bl)[4], float inv_row, f32x16 (&acc)[2], int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f16x8 ah = ldfrag_h(lw + (((t * 4 + s) * 2 + 0) * 64 + lane) * 4);
            const f16x8 al = ldfrag_h(lw + (((t * 4 + s) * 2 + 1) * 64 + lane) * 4);
            acc[t] = mfma_f16(al, bh[s], acc[t]);
            acc[t] = mfma_f16(ah, bl[s], acc[t]);
            acc[t] = mfma_f16(ah, bh[s], acc[t]);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 ti = ld4(ltinv + 32 * t + 8 * g + 4 * h), b = ld4(lbias + 32 * t + 8 * g + 4 * h);
            acc[t][4 * g + 0] = __builtin_fmaf(acc[t][4 * g + 0], inv_row * ti.x, b.x);
            acc[t][4 * g + 1] = __builtin_fmaf(acc[t][4 * g + 1], inv_row * ti.y, b.y);
            acc[t][4 * g + 2] = __builtin_fmaf(acc[t][4 * g + 2], inv_row * ti.z, b.z);
            acc[t][4 * g + 3] = __builtin_fmaf(acc[t][4 * g + 3], inv_row * ti.w, b.w);
        }
    }
}

// 64->64 layer whose B operand is the previous layer's accumulators: k-step (tin, s') = registers 8s'..8s'+7 of tile tin.
// The lane holds 32 of its point's 64 channels, lane ^ 32 the others: the row maximum is one exchange.
__device__ __forceinline__ void layer_acc64(const float* lw, const float* lbias, const float* ltinv, const f32x16 (&in)[2],
                                            f32x16 (&acc)[2], int lane) {
    float m = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) m = fmaxf(m, fabsf(in[t][r]));
    m = fmaxf(m, __shfl_xor(m, 32));
    float sc, inv_row;
    row_scale_pow2(m, sc, inv_row);
    f16x8 fh[4], fl[4];
#pragma unroll
    for (int st = 0; st < 4; ++st) {
        float v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) v[q] = in[st >> 1][8 * (st & 1) + q];
        split8_f16s(v, sc, fh[st], fl[st]);
    }
    layer_s64(lw, lbias, ltinv, fh, fl, inv_row, acc, lane);
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

// staged [pt][64] row of the lane's point -> scaled split B fragments (k-step s = channels 16s + 8h .. +7); returns the
// row's inverse scale
__device__ __forceinline__ float stage_to_bop(const float* st, f16x8 (&bh)[4], f16x8 (&bl)[4], int lane) {
    const int j = lane & 31, h = lane >> 5;
    float v[4][8];
    float m = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const float4 a = ld4(st + j * ST_STRIDE + 16 * s + 8 * h), b = ld4(st + j * ST_STRIDE + 16 * s + 8 * h + 4);
        v[s][0] = a.x, v[s][1] = a.y, v[s][2] = a.z, v[s][3] = a.w, v[s][4] = b.x, v[s][5] = b.y, v[s][6] = b.z, v[s][7] = b.w;
#pragma unroll
        for (int q = 0; q < 8; ++q) m = fmaxf(m, fabsf(v[s][q]));
    }
    m = fmaxf(m, __shfl_xor(m, 32));   // the other 32 channels of the point
    float sc, inv_row;
    row_scale_pow2(m, sc, inv_row);
#pragma unroll
    for (int s = 0; s < 4; ++s) split8_f16s(v[s], sc, bh[s], bl[s]);
    return inv_row;
}

// ---- shared pieces of the two block kernels ------------------------------------------------------------------
// A point's first 20 neighbour indices from its list row of `cap` slots: 4-byte entries (epc_knn_topk) or 2-byte entries
// (the fused pipeline, epc_knn_topk_conv1 with idx_u16: 64-B rows, half the index traffic).  `u16` is wave-uniform.
__device__ __forceinline__ void load_nb20(const char* lists, unsigned row, int cap, bool u16, int (&nb)[EPC_KNN_SELECT]) {
    static_assert(EPC_KNN_SELECT == 20, "five int4 / two int4 + one int2 per row");
    if (u16) {
        const char* r = lists + (size_t)row * cap * 2;
        const uint4 a = *reinterpret_cast<const uint4*>(r), b = *reinterpret_cast<const uint4*>(r + 16);
        const uint2 c = *reinterpret_cast<const uint2*>(r + 32);
        const unsigned int w[10] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y};
#pragma unroll
        for (int k = 0; k < 10; ++k) {
            nb[2 * k] = (int)(w[k] & 0xffffu);
            nb[2 * k + 1] = (int)(w[k] >> 16);
        }
    } else {
        const int4* il = reinterpret_cast<const int4*>(lists + (size_t)row * cap * 4);
#pragma unroll
        for (int m4 = 0; m4 < EPC_KNN_SELECT / 4; ++m4) {
            const int4 t = il[m4];
            nb[4 * m4] = t.x;
            nb[4 * m4 + 1] = t.y;
            nb[4 * m4 + 2] = t.z;
            nb[4 * m4 + 3] = t.w;
        }
    }
}
__device__ __forceinline__ int list_entry(const void* lists, size_t slot, bool u16) {
    return u16 ? (int)reinterpret_cast<const unsigned short*>(lists)[slot] : reinterpret_cast<const int32_t*>(lists)[slot];
}

// ---- f32 rows, scaled split-fp16 layers (EPC_PRECISION_F32 and EPC-Net-L; f32-equivalent at every stage boundary) ----
__global__ __launch_bounds__(BLK_THREADS) void proxyconv_block_kernel(
    const float* __restrict__ x, const float* __restrict__ xyz, const void* __restrict__ idx, int idx_u16,
    const int32_t* __restrict__ cnt, const float* __restrict__ kth, int cap, const float* __restrict__ pack,
    int has_next, int total_points, int n, float kdiv, float* __restrict__ out, int out_stride, int out_off,
    float* __restrict__ x_next) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    __shared__ int s_next_tile;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_next_tile = 0;
    for (int o = tid * 4; o < BLK_PACK_S; o += BLK_THREADS * 4) st4(lds + o, ld4(pack + o));
    __syncthreads();
    const float* wa = lds;
    const float* ba = lds + 4096;
    const float* wb = lds + 4160;
    const float* bb = lds + 4160 + 4096;
    const float* wn = lds + 8320;
    const float* bn = lds + 8320 + 4096;
    const float* tia = lds + BLK_PACK;        // inverse column scales of conv_a, conv_b, conv_next
    const float* tib = tia + 64;
    const float* tin = tia + 128;
    float* st = lds + BLK_PACK_S + wave * 32 * ST_STRIDE;

    // A workgroup owns a contiguous range of 32-point tiles -- the ranges differ by at most one tile -- and its waves draw
    // tiles from it one at a time (an LDS counter) until it is empty.  The launch sizes the grid (epc_proxyconv_block_fwd):
    // one workgroup per 12 tiles (every wave one tile), or -- for grids of a few rounds -- ONE persistent workgroup per CU, so
    // that every CU gets the same work and stages the 49-KB weight pack once.  Which wave computes a tile does not enter
    // its arithmetic.
    // The workgroups of one XCD (blockIdx % 8: they share an L2) sweep their XCD's range TOGETHER, four tiles per workgroup
    // and step -- at any time the XCD works inside a window of a few hundred tiles (a cloud or two of gathered rows: the 4-MB
    // L2 holds them), as the short workgroups did.  (A private contiguous quarter-cloud per workgroup put eight clouds behind
    // one L2 at a time: FETCH_SIZE doubled.)
    const int ntiles = total_points / 32;
    const int nwg = (int)gridDim.x, per = ntiles / nwg, rem = ntiles % nwg;
    auto first_tile_of = [&](int wg) { return wg * per + min(wg, rem); };   // balanced contiguous partition, in workgroup order
    const int xq = nwg >> 3, xr = nwg & 7, xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int xcd_wgs = xq + (xcd < xr ? 1 : 0);
    const int xcd_first = xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq;     // (xcd_contiguous_block's order)
    const int t_begin = first_tile_of(xcd_first), t_end = first_tile_of(xcd_first + xcd_wgs);
    const int p = lane >> 4, q = lane & 15;
    const bool u16 = idx_u16 != 0;
    const float rk = 1.0f / kdiv;
    // a / kdiv, correctly rounded for ordinary operands: quotient estimate + one exact-remainder correction
    auto div_k = [&](float a) {
        const float q0 = a * rk;
        return __builtin_fmaf(__builtin_fmaf(-q0, kdiv, a), rk, q0);
    };
  for (;;) {   // (no workgroup barriers inside)
    int t_draw = 0;
    if (lane == 0) t_draw = atomicAdd(&s_next_tile, 1);
    const int d = __builtin_amdgcn_readfirstlane(t_draw);
    const int tile = t_begin + (d >> 2) * (xcd_wgs * 4) + slot * 4 + (d & 3);
    if (tile >= t_end) break;
    const int g0 = tile * 32;
    // a wave's 32 points lie in one cloud: wave-uniform bases + 32-bit lane offsets (saddr addressing, no 64-bit VALU math)
    const int cloud_base = __builtin_amdgcn_readfirstlane((g0 / n) * n);
    const char* xc = reinterpret_cast<const char*>(x) + (size_t)cloud_base * 256;  // 256-B rows
    auto row32 = [&](int row) { return *reinterpret_cast<const float4*>(xc + (unsigned)(row * 256 + q * 16)); };
    const int wg0 = __builtin_amdgcn_readfirstlane(g0);

    // ---- gather-mean, 4 points per pass ----
    float4 xm[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int g = g0 + 4 * s + p;
#ifdef BLK_ABL_NOIDX   // timing only
        const int c = 20;
#else
        const int c = cnt[g];
#endif
        const bool ovf = c > cap;
        // every row holds >= 20 valid entries (cnt >= 20 by construction); ties beyond 20 are the rare tail
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (!ovf) {
            int nb[EPC_KNN_SELECT];
#ifdef BLK_ABL_NOIDX
#pragma unroll
            for (int m = 0; m < EPC_KNN_SELECT; ++m) nb[m] = 0;
#else
            load_nb20(reinterpret_cast<const char*>(idx), (unsigned)(wg0 + 4 * s + p), cap, u16, nb);
#endif
            float4 v[EPC_KNN_SELECT];
#pragma unroll
#ifdef BLK_ABL_LDSGATHER   // timing only: every neighbour row read from LDS (the weight pack's bytes stand in for a row cache)
            for (int m = 0; m < EPC_KNN_SELECT; ++m) v[m] = ld4(lds + (((unsigned)nb[m] % 180u) * 64 + q * 4));
#elif defined(BLK_ABL_NOGATHER)   // timing only: every neighbour row = the point's own row (L1-resident after the first)
            for (int m = 0; m < EPC_KNN_SELECT; ++m) v[m] = row32(g - cloud_base + 0 * nb[m]);
#else
            for (int m = 0; m < EPC_KNN_SELECT; ++m) v[m] = row32(nb[m]);
#endif
#pragma unroll
            for (int m = 0; m < EPC_KNN_SELECT; ++m) {  // ascending j, one rounding per add
                acc.x += v[m].x;
                acc.y += v[m].y;
                acc.z += v[m].z;
                acc.w += v[m].w;
            }
            for (int m = EPC_KNN_SELECT; m < c; ++m) {
                const float4 w = row32(list_entry(idx, (size_t)g * cap + m, u16));
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
                // hundreds to thousands of rows (clumps of identical points, zero padding): compensated (Kahan) summation, so
                // that the sum is as good as the reference's blocked matmul(mask, x) -- a plain running f32 sum of 3072 rows
                // is 50x noisier, and on ill-conditioned weights the blocks amplify that to 3e-4 of the descriptor
                float4 comp = make_float4(0.f, 0.f, 0.f, 0.f);
                for (int j = 0; j < n; ++j) {
                    const float xj = pc[3 * j], yj = pc[3 * j + 1], zj = pc[3 * j + 2];
                    const float a = neg_sq_dist(sqi, xi, yi, zi, xj, yj, zj, sq3(xj, yj, zj));
                    if (a >= kv) {
                        const float4 w = row32(j);
                        kahan_add(acc.x, comp.x, w.x);
                        kahan_add(acc.y, comp.y, w.y);
                        kahan_add(acc.z, comp.z, w.z);
                        kahan_add(acc.w, comp.w, w.w);
                    }
                }
            }
        }
        acc.x = div_k(acc.x), acc.y = div_k(acc.y), acc.z = div_k(acc.z), acc.w = div_k(acc.w);
        xm[s] = acc;
        const float4 xi4 = row32(g - cloud_base);
        st4(st + (4 * s + p) * ST_STRIDE + 4 * q,
            make_float4(acc.x - xi4.x, acc.y - xi4.y, acc.z - xi4.z, acc.w - xi4.w));
    }

    // ---- conv_a, conv_b ----
    f16x8 bh[4], bl[4];
    f32x16 a1[2], a2[2];
    float inv_row = stage_to_bop(st, bh, bl, lane);
#ifdef BLK_ABL_NOLAYERS   // timing only
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) a2[t][r] = inv_row + (float)r;
#else
    layer_s64(wa, ba, tia, bh, bl, inv_row, a1, lane);
    relu16(a1[0]);
    relu16(a1[1]);
    layer_acc64(wb, bb, tib, a1, a2, lane);
#endif
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
    if (!has_next) continue;

    // ---- next block's leading conv ----
    inv_row = stage_to_bop(st, bh, bl, lane);
    layer_s64(wn, bn, tin, bh, bl, inv_row, a1, lane);
    relu16(a1[0]);
    relu16(a1[1]);
    acc_to_stage(st, a1, lane);
#pragma unroll
    for (int s = 0; s < 8; ++s)
        st4(x_next + (size_t)(g0 + 4 * s + p) * 64 + 4 * q, ld4(st + (4 * s + p) * ST_STRIDE + 4 * q));
  }
}

// ---- fp16 rows, fp16 activations (EPC-Net) ---------------------------------------------------------------------------
// Every tensor that crosses HBM is fp16 (x rows, the concat slice, x_next: the kernel's HBM bytes halve, and so do the
// gather's), every MFMA operand is ONE fp16 value per activation against fp16 hi + lo weights (W5_SCALE comment in
// common.h: two MFMAs per product, 16 per layer), and everything in between (neighbour sum, mean, xm - x, t + xm,
// accumulators) is f32.  The 2^-12 roundings are independent per point and channel and average out in the VLAD
// aggregation over the cloud: measured descriptor effect 6e-7.  EPC-Net-L's max-pool head keeps single
// points, so it stays on the f32 kernel above.
// Lane mapping of the row phases: 8 lanes x 8 channels (16 B of fp16) per point, 8 points per wave-instruction.
#define ST16 72  // halfs per staged row: 64 + 8 pad (144 B: conflict-free b128 / b64 accesses in both layouts)


// 64->64 layer on fp16 fragments: bop[s] = the lane's B fragment of k-step s (order fixed by the pack mode)
__device__ __forceinline__ void layer_f16(const float* lw, const float* lbias, const f16x8 (&bop)[4], f32x16 (&acc)[2],
                                          int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        acc_init_bias(acc[t], lbias + 32 * t, h);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f16x8 ah = ldfrag16(lw + (((t * 4 + s) * 2 + 0) * 64 + lane) * 4);
            const f16x8 al = ldfrag16(lw + (((t * 4 + s) * 2 + 1) * 64 + lane) * 4);
            acc[t] = mfma_f16(al, bop[s], acc[t]);
            acc[t] = mfma_f16(ah, bop[s], acc[t]);
        }
    }
}

// fp16 range guard (EPC_STATUS_FP16_RANGE): every value this kernel rounds to fp16 passes through range_track, which
// keeps the packed maximum of the |bit patterns| (|fp16| orders like its bit pattern; Inf = 0x7c00, NaN above it) -- two
// VALU instructions per pair of values; one compare per lane at the end decides.
typedef unsigned short u16x2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void range_track(u16x2_t& m, unsigned int packed_pair) {
    m = __builtin_elementwise_max(m, __builtin_bit_cast(u16x2_t, packed_pair & 0x7fff7fffu));
}
__device__ __forceinline__ void range_track4(u16x2_t& m, const u32x4& w) {
    range_track(m, w[0]), range_track(m, w[1]), range_track(m, w[2]), range_track(m, w[3]);
}

// ReLU + removal of W5_SCALE + rounding to fp16 of 4 consecutive accumulator registers (= 4 consecutive channels)
__device__ __forceinline__ uint2 relu_descale_pack4(const f32x16& a, int r0) {
    _Float16 hv[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) hv[e] = (_Float16)(fmaxf(a[r0 + e], 0.f) * (1.0f / W5_SCALE));
    uint2 w;
    w.x = (unsigned)__builtin_bit_cast(unsigned short, hv[0]) | ((unsigned)__builtin_bit_cast(unsigned short, hv[1]) << 16);
    w.y = (unsigned)__builtin_bit_cast(unsigned short, hv[2]) | ((unsigned)__builtin_bit_cast(unsigned short, hv[3]) << 16);
    return w;
}

// accumulators (channel in register, point on lane) -> staged fp16 rows [pt][64]
__device__ __forceinline__ void acc_to_stage16(unsigned short* st, const f32x16 (&acc)[2], int lane) {
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<uint2*>(st + j * ST16 + 32 * t + 8 * g + 4 * h) = relu_descale_pack4(acc[t], 4 * g);
}

// staged fp16 row of the lane's point -> B fragments (k-step s = channels 16s + 8h .. +7)
__device__ __forceinline__ void stage16_to_bop(const unsigned short* st, f16x8 (&bop)[4], int lane) {
    const int j = lane & 31, h = lane >> 5;
#pragma unroll
    for (int s = 0; s < 4; ++s)
        bop[s] = __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(st + j * ST16 + 16 * s + 8 * h));
}

// waves per workgroup of the fp16 kernel (its staging tile is 4.5 KB per wave, so more waves fit beside the 49-KB
// weight pack than in the f32 kernel): tuned on MI355X, scripts/tune_lib.sh
#ifndef BLK16_WAVES
#define BLK16_WAVES 16
#endif
#define BLK16_THREADS (64 * BLK16_WAVES)
#define BLK16_LDS_BYTES (BLK_PACK * 4 + BLK16_WAVES * 32 * ST16 * 2)

__global__ __launch_bounds__(BLK16_THREADS) void proxyconv_block_f16_kernel(
    const unsigned short* __restrict__ x16, const float* __restrict__ xyz, const void* __restrict__ idx, int idx_u16,
    const int32_t* __restrict__ cnt, const float* __restrict__ kth, int cap, const float* __restrict__ pack,
    int has_next, int total_points, int n, float kdiv, unsigned short* __restrict__ out16, int out_stride, int out_off,
    unsigned short* __restrict__ x_next16, int32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int o = tid * 4; o < BLK_PACK; o += BLK16_THREADS * 4) st4(lds + o, ld4(pack + o));
    __syncthreads();
    const float* wa = lds;
    const float* ba = lds + 4096;
    const float* wb = lds + 4160;
    const float* bb = lds + 4160 + 4096;
    const float* wn = lds + 8320;
    const float* bn = lds + 8320 + 4096;
    unsigned short* st = reinterpret_cast<unsigned short*>(lds + BLK_PACK) + wave * 32 * ST16;

    const int bid = xcd_contiguous_block(blockIdx.x, gridDim.x);
    const int g0 = (bid * BLK16_WAVES + wave) * 32;
    if (g0 >= total_points) return;  // no further workgroup barriers below
    const int cloud_base = __builtin_amdgcn_readfirstlane((g0 / n) * n);
    const int p = lane >> 3, q = lane & 7;
    const char* xc = reinterpret_cast<const char*>(x16) + (size_t)cloud_base * 128;  // 128-B rows
    auto row16 = [&](int row) { return *reinterpret_cast<const u32x4*>(xc + (unsigned)(row * 128 + q * 16)); };
    // acc[0..7] += the 8 halfs of a row slice: v_fma_mix_f32 converts and adds in one instruction (x * 1.0 + acc: the only
    // rounding is the add's)
    auto add_row = [&](float (&acc)[8], const u32x4& raw) {
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) {
            asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[2 * w2]) : "v"(raw[w2]));
            asm("v_fma_mix_f32 %0, %1, 1.0, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(acc[2 * w2 + 1]) : "v"(raw[w2]));
        }
    };
    const int wg0 = __builtin_amdgcn_readfirstlane(g0);
    const bool u16 = idx_u16 != 0;
    const float rk = 1.0f / kdiv;
    auto div_k = [&](float a) {
        const float q0 = a * rk;
        return __builtin_fmaf(__builtin_fmaf(-q0, kdiv, a), rk, q0);
    };

    u16x2_t rmax = {0, 0};
    auto report_range = [&]() {   // wave-uniform cloud: one atomic per wave, and only when something overflowed
        const bool over = max(rmax[0], rmax[1]) >= 0x7c00;
        if (status && __builtin_amdgcn_ballot_w64(over) != 0ull && lane == 0) atomicOr(status + g0 / n, EPC_STATUS_FP16_RANGE);
    };
    // ---- gather-mean, 8 points per pass; t = xm - x staged as fp16 rows ----
    float xm[4][8];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int g = g0 + 8 * s + p;
        const int c = cnt[g];
        const bool ovf = c > cap;
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.f;
        if (!ovf) {
            int nb[EPC_KNN_SELECT];
            load_nb20(reinterpret_cast<const char*>(idx), (unsigned)(wg0 + 8 * s + p), cap, u16, nb);
            u32x4 raw[EPC_KNN_SELECT];
#pragma unroll
            for (int m = 0; m < EPC_KNN_SELECT; ++m) raw[m] = row16(nb[m]);
#pragma unroll
            for (int m = 0; m < EPC_KNN_SELECT; ++m) add_row(acc, raw[m]);  // ascending j, one rounding per add
            for (int m = EPC_KNN_SELECT; m < c; ++m) add_row(acc, row16(list_entry(idx, (size_t)g * cap + m, u16)));
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
                    if (a >= kv) add_row(acc, row16(j));   // (fast arithmetic: plain running sum, its rows are fp16 anyway)
                }
            }
        }
        const f16x8 self = __builtin_bit_cast(f16x8, row16(g - cloud_base));
        f16x8 t16;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            xm[s][e] = div_k(acc[e]);
            t16[e] = (_Float16)(xm[s][e] - (float)self[e]);
        }
        const u32x4 tw = __builtin_bit_cast(u32x4, t16);
        range_track4(rmax, tw);
        *reinterpret_cast<u32x4*>(st + (8 * s + p) * ST16 + 8 * q) = tw;
    }

    // ---- conv_a, conv_b ----
    f16x8 bop[4];
    f32x16 a1[2], a2[2];
    stage16_to_bop(st, bop, lane);
    layer_f16(wa, ba, bop, a1, lane);
    {   // accumulators -> B fragments of conv_b: k-step st = registers 8(st&1) .. +7 of tile st>>1 (PACK_ACC order)
#pragma unroll
        for (int k4 = 0; k4 < 4; ++k4) {
            const uint2 lo = relu_descale_pack4(a1[k4 >> 1], 8 * (k4 & 1)), hi = relu_descale_pack4(a1[k4 >> 1], 8 * (k4 & 1) + 4);
            u32x4 w;
            w[0] = lo.x, w[1] = lo.y, w[2] = hi.x, w[3] = hi.y;
            range_track4(rmax, w);
            bop[k4] = __builtin_bit_cast(f16x8, w);
        }
    }
    layer_f16(wb, bb, bop, a2, lane);

    // ---- out = t + xm: fp16 rows to the concat slice, and staged again as the next conv's operand ----
    acc_to_stage16(st, a2, lane);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        unsigned short* row = st + (8 * s + p) * ST16 + 8 * q;
        const u32x4 t16w = *reinterpret_cast<const u32x4*>(row);   // conv_b's activations as acc_to_stage16 rounded them
        const f16x8 t16 = __builtin_bit_cast(f16x8, t16w);
        f16x8 o16;
#pragma unroll
        for (int e = 0; e < 8; ++e) o16[e] = (_Float16)((float)t16[e] + xm[s][e]);
        const u32x4 ow = __builtin_bit_cast(u32x4, o16);
        range_track4(rmax, t16w);
        range_track4(rmax, ow);
        *reinterpret_cast<u32x4*>(out16 + (size_t)(g0 + 8 * s + p) * out_stride + out_off + 8 * q) = ow;
        *reinterpret_cast<u32x4*>(row) = ow;
    }
    if (!has_next) {
        report_range();
        return;
    }

    // ---- next block's leading conv -> fp16 rows ----
    stage16_to_bop(st, bop, lane);
    layer_f16(wn, bn, bop, a1, lane);
    acc_to_stage16(st, a1, lane);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const u32x4 xw = *reinterpret_cast<const u32x4*>(st + (8 * s + p) * ST16 + 8 * q);
        range_track4(rmax, xw);
        *reinterpret_cast<u32x4*>(x_next16 + (size_t)(g0 + 8 * s + p) * 64 + 8 * q) = xw;
    }
    report_range();
}

// conv1 (models/epc-net.py:66-69): 3 -> 64, folded BN, ReLU.  16 lanes x float4 per point.
__global__ __launch_bounds__(256) void conv1_kernel(const float* __restrict__ xyz, const float* __restrict__ pack,
                                                    int total_points, float* __restrict__ x,
                                                    unsigned short* __restrict__ x16, int32_t* __restrict__ status,
                                                    int n) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int g = t >> 4, q = t & 15;
    if (g >= total_points) return;
    const float px = xyz[3 * (size_t)g], py = xyz[3 * (size_t)g + 1], pz = xyz[3 * (size_t)g + 2];
    const float4 w0 = ld4(pack + 4 * q), w1 = ld4(pack + 64 + 4 * q), w2 = ld4(pack + 128 + 4 * q);
    const float4 b = ld4(pack + 192 + 4 * q);
    const float4 y = conv1_quad(px, py, pz, w0, w1, w2, b);
    if (x) st4(x + (size_t)g * 64 + 4 * q, y);
    if (x16) {
        reinterpret_cast<uint2*>(x16)[(size_t)g * 16 + q] = pack_half4(y);  // fp16 rows: block 1's gather source
        if (status && fmaxf(fmaxf(y.x, y.y), fmaxf(y.z, y.w)) > 65504.0f) atomicOr(status + g / n, EPC_STATUS_FP16_RANGE);
    }
}

// (library-internal form: with the per-cloud status words of the pipeline; n = points per cloud)
int epc_conv1_launch(const float* xyz, const void* packed_conv1, int num_points_total, float* x, void* x16,
                     int32_t* status, int n, void* stream) {
    EPC_CHECK_ARG(xyz && packed_conv1 && (x || x16), "null pointer");
    EPC_CHECK_ARG(num_points_total >= 0 && (!status || n > 0), "bad shape");
    if (num_points_total == 0) return EPC_OK;
    const long threads = (long)num_points_total * 16;
    hipLaunchKernelGGL(conv1_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       xyz, (const float*)packed_conv1, num_points_total, x, (unsigned short*)x16, status, n);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_conv1_fwd(const float* xyz, const void* packed_conv1, int num_points_total, float* x, void* x16,
                             void* stream) {
    return epc_conv1_launch(xyz, packed_conv1, num_points_total, x, x16, nullptr, 0, stream);
}

extern "C" int epc_proxyconv_block_fwd(const float* x, const void* x16, const float* xyz, const void* idx, int idx_u16,
                                       const int32_t* cnt, const float* kth, int cap, const void* packed_block,
                                       int has_next, int num_clouds, int n, int knn, float* out, void* out16,
                                       int out_stride, int out_off, float* x_next, void* x_next16, int32_t* status,
                                       void* stream) {
    const bool f16 = x16 != nullptr;
    EPC_CHECK_ARG(xyz && idx && cnt && kth && packed_block, "null pointer");
    if (f16) {
        EPC_CHECK_ARG(out16 && (!has_next || x_next16), "fp16 mode needs out16 (and x_next16 when has_next)");
        EPC_CHECK_ARG(out_stride % 8 == 0 && out_off % 8 == 0 && out_off + 64 <= out_stride, "bad output slice");
    } else {
        EPC_CHECK_ARG(x && out && (!has_next || x_next), "f32 mode needs x, out (and x_next when has_next)");
        EPC_CHECK_ARG(out_stride % 4 == 0 && out_off % 4 == 0 && out_off + 64 <= out_stride, "bad output slice");
    }
    EPC_CHECK_ARG(cap == EPC_KNN_CAP, "neighbour-list capacity must be EPC_KNN_CAP (32)");
    EPC_CHECK_ARG(n > 0 && n % 32 == 0, "num_points must be a multiple of 32");
    EPC_CHECK_ARG(knn > 0, "KNN divisor must be positive");
    if (num_clouds <= 0) return num_clouds == 0 ? EPC_OK : EPC_EINVAL;
    const long total = (long)num_clouds * n;
    EPC_CHECK_ARG(total < (1L << 31), "too many points");
    const size_t lds_bytes = f16 ? (size_t)BLK16_LDS_BYTES : BLK_LDS_FLOATS * sizeof(float);
    const void* fn = f16 ? reinterpret_cast<const void*>(proxyconv_block_f16_kernel)
                         : reinterpret_cast<const void*>(proxyconv_block_kernel);
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
        epc_set_error("epc_proxyconv_block_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return EPC_EHIP;
    }
    const int wpb = f16 ? BLK16_WAVES : BLK_WAVES;
    unsigned blocks = (unsigned)((total + wpb * 32 - 1) / (wpb * 32));
    if (!f16) {
        // The f32 kernel's workgroups walk a tile range (its comment): with one workgroup per 12 tiles a grid of a few rounds
        // leaves CUs idle in the last one (64 clouds: 683 workgroups on 256 CUs = 2.67 rounds), and every workgroup stages the
        // 49-KB weight pack.  So a grid beyond one round is ONE persistent workgroup per CU (154 KB of LDS admit no second
        // one).  Same box, same bits: EPC-Net, 64 clouds: 0.097 -> 0.084 ms per block, step 1.168 -> 1.123 ms; EPC-Net-L,
        // 256 clouds (10.7 rounds): 1.839 -> 1.769 ms.
        const int num_cus = epc_device_cu_count();   // per device id (common.h; ADVICE r3: was one static for the first device seen)
        if (blocks > (unsigned)num_cus && blocks <= (unsigned)BLK_PERSIST_MAX_ROUNDS * (unsigned)num_cus) blocks = (unsigned)num_cus;
    }
    if (f16)
        hipLaunchKernelGGL(proxyconv_block_f16_kernel, dim3(blocks), dim3(BLK16_THREADS), lds_bytes, (hipStream_t)stream,
                           (const unsigned short*)x16, xyz, idx, idx_u16, cnt, kth, cap, (const float*)packed_block, has_next,
                           (int)total, n, (float)knn, (unsigned short*)out16, out_stride, out_off,
                           (unsigned short*)x_next16, status);
    else
        hipLaunchKernelGGL(proxyconv_block_kernel, dim3(blocks), dim3(BLK_THREADS), lds_bytes, (hipStream_t)stream, x,
                           xyz, idx, idx_u16, cnt, kth, cap, (const float*)packed_block, has_next, (int)total, n, (float)knn,
                           out, out_stride, out_off, x_next);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}
