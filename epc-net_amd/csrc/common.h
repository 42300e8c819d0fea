// Shared device/host helpers for libepcnet_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/epcnet.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Row of a 32x32 MFMA C/D tile held by register r of lane-half h (column = lane & 31).
__host__ __device__ __forceinline__ constexpr int mfma_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// D = A(32x2) * B(2x32) + C, exact f32 (v_mfma_f32_32x32x2_f32).
// lane l supplies A[i = l & 31][k = l >> 5] and B[k = l >> 5][j = l & 31].
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// a_ij = -((sq_i + (-2 * inner)) + sq_j), inner = (x x' + y y') + z z', one rounding per operation.
// Restates utils/tf_util.py:650-656; must stay bit-identical to oracle/epcnet_oracle.py:neg_sq_dist.
__device__ __forceinline__ float sq3(float x, float y, float z) {
#pragma clang fp contract(off)
    return (x * x + y * y) + z * z;
}
__device__ __forceinline__ float neg_sq_dist(float sqi, float xi, float yi, float zi, float xj, float yj,
                                             float zj, float sqj) {
#pragma clang fp contract(off)
    float inner = (xi * xj + yi * yj) + zi * zj;
    float t = -2.0f * inner;
    return -((sqi + t) + sqj);
}

// D = A(32x16) * B(16x32) + C on bf16 inputs, f32 accumulate (v_mfma_f32_32x32x16_bf16, 32 cycles/SIMD: 16x the f32
// MFMA rate).  lane l supplies A[i = l & 31][k = 8*(l>>5) + j] and B[k = 8*(l>>5) + j][n = l & 31], j = 0..7.
__device__ __forceinline__ f32x16 mfma_bf16(bf16x8 a, bf16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}

// fp16 form of the same instruction (v_mfma_f32_32x32x16_f16): same rate and operand layout as the bf16 one.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x16 mfma_f16(f16x8 a, f16x8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
__device__ __forceinline__ bf16x8 ldfrag(const float* p) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(p));
}
__device__ __forceinline__ f16x8 ldfrag16(const float* p) {
    return __builtin_bit_cast(f16x8, *reinterpret_cast<const u32x4*>(p));
}
// 16 bytes per lane, global -> LDS without a VGPR destination (LDS-DMA): lane l reads uniform_base + lane_byte_off and lands at LDS byte
// address lds_dst + 16*l (uniform_base and lds_dst wave-uniform, in SGPRs).  Written as inline asm on purpose: with the builtin hipcc (ROCm 7.2) drains
// vmcnt(0) before the next ds_read of the same __shared__ array, which would serialise the prefetch; the asm form is
// invisible to its wait bookkeeping, so completion is tracked by the hand-placed counted s_waitcnt in the chunk loop.
__device__ __forceinline__ void glds16(const float* uniform_base, unsigned lane_byte_off, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(lane_byte_off), "s"(uniform_base), "s"(lds_dst)
                 : "memory");
}

// conv5 / assignment weights are packed as fp16 hi + lo fragments of W * W5_SCALE (2^8 keeps the lo parts of ordinary
// weights in fp16's normal range; biases are packed with the same factor and the kernel multiplies the result by
// 2^-8).  With ONE fp16 value per activation the product is x16*W_hi + x16*W_lo, f32 accumulate: two MFMAs per f32 product.
// The activation's rounding (2^-12 relative, independent per point and channel) averages out over the 256-term dot
// product and the cloud's points: measured descriptor error 1e-6; a weight rounded to one fp16 would be a
// systematic error (2e-5) -- hence the split on the weight side.
#define W5_SCALE 256.0f
// lo parts of the conv5 weights as MX fp6 (e2m3): lo = W * W5_SCALE - hi is stored per (output channel, block of 32 input
// channels) as 32 six-bit values q with one E8M0 block scale, lo ~ q * 2^e, |q| <= 7.5.  The lo term is a 2^-11
// correction, so e2m3's 4 significant bits keep it to 2^-15 of the product -- as good as the fp8 e4m3 form it replaced --
// while v_mfma_scale_f32_32x32x64_f8f6f4 runs fp6 operands in 8 passes instead of fp8's 16.  Operand facts pinned by
// scripts/probe/mfma_fp6_probe.hip: lane l supplies row / column l & 31 and k = 32 * (l >> 5) + j, value j in bits
// [6j, 6j + 6) of the lane's six dwords; each lane passes its own scale byte (2^(byte - 127)), the byte of the scale
// dword is picked by op_sel.  Six dwords per lane per 64-wide k-step, kept in LDS as 16 B + 8 B pieces.
typedef int i32x8 __attribute__((ext_vector_type(8)));
typedef int i32x6 __attribute__((ext_vector_type(6)));
typedef _Float16 f16x32 __attribute__((ext_vector_type(32)));
#define FP6_MAX 7.5f
// exponent e (clamped to [lo_e, hi_e]) with m * 2^-e <= 7.5 and > 3.75 when not clamped: e = ceil(log2(m / 7.5))
__device__ __forceinline__ int fp6_block_exponent(float m, int lo_e, int hi_e) {
    const int bits = __float_as_int(m * (1.0f / FP6_MAX));
    const int e = ((bits >> 23) & 0xff) - 127 + ((bits & 0x7fffff) ? 1 : 0);
    return min(max(e, lo_e), hi_e);
}
__device__ __forceinline__ float exp2i(int e) { return __int_as_float((127 + e) << 23); }   // 2^e, -126 <= e <= 127

// scale of the fp16 assignment fragments between epc_conv5_assign_fwd and epc_vlad_aggregate_fwd (exact power of two)
#define AGG_ASSIGN_SCALE 16384.0f
// Split-bf16 ("bf16x3") arithmetic: x = hi + lo + O(2^-17 |x|) with hi = bf16(x), lo = bf16(x - hi);
// a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, accumulated in f32 by the MFMA.  Measured on the whole network this
// keeps the descriptor within 4e-7 of the f32 oracle (plain bf16: 1.7e-4, over the 1e-4 budget) -- DESIGN.md 4.
__device__ __forceinline__ void split8(const float (&v)[8], bf16x8& hi, bf16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        hi[j] = (__bf16)v[j];
        lo[j] = (__bf16)(v[j] - (float)hi[j]);
    }
}
// Scaled split-fp16 ("f16x3s") arithmetic -- the f32-equivalent form of EPC_PRECISION_F32's conv layers: every ROW of the
// activation operand and every COLUMN of the weight operand is multiplied by a power of two that brings its largest
// magnitude into [2^14, 2^15) (exact), then split as hi = fp16(v), lo = fp16(v - hi): 22 significant bits for every element
// within 2^-28 of its row's maximum (smaller ones lose bits gradually -- their share of any dot product is below float32's
// own rounding of it), no range restriction.  Products lo*hi + hi*lo + hi*hi on the fp16 MFMA (f32 accumulate): 2^-21
// relative per product against 2^-16.5 for the split-bf16 form at the same three MFMAs -- which is what an ill-conditioned
// network needs (heavy-tailed weights: split-bf16 layers land 2.6e-4 from the float64 result where float32 itself is at
// 2e-5; this form at 2e-5; tests/test_gpu_adversarial.py).  The accumulator is un-scaled by inv_row * inv_col afterwards.
__device__ __forceinline__ void row_scale_pow2(float maxabs, float& s, float& inv_s) {
    int E = ((__float_as_int(maxabs) >> 23) & 0xff) - 127;   // maxabs >= 0: floor(log2) for normal numbers
    E = min(max(E, -100), 112);                                // all-zero / denormal rows; Inf / NaN rows (flagged elsewhere)
    s = exp2i(14 - E);
    inv_s = exp2i(E - 14);
}
__device__ __forceinline__ void split8_f16s(const float (&v)[8], float s, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float xs = v[j] * s;
        hi[j] = (_Float16)xs;
        lo[j] = (_Float16)(xs - (float)hi[j]);
    }
}

__host__ __device__ __forceinline__ unsigned short bf16_bits_rne(float f) {
    union { float f; unsigned int u; } c;
    c.f = f;
    return (unsigned short)((c.u + 0x7fffu + ((c.u >> 16) & 1u)) >> 16);
}
__host__ __device__ __forceinline__ float bf16_bits_to_float(unsigned short b) {
    union { float f; unsigned int u; } c;
    c.u = (unsigned int)b << 16;
    return c.f;
}

// Tail splitting of a row-tiled launch: `tiles` tiles of `chunks` equal column chunks on `slots` co-resident workgroups.  The first
// `whole` tiles (the largest multiple of `slots` that fits) take a workgroup each; each of the rest is cut into `parts` workgroups of
// chunks / parts chunks -- the last, partly filled round then costs a fraction of a tile's time instead of a whole one (576 tiles on
// 512 slots: 1 + 1/8 rounds instead of 2; on 768 slots: four parts each, three full rounds of quarter tiles instead of a round in which a
// quarter of the CUs carry a third more).  `parts` = the power of two that minimises the slot-time  rounds x (1 / parts + overhead), with
// a workgroup's fixed cost (its prologue: the resident operand, the first stage) priced at 6 % of a tile.
// Workgroup b < whole: tile b, every chunk; else tile whole + (b - whole) / parts, part (b - whole) % parts.
static inline void tail_split(int tiles, int slots, int chunks, int& whole, int& parts) {
    whole = slots > 0 ? tiles / slots * slots : tiles;
    const int rem = tiles - whole;
    parts = 1;
    if (rem <= 0 || slots <= 0) return;
    const double overhead = 0.06;
    double best = 1e30;
    for (int p = 1; p <= chunks; p *= 2) {
        if (chunks % p) break;
        const long rounds = ((long)rem * p + slots - 1) / slots;
        const double cost = (double)rounds * (1.0 / p + overhead);
        if (cost < best - 1e-9) best = cost, parts = p;
    }
}

// Waves per workgroup (3 or 4: 96- or 128-row tiles) of a row-tiled launch whose workgroups have no independent parts to split: the one
// that minimises  rounds x rows per workgroup  on `slots` co-resident workgroups (18 x 4096 rows on 512 slots: 768 tiles of 96 rows are two
// rounds of 96, 576 of 128 two rounds of 128; on 768 slots the former is exactly one round).
static inline int rows_tile_waves(int rows, int slots) {
    int best = 4;
    long best_cost = 0;
    for (int nw = 4; nw >= 3; --nw) {
        const long wgs = (rows + 32 * nw - 1) / (32 * nw), rounds = (wgs + slots - 1) / slots, cost = rounds * 32 * nw;
        if (nw == 4 || cost < best_cost) best = nw, best_cost = cost;
    }
    return best;
}

void epc_set_error(const char* fmt, ...);
// Compute units of the CURRENT device, cached per device id (a read-mostly table of ints: a racing first call writes the same
// value twice).  256 when the query fails.
static inline int epc_device_cu_count() {
    static int cus[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int v = cus[dev];
    if (v == 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cus[dev] = v;
    }
    return v;
}
// library-internal launchers (not part of the C ABI)
int epc_conv1_launch(const float* xyz, const void* packed_conv1, int num_points_total, float* x, void* x16,
                     int32_t* status, int n, void* stream);
int epc_moments_finalize_launch(const float* stats, int tiles, int N, int rows, int tile_rows, const float* bias, float* mean,
                                float* var, void* stream, int group_rows = 0);
int epc_partial_sum_wide_launch(const float* partials, int P, int E, float* out, void* stream);
int epc_sort_launch(const float* xyz, int num_clouds, int n, float* xyz_sorted, int32_t* perm, int32_t* status_zero,
                    void* stream);

#define EPC_CHECK_ARG(cond, msg)                                  \
    do {                                                          \
        if (!(cond)) {                                            \
            epc_set_error("%s: %s", __func__, msg);               \
            return EPC_EINVAL;                                    \
        }                                                         \
    } while (0)

#define EPC_CHECK_LAUNCH()                                                        \
    do {                                                                          \
        hipError_t e__ = hipGetLastError();                                       \
        if (e__ != hipSuccess) {                                                  \
            epc_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e__)); \
            return EPC_EHIP;                                                      \
        }                                                                         \
    } while (0)

// ---- packed inference-weight layout (floats) --------------------------------------------------------------
// A "layer pack" for a Cin->Cout 1x1 conv consumed by mfma32 in the transposed orientation
//   out^T[ch][pt] = sum_k Wf[k][ch] * in[pt][k]
// is   Wp[tile_out][e/4][lane][e%4] = Wf[chan(e, lane>>5)][32*tile_out + (lane&31)],  e = k-step (Cin/2 of them)
// followed by the folded bias bf[Cout].  Two k-step -> channel maps:
//   PACK_SPLIT : chan(e,h) = (Cin/2)*h + e                  (B operand read from a [pt][ch] row, half per lane-half)
//   PACK_ACC   : chan(e,h) = 32*(e/16) + mfma_row(e%16, h)  (B operand = accumulator registers of the previous layer)
enum { PACK_SPLIT = 0, PACK_ACC = 1 };
__host__ __device__ __forceinline__ constexpr int pack_chan(int mode, int cin, int e, int h) {
    return mode == PACK_SPLIT ? (cin / 2) * h + e : 32 * (e / 16) + mfma_row(e % 16, h);
}
__host__ __device__ __forceinline__ constexpr size_t layer_pack_floats(int cin, int cout) {
    return (size_t)cin * cout + cout;
}

// conv5 / cluster_weights are consumed by the bf16 MFMA as split hi/lo fragments (same byte count as f32):
//   W5p[chunk c][k-step s][part p (0 hi, 1 lo)][lane][8 bf16], value = part(W5f[16s + 8(lane>>5) + j][32c + (lane&31)])
//   Wcp[chunk c][s' (2)][tile t (2)][lane][8 fp16] (ONE fp16 per cluster weight, x W5_SCALE),
//        value = Wc[32c + 16s' + 8(j>>2) + 4(lane>>5) + (j&3)][32t + (lane&31)]
//        (the k order of an accumulator tile used as B operand: element j of lane-half h is row 16s'+8(j>>2)+4h+(j&3))
// Block pack: [conv_a SPLIT 64x64][conv_b ACC 64x64][conv_next SPLIT 64x64 (zeros when absent)], then (scaled split-fp16
// form only) the three layers' inverse column scales [3][64]
#define EPC_BLOCK_PACK_FLOATS (3 * (64 * 64 + 64))
#define EPC_BLOCK_PACK_FLOATS_S (EPC_BLOCK_PACK_FLOATS + 3 * 64)
// conv1 pack: Wf[3][64] + bf[64]
#define EPC_CONV1_PACK_FLOATS (3 * 64 + 64)
// conv1 (models/epc-net.py:66-69: 3 -> 64, folded BN, ReLU) for four consecutive output channels of one point; ONE
// definition shared by conv1_kernel (block.hip) and the fused form inside the kNN kernel (knn.hip), so both round alike.
__device__ __forceinline__ float4 conv1_quad(float px, float py, float pz, const float4& w0, const float4& w1,
                                             const float4& w2, const float4& b) {
    float4 y;
    y.x = fmaxf(((px * w0.x + py * w1.x) + pz * w2.x) + b.x, 0.f);
    y.y = fmaxf(((px * w0.y + py * w1.y) + pz * w2.y) + b.y, 0.f);
    y.z = fmaxf(((px * w0.z + py * w1.z) + pz * w2.z) + b.z, 0.f);
    y.w = fmaxf(((px * w0.w + py * w1.w) + pz * w2.w) + b.w, 0.f);
    return y;
}
// the four channels as fp16, packed for an 8-byte store (the fp16 row layout of EPC-Net's block chain)
__device__ __forceinline__ uint2 pack_half4(const float4& y) {
    const _Float16 h0 = (_Float16)y.x, h1 = (_Float16)y.y, h2 = (_Float16)y.z, h3 = (_Float16)y.w;
    uint2 w;
    w.x = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
    w.y = (unsigned)__builtin_bit_cast(unsigned short, h2) | ((unsigned)__builtin_bit_cast(unsigned short, h3) << 16);
    return w;
}

// XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 labels the XCD group), so
// giving each group a CONTIGUOUS range of tiles keeps all tiles of a cloud -- which gather from the same rows of x --
// behind one L2 instead of eight.  Speed only: any mapping is correct.
__device__ __forceinline__ int xcd_contiguous_block(int bid, int nb) {
#ifndef BLK_NO_XCD_REMAP
    const int q = nb >> 3, r = nb & 7, xcd = bid & 7, slot = bid >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;  // bijective for any grid size
#else
    return bid;
#endif
}

