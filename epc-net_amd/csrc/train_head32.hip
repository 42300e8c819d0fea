// The head of the training step -- conv5, the per-point l2 norm, the VLAD soft assignment and aggregation (models/epc-net.py:136-148,
// loupe.py:249-291), forward and backward -- in the DEFAULT, f32-accurate arithmetic on f32 tensors, as the streaming kernels of
// train_head16.hip: one pass over the (rows, 1024) tensor per product, the small operand resident or streamed through LDS, and the feature
// map f = l2_normalize(relu(bn(z5))) never written (BatchNorm + ReLU applied to z5 as it is loaded, the row norm in the epilogue or on the
// other operand).  What it replaces: the generic tile GEMMs of train_ops.hip on this path -- conv5's forward 202 us, the BatchNorm /
// ReLU / row-norm pass that WROTE f 114 us, the assignment product 100 us, the aggregation 95 us, f dvlad 73 us, f^T dz 94 us, dz5 W5^T
// 153 us at 18 x 4096 rows -- every one of them re-reading and re-splitting f32 operand tiles per output tile.
//
//   arithmetic   conv5's forward: scaled split-fp16 (common.h "f16x3s": every row of cat and every column of W5 brought to [2^14, 2^15)
//                by a power of two, hi + lo fp16, three products, 2^-21 per product, no range restriction) -- the inference kernels' form;
//                the assignment, the aggregation and every backward product: two bf16 pieces per operand, three products
//                (epc_gemm_f32_fast's: 2^-16 per product, averaged over the 1024 channels / a cloud's 4096 points -- the arithmetic of
//                the inference path's assignment and aggregate).
//   tensors      all f32: cat (rows, 256), z5 / du / dz5 (rows, 1024), za / a / dz / da (rows, 64), rn (rows), dcat (rows, 256).
//   the rest of the head's backward (the feature gradient through conv5's tail: epc_vlad_df_tail; dz5: epc_bn_apply_bwd_given; dW5: the
//   split-K tile product) stays on train_head.hip / train_ops.hip.
#include "train_head_common.h"

// ----------------------------------------------------------------------------------------------------------------
// conv5's weights as scaled split-fp16 B fragments: [chunk c < 16][k-step s < 16][nt < 2][hi, lo][lane] (v_mfma_f32_32x32x16_f16:
// lane (i, h) holds W[16 s + 8 h + j][64 c + 32 nt + i] x colscale), and the inverse column scales [1024].
// ----------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void h32_pack_conv5_kernel(const float* __restrict__ W, u32x4* __restrict__ out, float* __restrict__ inv_col) {
    // grid (16 chunks, 2 column tiles, 4 k quarters): 128 small workgroups (sixteen large ones took 21 us of the step's critical path);
    // each finds the scales of its 32 columns (every quarter redundantly: 32 KB of reads) and packs 4 k-steps of them
    __shared__ float cmax[8][32];
    __shared__ float cscale[32];
    const int c = blockIdx.x, nt = blockIdx.y, kq = blockIdx.z, tid = threadIdx.x;
    {
        const int col = tid & 31, part = tid >> 5;
        float m = 0.f;
        for (int k = 32 * part; k < 32 * part + 32; ++k) m = fmaxf(m, fabsf(W[(size_t)k * 1024 + 64 * c + 32 * nt + col]));
        cmax[part][col] = m;
    }
    __syncthreads();
    if (tid < 32) {
        float m = cmax[0][tid];
#pragma unroll
        for (int q = 1; q < 8; ++q) m = fmaxf(m, cmax[q][tid]);
        float sc, inv;
        row_scale_pow2(m, sc, inv);
        cscale[tid] = sc;
        if (kq == 0) inv_col[64 * c + 32 * nt + tid] = inv;
    }
    __syncthreads();
    {
        const int l = tid & 63, s = 4 * kq + (tid >> 6), i = l & 31, h = l >> 5;
        const float* src = W + (size_t)(16 * s + 8 * h) * 1024 + 64 * c + 32 * nt + i;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = src[(size_t)j * 1024];
        f16x8 hi, lo;
        split8_f16s(v, cscale[i], hi, lo);
        u32x4* dst = out + ((size_t)(c * 16 + s) * 2 + nt) * 2 * 64 + l;
        dst[0] = __builtin_bit_cast(u32x4, hi);
        dst[64] = __builtin_bit_cast(u32x4, lo);
    }
}

// ----------------------------------------------------------------------------------------------------------------
// conv5's forward in the training step: z5 (rows, 1024) f32 = cat W5 + b5 and the batch statistics of the product (models/epc-net.py:136
// with is_training; utils/tf_util.py:94-106, 472-476).  h16_conv5_fwd_kernel's shape in the scaled split-fp16 arithmetic: a wave's 32 rows
// resident as hi + lo fragments (128 registers: the row's largest magnitude meets over the two lane halves, every eight raw values are
// split in place), W5 streaming through double-buffered 32-KB LDS stages of 64 columns x 128 k (hi + lo), three products per k-step,
// the accumulators un-scaled by inv_row x inv_col in the epilogue; lane = column: a row's 32 values leave as one 128-byte run.
// ----------------------------------------------------------------------------------------------------------------
#define C32_STAGE_U4 (8 * 2 * 2 * 64)   // 32 KB: [k-step 8][nt 2][hi, lo][lane]

// Work split (round 5): EIGHT waves = a 256-row tile per workgroup, one workgroup per CU (256 registers a lane: two waves per SIMD).  What
// bounds this kernel is the weights' stream: every workgroup draws the whole 1-MB pack through its CU's miss path (~10 B/clk), 604 MB per
// launch with 128-row tiles -- twice the bytes of z5 -- and 302 with 256-row ones.  A workgroup takes a tile and a RANGE of its sixteen
// 64-column chunks: 18 x 4096 rows are 288 tiles, and a tile per workgroup on 256 CUs is two rounds with the second one eighth full;
// instead the first `whole` tiles (a multiple of the slot count) go to one workgroup each and every remaining tile to `parts` workgroups
// of 16 / parts chunks -- 256 + 32 x 8 workgroups, 18 chunk-times per CU instead of 32.  Statistics partials stay [tile][3][1024]: every
// (tile, chunk) is produced by exactly one workgroup.
// The 48 products of a half-stage: a k-step's four B fragments ([nt][hi, lo]) are read ONE k-step ahead of the six products that use them
// (the compiler's own schedule read two fragments, waited, multiplied three times)
template <int S>
__device__ __forceinline__ void c32_step(const u32x4* __restrict__ B, int lane, const f16x8* ah, const f16x8* al, u32x4 (&f)[2][4],
                                         f32x16& acc0, f32x16& acc1) {
    if constexpr (S < 8) {
        if constexpr (S + 1 < 8) {
#pragma unroll
            for (int q = 0; q < 4; ++q) f[(S + 1) & 1][q] = B[((S + 1) * 4 + q) * 64 + lane];
        }
        const f16x8 bh0 = __builtin_bit_cast(f16x8, f[S & 1][0]), bl0 = __builtin_bit_cast(f16x8, f[S & 1][1]);
        const f16x8 bh1 = __builtin_bit_cast(f16x8, f[S & 1][2]), bl1 = __builtin_bit_cast(f16x8, f[S & 1][3]);
        acc0 = mfma_f16(al[S], bh0, acc0);
        acc1 = mfma_f16(al[S], bh1, acc1);
        acc0 = mfma_f16(ah[S], bl0, acc0);
        acc1 = mfma_f16(ah[S], bl1, acc1);
        acc0 = mfma_f16(ah[S], bh0, acc0);
        acc1 = mfma_f16(ah[S], bh1, acc1);
        if constexpr (S + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
        c32_step<S + 1>(B, lane, ah, al, f, acc0, acc1);
    }
}
__device__ __forceinline__ void c32_half(const u32x4* __restrict__ B, int lane, const f16x8* ah, const f16x8* al, f32x16& acc0, f32x16& acc1) {
    u32x4 f[2][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) f[0][q] = B[q * 64 + lane];
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    c32_step<0>(B, lane, ah, al, f, acc0, acc1);
}

#define C32_WAVES 8
__global__ __launch_bounds__(64 * C32_WAVES, 1) void h32_conv5_fwd_kernel(const float* __restrict__ A, int rows, const u32x4* __restrict__ Bp,
                                                               const float* __restrict__ inv_col, const float* __restrict__ bias,
                                                               float* __restrict__ Z, float* __restrict__ stats, int whole, int parts) {
    __shared__ u32x4 Bs[2][C32_STAGE_U4];
    __shared__ float wst[2][C32_WAVES][3][64];
    __shared__ float rowc[C32_WAVES][32];
    __shared__ float colc[2][1024];     // inverse column scales, bias
    __shared__ bool wlive[C32_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int i = lane & 31, h = lane >> 5;
    const int b = blockIdx.x;
    const int tile = b < whole ? b : whole + (b - whole) / parts;
    const int st_begin = b < whole ? 0 : ((b - whole) % parts) * (16 / parts), st_end = b < whole ? 16 : st_begin + 16 / parts;
    const int r0 = tile * (32 * C32_WAVES) + wave * 32;
    const bool live = r0 < rows;
    if (lane == 0) wlive[wave] = live;
    // W5's stages travel global -> LDS by LDS-DMA (round 6; h16_conv5_fwd_kernel's scheme): a wave moves four 1-KB pieces of a stage, and
    // waits for them with a counted s_waitcnt in front of the barrier that publishes the buffer -- the chunk's 32 z5 stores stay in flight
    constexpr int PER = C32_STAGE_U4 / (64 * C32_WAVES);
    const unsigned bs_base = (unsigned)(size_t)(const __attribute__((address_space(3))) u32x4*)&Bs[0][0];
    auto request = [&](int step, int buf) {   // step = 2 chunk + half: [chunk][k-step][nt][piece][lane], so halves are contiguous
        const float* from = reinterpret_cast<const float*>(Bp + (size_t)step * C32_STAGE_U4);
#pragma unroll
        for (int u = 0; u < PER; ++u)
            glds16(from, 16u * ((u * C32_WAVES + wave) * 64 + lane), bs_base + 16u * (buf * C32_STAGE_U4 + (u * C32_WAVES + wave) * 64));
    };
    f16x8 ah[16], al[16];
    {
        const float* p = A + (size_t)min(r0 + i, rows - 1) * 256 + 8 * h;
        float4 raw[16][2];
        float m = 0.f;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            raw[s][0] = ld4(p + 16 * s), raw[s][1] = ld4(p + 16 * s + 4);
            m = fmaxf(m, fmaxf(fmaxf(fabsf(raw[s][0].x), fabsf(raw[s][0].y)), fmaxf(fabsf(raw[s][0].z), fabsf(raw[s][0].w))));
            m = fmaxf(m, fmaxf(fmaxf(fabsf(raw[s][1].x), fabsf(raw[s][1].y)), fmaxf(fabsf(raw[s][1].z), fabsf(raw[s][1].w))));
        }
        m = fmaxf(m, __shfl_xor(m, 32));
        float sc, inv;
        row_scale_pow2(m, sc, inv);
        if (h == 0) rowc[wave][i] = inv;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float v[8] = {raw[s][0].x, raw[s][0].y, raw[s][0].z, raw[s][0].w, raw[s][1].x, raw[s][1].y, raw[s][1].z, raw[s][1].w};
            split8_f16s(v, sc, ah[s], al[s]);
        }
    }
    request(2 * st_begin, 0);
    // the epilogue's column tables from LDS (the loop holds no load the compiler can see); the resident fragments pinned as complete --
    // the compiler's counters do not see the loop's asm waits and would guard every first use of a fragment with a counted vmcnt
    for (int e = tid; e < 1024; e += 64 * C32_WAVES) colc[0][e] = inv_col[e], colc[1][e] = bias[e];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        asm volatile("" : "+v"(ah[s]));
        asm volatile("" : "+v"(al[s]));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto flush_stats = [&](int st, int buf) {
        if (tid < 64) {
            float S1, S2, P;
            h16_merge_stats<C32_WAVES>(wst[buf], wlive, tid, S1, S2, P);
            float* o = stats + (size_t)tile * 3 * 1024 + 64 * st + tid;
            o[0] = S1, o[1024] = S2, o[2048] = P;
        }
    };
    float* const zrow = Z + (size_t)r0 * 1024;   // (wave-uniform base: the stores take 32-bit lane offsets)
    for (int st = st_begin; st < st_end; ++st) {
        const int sb = st & 1;
        f32x16 acc[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.f;
        auto half = [&](int buf, int s0) { c32_half(Bs[buf], lane, ah + s0, al + s0, acc[0], acc[1]); };
        if (st > st_begin) flush_stats(st - 1, sb ^ 1);
        request(2 * st + 1, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (live) half(0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (nothing younger than the request in this half)
        __syncthreads();
        if (st + 1 < st_end) request(2 * st + 2, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (live) {
            half(1, 8);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const int col = 64 * st + 32 * nt + i;
                const float ic = colc[0][col], bv = colc[1][col];
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[nt][r] *= rowc[wave][mfma_row(r, h)] * ic;       // (powers of two: exact)
                const float p = __shfl(acc[nt][0], i);                         // the wave's row 0
                float s1 = 0.f, s2 = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float d = acc[nt][r] - p;
                    s1 += d, s2 += d * d;
                }
                s1 += __shfl_xor(s1, 32), s2 += __shfl_xor(s2, 32);
                if (h == 0) wst[sb][wave][0][32 * nt + i] = s1, wst[sb][wave][1][32 * nt + i] = s2, wst[sb][wave][2][32 * nt + i] = p;
#pragma unroll
                for (int r = 0; r < 16; ++r) zrow[mfma_row(r, h) * 1024 + col] = acc[nt][r] + bv;
            }
        }
        if (live) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");   // (the 32 z5 stores of this chunk may still travel)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    flush_stats(st_end - 1, (st_end - 1) & 1);
}

// ----------------------------------------------------------------------------------------------------------------
// conv5's weight gradient dW5 (256, 1024) = cat^T dz5 on f32 rows (models/epc-net.py:136's kernel, backward), two bf16 pieces per operand and
// three products -- the arithmetic of every backward product of this file -- in h16_dw5_kernel's shape (train_head16.hip): a workgroup of
// eight waves owns all 256 input channels x 256 output columns over a slice of the rows and moves every byte ONCE, 16 bytes per lane.
// What differs: a step is 32 rows (2 k-steps; hi + lo fragments of both operands: 2 x 64 KB of LDS); a loader wave takes 8 rows of its
// operand, a whole 1-KB row per load instruction, lane l the values 4 l .. 4 l + 3 -- the MFMA wants 8 consecutive rows of ONE value per lane,
// and with f32 in registers that is the split itself (no byte permutes): value q of the lane's 8 rows -> one hi and one lo entry.  Entry
// (tile T, lane L) sits at position L ^ (T & 3) of its tile: a write instruction (fixed q) then spreads its 64 lanes over all sixteen
// 16-byte bank groups, four lanes each (in lane order: four groups, sixteen lanes each).  Per k-step and wave 12 fragment reads feed 24
// MFMAs (lo hi + hi lo + hi hi into the same eight accumulators).  Slices are added in ascending order by h16_partial_reduce_kernel.
// Replaces the split-K tile product of train_ops.hip on this path (gemm_split_kernel<2,2,2,false,2>: operand tiles re-read and re-split per
// output tile, 159 + 10 us at 18 x 4096 rows; this one 129 + 21 for the slices' sum).  What bounds it: 602 MB per launch -- dz5 once and
// cat once per column tile (a 256 x 256 tile per workgroup is the square that minimises that sum; its accumulators are half a CU's
// registers) -- at 4.7 TB/s, with one 64-KB step per CU in flight.  Rows travelling TWO steps ahead (a second register set) spill at 256
// registers a lane (128 accumulators + 64 + the fragments): built, not kept.
// ----------------------------------------------------------------------------------------------------------------
#define DW32_STEP_U4 (2 * 8 * 2 * 64)   // one operand of one 32-row step: [k-step 2][tile 8][hi, lo][lane 64] x 16 bytes = 32 KB

__global__ __launch_bounds__(512, 1) void h32_dw5_kernel(const float* __restrict__ cat, const float* __restrict__ dz5, int rows,
                                                         int rows_per_wg, float* __restrict__ P) {
    extern __shared__ u32x4 dw32_lds[];                // [buffer 2][operand 2][DW32_STEP_U4]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;
    const int n0 = blockIdx.x * 256;
    const int rbeg = blockIdx.y * rows_per_wg, rend = min(rbeg + rows_per_wg, rows);
    // loader role: operand (0 = cat, 1 = dz5), rows 16 ks + 8 hh + j of the step, values 4 lane .. 4 lane + 3
    const int oper = wave >> 2, ks_ld = (wave >> 1) & 1, hh = wave & 1;
    const int last = max(rend - 1, rbeg);
    const float* src = oper == 0 ? cat + 4 * lane : dz5 + n0 + 4 * lane;
    const size_t pitch = oper == 0 ? 256 : 1024;
    float4 in[8];
    auto load = [&](int rb) {       // (unconditional loads, rows past the slice zeroed at deposit time: h16_dw5_kernel's comment)
#pragma unroll
        for (int j = 0; j < 8; ++j) in[j] = ld4(src + (size_t)min(rb + 16 * ks_ld + 8 * hh + j, last) * pitch);
    };
    // value 4 lane + q: tile T = lane >> 3, fragment lane L = 32 hh + 4 (lane & 7) + q
    const int wr_tile = lane >> 3;
    const int wr_base = oper * DW32_STEP_U4 + (ks_ld * 8 + wr_tile) * 128;
    const int wr_lane = 32 * hh + 4 * (lane & 7);
    auto deposit = [&](int buf, int rb) {         // rb: the step whose rows `in` holds
        if (oper == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (rb + 16 * ks_ld + 8 * hh + j >= rend) in[j] = make_float4(0.f, 0.f, 0.f, 0.f);   // (a row past the slice contributes zeros)
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float v[8] = {(&in[0].x)[q], (&in[1].x)[q], (&in[2].x)[q], (&in[3].x)[q],
                                (&in[4].x)[q], (&in[5].x)[q], (&in[6].x)[q], (&in[7].x)[q]};
            bf16x8 hi, lo;
            split8(v, hi, lo);
            u32x4* dst = dw32_lds + buf * 2 * DW32_STEP_U4 + wr_base + ((wr_lane + q) ^ (wr_tile & 3));
            dst[0] = __builtin_bit_cast(u32x4, hi);
            dst[64] = __builtin_bit_cast(u32x4, lo);
        }
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[q][t][r] = 0.f;
    if (rbeg < rend) {
        load(rbeg);
        deposit(0, rbeg);
    }
    __syncthreads();
    int buf = 0;
    for (int rb = rbeg; rb < rend; rb += 32, buf ^= 1) {
        const bool more = rb + 32 < rend;
        if (more) load(rb + 32);
        __builtin_amdgcn_sched_barrier(0);
        const u32x4* As = dw32_lds + buf * 2 * DW32_STEP_U4;
        const u32x4* Bs = As + DW32_STEP_U4;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int T = 4 * wm + q;
                const u32x4* f = As + (ks * 8 + T) * 128 + (lane ^ (T & 3));
                ah[q] = __builtin_bit_cast(bf16x8, f[0]), al[q] = __builtin_bit_cast(bf16x8, f[64]);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int T = 2 * wn + t;
                const u32x4* f = Bs + (ks * 8 + T) * 128 + (lane ^ (T & 3));
                bh[t] = __builtin_bit_cast(bf16x8, f[0]), bl[t] = __builtin_bit_cast(bf16x8, f[64]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[q][t] = mfma_bf16(al[q], bh[t], acc[q][t]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[q][t] = mfma_bf16(ah[q], bl[t], acc[q][t]);
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int t = 0; t < 2; ++t) acc[q][t] = mfma_bf16(ah[q], bh[t], acc[q][t]);
        }
        if (more) deposit(buf ^ 1, rb + 32);    // (the other buffer: its last readers passed the barrier at the end of the previous step)
        __syncthreads();
    }
    // D: lane (i, h), register r of (q, t) = dW5[channel 128 wm + 32 q + mfma_row(r, h)][column n0 + 64 wn + 32 t + i]
    const int i = lane & 31, h = lane >> 5;
    float* o = P + (size_t)blockIdx.y * 256 * 1024 + n0 + 64 * wn + i;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = 128 * wm + 32 * q + mfma_row(r, h);
#pragma unroll
            for (int t = 0; t < 2; ++t) o[(size_t)ch * 1024 + 32 * t] = acc[q][t][r];
        }
}

// ---- C ABI ---------------------------------------------------------------------------------------------------------------------
extern "C" size_t epc_h32_conv5_fwd_scratch_bytes(int rows) {
    if (rows <= 0) return 0;
    return (size_t)256 * 1024 * 4 + 1024 * sizeof(float) + (size_t)((rows + 127) / 128) * 3 * 1024 * sizeof(float);   // (>= the 256-row tiles' partials)
}

extern "C" int epc_h32_conv5_fwd(const float* cat, const float* W5, const float* b5, int rows, float* z5, float* mean, float* var,
                                 void* scratch, size_t scratch_bytes, void* stream) {
    EPC_CHECK_ARG(cat && W5 && b5 && z5 && mean && var && scratch, "null pointer");
    EPC_CHECK_ARG(rows > 0 && rows % 32 == 0 && (long)rows * 1024 < (1L << 32), "rows must be a positive multiple of 32 (rows * 1024 < 2^32)");
    EPC_CHECK_ARG(scratch_bytes >= epc_h32_conv5_fwd_scratch_bytes(rows), "scratch too small (epc_h32_conv5_fwd_scratch_bytes)");
    EPC_CHECK_ARG(h16_aligned16(cat) && h16_aligned16(z5) && h16_aligned16(scratch), "tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    u32x4* pack = (u32x4*)scratch;
    float* inv_col = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + (size_t)256 * 1024 * 4);
    float* stats = inv_col + 1024;
    hipLaunchKernelGGL(h32_pack_conv5_kernel, dim3(16, 2, 4), dim3(256), 0, st, W5, pack, inv_col);
    const int tile_rows = 32 * C32_WAVES, tiles = (rows + tile_rows - 1) / tile_rows;
    int whole, parts;
    tail_split(tiles, epc_device_cu_count(), 16, whole, parts);   // (one workgroup per CU)
    hipLaunchKernelGGL(h32_conv5_fwd_kernel, dim3(whole + (tiles - whole) * parts), dim3(64 * C32_WAVES), 0, st, cat, rows, (const u32x4*)pack,
                       inv_col, b5, z5, stats, whole, parts);
    epc_moments_finalize_launch(stats, tiles, 1024, rows, tile_rows, b5, mean, var, stream);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" size_t epc_h32_assign_scratch_bytes(int num_clouds, int n_points, int per_cloud_operand) {
    if (num_clouds <= 0 || n_points <= 0) return 0;
    const size_t pack = (size_t)(per_cloud_operand ? num_clouds : 1) * 1024 * 64 * 2 * 3;
    const size_t tiles = (size_t)num_clouds * ((n_points + 127) / 128), tiles96 = ((size_t)num_clouds * n_points + 95) / 96;
    return pack + (tiles > tiles96 ? tiles : tiles96) * 3 * 64 * sizeof(float);   // (the shared-operand launch may tile the rows by 96)
}

// epc_h16_assign on f32 rows: out (rows, 64) = rn (relu(bn(z5)) B), three products.  per_cloud_operand = 0: B = cluster_weights (the
// forward's logits; rn_out / mean_out / var_out written when given); 1: B = dvlad (num_clouds, 1024, 64) (da).
extern "C" int epc_h32_assign(const float* z5, const float* mean5, const float* var5, const float* gamma5, const float* beta5, float eps,
                              const float* B, int per_cloud_operand, int num_clouds, int n_points, float* out, float* rn_out,
                              float* mean_out, float* var_out, void* scratch, size_t scratch_bytes, void* stream) {
    EPC_CHECK_ARG(z5 && mean5 && var5 && gamma5 && beta5 && B && out && scratch, "null pointer");
    EPC_CHECK_ARG(num_clouds > 0 && num_clouds <= 65535 && n_points > 0 && n_points % 32 == 0, "n_points must be a positive multiple of 32");
    EPC_CHECK_ARG((mean_out == nullptr) == (var_out == nullptr), "mean_out and var_out come together");
    EPC_CHECK_ARG(scratch_bytes >= epc_h32_assign_scratch_bytes(num_clouds, n_points, per_cloud_operand), "scratch too small (epc_h32_assign_scratch_bytes)");
    EPC_CHECK_ARG(h16_aligned16(z5) && h16_aligned16(scratch) && h16_aligned16(out), "tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int nb = per_cloud_operand ? num_clouds : 1;
    // two pieces (three products, 2^-16 per product): the logits are sums of 1024 products whose errors average out (5e-7 of a logit,
    // measured against the float64 graph: tests/test_gpu_head_stream.py) -- the six-product form took 116 us against 93
    float* stats = reinterpret_cast<float*>(reinterpret_cast<char*>(scratch) + (size_t)nb * 1024 * 64 * 2 * 3);
    const dim3 grid((n_points + 127) / 128, num_clouds);
    const H16Bn bn{mean5, var5, gamma5, beta5, eps};
    h16_pack<2>(B, 64, 1, (long)1024 * 64, nb, 1024, 64, 1, 4, scratch, st);
    const long rows = (long)num_clouds * n_points;
    if (!per_cloud_operand && rows < (1L << 31) && rows_tile_waves((int)rows, 3 * epc_device_cu_count()) == 3) {
        // one operand for every row: the tiles need not respect the clouds -- 96-row workgroups where they fill the CU's three slots evenly
        // (18 x 4096 rows: 768 of them, three per CU; 576 of 128 rows leave a quarter of the CUs a third more)
        const int wgs = (int)((rows + 95) / 96);
        hipLaunchKernelGGL((hx_rowgemm_kernel<2, true, float, 2, 4, false, 3>), dim3(wgs, 1), dim3(192), 0, st, z5, (int)rows, (const u32x4*)scratch,
                           0L, bn, out, rn_out, mean_out ? stats : nullptr, HxBnb<float>{});
        if (mean_out) epc_moments_finalize_launch(stats, wgs, 64, (int)rows, 96, nullptr, mean_out, var_out, stream);
        EPC_CHECK_LAUNCH();
        return EPC_OK;
    }
    hipLaunchKernelGGL((hx_rowgemm_kernel<2, true, float, 2, 4>), grid, dim3(256), 0, st, z5, n_points, (const u32x4*)scratch,
                       per_cloud_operand ? (long)(1024 * 64 * 2 * 2 / 16) : 0L, bn, out, rn_out, mean_out ? stats : nullptr, HxBnb<float>{});
    if (mean_out) epc_moments_finalize_launch(stats, (int)(grid.x * grid.y), 64, num_clouds * n_points, 128, nullptr, mean_out, var_out, stream, n_points);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" size_t epc_h32_colgemm_scratch_bytes(int num_clouds, int n_points) {
    if (num_clouds <= 0 || n_points <= 0) return 0;
    return (size_t)num_clouds * h16_splits(num_clouds, n_points, 16) * 1024 * 64 * sizeof(float);
}

// epc_h16_colgemm on f32 rows, two bf16 pieces per operand (three products): out = relu(bn(z5))^T (rn C)
extern "C" int epc_h32_colgemm(const float* z5, const float* mean5, const float* var5, const float* gamma5, const float* beta5, float eps,
                               const float* C, const float* rn, int num_clouds, int n_points, int per_cloud, float* out, void* scratch,
                               size_t scratch_bytes, void* stream) {
    EPC_CHECK_ARG(z5 && mean5 && var5 && gamma5 && beta5 && C && rn && out && scratch, "null pointer");
    EPC_CHECK_ARG(num_clouds > 0 && n_points > 0 && n_points % 32 == 0 && (long)num_clouds * h16_splits(num_clouds, n_points, 16) <= 65535, "bad shape");
    EPC_CHECK_ARG(scratch_bytes >= epc_h32_colgemm_scratch_bytes(num_clouds, n_points), "scratch too small (epc_h32_colgemm_scratch_bytes)");
    EPC_CHECK_ARG(h16_aligned16(z5) && h16_aligned16(scratch) && h16_aligned16(out), "tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int S = h16_splits(num_clouds, n_points, 16);
    const int rows_per_wg = (n_points + S - 1) / S;
    const H16Bn bn{mean5, var5, gamma5, beta5, eps};
    hipLaunchKernelGGL((hx_colgemm_kernel<float, 2>), dim3(16, num_clouds * S), dim3(256), 0, st, z5, bn, C, rn, rows_per_wg, n_points, S,
                       (float*)scratch);
    const long per = 1024 * 64;
    if (per_cloud)
        hipLaunchKernelGGL(h16_partial_reduce_kernel, dim3((unsigned)(per / 4 / 256), num_clouds), dim3(256), 0, st, (const float*)scratch, S, per, out);
    else
        hipLaunchKernelGGL(h16_partial_reduce_kernel, dim3((unsigned)(per / 4 / 256), 1), dim3(256), 0, st, (const float*)scratch, num_clouds * S, per, out);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" size_t epc_h32_dx_scratch_bytes(void) { return (size_t)1024 * 256 * 2 * 2; }

// dcat (rows, 256) f32 = dz5 (rows, 1024) f32 times W5^T, two bf16 pieces per operand (three products)
extern "C" int epc_h32_conv5_dx(const float* dz5, const float* W5, int rows, float* dcat, void* scratch, size_t scratch_bytes, void* stream) {
    EPC_CHECK_ARG(dz5 && W5 && dcat && scratch, "null pointer");
    EPC_CHECK_ARG(rows > 0 && rows % 32 == 0, "rows must be a positive multiple of 32");
    EPC_CHECK_ARG(scratch_bytes >= epc_h32_dx_scratch_bytes(), "scratch too small (epc_h32_dx_scratch_bytes)");
    EPC_CHECK_ARG(h16_aligned16(dz5) && h16_aligned16(scratch) && h16_aligned16(dcat), "tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    h16_pack<2>(W5, 1, 1024, 0, 1, 1024, 256, 1, 2, scratch, st);      // B[k = output channel][n = input channel] = W5[n][k]
    const H16Bn none{nullptr, nullptr, nullptr, nullptr, 0.f};
    hipLaunchKernelGGL((hx_rowgemm_kernel<8, false, float, 2, 2>), dim3((rows + 127) / 128, 1), dim3(256), 0, st, dz5, rows,
                       (const u32x4*)scratch, 0L, none, dcat, (float*)nullptr, (float*)nullptr, HxBnb<float>{});
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// epc_bn_apply_bwd_given and epc_h32_conv5_dx in ONE pass (epc_h16_conv5_dx_bn on f32 tensors, two bf16 pieces per operand)
extern "C" int epc_h32_conv5_dx_bn(const float* du, const float* z5, const float* mean5, const float* var5, const float* gamma5, float eps,
                                   const float* dbeta, const float* dgamma, const float* W5, int rows, float* dz5, float* dcat, void* scratch,
                                   size_t scratch_bytes, void* stream) {
    EPC_CHECK_ARG(du && z5 && mean5 && var5 && gamma5 && dbeta && dgamma && W5 && dz5 && dcat && scratch, "null pointer");
    EPC_CHECK_ARG(rows > 0 && rows % 32 == 0, "rows must be a positive multiple of 32");
    EPC_CHECK_ARG(scratch_bytes >= epc_h32_dx_scratch_bytes(), "scratch too small (epc_h32_dx_scratch_bytes)");
    EPC_CHECK_ARG(h16_aligned16(du) && h16_aligned16(z5) && h16_aligned16(dz5) && h16_aligned16(scratch) && h16_aligned16(dcat),
                  "tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    h16_pack<2>(W5, 1, 1024, 0, 1, 1024, 256, 1, 2, scratch, st);      // B[k = output channel][n = input channel] = W5[n][k]
    const H16Bn bn{mean5, var5, gamma5, nullptr, eps};
    const HxBnb<float> bnb{z5, dbeta, dgamma, 1.0f / rows, dz5};
    // 96- or 128-row workgroups, whichever leaves fewer rows on the busiest slot (two workgroups per CU: 235-248 registers a lane)
    if (rows_tile_waves(rows, 2 * epc_device_cu_count()) == 3)
        hipLaunchKernelGGL((hx_rowgemm_kernel<8, false, float, 2, 2, true, 3>), dim3((rows + 95) / 96, 1), dim3(192), 0, st, du, rows,
                       (const u32x4*)scratch, 0L, bn, dcat, (float*)nullptr, (float*)nullptr, bnb);
    else
        hipLaunchKernelGGL((hx_rowgemm_kernel<8, false, float, 2, 2, true, 4>), dim3((rows + 127) / 128, 1), dim3(256), 0, st, du, rows,
                       (const u32x4*)scratch, 0L, bn, dcat, (float*)nullptr, (float*)nullptr, bnb);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

static int h32_dw5_splits(int rows) {
    const int cus = epc_device_cu_count();
    int s = max(1, cus / 4);                              // 4 column tiles per slice: one workgroup (eight waves) per CU
    while (s > 1 && (rows + s - 1) / s < 32) s >>= 1;     // (short inputs: at least 32 rows per slice)
    return s;
}

extern "C" size_t epc_h32_conv5_dw_scratch_bytes(int rows) {
    return rows > 0 ? (size_t)h32_dw5_splits(rows) * 256 * 1024 * sizeof(float) : 0;
}

// dW5 (256, 1024) f32 = cat^T dz5: cat (rows, 256) f32, dz5 (rows, 1024) f32, two bf16 pieces per operand (three products)
extern "C" int epc_h32_conv5_dw(const float* cat, const float* dz5, int rows, float* dW5, void* scratch, size_t scratch_bytes, void* stream) {
    EPC_CHECK_ARG(cat && dz5 && dW5 && scratch, "null pointer");
    EPC_CHECK_ARG(rows > 0 && (long)rows * 1024 < (1L << 32), "bad shape");
    EPC_CHECK_ARG(scratch_bytes >= epc_h32_conv5_dw_scratch_bytes(rows), "scratch too small (epc_h32_conv5_dw_scratch_bytes)");
    EPC_CHECK_ARG(h16_aligned16(cat) && h16_aligned16(dz5) && h16_aligned16(dW5) && h16_aligned16(scratch), "tensors must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const int S = h32_dw5_splits(rows);
    const int rows_per_wg = ((rows + S - 1) / S + 31) / 32 * 32;
    const size_t lds = (size_t)2 * 2 * DW32_STEP_U4 * sizeof(u32x4);     // 128 KB
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(h32_dw5_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        epc_set_error("epc_h32_conv5_dw: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return EPC_EHIP;
    }
    hipLaunchKernelGGL(h32_dw5_kernel, dim3(4, S), dim3(512), lds, st, cat, dz5, rows, rows_per_wg, (float*)scratch);
    const long per = 256 * 1024;
    hipLaunchKernelGGL(h16_partial_reduce_kernel, dim3((unsigned)(per / 4 / 256), 1), dim3(256), 0, st, (const float*)scratch, S, per, dW5);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}
