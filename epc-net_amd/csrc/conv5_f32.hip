// conv5 (+BN+ReLU) in the f32-equivalent arithmetic (EPC_PRECISION_F32; EPC-Net-L always), fused with what consumes it, on
// v_mfma_f32_16x16x32_f16 / _bf16.
//
//   conv5_vlad_f32_kernel (EPC-Net: models/epc-net.py:134-148 + loupe.py:249-272): conv5 256 -> 1024, per-point L2 norm, the
//             soft assignment (feat @ cluster_weights, cluster_bn, softmax over 64); outputs feat (3-byte values), rnorm, the
//             assignment as bf16 hi + lo B fragments of the aggregate GEMM, per-tile a_sum partials.
//   conv5_max_f32_kernel  (EPC-Net-L: models/epc-net-l.py:84-92): conv5 128 -> 1024 and the global max over the cloud's points.
//
// Arithmetic: the scaled split-fp16 form of common.h (row / column power-of-two scales, hi + lo fp16 parts, three products
// lo*hi + hi*lo + hi*hi, f32 accumulate): 2^-21 per product.  The assignment GEMM takes feat split into bf16 hi + lo against
// bf16 hi + lo cluster weights (three products).
//
// Why the 16x16x32 shape: the kernel is bound by the matrix pipe at the clock the chip holds under matrix load, and that clock
// depends on the MFMA shape (MI355X_MICROARCH.md, DVFS give-back item 7).  scripts/probe/mfma_shape_probe.hip runs this
// kernel's chunk loop bare (2 waves per SIMD, A fragments re-read from LDS, 3 products): 32x32x16 1.22 PFLOP/s, 16x16x32
// 1.49 PFLOP/s of executed products on random data -- the same FLOP per cycle, 1.22x the wall-clock rate.
//
// Geometry: 512 threads = 8 waves (two per SIMD), one 32-point tile per wave; the wave's input row block (32 points x CIN)
// lives in registers as fragments for all 32 output chunks (CIN = 256: 128 VGPRs); W5 (hi + lo, 1 MB at CIN = 256) streams
// through a double-buffered LDS chunk (32 output channels, 128*CIN bytes) shared by the 8 waves: LDS-DMA, ONE barrier per
// chunk, the next chunk's pieces in flight under this chunk's MFMAs.
//
// Schedule.  In-kernel stamps (-DC5_STAMPS, scripts/c5_stamps.py) of the lock-step form -- every wave: chain(c), epilogue(c),
// barrier -- showed the two waves of a SIMD serialising: the older wave wins the matrix pipe and runs its 96-MFMA chain, the
// younger one runs its chain afterwards (beside the older one's epilogue), then its own epilogue beside NOTHING while the older
// wave idles at the barrier (per wave: barrier wait 11 %, the un-overlapped epilogue 23 % of the cycles; matrix pipe busy 59 %
// of the chunk loop).  So the two waves of a SIMD run the chunk in different orders around the same single barrier:
//     waves 0-3:  [DMA c+1]  chain(c)      epilogue(c)  | barrier c
//     waves 4-7:  [DMA c+1]  epilogue(c-1) chain(c)     | barrier c
// -- each SIMD always has one wave on the matrix pipe and one in the VALU / store epilogue, and both reach the barrier
// together.  Hazards: a W5 buffer is overwritten only after the barrier that follows every wave's chain on it (as before);
// waves 4-7 read chunk c's cluster weights one interval late, so those sit in FOUR LDS slots (chunk c + 4's piece is issued
// after barrier c + 2); the max-pool form's per-wave maxima sit in four slabs for the same reason and are folded two
// intervals late.  Also built, measured and dropped (round 3):
//  * 256-thread workgroups, two per CU, the SIMD partners in different workgroups: no barrier ties them, but each wave issues
//    twice the LDS-DMA and the phases drift: 0.458 vs 0.449 ms (kept for the max-pool kernel below, where it wins);
//  * the chunk's 24 assignment MFMAs deferred to the head of the wave's next chain (pure VALU / pure MFMA phases): 0.456-0.468
//    vs 0.441-0.453 ms;
//  * first-round workgroups started a quarter phase apart (so that the CUs' 256-KB prologue bursts do not coincide): 0.430 vs
//    0.435 ms -- the prologue is bound by the CU's own miss queue at HBM latency (~9 B/clk/CU), not by the shared HBM rate;
//  * a PERSISTENT workgroup (one per CU, the W5 stream never stops, a wave starts its tile at whatever chunk the stream is at
//    and spreads the previous tile's final epilogue and the next tile's row loads over nine intervals while its SIMD partner,
//    eleven intervals out of phase, computes): bit-exact feat, but 0.492 ms as first built (the boundary intervals are
//    barrier-coupled: everyone waits for the loading wave's HBM latency and for the 9k-cycle final epilogue) and, decisive, a
//    tile that starts at chunk c0 adds |feat|^2 and the assignment logits in the ROTATED chunk order c0 .. c0 - 1, so a cloud's
//    descriptor would depend on its position in the batch in the last bits (tests/test_gpu_parity.py::
//    test_full_size_properties, and retrieval.evaluate_sharded == evaluate_runs bit for bit, rely on it not doing so); starts
//    aligned to chunk 0 need a 64-interval period, i.e. ONE computing wave per SIMD at a time: at best -6 %.
//  What the stamps leave (per wave, 212k cycles per tile): prologue 18 %, chain 27 %, epilogue 31 % (1 900 cycles per chunk
//  beside the partner's chain: a 16x16x32 MFMA blocks the SIMD's vector issue for 8 of its 16 cycles), barrier 12 %.
//
// Tile algebra.  A 32-channel x 32-point chunk tile is 2 x 2 tiles of 16 x 16, a k-step is 32 input channels.
//   fragment of lane l (q = l >> 4, li = l & 15) for k-step s:  8 consecutive k = 32 s + 8 q + 0..7 of row / column li
//   VLAD: D[channel][point] = W^T x^T:  A = weights (row = channel 16 g + li), B = inputs (column = point 16 p + li);
//         lane holds D[channel 16 g + 4 q + r][point 16 p + li], r = 0..3  -> acc[g][p][r]
//   MAX : D[point][channel] = x W (operands swapped: the max over points is register- and lane-group-wise);
//         lane holds D[point 16 p + 4 q + r][channel 16 g + li]             -> acc[g][p][r]
// The same register image serves as A or B operand, so both kernels read the same packed weights (pack.hip
// fold_pack_conv5_kernel, f16 = 0):  W5p[chunk c][g][k-step s][part (hi, lo)][lane][8 fp16] -- a (chunk, channel group)
// is contiguous: it is one LDS stage of the max-pool kernel, half a chunk buffer of the VLAD kernel.
//
// feat (VLAD) leaves as 3-byte values in accumulator order: the lane's 16 values of a chunk, value 4 t + r with t = 2 g + p,
// = feat[point 16 p + li][channel 32 c + 16 g + 4 q + r], packed into 12 dwords = three 16-byte pieces:
// [tile][chunk][piece][lane][16 B] (1 KB per wave-instruction).  The assignment GEMM reads the same accumulators as its B
// operand: for point group p, k index 8 q + e of the chunk's 32 channels <-> channel 16 (e >> 2) + 4 q + (e & 3); the
// cluster weights are packed in that k order (pack.hip pack_wc_bf16x2_kernel).
#include <type_traits>
#include "common.h"

#define C5_THREADS 512
#define C5_WAVES 8

typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4v mfma16_f16(f16x8 a, f16x8 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4v mfma16_bf16(bf16x8 a, bf16x8 b, f32x4v c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}

#ifdef C5_STAMPS
// Diagnostic build only (scripts/c5_stamps.py): per wave the shader cycles spent in each phase of the kernel, summed over the
// 32 chunks -- [prologue, DMA issue, MFMA chain, epilogue VALU + stores, vmcnt / lgkmcnt wait, barrier, final epilogue, total].
// The values go to a buffer of their own that nothing else reads; the product build contains no stamp.
__device__ unsigned int c5_stamp_buf[16384][8];
extern "C" int epc_debug_c5_stamps(void* host, size_t bytes) {
    return hipMemcpyFromSymbol(host, HIP_SYMBOL(c5_stamp_buf), bytes < sizeof(c5_stamp_buf) ? bytes : sizeof(c5_stamp_buf)) == hipSuccess ? 0 : -3;
}
#define C5_T(var) __builtin_amdgcn_sched_barrier(0); const unsigned long long var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#define C5_ADD(dst, a, b) dst += (unsigned)((b) - (a))
#else
#define C5_T(var)
#define C5_ADD(dst, a, b)
#endif
#ifndef C5_START_STAGGER
#define C5_START_STAGGER 0   // shader cycles between the start phases of first-round workgroups (0 = none)
#endif
#ifndef C5_ASYM
#define C5_ASYM 1   // 0: the lock-step schedule (every wave chain, epilogue, barrier), kept for the A/B measurement
#endif

template <int CIN>
struct C5fLds {  // VLAD kernel; offsets in floats (4 B)
    static constexpr int W5_CHUNK = 32 * CIN;              // hi + lo fragments of 32 output channels: 128 * CIN bytes
    static constexpr int WC_CHUNK = 2048;                  // cluster weights of the chunk's 32 channels: 4 groups x (hi, lo) x 1 KB
    static constexpr int OFF_W5 = 0;
    static constexpr int OFF_WC = 2 * W5_CHUNK;
    static constexpr int WC_SLOTS = 4;                     // waves 4-7 read a chunk's cluster weights one interval late
    static constexpr int OFF_B5 = OFF_WC + WC_SLOTS * WC_CHUNK;
    static constexpr int OFF_TI = OFF_B5 + 1024;           // the 1024 inverse column scales
    static constexpr int OFF_CBN = OFF_TI + 1024;          // cluster_bn scale[64], shift[64]
    // per-wave 32 x 32 f32 transpose tile (row stride 36) of the final epilogue: aliases the W5 stream buffers, dead by then
    static constexpr int OFF_T = OFF_W5;
    static constexpr int T_WAVE = 33 * 36;
    static constexpr int TOTAL = OFF_CBN + 128;
};

// The wave's 32 x CIN input block as scaled split-fp16 fragments (common.h): lane (li, q) holds, for point group p and k-step s,
// channels 32 s + 8 q .. + 7 of point 16 p + li.  ONE pass: the lane's 2 * STEPS quarter-row pieces are loaded (32 B each, the
// four q lanes of a point cover 128 contiguous bytes), the row's largest magnitude meets over the four q lanes, and every eight
// raw values are split IN PLACE into their hi and lo fragment registers -- the raw row and the fragments never coexist.
// A macro on purpose: as a function taking xh / xl by reference hipcc 7.2 keeps the raw rows and the fragments in separate
// registers (256 VGPRs + 39 spilled instead of 242).  Declares xh, xl, inv_row in the caller's scope.
#define C5_LOAD_ROWS_SPLIT_N(NP_, CIN_, cat_, g0_, active_, li_, q_)                                                                    \
    f16x8 xh[NP_][(CIN_) / 32], xl[NP_][(CIN_) / 32];                                                                                \
    float inv_row[NP_];                                                                                                           \
    {                                                                                                                            \
        float raw[NP_][(CIN_) / 32][8];                                                                                            \
        _Pragma("unroll") for (int p = 0; p < (NP_); ++p) {                                                                          \
            const float* row = (cat_) + (size_t)((active_) ? (g0_) + 16 * p + (li_) : 0) * (CIN_) + 8 * (q_);                    \
            _Pragma("unroll") for (int s = 0; s < (CIN_) / 32; ++s) {                                                            \
                const float4 a = (active_) ? ld4(row + 32 * s) : make_float4(0.f, 0.f, 0.f, 0.f);                                \
                const float4 b = (active_) ? ld4(row + 32 * s + 4) : make_float4(0.f, 0.f, 0.f, 0.f);                            \
                raw[p][s][0] = a.x, raw[p][s][1] = a.y, raw[p][s][2] = a.z, raw[p][s][3] = a.w;                                  \
                raw[p][s][4] = b.x, raw[p][s][5] = b.y, raw[p][s][6] = b.z, raw[p][s][7] = b.w;                                  \
            }                                                                                                                    \
        }                                                                                                                        \
        _Pragma("unroll") for (int p = 0; p < (NP_); ++p) {                                                                          \
            float m = 0.f;                                                                                                       \
            _Pragma("unroll") for (int s = 0; s < (CIN_) / 32; ++s)                                                              \
                _Pragma("unroll") for (int e = 0; e < 8; e += 2) m = fmaxf(fmaxf(m, fabsf(raw[p][s][e])), fabsf(raw[p][s][e + 1])); \
            m = fmaxf(m, __shfl_xor(m, 16));                                                                                     \
            m = fmaxf(m, __shfl_xor(m, 32));                                                                                     \
            float row_s;                                                                                                         \
            row_scale_pow2(m, row_s, inv_row[p]);                                                                                \
            _Pragma("unroll") for (int s = 0; s < (CIN_) / 32; ++s) split8_f16s(raw[p][s], row_s, xh[p][s], xl[p][s]);           \
        }                                                                                                                        \
    }
#define C5_LOAD_ROWS_SPLIT(CIN_, cat_, g0_, active_, li_, q_) C5_LOAD_ROWS_SPLIT_N(2, CIN_, cat_, g0_, active_, li_, q_)

// packed conv5 stage (4-byte units): [W5p CIN*1024][b5f 1024][Wcp 1024*64][cbn_s 64][cbn_t 64][tinv 1024]
template <int CIN>
__global__ __launch_bounds__(C5_THREADS) void conv5_vlad_f32_kernel(const float* __restrict__ cat, const float* __restrict__ pack,
                                                                    int total_points, float* __restrict__ feat,
                                                                    float* __restrict__ rnorm, float* __restrict__ assign,
                                                                    float* __restrict__ assign_frag, float* __restrict__ apart) {
    using L = C5fLds<CIN>;
    constexpr int STEPS = CIN / 32;
#ifdef C5_STAMPS
    unsigned int st_dma = 0, st_mfma = 0, st_epi = 0, st_wait = 0, st_bar = 0;
#endif
    C5_T(t_begin);
    static_assert(C5_WAVES * L::T_WAVE <= 2 * L::W5_CHUNK, "the transpose tiles must fit in the W5 buffers");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const float* gw5 = pack;
    const float* gb5 = pack + (size_t)CIN * 1024;
    const float* gwc = gb5 + 1024;
    const float* gcbn = gwc + 1024 * 64;
    const float* gti = gcbn + 128;

    // Weight chunks go global -> LDS directly (global_load_lds_dwordx4: each wave-instruction lands 1 KB at a wave-uniform LDS
    // base + lane * 16 = the packed fragment order).  Completion: a counted vmcnt in the chunk loop, then a raw s_barrier.
    constexpr int W5_PIECES = L::W5_CHUNK / (C5_WAVES * 256);  // 1-KB pieces per wave per chunk
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds_base = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)lds;
    const unsigned lane_off = lane * 16;
    auto stage_chunk = [&](int c, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
#pragma unroll
        for (int u = 0; u < W5_PIECES; ++u) {
            const int piece = u * C5_WAVES + wave_u;
            glds16(gw5 + (size_t)c * L::W5_CHUNK + piece * 256, lane_off,
                   lds_base + 4u * (L::OFF_W5 + buf * L::W5_CHUNK + piece * 256));
        }
        glds16(   // the chunk's cluster weights, 8 KB: one 1-KB piece from each wave
            gwc + (size_t)c * L::WC_CHUNK + wave_u * 256, lane_off,
                   lds_base + 4u * (L::OFF_WC + (c & (L::WC_SLOTS - 1)) * L::WC_CHUNK + wave_u * 256));
    };

#if C5_START_STAGGER
    // First-round workgroups start a quarter-phase apart: every workgroup's prologue is a 256-KB burst from HBM, all tiles take
    // the same time, so without this the 256 CUs load together (4.5 TB/s bursts with the matrix pipes idle) and then compute
    // together; a CU's later workgroups inherit its phase.
    if (blockIdx.x < 256 && (blockIdx.x & 3)) {
        const unsigned long long t0s = __builtin_amdgcn_s_memtime();
        const unsigned long long d = (unsigned long long)(blockIdx.x & 3) * C5_START_STAGGER;
        while (__builtin_amdgcn_s_memtime() - t0s < d) __builtin_amdgcn_s_sleep(32);
    }
#endif
    stage_chunk(0, std::integral_constant<int, 0>{});
    for (int o = tid; o < 1024; o += C5_THREADS) {
        lds[L::OFF_B5 + o] = gb5[o];
        lds[L::OFF_TI + o] = gti[o];
    }
    if (tid < 128) lds[L::OFF_CBN + tid] = gcbn[tid];

    const int g0 = (blockIdx.x * C5_WAVES + wave) * 32;
    const bool active = g0 < total_points;

    C5_LOAD_ROWS_SPLIT(CIN, cat, g0, active, li, q)

    f32x4v P[4][2];   // logits^T, P[cg][p][r] = cluster 16 cg + 4 q + r, point 16 p + li
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) P[cg][0] = P[cg][1] = f32x4v{0.f, 0.f, 0.f, 0.f};
    float ss[2] = {0.f, 0.f};
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    C5_T(t_pro);

    f32x4v acc[2][2];
    // ---- the chunk's MFMA chain: 2 x 2 accumulator tiles, STEPS k-steps, three products: 12 * STEPS MFMAs ----
    auto chain = [&](auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        const float* w5 = lds + L::OFF_W5 + buf * L::W5_CHUNK;
#pragma unroll
        for (int g = 0; g < 2; ++g) acc[g][0] = acc[g][1] = f32x4v{0.f, 0.f, 0.f, 0.f};
        // fragment reads run one (k-step, channel group) ahead of the MFMAs that consume them; the pack holds group g's k-steps
        // contiguously: fragment (s, g) is the (g * STEPS + s)-th hi / lo pair of the chunk
        constexpr int NF = 2 * STEPS;
        auto frag = [&](int f, int part) { return ldfrag16(w5 + ((((f & 1) * STEPS + (f >> 1)) * 2 + part) * 64 + lane) * 4); };
        f16x8 fa[2][2];   // [ring slot][hi, lo]
        fa[0][0] = frag(0, 0);
        fa[0][1] = frag(0, 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const int s = f >> 1, g = f & 1;
            if (f + 1 < NF) {
                fa[(f + 1) & 1][0] = frag(f + 1, 0);
                fa[(f + 1) & 1][1] = frag(f + 1, 1);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the reads AHEAD of this step's MFMAs (hipcc sinks them otherwise)
            const f16x8 wh = fa[f & 1][0], wl = fa[f & 1][1];
            acc[g][0] = mfma16_f16(wl, xh[0][s], acc[g][0]);
            acc[g][1] = mfma16_f16(wl, xh[1][s], acc[g][1]);
            acc[g][0] = mfma16_f16(wh, xl[0][s], acc[g][0]);
            acc[g][1] = mfma16_f16(wh, xl[1][s], acc[g][1]);
            acc[g][0] = mfma16_f16(wh, xh[0][s], acc[g][0]);
            acc[g][1] = mfma16_f16(wh, xh[1][s], acc[g][1]);
        }
    };
    // The epilogue of a chunk in two parts: epi_valu = VALU + stores (un-scale, ReLU, |feat|^2, feat stores, the bf16 split of
    // the accumulators into fh / fl); assign_mfma = the chunk's 24 assignment MFMAs.
    bf16x8 fh[2], fl[2];
    auto epi_valu = [&](int c) {
        // ---- epilogue: out = relu(acc * (inverse row scale * inverse column scale) + bias) ----
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const float4 bv = ld4(lds + L::OFF_B5 + 32 * c + 16 * g + 4 * q), tv = ld4(lds + L::OFF_TI + 32 * c + 16 * g + 4 * q);
            const float b4[4] = {bv.x, bv.y, bv.z, bv.w}, t4[4] = {tv.x, tv.y, tv.z, tv.w};
#pragma unroll
            for (int p = 0; p < 2; ++p)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float a = __builtin_fmaf(acc[g][p][r], inv_row[p] * t4[r], b4[r]);
                    const int vb = __float_as_int(a);
                    const float y = __int_as_float(vb > 0 ? vb : 0);      // ReLU on the bit pattern (NaN stays NaN)
                    acc[g][p][r] = y;
                    ss[p] += y * y;
                }
        }
        // feat leaves as 3-BYTE values (the upper 24 bits of the float, rounded: 16 significant bits -- its only reader, the
        // aggregate, multiplies by rnorm and splits the product into bf16 hi + lo, 16 significant bits as well): the lane's 16
        // values in order 4 t + r, t = 2 g + p, are 12 dwords = three 16-B stores, 1 KB per wave-instruction.
        if (active) {
            unsigned int w[12];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const f32x4v& v = acc[t >> 1][t & 1];
                const unsigned int a = __float_as_uint(v[0]) + 0x80u, b = __float_as_uint(v[1]) + 0x80u;
                const unsigned int cc = __float_as_uint(v[2]) + 0x80u, d = __float_as_uint(v[3]) + 0x80u;
                w[3 * t] = __builtin_amdgcn_perm(b, a, 0x05030201u);        // a.b1 a.b2 a.b3 b.b1
                w[3 * t + 1] = __builtin_amdgcn_perm(cc, b, 0x06050302u);   // b.b2 b.b3 c.b1 c.b2
                w[3 * t + 2] = __builtin_amdgcn_perm(d, cc, 0x07060503u);   // c.b3 d.b1 d.b2 d.b3
            }
            float* fdst = feat + ((size_t)(g0 >> 5) * 32 + c) * 768 + lane * 4;
#pragma unroll
            for (int pc = 0; pc < 3; ++pc)
                *reinterpret_cast<u32x4*>(fdst + pc * 256) = u32x4{w[4 * pc], w[4 * pc + 1], w[4 * pc + 2], w[4 * pc + 3]};
        }
        // assignment GEMM share of this chunk: P^T (64 clusters x 32 points) += Wc^T (64 x 32 ch) feat^T (32 ch x 32 points);
        // (feat * rn) @ Wc == (feat @ Wc) * rn, so it runs while the norm is still accumulating.  B operand of point group p:
        // k index 8 q + e <-> channel 16 (e >> 2) + 4 q + (e & 3) = the lane's own accumulators acc[e >> 2][p][e & 3].
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            const float v[8] = {acc[0][p][0], acc[0][p][1], acc[0][p][2], acc[0][p][3],
                                acc[1][p][0], acc[1][p][1], acc[1][p][2], acc[1][p][3]};
            split8(v, fh[p], fl[p]);
        }
    };
    auto assign_mfma = [&](int c) {
        const float* wc = lds + L::OFF_WC + (c & (L::WC_SLOTS - 1)) * L::WC_CHUNK;
        auto wfrag = [&](int cg, int part) { return ldfrag(wc + ((cg * 2 + part) * 64 + lane) * 4); };
        bf16x8 wq[2][2] = {{wfrag(0, 0), wfrag(0, 1)}, {wfrag(1, 0), wfrag(1, 1)}};
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            bf16x8 wn[2][2];
            if (half == 0) wn[0][0] = wfrag(2, 0), wn[0][1] = wfrag(2, 1), wn[1][0] = wfrag(3, 0), wn[1][1] = wfrag(3, 1);
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2) {
                const int cg = 2 * half + k2;
                const bf16x8 wh = wq[k2][0], wl = wq[k2][1];
                P[cg][0] = mfma16_bf16(wl, fh[0], P[cg][0]);
                P[cg][1] = mfma16_bf16(wl, fh[1], P[cg][1]);
                P[cg][0] = mfma16_bf16(wh, fl[0], P[cg][0]);
                P[cg][1] = mfma16_bf16(wh, fl[1], P[cg][1]);
                P[cg][0] = mfma16_bf16(wh, fh[0], P[cg][0]);
                P[cg][1] = mfma16_bf16(wh, fh[1], P[cg][1]);
            }
            if (half == 0) wq[0][0] = wn[0][0], wq[0][1] = wn[0][1], wq[1][0] = wn[1][0], wq[1][1] = wn[1][1];
        }
    };
    // (measured: the assignment MFMAs deferred to the head of the wave's next chain, so that each phase is pure VALU or pure
    // MFMA: 0.456-0.468 vs 0.441-0.453 ms with the epilogue in one piece -- the pure-VALU part still takes 1 400 cycles per
    // chunk beside the partner's chain, and the longer chain phase of the late waves lengthens the interval)
    auto epilogue = [&](int c) {
        epi_valu(c);
        assign_mfma(c);
    };
    // One interval = everything between two chunk barriers.  `late` waves (4-7) run the epilogue of the PREVIOUS chunk before this
    // chunk's chain (see Schedule at the top of the file).
    const bool late = C5_ASYM && wave_u >= 4;
    auto interval = [&](int c, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        C5_T(t0);
        if (c + 1 < 32) stage_chunk(c + 1, std::integral_constant<int, buf ^ 1>{});
        C5_T(t1);
        if (late && c > 0) epilogue(c - 1);
        C5_T(t2);
        chain(bufc);
        C5_T(t3);
        if (!late) epilogue(c);
        C5_T(t4);
        // the next chunk's LDS-DMA pieces are the OLDEST outstanding vector-memory operations of this wave; the 3 feat stores
        // issued after them may stay in flight (vmcnt counts in issue order).  Waves without stores drain everything.
        if (active && (!late || c > 0))
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        C5_T(t5);
        __builtin_amdgcn_s_barrier();
        C5_T(t6);
        C5_ADD(st_dma, t0, t1);
        C5_ADD(st_epi, t1, t2);
        C5_ADD(st_mfma, t2, t3);
        C5_ADD(st_epi, t3, t4);
        C5_ADD(st_wait, t4, t5);
        C5_ADD(st_bar, t5, t6);
    };
    for (int c = 0; c < 32; c += 2) {
        interval(c, std::integral_constant<int, 0>{});
        interval(c + 1, std::integral_constant<int, 1>{});
    }
    if (late) epilogue(31);
    C5_T(t_loop);

    if (active) {
        // per-point inverse norm (models/epc-net.py:148): the point's 1024 squares sit in its four q lanes
        float rn[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float s_ = ss[p];
            s_ += __shfl_xor(s_, 16);
            s_ += __shfl_xor(s_, 32);
            rn[p] = 1.0f / sqrtf(fmaxf(s_, 1e-12f));
        }
        // cluster_bn (folded: logit * s + t) then softmax over the 64 clusters (16 here, 48 in the other three q lanes)
        const float* cs = lds + L::OFF_CBN;
        const float* ct = cs + 64;
        float inv_sum[2];
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float mx = -INFINITY;
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 16 * cg + 4 * q + r;
                    const float v = (P[cg][p][r] * rn[p]) * cs[k] + ct[k];
                    P[cg][p][r] = v;
                    mx = fmaxf(mx, v);
                }
            mx = fmaxf(mx, __shfl_xor(mx, 16));
            mx = fmaxf(mx, __shfl_xor(mx, 32));
            float sum = 0.f;
#pragma unroll
            for (int cg = 0; cg < 4; ++cg)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float e = expf(P[cg][p][r] - mx);
                    P[cg][p][r] = e;
                    sum += e;
                }
            sum += __shfl_xor(sum, 16);
            sum += __shfl_xor(sum, 32);
            inv_sum[p] = sum;
        }
        if (assign) {  // the f32 point-major copy is for op-level callers; the fused pipeline passes NULL (67 MB less HBM)
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                float* arow = assign + (size_t)(g0 + 16 * p + li) * 64 + 4 * q;
#pragma unroll
                for (int cg = 0; cg < 4; ++cg)
                    st4(arow + 16 * cg, make_float4(P[cg][p][0] / inv_sum[p], P[cg][p][1] / inv_sum[p], P[cg][p][2] / inv_sum[p],
                                                    P[cg][p][3] / inv_sum[p]));
            }
        }
        if (q == 0) {
            rnorm[g0 + li] = rn[0];
            rnorm[g0 + 16 + li] = rn[1];
        }
        // the assignments as bf16 hi + lo B fragments of the aggregate GEMM (cluster -> lane, 8 consecutive points -> fragment:
        // [tile][cluster tile t][k-step][part][lane][8], a[32 g + 16 ks + 8 (l >> 5) + e][32 t + (l & 31)]) and the tile's partial
        // a_sum (loupe.py:276), through a per-wave [cluster][point] LDS tile.  rnorm is applied on the feature side there.
        float* T = lds + L::OFF_T + wave * L::T_WAVE;
        float* fdst = assign_frag + (size_t)(g0 >> 5) * 2048 + lane * 4;
        const int j = lane & 31, h = lane >> 5;
        float asum[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
                for (int p = 0; p < 2; ++p)
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[(16 * k2 + 4 * q + r) * 36 + 16 * p + li] = P[2 * t + k2][p][r] / inv_sum[p];
            float s_ = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float v[8];
                const float4 a0 = ld4(T + j * 36 + 16 * ks + 8 * h), a1 = ld4(T + j * 36 + 16 * ks + 8 * h + 4);
                v[0] = a0.x, v[1] = a0.y, v[2] = a0.z, v[3] = a0.w, v[4] = a1.x, v[5] = a1.y, v[6] = a1.z, v[7] = a1.w;
#pragma unroll
                for (int e = 0; e < 8; ++e) s_ += v[e];
                bf16x8 ah, al;
                split8(v, ah, al);
                *reinterpret_cast<u32x4*>(fdst + ((t * 2 + ks) * 2 + 0) * 256) = __builtin_bit_cast(u32x4, ah);
                *reinterpret_cast<u32x4*>(fdst + ((t * 2 + ks) * 2 + 1) * 256) = __builtin_bit_cast(u32x4, al);
            }
            asum[t] = s_ + __shfl_xor(s_, 32);
        }
        if (h == 0) {
            apart[(size_t)(g0 >> 5) * 64 + j] = asum[0];
            apart[(size_t)(g0 >> 5) * 64 + 32 + j] = asum[1];
        }
    }
#ifdef C5_STAMPS
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        C5_T(t_end);
        const unsigned wv = blockIdx.x * C5_WAVES + wave;
        if (lane == 0 && wv < 16384) {
            unsigned int* o = c5_stamp_buf[wv];
            o[0] = (unsigned)(t_pro - t_begin), o[1] = st_dma, o[2] = st_mfma, o[3] = st_epi, o[4] = st_wait, o[5] = st_bar;
            o[6] = (unsigned)(t_end - t_loop), o[7] = (unsigned)(t_end - t_begin);
        }
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------------------
// conv5 + global max-pool (EPC-Net-L).  Operands swapped (D[point][channel]: lane holds points 16 p + 4 q + r of channel
// 16 g + li), so the max over a tile's points is over registers, the two point groups and the four q lanes.
//
// Geometry: 256 threads = 4 waves, ONE wave per SIMD, one 32-point tile per wave; at CIN = 128 the kernel needs ~122 VGPRs
// and 30 KB of LDS, so FOUR workgroups share a CU and a SIMD's four waves belong to four different workgroups: no barrier
// ties them, and one's prologue, epilogue or barrier wait runs beside another's MFMA chain (measured at batch 256: 0.604 ms
// against 0.69 ms for the 8-wave form above with this epilogue).  W5 streams in stages of ONE channel group (16 output
// channels, 64 * CIN bytes, contiguous in the pack), double-buffered, one barrier per stage.  When the workgroup's four tiles
// lie in one cloud the per-wave maxima meet in LDS (one slab per stage parity) and the workgroup's 1024 maxima leave as
// 256-B atomic wave-instructions at the very end; otherwise each wave issues its own atomics.  Values are >= 0 (ReLU) and
// pooled starts at 0, so unsigned-integer max on the bit patterns is the float max and 0 is the neutral element.
// ---------------------------------------------------------------------------------------------------------------------------
#define C5M_THREADS 256
#define C5M_WAVES 4
template <int CIN>
struct C5mLds {  // offsets in floats
    static constexpr int W5_STAGE = 16 * CIN;
    static constexpr int OFF_W5 = 0;                                   // two stage buffers: buffer g holds channel group g
    static constexpr int OFF_B5 = 2 * W5_STAGE;
    static constexpr int OFF_TI = OFF_B5 + 1024;
    static constexpr int OFF_MAX = OFF_TI + 1024;                      // 2 x (4 waves x 16) per-wave maxima + 1024 workgroup maxima
    static constexpr int OFF_IS = OFF_MAX + 2 * C5M_WAVES * 16 + 1024; // per wave the 32 inverse row scales of its tile
    static constexpr int TOTAL = OFF_IS + C5M_WAVES * 32;
};

// packed conv5 stage (4-byte units): [W5p CIN*1024][b5f 1024][tinv 1024]
template <int CIN>
__global__ __launch_bounds__(C5M_THREADS, 4) void conv5_max_f32_kernel(const float* __restrict__ cat, const float* __restrict__ pack,
                                                                       int total_points, int n, float* __restrict__ pooled) {
    using L = C5mLds<CIN>;
    constexpr int STEPS = CIN / 32;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, q = lane >> 4;
    const float* gw5 = pack;
    const float* gb5 = pack + (size_t)CIN * 1024;
    const float* gti = gb5 + 1024;
    constexpr int W5_PIECES = L::W5_STAGE / (C5M_WAVES * 256);         // 1-KB LDS-DMA pieces per wave per stage
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const unsigned lds_base = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)lds;
    const unsigned lane_off = lane * 16;
    auto stage_w5 = [&](int c, auto gc) {      // channel group g of chunk c -> stage buffer g
        constexpr int g = decltype(gc)::value;
#pragma unroll
        for (int u = 0; u < W5_PIECES; ++u) {
            const int piece = u * C5M_WAVES + wave_u;
            glds16(gw5 + (size_t)c * (2 * L::W5_STAGE) + g * L::W5_STAGE + piece * 256, lane_off,
                   lds_base + 4u * (L::OFF_W5 + g * L::W5_STAGE + piece * 256));
        }
    };
    stage_w5(0, std::integral_constant<int, 0>{});
    for (int o = tid; o < 1024; o += C5M_THREADS) {
        lds[L::OFF_B5 + o] = gb5[o];
        lds[L::OFF_TI + o] = gti[o];
    }
    const int g0 = (blockIdx.x * C5M_WAVES + wave) * 32;
    const bool active = g0 < total_points;
    const bool wg_one_cloud = n % (C5M_WAVES * 32) == 0;               // the workgroup's four tiles share a cloud
    C5_LOAD_ROWS_SPLIT(CIN, cat, g0, active, li, q)
    // register r of acc[p] belongs to point 16 p + 4 q + r, whose inverse row scale lives in the lane that loaded that point:
    // through a wave-private LDS row, read back once
    if (q == 0) {
        lds[L::OFF_IS + wave * 32 + li] = inv_row[0];
        lds[L::OFF_IS + wave * 32 + 16 + li] = inv_row[1];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float isr[2][4];
#pragma unroll
    for (int p = 0; p < 2; ++p) {
        const float4 v = ld4(lds + L::OFF_IS + wave * 32 + 16 * p + 4 * q);
        isr[p][0] = v.x, isr[p][1] = v.y, isr[p][2] = v.z, isr[p][3] = v.w;
    }
    // fold the per-wave maxima of stage (c, g) (written before the barrier that ended it) into the workgroup's
    auto fold_stage_max = [&](int c, int g) {
        if (wg_one_cloud && tid < 16) {
            const float* red = lds + L::OFF_MAX + g * (C5M_WAVES * 16) + tid;
            float m = red[0];
#pragma unroll
            for (int w = 1; w < C5M_WAVES; ++w) m = fmaxf(m, red[16 * w]);
            lds[L::OFF_MAX + 2 * C5M_WAVES * 16 + 32 * c + 16 * g + tid] = m;
        }
    };
    // One stage = channel group g of chunk c: 6 * STEPS MFMAs from stage buffer g while the NEXT stage's pieces land in the other
    // buffer (last read one stage ago; the barrier that ended that stage orders it), then the stage's maxima.
    auto do_stage = [&](int c, auto gc) {
        constexpr int g = decltype(gc)::value;
        if constexpr (g == 0) {
            if (c > 0) fold_stage_max(c - 1, 1);
            stage_w5(c, std::integral_constant<int, 1>{});
        } else {
            fold_stage_max(c, 0);
            if (c + 1 < 32) stage_w5(c + 1, std::integral_constant<int, 0>{});
        }
        const float* w5 = lds + L::OFF_W5 + g * L::W5_STAGE;
        f32x4v acc[2] = {f32x4v{0.f, 0.f, 0.f, 0.f}, f32x4v{0.f, 0.f, 0.f, 0.f}};
        f16x8 fa[2][2];   // fragment reads run one k-step ahead of their MFMAs: [ring slot][hi, lo]
        fa[0][0] = ldfrag16(w5 + (0 * 64 + lane) * 4);
        fa[0][1] = ldfrag16(w5 + (1 * 64 + lane) * 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            if (s + 1 < STEPS) {
                fa[(s + 1) & 1][0] = ldfrag16(w5 + (((s + 1) * 2 + 0) * 64 + lane) * 4);
                fa[(s + 1) & 1][1] = ldfrag16(w5 + (((s + 1) * 2 + 1) * 64 + lane) * 4);
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the reads AHEAD of this step's MFMAs (hipcc sinks them otherwise)
            const f16x8 wh = fa[s & 1][0], wl = fa[s & 1][1];
            acc[0] = mfma16_f16(xh[0][s], wl, acc[0]);
            acc[1] = mfma16_f16(xh[1][s], wl, acc[1]);
            acc[0] = mfma16_f16(xl[0][s], wh, acc[0]);
            acc[1] = mfma16_f16(xl[1][s], wh, acc[1]);
            acc[0] = mfma16_f16(xh[0][s], wh, acc[0]);
            acc[1] = mfma16_f16(xh[1][s], wh, acc[1]);
        }
        // out = relu(acc * (inverse row scale * inverse column scale) + bias), then the maximum over the tile's points
        const float ti = lds[L::OFF_TI + 32 * c + 16 * g + li], bv = lds[L::OFF_B5 + 32 * c + 16 * g + li];
        float m = 0.f;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a = __builtin_fmaf(acc[p][r], isr[p][r] * ti, bv);
                const int vb = __float_as_int(a);
                m = fmaxf(m, __int_as_float(vb > 0 ? vb : 0));      // ReLU on the bit pattern
            }
        m = fmaxf(m, __shfl_xor(m, 16));
        m = fmaxf(m, __shfl_xor(m, 32));
        if (!active) m = 0.f;
        if (wg_one_cloud) {
            if (q == 0) lds[L::OFF_MAX + g * (C5M_WAVES * 16) + wave * 16 + li] = m;
        } else if (active && q == 0) {
            atomicMax(reinterpret_cast<unsigned int*>(pooled + (size_t)(g0 / n) * 1024 + 32 * c + 16 * g + li), __float_as_uint(m));
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next stage's pieces have landed (and this wave's atomics have left)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    for (int c = 0; c < 32; ++c) {
        do_stage(c, std::integral_constant<int, 0>{});
        do_stage(c, std::integral_constant<int, 1>{});
    }
    if (wg_one_cloud) {
        fold_stage_max(31, 1);
        __syncthreads();
        unsigned int* dst = reinterpret_cast<unsigned int*>(pooled + (size_t)((blockIdx.x * C5M_WAVES * 32) / n) * 1024);
        for (int o = tid; o < 1024; o += C5M_THREADS) atomicMax(dst + o, __float_as_uint(lds[L::OFF_MAX + 2 * C5M_WAVES * 16 + o]));
    }
}

static int c5_set_lds(const void* fn, size_t lds_bytes, const char* who) {
    hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
        epc_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
        return EPC_EHIP;
    }
    return EPC_OK;
}

extern "C" int epc_conv5_assign_f32_fwd(const float* cat, int cin, const void* packed_conv5, int num_points_total,
                                        void* feat_frag, float* rnorm, float* assign, void* assign_frag, float* apart,
                                        void* stream) {
    EPC_CHECK_ARG(cat && packed_conv5 && feat_frag && rnorm && assign_frag && apart, "null pointer");
    EPC_CHECK_ARG(cin == 256, "EPC-Net conv5 takes the 256-channel concat (models/epc-net.py:134)");
    EPC_CHECK_ARG(num_points_total >= 0 && num_points_total % 32 == 0, "point count must be a multiple of 32");
    if (num_points_total == 0) return EPC_OK;
    const size_t lds_bytes = C5fLds<256>::TOTAL * sizeof(float);
    if (int rc = c5_set_lds(reinterpret_cast<const void*>(conv5_vlad_f32_kernel<256>), lds_bytes, __func__)) return rc;
    const unsigned blocks = (unsigned)((num_points_total + C5_WAVES * 32 - 1) / (C5_WAVES * 32));
    hipLaunchKernelGGL((conv5_vlad_f32_kernel<256>), dim3(blocks), dim3(C5_THREADS), lds_bytes, (hipStream_t)stream, cat,
                       (const float*)packed_conv5, num_points_total, (float*)feat_frag, rnorm, assign, (float*)assign_frag, apart);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_conv5_maxpool_fwd(const float* cat, int cin, const void* packed_conv5, int num_clouds, int n,
                                     float* pooled, void* stream) {
    EPC_CHECK_ARG(cat && packed_conv5 && pooled, "null pointer");
    EPC_CHECK_ARG(cin == 128, "EPC-Net-L conv5 takes the 128-channel concat (models/epc-net-l.py:84)");
    EPC_CHECK_ARG(n > 0 && n % 32 == 0 && num_clouds >= 0, "num_points must be a multiple of 32");
    if (num_clouds == 0) return EPC_OK;
    const long total = (long)num_clouds * n;
    hipError_t e = hipMemsetAsync(pooled, 0, (size_t)num_clouds * 1024 * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) {
        epc_set_error("epc_conv5_maxpool_fwd: hipMemsetAsync: %s", hipGetErrorString(e));
        return EPC_EHIP;
    }
    const size_t lds_bytes = C5mLds<128>::TOTAL * sizeof(float);
    if (int rc = c5_set_lds(reinterpret_cast<const void*>(conv5_max_f32_kernel<128>), lds_bytes, __func__)) return rc;
    const unsigned blocks = (unsigned)((total + C5M_WAVES * 32 - 1) / (C5M_WAVES * 32));
    hipLaunchKernelGGL((conv5_max_f32_kernel<128>), dim3(blocks), dim3(C5M_THREADS), lds_bytes, (hipStream_t)stream, cat,
                       (const float*)packed_conv5, (int)total, n, pooled);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}
