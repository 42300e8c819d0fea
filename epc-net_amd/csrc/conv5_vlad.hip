// EPC_PRECISION_FAST form of conv5 (+BN+ReLU) + per-point L2 norm + soft assignment (EPC-Net: models/epc-net.py:136-139,
// 147-148 + loupe.py:249-272), and the two VLAD aggregate kernels (fast and f32-equivalent).  The f32-equivalent conv5 kernels
// (EPC-Net and EPC-Net-L) live in conv5_f32.hip.
//
// Arithmetic ("f16 + f6"): ONE fp16 value per activation; weights as fp16 hi + MX-fp6 lo of W * 2^8: per 32-channel chunk 16
// v_mfma_f32_32x32x16_f16 (hi) + 4 v_mfma_scale_f32_32x32x64_f8f6f4 (lo), f32 accumulation.
//   feat^T chunk (32 ch x 32 pts) = W5f^T X^T; epilogue per chunk: ReLU, 2^-8, |feat|^2 partial, rounding to fp16: that
//   fragment is stored (feat, accumulator-fragment order) AND is the B operand of P^T (64 clusters x 32 pts) += Wc^T feat^T
//   ((feat*rn) @ Wc == (feat @ Wc) * rn, so the assignment GEMM runs while the norm is still being accumulated).
//   Final: rn = rsqrt(max(|feat|^2,1e-12)), cluster_bn (folded), softmax over 64; assign (f32) + fp16 fragments.
//
// Geometry: 512 threads = 8 waves, one 32-point tile per wave; the wave's input row block (32 pts x 256) lives in
// registers as B fragments for all 32 output chunks; W5 (hi + lo: 3 B per weight, 768 KB) streams through a
// double-buffered LDS chunk shared by the 8 waves (LDS-DMA, one barrier per chunk).
#include <type_traits>
#include "common.h"

#define C5_THREADS 512
#define C5_WAVES 8

// Per 32-channel chunk the weights are [fp16 hi fragments: CIN/16 k-steps x 1 KB][MX fp6 lo fragments: CIN/64 k-steps x
// (1 KB + 512 B), then 256 B of block scales; pack.hip pack_conv5_lo6_kernel] inside 96*CIN bytes.
template <int CIN>
struct C5Lds {  // offsets in floats (4 B)
    static constexpr int W5_CHUNK = 24 * CIN;
    static constexpr int W5_LO8 = 16 * CIN;    // float offset of the fp6 lo fragments inside a chunk
    static constexpr int LO6_KS = 384;         // floats per k-step of lo fragments (1 KB of 16-B pieces + 512 B of 8-B pieces)
    static constexpr int W5_LOSC = W5_LO8 + (CIN / 64) * LO6_KS;   // the block-scale dwords (one per lane)
    static constexpr int WC_CHUNK = 1024;      // cluster-weight chunk: 32 ch x 64 clusters x 2 B (ONE fp16 per cluster weight, see the epilogue)
    static constexpr int OFF_W5 = 0;
    static constexpr int WC_SLOTS = 2;         // chunk parity
    static constexpr int OFF_WC = 2 * W5_CHUNK;
    static constexpr int OFF_B5 = OFF_WC + WC_SLOTS * WC_CHUNK;
    static constexpr int OFF_CBN = OFF_B5 + 1024;
    // per-wave 32 x 32 f32 transpose tile (row stride 36) of the FINAL epilogue: it aliases the W5 stream buffers, which
    // are dead by then (a barrier separates the last chunk from the first tile write) -- 38 KB less LDS, so that a
    // kNN workgroup (65 KB) of another stream can share the CU.
    static constexpr int OFF_T = OFF_W5;
    static constexpr int T_WAVE = 33 * 36;
    static constexpr int TOTAL = OFF_CBN + 128;
};

// packed conv5 stage (4-byte units): [W5p: 24*CIN*32 floats][b5f 1024][Wcp 1024*32 (fp16)][cbn_s 64][cbn_t 64]
// CAT16: the input rows are fp16 (the fast block chain's out16) instead of f32.
template <int CIN, bool CAT16>
__global__ __launch_bounds__(C5_THREADS) void conv5_kernel(const float* __restrict__ cat,
                                                           const float* __restrict__ pack, int total_points,
                                                           int n, float* __restrict__ feat,
                                                           float* __restrict__ rnorm,
                                                           float* __restrict__ assign,
                                                           float* __restrict__ assign_frag,
                                                           float* __restrict__ apart,
                                                           int32_t* __restrict__ status) {
    using L = C5Lds<CIN>;
    constexpr int STEPS = CIN / 16;
    static_assert(8 * L::T_WAVE <= 2 * L::W5_CHUNK, "the transpose tiles must fit in the W5 buffers");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const float* gw5 = pack;
    const float* gb5 = pack + (size_t)CIN * 1024;
    const float* gwc = gb5 + 1024;
    const float* gcbn = gwc + 1024 * 32;   // past 1024 x 64 fp16

    // Weight chunks go global -> LDS directly (global_load_lds_dwordx4: each wave-instruction writes 1 KB at a
    // wave-uniform LDS base + lane*16, which is exactly the packed fragment order), so no VGPRs are spent on staging
    // and the loads of chunk c+1 stay in flight under chunk c's MFMAs.  Completion is a counted vmcnt (the chunk's
    // 2 feat stores are younger and may stay in flight) followed by a raw s_barrier.
    constexpr int W5_PIECES = L::W5_CHUNK / (C5_WAVES * 256);  // 1-KB pieces per wave per chunk (256 floats each)
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
        if (wave_u < L::WC_CHUNK / 256)   // 4 KB per chunk: one 1-KB piece from each of four waves
            glds16(gwc + (size_t)c * L::WC_CHUNK + wave_u * 256, lane_off,
                   lds_base + 4u * (L::OFF_WC + (c & (L::WC_SLOTS - 1)) * L::WC_CHUNK + wave_u * 256));
    };

    stage_chunk(0, std::integral_constant<int, 0>{});
    for (int o = tid; o < 1024; o += C5_THREADS) lds[L::OFF_B5 + o] = gb5[o];
    if (tid < 128) lds[L::OFF_CBN + tid] = gcbn[tid];

    const int g0 = (blockIdx.x * C5_WAVES + wave) * 32;
    const bool active = g0 < total_points;

    // this lane's B fragments: point j, k-step s covers input channels 16s + 8h .. +7.  ONE fp16 value per input, the weights
    // carry the hi + lo split (W5_SCALE comment in common.h) -- the input rounding averages out over the cloud's points in the
    // aggregation.
    constexpr float kDescale = 1.0f / W5_SCALE;
    f16x8 xf[STEPS];
    // the same inputs as MX fp6 (B operand of the lo-term MFMA): per 64-wide k-step the lane's 32 consecutive channels
    // 64ks + 32h .. +31 in six dwords, their block scale in byte ks of xsc
    i32x6 x6[CIN / 64];
    int xsc = 0;
    auto to_fp6 = [&](f16x32 v, int ks) {
        typedef unsigned short u16x2 __attribute__((ext_vector_type(2)));
        typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));
        const u32x16 w = __builtin_bit_cast(u32x16, v);
        u16x2 m2 = {0, 0};
#pragma unroll
        for (int i = 0; i < 16; ++i)   // |fp16| orders like its bit pattern: packed 16-bit max
            m2 = __builtin_elementwise_max(m2, __builtin_bit_cast(u16x2, w[i] & 0x7fff7fffu));
        const unsigned short mb = m2[0] > m2[1] ? m2[0] : m2[1];
        const float m = (float)__builtin_bit_cast(_Float16, mb);
        const int e = fp6_block_exponent(m, -12, 14);          // 2^-e stays an fp16 normal: the scaling below is exact
        const _Float16 down = (_Float16)exp2i(-e);
        x6[ks] = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(v * down, 1.0f);
        xsc |= (127 + e) << (8 * ks);
    };
    if constexpr (CAT16) {  // fp16 rows (the blocks' out16): the 16 B a lane reads ARE its fragment
        const unsigned short* row = reinterpret_cast<const unsigned short*>(cat) + (size_t)(active ? g0 + j : 0) * CIN + 8 * h;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            u32x4 w = *reinterpret_cast<const u32x4*>(row + 16 * s);
            if (!active) w = u32x4{0u, 0u, 0u, 0u};
            xf[s] = __builtin_bit_cast(f16x8, w);
        }
#pragma unroll
        for (int ks = 0; ks < CIN / 64; ++ks) {   // channels 64ks + 32h .. +31 of the lane's point
            typedef unsigned int u32x16 __attribute__((ext_vector_type(16)));
            u32x16 w16;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {
                u32x4 w = *reinterpret_cast<const u32x4*>(row + 64 * ks + 24 * h + 8 * q4);   // row already holds +8h
                if (!active) w = u32x4{0u, 0u, 0u, 0u};
                w16[4 * q4] = w[0], w16[4 * q4 + 1] = w[1], w16[4 * q4 + 2] = w[2], w16[4 * q4 + 3] = w[3];
            }
            to_fp6(__builtin_bit_cast(f16x32, w16), ks);
        }
    } else {
        const float* row = cat + (size_t)(active ? g0 + j : 0) * CIN + 8 * h;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            const float4 a = active ? ld4(row + 16 * s) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 b = active ? ld4(row + 16 * s + 4) : make_float4(0.f, 0.f, 0.f, 0.f);
            xf[s][0] = (_Float16)a.x, xf[s][1] = (_Float16)a.y, xf[s][2] = (_Float16)a.z, xf[s][3] = (_Float16)a.w;
            xf[s][4] = (_Float16)b.x, xf[s][5] = (_Float16)b.y, xf[s][6] = (_Float16)b.z, xf[s][7] = (_Float16)b.w;
        }
        const float* row0 = row - 8 * h;   // channel 0 of the lane's point
#pragma unroll
        for (int ks = 0; ks < CIN / 64; ++ks) {
            f16x32 v;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) {
                const float4 a = active ? ld4(row0 + 64 * ks + 32 * h + 4 * w8) : make_float4(0.f, 0.f, 0.f, 0.f);
                // (through fp16: the fp6 copy must describe the same input the hi term sees)
                v[4 * w8] = (_Float16)a.x, v[4 * w8 + 1] = (_Float16)a.y, v[4 * w8 + 2] = (_Float16)a.z, v[4 * w8 + 3] = (_Float16)a.w;
            }
            to_fp6(v, ks);
        }
    }

    f32x16 P[2];
#pragma unroll
    for (int r = 0; r < 16; ++r) P[0][r] = P[1][r] = 0.f;
    float ss = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // A chunk is three pieces: its MFMA chain (chunk_mfma: fragments from the LDS buffer -- the buffer index is a compile-time
    // constant so that the compiler can see that the DMA destination, the other buffer, never aliases the fragments being read;
    // otherwise it drains vmcnt before every ds_read -- accumulator left in `acc`), its epilogue (chunk_epi: ReLU, |feat|^2, feat
    // stores, the assignment GEMM's share) and the wait that lets the next chunk's weights land (chunk_wait).
    f32x16 acc;
    auto chunk_mfma = [&](int c, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
        const float* w5 = lds + L::OFF_W5 + buf * L::W5_CHUNK;
        const float* b = lds + L::OFF_B5 + 32 * c;      // the (scaled) bias initialises the accumulator
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bv = ld4(b + 8 * g + 4 * h);
            acc[4 * g] = bv.x;
            acc[4 * g + 1] = bv.y;
            acc[4 * g + 2] = bv.z;
            acc[4 * g + 3] = bv.w;
        }
        // lo term first (small): W_lo and the inputs as MX fp6, K = 64 per instruction in 8 passes (the fp16 K = 16
        // instruction takes 8 as well) -- the lo term is a 2^-11 correction, so 4 significant bits on each side keep it
        // to 2^-15 of the product.  Block scales: byte ks of the lanes' scale dwords (op_sel).
        const float* wl = w5 + L::W5_LO8;
#ifndef CONV5_ABL_NO_LO
        const int wsc = __float_as_int(w5[L::W5_LOSC + lane]);
        auto lo_step = [&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            if constexpr (ks < CIN / 64) {
                const u32x4 l0 = *reinterpret_cast<const u32x4*>(wl + ks * L::LO6_KS + lane * 4);
                const uint2 l1 = *reinterpret_cast<const uint2*>(wl + ks * L::LO6_KS + 256 + lane * 2);
                const i32x8 a = {(int)l0[0], (int)l0[1], (int)l0[2], (int)l0[3], (int)l1.x, (int)l1.y, 0, 0};
                const i32x8 b6 = {x6[ks][0], x6[ks][1], x6[ks][2], x6[ks][3], x6[ks][4], x6[ks][5], 0, 0};
                acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b6, acc, 2, 2, ks, wsc, ks, xsc);
            }
        };
        lo_step(std::integral_constant<int, 0>{});
        lo_step(std::integral_constant<int, 1>{});
        lo_step(std::integral_constant<int, 2>{});
        lo_step(std::integral_constant<int, 3>{});
        static_assert(CIN / 64 <= 4, "one scale dword holds four block scales");
#endif
        // fragment reads run C5_PF k-steps ahead of the MFMAs that consume them: eight waves share the LDS port, so a
        // read returns after ~8 other 1-KB reads (64+ cycles) while a k-step's MFMA takes 32 -- with one read in flight
        // per wave the port idles
#ifndef C5_PF
#define C5_PF 1   // measured 1, 2, 3, 4, 6: 0.258-0.260 ms alike, so the shallowest (fewest registers) stays
#endif
        constexpr int PF = C5_PF < STEPS ? C5_PF : STEPS - 1;
        constexpr int RING = PF + 1;
        f16x8 fa[RING];
#pragma unroll
        for (int s = 0; s < PF; ++s) fa[s] = ldfrag16(w5 + (s * 64 + lane) * 4);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            if (s + PF < STEPS) fa[(s + PF) % RING] = ldfrag16(w5 + ((s + PF) * 64 + lane) * 4);
            __builtin_amdgcn_sched_barrier(0);  // keep the reads AHEAD of this step's MFMAs (hipcc sinks them otherwise)
            acc = mfma_f16(fa[s % RING], xf[s], acc);
        }
    };
    auto chunk_epi = [&](int c) {
        // ReLU and the 2^-8 that removes W5_SCALE (bias and weights are packed scaled): max on the bit pattern
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float a = acc[r];  // (a scalar copy: __builtin_bit_cast on a vector ELEMENT reads element 0 with hipcc 7.2)
            const int vb = __float_as_int(a);
            acc[r] = __int_as_float(vb > 0 ? vb : 0) * kDescale;
        }
#ifdef C5_ABL_NOEPI
        constexpr bool kEpi = false;
        asm volatile("" :: "v"(acc));
#else
        constexpr bool kEpi = true;
#endif
        if constexpr (kEpi) {
            const float* wc = lds + L::OFF_WC + (c & (L::WC_SLOTS - 1)) * L::WC_CHUNK;
            // The cluster weights are ONE fp16 value each (x 2^8): the soft assignment only enters through a softmax whose
            // logits tolerate a 2^-12 weight rounding -- emulated descriptor effect 5e-9 on top of the 9.4e-7 of the
            // two-product conv5 -- so the lo product of the hi+lo form is not spent here.
            auto wfrag = [&](int sp, int t) { return ldfrag16(wc + ((sp * 2 + t) * 64 + lane) * 4); };
            // cluster-weight fragments of k-step 0: issued now, they land under the VALU work below
            f16x8 wf[2];
            wf[0] = wfrag(0, 0), wf[1] = wfrag(0, 1);
#pragma unroll
            for (int r = 0; r < 16; ++r) ss += acc[r] * acc[r];
            // accumulators -> fp16 B fragments: k-step s' = registers 8s' .. 8s'+7 (k order: common.h, Wcp).  The same
            // fragment is the assignment GEMM's operand and what is stored.
#pragma unroll
            for (int sp = 0; sp < 2; ++sp) {
                f16x8 fs;
#pragma unroll
                for (int q = 0; q < 8; ++q) fs[q] = (_Float16)acc[8 * sp + q];
#ifndef C5_ABL_NOSTORE
                // feat leaves the kernel as ONE fp16 fragment per accumulator half (accumulator order: lane = point,
                // element q = channel 32c + 16sp + 8(q>>2) + 4h + (q&3)): 1 KB per wave-instruction, 2 B per value; the
                // aggregate kernel scales by rnorm and transposes.  (fp16 keeps 11 significant bits of a value that is
                // then averaged over the cloud's points: measured descriptor effect 7e-7)
                if (active) {
                    float* fdst = feat + ((size_t)(g0 >> 5) * 32 + c) * 512 + lane * 4;
                    *reinterpret_cast<u32x4*>(fdst + sp * 256) = __builtin_bit_cast(u32x4, fs);
                }
#endif
                f16x8 wn[2];
                if (sp == 0) wn[0] = wfrag(1, 0), wn[1] = wfrag(1, 1);
#ifndef C5_ABL_NOASSIGN
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    P[t] = mfma_f16(wf[t], fs, P[t]);
                }
#else
                asm volatile("" :: "v"(fs), "v"(wf[0]), "v"(wf[1]));
#endif
                if (sp == 0) wf[0] = wn[0], wf[1] = wn[1];
            }
        }
    };
    // the next chunk's LDS-DMA pieces are the OLDEST outstanding vector-memory operations of this wave; the 2 feat stores
    // issued after them may stay in flight (vmcnt counts in issue order).  Waves without stores (tail of the grid) drain
    // everything.
    auto chunk_wait = [&]() {
#ifdef C5_ABL_NOSTORE
        if (false)
#else
        if (active)
#endif
            asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    // (measured on this kernel and not adopted: the second wave of each SIMD deferring its epilogue across the chunk barrier,
    // 0.32 vs 0.31 ms; a one-wave-per-SIMD, two-tiles-per-wave form with a software-pipelined epilogue, 0.50-0.52 vs 0.48 ms)
    auto do_chunk = [&](int c, auto bufc) {
        constexpr int buf = decltype(bufc)::value;
#ifndef C5_ABL_NODMA
        if (c + 1 < 32) stage_chunk(c + 1, std::integral_constant<int, buf ^ 1>{});
#endif
        chunk_mfma(c, bufc);
        chunk_epi(c);
        chunk_wait();
#ifndef C5_ABL_NOBARRIER
        __builtin_amdgcn_s_barrier();
#endif
    };
    for (int c = 0; c < 32; c += 2) {
        do_chunk(c, std::integral_constant<int, 0>{});
        do_chunk(c + 1, std::integral_constant<int, 1>{});
    }

    if (active) {
        // per-point inverse norm (models/epc-net.py:148)
        ss += __shfl_xor(ss, 32);
        {
            // fp16 range guard: no element of the row exceeds 65504 unless |feat|^2 does (NaN / Inf rows fail the compare too)
            const bool over = !(ss <= 65504.0f * 65504.0f);
            if (status && __builtin_amdgcn_ballot_w64(over) != 0ull && lane == 0) atomicOr(status + g0 / n, EPC_STATUS_FP16_RANGE);
        }
        const float rn = 1.0f / sqrtf(fmaxf(ss, 1e-12f));
        // cluster_bn (folded: logit*s + t) then softmax over the 64 clusters (32 here, 32 in lane^32)
        const float* cs = lds + L::OFF_CBN;
        const float* ct = cs + 64;
        float mx = -INFINITY;
        const float rn_w = rn * (1.0f / W5_SCALE);  // (the cluster weights are packed scaled as well)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = 32 * t + mfma_row(r, h);
                const float v = (P[t][r] * rn_w) * cs[k] + ct[k];
                P[t][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        float sum = 0.f;
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = expf(P[t][r] - mx);
                P[t][r] = e;
                sum += e;
            }
        sum += __shfl_xor(sum, 32);
        if (assign) {  // the f32 point-major copy is for op-level callers; the fused pipeline passes NULL (67 MB less HBM)
            float* arow = assign + (size_t)(g0 + j) * 64 + 4 * h;
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    st4(arow + 32 * t + 8 * g, make_float4(P[t][4 * g] / sum, P[t][4 * g + 1] / sum,
                                                           P[t][4 * g + 2] / sum, P[t][4 * g + 3] / sum));
        }
        if (h == 0) rnorm[g0 + j] = rn;
        // a * 2^14 as fp16 B fragments of the aggregate GEMM (cluster -> lane, 8 consecutive points -> fragment) and the
        // tile's partial a_sum (loupe.py:276).  The 2^14 keeps small assignments in fp16's normal range; it is exact
        // and the aggregate kernel removes it.  rnorm is applied on the feature side there.
        float* T = lds + L::OFF_T + wave * L::T_WAVE;
        float* fdst = assign_frag + (size_t)(g0 >> 5) * 1024 + lane * 4;
        float asum[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) T[mfma_row(r, h) * 36 + j] = P[t][r] / sum;
            float s_ = 0.f;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                float v[8];
                const float4 a0 = ld4(T + j * 36 + 16 * ks + 8 * h), a1 = ld4(T + j * 36 + 16 * ks + 8 * h + 4);
                v[0] = a0.x, v[1] = a0.y, v[2] = a0.z, v[3] = a0.w, v[4] = a1.x, v[5] = a1.y, v[6] = a1.z, v[7] = a1.w;
#pragma unroll
                for (int q = 0; q < 8; ++q) s_ += v[q];
                f16x8 th;
#pragma unroll
                for (int q = 0; q < 8; ++q) th[q] = (_Float16)(v[q] * AGG_ASSIGN_SCALE);
                *reinterpret_cast<u32x4*>(fdst + (t * 2 + ks) * 256) = __builtin_bit_cast(u32x4, th);
            }
            asum[t] = s_ + __shfl_xor(s_, 32);
        }
        if (h == 0) {
            apart[(size_t)(g0 >> 5) * 64 + j] = asum[0];
            apart[(size_t)(g0 >> 5) * 64 + 32 + j] = asum[1];
        }
    }
}

template <int CIN, bool CAT16>
static int launch_conv5(const float* cat, const float* pack, long total, int n, float* feat, float* rnorm,
                        float* assign, float* assign_frag, float* apart, int32_t* status, hipStream_t stream, const char* who) {
    const size_t lds_bytes = C5Lds<CIN>::TOTAL * sizeof(float);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv5_kernel<CIN, CAT16>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
        epc_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
        return EPC_EHIP;
    }
    const unsigned blocks = (unsigned)((total + C5_WAVES * 32 - 1) / (C5_WAVES * 32));
    hipLaunchKernelGGL((conv5_kernel<CIN, CAT16>), dim3(blocks), dim3(C5_THREADS), lds_bytes, stream, cat, pack, (int)total, n, feat,
                       rnorm, assign, assign_frag, apart, status);
    hipError_t le = hipGetLastError();
    if (le != hipSuccess) {
        epc_set_error("%s: launch failed: %s", who, hipGetErrorString(le));
        return EPC_EHIP;
    }
    return EPC_OK;
}

extern "C" int epc_conv5_assign_fwd(const void* cat, int cat_fp16, int cin, const void* packed_conv5,
                                    int num_points_total, int n, void* feat_frag, float* rnorm, float* assign,
                                    void* assign_frag, float* apart, int32_t* status, void* stream) {
    EPC_CHECK_ARG(cat && packed_conv5 && feat_frag && rnorm && assign_frag && apart, "null pointer");
    EPC_CHECK_ARG(cin == 256, "EPC-Net conv5 takes the 256-channel concat (models/epc-net.py:134)");
    EPC_CHECK_ARG(num_points_total >= 0 && num_points_total % 32 == 0, "point count must be a multiple of 32");
    EPC_CHECK_ARG(!status || (n > 0 && n % 32 == 0 && num_points_total % n == 0),
                  "status needs the points per cloud (a multiple of 32 dividing the point count)");
    if (num_points_total == 0) return EPC_OK;
    if (!status) n = 32;   // (only used to find a tile's status word)
    if (cat_fp16)
        return launch_conv5<256, true>((const float*)cat, (const float*)packed_conv5, num_points_total, n, (float*)feat_frag, rnorm,
                                       assign, (float*)assign_frag, apart, status, (hipStream_t)stream, __func__);
    return launch_conv5<256, false>((const float*)cat, (const float*)packed_conv5, num_points_total, n, (float*)feat_frag, rnorm,
                                    assign, (float*)assign_frag, apart, status, (hipStream_t)stream, __func__);
}

// ---------------------------------------------------------------------------------------------------------------
// VLAD aggregate (loupe.py:276-292): V[f][k] = sum_n (feat[n][f] * rnorm[n]) * a[n][k] - a_sum[k] * centres[f][k] per
// cloud, and the per-cluster sums of squares of V (per 32-feature chunk) that the intra-normalisation (:295) needs.  The product is an fp16
// MFMA GEMM (f32 accumulate) with the point index as K.  a arrives as ready-made fp16 B fragments (scaled by 2^14,
// removed at the end); feat arrives as fp16 in conv5's accumulator-fragment order (lane = point, 8 channels per
// fragment): each lane multiplies its 8 values by its point's rnorm in f32 (the normalised feature is <= 1: no range
// concern), rounds to fp16 and the wave transposes them into A fragments (lane = channel, 8 consecutive points).  The
// kernel streams feat once from HBM (8.4 MB per cloud) and is bound by that read.
// One workgroup = 8 waves = 4 feature groups of 64 (AGG_FT = 2 chunks of 32: one slab of the column norms) x the two
// halves of the cloud's 32-point tiles; the halves meet in LDS at the end, so no partial slabs travel through HBM and
// the centre subtraction and the column norms need no kernel of their own.
// ---------------------------------------------------------------------------------------------------------------
#define AGG_THREADS 512
#define AGG_FT 2
#define AGG_ROW 36  // fp16 per LDS row: 32 channels + 4 pad (72 B: conflict-free 8-byte stores, 8-byte aligned rows)
#define AGG_XCH_FLOATS (4 * AGG_FT * 2 * 16 * 64)   // the second half's accumulators: [wave][register][lane]

__global__ __launch_bounds__(AGG_THREADS) void vlad_aggregate_kernel(const float* __restrict__ feat_frag,
                                                                     const float* __restrict__ assign_frag,
                                                                     const float* __restrict__ rnorm,
                                                                     const float* __restrict__ apart,
                                                                     const float* __restrict__ centres, int n,
                                                                     int num_clouds, float* __restrict__ V,
                                                                     float* __restrict__ colss) {
    extern __shared__ __attribute__((aligned(16))) float agg_lds[];
    float* xch = agg_lds;                                    // [4][AGG_FT * 2 * 16][64] f32
    float* s_asum = agg_lds + AGG_XCH_FLOATS;                // [8][64]
    unsigned short* xt = reinterpret_cast<unsigned short*>(s_asum + 8 * 64);   // [8 waves][32 points][AGG_ROW]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    // Workgroup -> (cloud, 256-feature group): the four workgroups of a cloud read the same assignment fragments, so they are
    // placed behind ONE L2: workgroups are dealt round-robin over the 8 XCDs, hence ids i, i + 8, i + 16, i + 24 share an XCD
    // (speed only; any bijection is correct).  Grid = 4 * clouds, padded to a multiple of 32 with idle workgroups.
    const int wg = blockIdx.x, fgi = (wg >> 3) & 3;
    const int cloud = (wg >> 5) * 8 + (wg & 7);
    if (cloud >= num_clouds) return;
    const int fg = fgi * 4 + (wave & 3);         // group of AGG_FT chunks (32 features each): 64 features, 16 groups
    const int sp = wave >> 2;                    // which half of the cloud's tiles
    const int tiles = n / 32, half = (tiles + 1) / 2;
    const int per = sp ? tiles - half : half;

    // a_sum partials (loupe.py:276): wave w adds the per-tile sums of its eighth of the tiles, 8 loads in flight
    {
        const int t8 = (tiles + 7) / 8, ta = wave * t8, tb = min(tiles, ta + t8);
        const float* ap = apart + (size_t)cloud * tiles * 64 + lane;
        float sum = 0.f;
        int t = ta;
        for (; t + 8 <= tb; t += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ap[(size_t)(t + u) * 64];
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += v[u];
        }
        for (; t < tb; ++t) sum += ap[(size_t)t * 64];
        s_asum[wave * 64 + lane] = sum;
    }

    f32x16 acc[AGG_FT][2];
#pragma unroll
    for (int t = 0; t < AGG_FT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][0][r] = acc[t][1][r] = 0.f;

    // Ping-pong over the tiles: tile t+1's loads (4 KB of feat, 4 KB of assignment fragments per wave) are issued before
    // tile t is scaled, transposed and multiplied, so the wave always has a tile in flight (a pure streaming read reaches
    // 6.9 TB/s on this device with as little as 16 KB in flight per CU -- scripts/probe/read_bw.hip).
    // A tile in a wave's registers is its `feat` bytes (HBM, read once, non-temporal).  The tile's assignment fragments (4 KB)
    // are the same for the four waves of a half: each brings one 1-KB piece into LDS by LDS-DMA and all four read them there
    // (three slots per half aliasing the epilogue's exchange buffer, one workgroup barrier per tile; vlad_aggregate_f32_kernel
    // has the reasoning and the hazard analysis).  The 6 vector-memory operations of a tile are inline asm, their completion a
    // hand-counted s_waitcnt.
    struct Tile {
        u32x4 raw[AGG_FT][2];  // [chunk][s']
        float rn;
    };
    const int sp_u = __builtin_amdgcn_readfirstlane(sp), w3_u = __builtin_amdgcn_readfirstlane(wave & 3);
    const int per_u = __builtin_amdgcn_readfirstlane(per);
    const size_t gt0_u = (size_t)cloud * tiles + (sp_u ? half : 0);
    float* stg = xch + sp_u * (3 * 1024);                       // [3 slots][1024 floats] of this half
    const unsigned stg_lds = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)stg;
    auto load = [&](Tile& t, int tt, int slot) {
        const char* fa = reinterpret_cast<const char*>(feat_frag + ((gt0_u + tt) * 32 + (size_t)fg * AGG_FT) * 512 + lane * 4);
#pragma unroll
        for (int c = 0; c < AGG_FT; ++c)
#pragma unroll
            for (int q = 0; q < 2; ++q)
                asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(t.raw[c][q]) : "v"(fa + (c * 512 + q * 256) * 4));
        asm volatile("global_load_dword %0, %1, off" : "=v"(t.rn) : "v"(rnorm + (gt0_u + tt) * 32 + j));
        glds16(assign_frag + (gt0_u + tt) * 1024 + w3_u * 256, lane * 16, stg_lds + 4u * (slot * 1024 + w3_u * 256));
    };
    static_assert(AGG_FT == 2, "a tile is 4 + 1 + 1 = 6 vector-memory operations: the counted wait below says 6");
    auto landed = [&](Tile& t) {   // every operation issued before the LAST 6 has completed
        asm volatile("s_waitcnt vmcnt(6)" : "+v"(t.raw[0][0]), "+v"(t.raw[0][1]), "+v"(t.raw[1][0]), "+v"(t.raw[1][1]), "+v"(t.rn) : : "memory");
    };
    // Transposition lane = point -> lane = channel: every lane stores its 4-channel groups (8 bytes) into a
    // [point][channel] image and the A fragments (lane = channel, 8 consecutive points) come back through the hardware
    // transposing read ds_read_b64_tr_b16 (per 16-lane group a 4-point x 16-channel block, delivered channel-major).
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    unsigned short* img = xt + wave * 32 * AGG_ROW;
    const int li = lane & 15;
    // transposing-read address of this lane: block row q = li >> 2 (point), columns 16*((lane >> 4) & 1) + 4*(li & 3)
    const int tr_off = (8 * h + (li >> 2)) * AGG_ROW + 16 * ((lane >> 4) & 1) + 4 * (li & 3);
    auto process = [&](const Tile& t, int slot) {
        const float* fbl = stg + slot * 1024 + lane * 4;
        u32x4 bfr[2][2];       // [cluster tile][k-step]: this tile's assignment fragments, from the half's LDS slot
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) bfr[ct][ks] = *reinterpret_cast<const u32x4*>(fbl + (ct * 2 + ks) * 256);
#pragma unroll
        for (int c = 0; c < AGG_FT; ++c) {
            // scale: element q of fragment s' is channel 16s' + 8(q>>2) + 4h + (q&3) of point j
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const f16x8 fv = __builtin_bit_cast(f16x8, t.raw[c][s2]);
                f16x8 y;
#pragma unroll
                for (int q = 0; q < 8; ++q) y[q] = (_Float16)((float)fv[q] * t.rn);
                const u32x4 packed = __builtin_bit_cast(u32x4, y);
                unsigned short* dst = img + j * AGG_ROW + 16 * s2 + 4 * h;
                *reinterpret_cast<uint2*>(dst) = make_uint2(packed[0], packed[1]);       // channels 16s' + 4h + 0..3
                *reinterpret_cast<uint2*>(dst + 8) = make_uint2(packed[2], packed[3]);   // channels 16s' + 8 + 4h + 0..3
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + tr_off + (16 * ks) * AGG_ROW));
                const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + tr_off + (16 * ks + 4) * AGG_ROW));
                typedef short s16x8 __attribute__((ext_vector_type(8)));
                const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
                const f16x8 af = __builtin_bit_cast(f16x8, both);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) acc[c][ct] = mfma_f16(af, __builtin_bit_cast(f16x8, bfr[ct][ks]), acc[c][ct]);
            }
        }
    };
    Tile t0, t1;
    {
        // every wave of the workgroup walks `half` (the longer half's) tiles and meets the others at one barrier per tile; a wave
        // whose own half is shorter requests its last tile again and skips the processing.  Loads are unconditional.
        const int last = max(per_u - 1, 0);
        const bool any = per_u > 0;                                 // (a one-tile cloud leaves the second half without work)
        if (any) load(t0, 0, 0);
        for (int tt = 0; tt < half; tt += 2) {
            if (any) load(t1, min(tt + 1, last), (tt + 1) % 3);
            landed(t0);
            __builtin_amdgcn_s_barrier();                           // the four waves' DMA pieces of tile tt are in LDS
            asm volatile("" ::: "memory");                          // (the raw barrier is no compiler fence: keep the LDS reads below it)
            if (tt < per_u) process(t0, tt % 3);
            if (any) load(t0, min(tt + 2, last), (tt + 2) % 3);
            landed(t1);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (tt + 1 < per_u) process(t1, (tt + 1) % 3);
        }
        // the tail's spare requests are still in flight: the drain names both tiles' registers so that the compiler cannot hand
        // them to the epilogue before the returning loads have written them (the spare DMA pieces target LDS slots the epilogue's
        // exchange buffer aliases: hence the barrier)
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(t0.raw[0][0]), "+v"(t0.raw[0][1]), "+v"(t0.raw[1][0]), "+v"(t0.raw[1][1]), "+v"(t0.rn), "+v"(t1.raw[0][0]),
                       "+v"(t1.raw[0][1]), "+v"(t1.raw[1][0]), "+v"(t1.raw[1][1]), "+v"(t1.rn)
                     :
                     : "memory");
        __syncthreads();
    }

    // ---- the two halves meet: V = (first + second) * 2^-14 - a_sum * centres, column sums of squares ----
    // The wave of half `sp` finishes chunk c = sp of its feature group and hands its accumulators of the other chunk to
    // its partner wave (same feature group, other half) through LDS: all eight waves share the epilogue.
    static_assert(AGG_FT == 2, "one chunk per half in the epilogue");
    const int w3 = wave & 3;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float mine0 = acc[0][ct][r], mine1 = acc[1][ct][r];
            xch[(((w3 * 2 + sp) * 2 + ct) * 16 + r) * 64 + lane] = sp ? mine0 : mine1;   // the chunk the partner finishes
        }
    __syncthreads();
    float asum[2] = {0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        asum[0] += s_asum[w * 64 + j];
        asum[1] += s_asum[w * 64 + 32 + j];
    }
    constexpr float unscale = 1.0f / AGG_ASSIGN_SCALE;
    float* vout = V + (size_t)cloud * 1024 * 64;
    float ss[2] = {0.f, 0.f};
    const int chunk = fg * AGG_FT + sp;   // 32 features
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = chunk * 32 + mfma_row(r, h);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const float other = xch[(((w3 * 2 + (sp ^ 1)) * 2 + ct) * 16 + r) * 64 + lane];
            const float own = sp ? acc[1][ct][r] : acc[0][ct][r];
            const float first = sp ? other : own, second = sp ? own : other;   // fixed order: first half + second half
            const float v = (first + second) * unscale - asum[ct] * centres[f * 64 + 32 * ct + j];
            vout[(size_t)f * 64 + 32 * ct + j] = v;
            ss[ct] += v * v;
        }
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        ss[ct] += __shfl_xor(ss[ct], 32);   // the other 16 of the chunk's 32 feature rows
        if (h == 0) colss[((size_t)cloud * 32 + chunk) * 64 + 32 * ct + j] = ss[ct];
    }
}

extern "C" int epc_vlad_aggregate_fwd(const void* feat_frag, const void* assign_frag, const float* rnorm,
                                      const float* apart, const float* centres, int num_clouds, int n, float* V,
                                      float* colss, void* stream) {
    EPC_CHECK_ARG(feat_frag && assign_frag && rnorm && apart && centres && V && colss, "null pointer");
    EPC_CHECK_ARG(n > 0 && n % 32 == 0, "num_points must be a multiple of 32");
    EPC_CHECK_ARG(num_clouds >= 0 && num_clouds <= 65535, "bad shape");
    if (num_clouds == 0) return EPC_OK;
    const size_t lds_bytes = (AGG_XCH_FLOATS + 8 * 64) * sizeof(float) + 8 * 32 * AGG_ROW * sizeof(unsigned short);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(vlad_aggregate_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
        epc_set_error("epc_vlad_aggregate_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return EPC_EHIP;
    }
    hipLaunchKernelGGL(vlad_aggregate_kernel, dim3(32 * ((num_clouds + 7) / 8)), dim3(AGG_THREADS), lds_bytes,
                       (hipStream_t)stream, (const float*)feat_frag, (const float*)assign_frag, rnorm, apart, centres, n,
                       num_clouds, V, colss);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// EPC_PRECISION_F32 form of the aggregate: feat arrives as 3-byte values (accumulator order, epc_conv5_assign_f32_fwd), the
// assignments as bf16 hi + lo B fragments.  (feat * rnorm) is split into bf16 hi + lo per lane, both halves go through
// their own [point][channel] LDS image and come back as A fragments via the transposing read; three products per
// k-step (lo*hi + hi*lo + hi*hi, f32 accumulate).  Same workgroup geometry, epilogue and outputs as the fp16 form; it
// streams twice the bytes (16 MB of feat per cloud) and is bound by that read.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(AGG_THREADS) void vlad_aggregate_f32_kernel(const float* __restrict__ feat_frag,
                                                                         const float* __restrict__ assign_frag,
                                                                         const float* __restrict__ rnorm,
                                                                         const float* __restrict__ apart,
                                                                         const float* __restrict__ centres, int n,
                                                                         int num_clouds, float* __restrict__ V,
                                                                         float* __restrict__ colss) {
    extern __shared__ __attribute__((aligned(16))) float agg_lds[];
    float* xch = agg_lds;                                    // [4][AGG_FT * 2 * 16][64] f32
    float* s_asum = agg_lds + AGG_XCH_FLOATS;                // [8][64]
    unsigned short* xt = reinterpret_cast<unsigned short*>(s_asum + 8 * 64);   // [8 waves][hi, lo][32 points][AGG_ROW]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int li = lane & 15, q4 = lane >> 4;            // conv5_f32.hip's lane roles: point-in-group, channel quad
    const int wg = blockIdx.x, fgi = (wg >> 3) & 3;      // XCD-aware (cloud, feature group) mapping: see vlad_aggregate_kernel
    const int cloud = (wg >> 5) * 8 + (wg & 7);
    if (cloud >= num_clouds) return;
    const int fg = fgi * 4 + (wave & 3);
    const int sp = wave >> 2;
    const int tiles = n / 32, half = (tiles + 1) / 2;
    const int per = sp ? tiles - half : half;

    {
        const int t8 = (tiles + 7) / 8, ta = wave * t8, tb = min(tiles, ta + t8);
        const float* ap = apart + (size_t)cloud * tiles * 64 + lane;
        float sum = 0.f;
        int t = ta;
        for (; t + 8 <= tb; t += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = ap[(size_t)(t + u) * 64];
#pragma unroll
            for (int u = 0; u < 8; ++u) sum += v[u];
        }
        for (; t < tb; ++t) sum += ap[(size_t)t * 64];
        s_asum[wave * 64 + lane] = sum;
    }

    f32x16 acc[AGG_FT][2];
#pragma unroll
    for (int t = 0; t < AGG_FT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][0][r] = acc[t][1][r] = 0.f;

    // A tile in a wave's registers is its `feat` bytes (HBM, read once).  The tile's ASSIGNMENT fragments (8 KB) are the same for
    // the four waves of a half (same tiles, other features): each of the four brings a quarter of them into LDS by LDS-DMA and all
    // four read them there -- as per-wave global loads they were 53 % of the kernel's load traffic (1.07 GB through L2 per launch
    // beside the 0.8 GB of feat) and cost it 0.035 of its 0.19 ms (ablation: every wave re-reading one L1-hot fragment tile).
    // Three LDS slots per half (aliasing the exchange buffer of the epilogue), one workgroup barrier per tile: a wave may run one
    // tile ahead of the slowest, and the slot it fills (t + 2) is neither the one being read (t) nor the one landed (t + 1).
    struct Tile {
        // [chunk][piece]: the lane's 16 three-byte values of a chunk in conv5_f32.hip's accumulator order: with li = lane & 15,
        // q = lane >> 4, value 4 t + r (t = 2 g + p) = channel 16 g + 4 q + r of point 16 p + li
        u32x4 raw[AGG_FT][3];
        float rn[2];           // rnorm of points li and 16 + li
    };
    const int sp_u = __builtin_amdgcn_readfirstlane(sp), w3_u = __builtin_amdgcn_readfirstlane(wave & 3);
    const int per_u = __builtin_amdgcn_readfirstlane(per);
    const size_t gt0_u = (size_t)cloud * tiles + (sp_u ? half : 0);
    float* stg = xch + sp_u * (3 * 2048);                       // [3 slots][2048 floats] of this half
    const unsigned stg_lds = (unsigned)(size_t)(const __attribute__((address_space(3))) float*)stg;
    // A tile's 10 vector-memory operations (6 feat loads, 2 rnorm, 2 LDS-DMA pieces) are inline asm and their completion a
    // hand-counted s_waitcnt: as C++ loads hipcc 7.2 ended every tile on `s_waitcnt vmcnt(0)` (its wait bookkeeping merges the
    // exec-masked prefetch branches) and the ping-pong overlapped nothing.  Operations complete in issue order, so `vmcnt(10)`
    // after the NEXT tile's 10 are issued is "this tile has landed".
    auto load = [&](Tile& t, int tt, int slot) {
        const char* fa = reinterpret_cast<const char*>(feat_frag + ((gt0_u + tt) * 32 + (size_t)fg * AGG_FT) * 768 + lane * 4);
#pragma unroll
        for (int c = 0; c < AGG_FT; ++c)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(t.raw[c][q]) : "v"(fa + (c * 768 + q * 256) * 4));
        asm volatile("global_load_dword %0, %1, off" : "=v"(t.rn[0]) : "v"(rnorm + (gt0_u + tt) * 32 + li));
        asm volatile("global_load_dword %0, %1, off" : "=v"(t.rn[1]) : "v"(rnorm + (gt0_u + tt) * 32 + 16 + li));
        const float* fb = assign_frag + (gt0_u + tt) * 2048 + (2 * w3_u) * 256;      // this wave's two 1-KB pieces of the tile
        glds16(fb, lane * 16, stg_lds + 4u * (slot * 2048 + (2 * w3_u) * 256));
        glds16(fb + 256, lane * 16, stg_lds + 4u * (slot * 2048 + (2 * w3_u + 1) * 256));
    };
    static_assert(AGG_FT == 2, "a tile is 6 + 2 + 2 = 10 vector-memory operations: the counted wait below says 10");
    auto landed = [&](Tile& t) {
        asm volatile("s_waitcnt vmcnt(10)"
                     : "+v"(t.raw[0][0]), "+v"(t.raw[0][1]), "+v"(t.raw[0][2]), "+v"(t.raw[1][0]), "+v"(t.raw[1][1]), "+v"(t.raw[1][2]),
                       "+v"(t.rn[0]), "+v"(t.rn[1])
                     :
                     : "memory");
    };
    typedef short s16x4 __attribute__((ext_vector_type(4)));
    typedef short s16x8 __attribute__((ext_vector_type(8)));
    typedef __attribute__((address_space(3))) s16x4 lds_s16x4;
    unsigned short* img_hi = xt + (wave * 2 + 0) * 32 * AGG_ROW;
    unsigned short* img_lo = xt + (wave * 2 + 1) * 32 * AGG_ROW;
    const int tr_off = (8 * h + (li >> 2)) * AGG_ROW + 16 * ((lane >> 4) & 1) + 4 * (li & 3);
    auto tr_read = [&](const unsigned short* img, int ks) {
        const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + tr_off + (16 * ks) * AGG_ROW));
        const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(img + tr_off + (16 * ks + 4) * AGG_ROW));
        const s16x8 both = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(bf16x8, both);
    };
    auto process = [&](const Tile& t, int slot) {
        const float* fbl = stg + slot * 2048 + lane * 4;
        u32x4 bfr[2][2][2];    // [cluster tile][k-step][hi, lo]: this tile's assignment fragments, from the half's LDS slot
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int part = 0; part < 2; ++part)
                    bfr[ct][ks][part] = *reinterpret_cast<const u32x4*>(fbl + ((ct * 2 + ks) * 2 + part) * 256);
#pragma unroll
        for (int c = 0; c < AGG_FT; ++c) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                unsigned short hb[4], lb[4];
                // dwords 3r .. 3r + 2 of the lane's twelve hold values 4r .. 4r + 3 (conv5's packing)
                const unsigned int w0 = t.raw[c][(3 * r4) >> 2][(3 * r4) & 3], w1 = t.raw[c][(3 * r4 + 1) >> 2][(3 * r4 + 1) & 3],
                                   w2 = t.raw[c][(3 * r4 + 2) >> 2][(3 * r4 + 2) & 3];
                const unsigned int fv[4] = {w0 << 8, __builtin_amdgcn_perm(w1, w0, 0x0504030cu),
                                            __builtin_amdgcn_perm(w2, w1, 0x0403020cu), w2 & 0xffffff00u};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float y = __uint_as_float(fv[e]) * t.rn[r4 & 1];
                    const __bf16 yh = (__bf16)y;
                    const __bf16 yl = (__bf16)(y - (float)yh);
                    hb[e] = __builtin_bit_cast(unsigned short, yh);
                    lb[e] = __builtin_bit_cast(unsigned short, yl);
                }
                const int off = (16 * (r4 & 1) + li) * AGG_ROW + 16 * (r4 >> 1) + 4 * q4;   // t = r4 = 2 g + p: channels 16 g + 4 q + 0..3 of point 16 p + li
                *reinterpret_cast<uint2*>(img_hi + off) = make_uint2(hb[0] | ((unsigned)hb[1] << 16), hb[2] | ((unsigned)hb[3] << 16));
                *reinterpret_cast<uint2*>(img_lo + off) = make_uint2(lb[0] | ((unsigned)lb[1] << 16), lb[2] | ((unsigned)lb[3] << 16));
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 ah = tr_read(img_hi, ks), al = tr_read(img_lo, ks);
#pragma unroll
                for (int ct = 0; ct < 2; ++ct) {
                    const bf16x8 bh = __builtin_bit_cast(bf16x8, bfr[ct][ks][0]), bl = __builtin_bit_cast(bf16x8, bfr[ct][ks][1]);
                    acc[c][ct] = mfma_bf16(al, bh, acc[c][ct]);
                    acc[c][ct] = mfma_bf16(ah, bl, acc[c][ct]);
                    acc[c][ct] = mfma_bf16(ah, bh, acc[c][ct]);
                }
            }
        }
    };
    Tile t0, t1;
    {
        // every wave of the workgroup walks `half` (the longer half's) tiles and meets the others at one barrier per tile; a wave
        // whose own half is shorter requests its last tile again and skips the processing.  Loads are unconditional.
        const int last = max(per_u - 1, 0);
        const bool any = per_u > 0;                                 // (a one-tile cloud leaves the second half without work)
        if (any) load(t0, 0, 0);
        for (int tt = 0; tt < half; tt += 2) {
            if (any) load(t1, min(tt + 1, last), (tt + 1) % 3);
            landed(t0);
            __builtin_amdgcn_s_barrier();                           // the four waves' DMA pieces of tile tt are in LDS
            asm volatile("" ::: "memory");                          // (the raw barrier is no compiler fence: keep the LDS reads below it)
            if (tt < per_u) process(t0, tt % 3);
            if (any) load(t0, min(tt + 2, last), (tt + 2) % 3);
            landed(t1);
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (tt + 1 < per_u) process(t1, (tt + 1) % 3);
        }
        // The tail's spare requests are still in flight: their destination registers are dead to the compiler -- it would hand
        // them to the epilogue, and the returning loads would overwrite its values.  The drain names both tiles' registers, so they
        // stay allocated until it has executed.  (The spare DMA pieces target LDS slots the epilogue's exchange buffer aliases.)
        asm volatile("s_waitcnt vmcnt(0)"
                     : "+v"(t0.raw[0][0]), "+v"(t0.raw[0][1]), "+v"(t0.raw[0][2]), "+v"(t0.raw[1][0]), "+v"(t0.raw[1][1]), "+v"(t0.raw[1][2]),
                       "+v"(t0.rn[0]), "+v"(t0.rn[1]), "+v"(t1.raw[0][0]), "+v"(t1.raw[0][1]), "+v"(t1.raw[0][2]), "+v"(t1.raw[1][0]),
                       "+v"(t1.raw[1][1]), "+v"(t1.raw[1][2]), "+v"(t1.rn[0]), "+v"(t1.rn[1])
                     :
                     : "memory");
        __syncthreads();
    }

    static_assert(AGG_FT == 2, "one chunk per half in the epilogue");
    const int w3 = wave & 3;
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float mine0 = acc[0][ct][r], mine1 = acc[1][ct][r];
            xch[(((w3 * 2 + sp) * 2 + ct) * 16 + r) * 64 + lane] = sp ? mine0 : mine1;
        }
    __syncthreads();
    float asum[2] = {0.f, 0.f};
#pragma unroll
    for (int w = 0; w < 8; ++w) {
        asum[0] += s_asum[w * 64 + j];
        asum[1] += s_asum[w * 64 + 32 + j];
    }
    float* vout = V + (size_t)cloud * 1024 * 64;
    float ss[2] = {0.f, 0.f};
    const int chunk = fg * AGG_FT + sp;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int f = chunk * 32 + mfma_row(r, h);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const float other = xch[(((w3 * 2 + (sp ^ 1)) * 2 + ct) * 16 + r) * 64 + lane];
            const float own = sp ? acc[1][ct][r] : acc[0][ct][r];
            const float first = sp ? other : own, second = sp ? own : other;
            const float v = (first + second) - asum[ct] * centres[f * 64 + 32 * ct + j];
            vout[(size_t)f * 64 + 32 * ct + j] = v;
            ss[ct] += v * v;
        }
    }
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
        ss[ct] += __shfl_xor(ss[ct], 32);
        if (h == 0) colss[((size_t)cloud * 32 + chunk) * 64 + 32 * ct + j] = ss[ct];
    }
}

extern "C" int epc_vlad_aggregate_f32_fwd(const void* feat_frag, const void* assign_frag, const float* rnorm,
                                          const float* apart, const float* centres, int num_clouds, int n, float* V,
                                          float* colss, void* stream) {
    EPC_CHECK_ARG(feat_frag && assign_frag && rnorm && apart && centres && V && colss, "null pointer");
    EPC_CHECK_ARG(n > 0 && n % 32 == 0, "num_points must be a multiple of 32");
    EPC_CHECK_ARG(num_clouds >= 0 && num_clouds <= 65535, "bad shape");
    if (num_clouds == 0) return EPC_OK;
    const size_t lds_bytes = (AGG_XCH_FLOATS + 8 * 64) * sizeof(float) + 2 * 8 * 32 * AGG_ROW * sizeof(unsigned short);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(vlad_aggregate_f32_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
        epc_set_error("epc_vlad_aggregate_f32_fwd: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return EPC_EHIP;
    }
    hipLaunchKernelGGL(vlad_aggregate_f32_kernel, dim3(32 * ((num_clouds + 7) / 8)), dim3(AGG_THREADS), lds_bytes,
                       (hipStream_t)stream, (const float*)feat_frag, (const float*)assign_frag, rnorm, apart, centres, n, num_clouds, V,
                       colss);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}
