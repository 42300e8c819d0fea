// Static xyz kNN graph in index form -- replaces utils/tf_util.py:647-666 (pairwise_distance_mask), which
// materialises a dense (B,N,N) float mask (64 MB per 4096-point cloud).
//
// Semantics (bit-exact w.r.t. oracle/epcnet_oracle.py): a_ij = -((sq_i + -2*inner_ij) + sq_j) in f32 with one
// rounding per operation; kth_i = 20th largest a_i. with multiplicity (tf.nn.top_k + reduce_min); the selected set is
// {j : a_ij >= kth_i} (greater_equal), emitted in ascending j, counted past the list capacity.
//
// knn_topk_culled_kernel (N <= 8192): the whole cloud sits in LDS as (x,y,z,|p|^2); one thread per query; candidates
// are visited in 32-point tiles, nearest tiles (by index distance from the query's own tile) first.  Before a tile is
// scanned every lane computes the exact lower bound of its distance to the tile's bounding box and the wave votes:
// the tile is skipped when no lane can improve its current 20th distance (minus a rounding margin that covers the
// difference between the computed a_ij and the true squared distance).  On spatially ordered clouds (the pipeline
// Morton-sorts them first, sort.hip) a wave's 64 queries are neighbours and ~3/4 of the tiles are skipped; on
// arbitrary order the bounds are loose and the kernel degrades gracefully to the full scan with identical results.
// Pass 1 keeps the 20 largest a_ij in a sorted register file (v_max/v_min insertion network, skipped wave-wide when
// no lane improves) and records, per wave, which 8-candidate batches produced a hit; pass 2 re-visits exactly those
// batches in ascending order and emits the list.  (Packed-f32 distance arithmetic -- v_pk_mul/add_f32 on candidate
// pairs -- was measured slower on gfx950: 0.35 vs 0.30 ms.)
//
// knn_topk_stream_kernel (any N): same two passes with candidates streamed through an LDS tile, no culling.
#include "common.h"
#include <stdlib.h>

#ifndef KNN_THREADS
#define KNN_THREADS 1024   // re-tuned on Hilbert-ordered clouds (r01_o): 1024 0.248 ms, 512 0.262, 256 0.372 per 64 clouds
#endif
#define KNN_TILE 1024
#ifndef KNN_CT
#define KNN_CT 32             // candidate tile (culled kernel)
#endif
#define KNN_WAVES (KNN_THREADS / 64)
#define KNN_LDS_MAX_N 8192    // (N * 16 B + N/32 * 32 B) <= 160 KB

// Order-preserving float -> int key (involution): integer compare == float compare for non-NaN values once -0.0 has
// been folded into +0.0 (callers add +0.0f).  The insertion network then runs on v_max_i32 / v_min_i32 (the float
// forms cost an extra canonicalising v_max per slot).
__device__ __forceinline__ int fkey(float v) {
    const int b = __float_as_int(v);
    return b ^ ((b >> 31) & 0x7fffffff);
}
__device__ __forceinline__ float fkey_inv(int k) { return __int_as_float(k ^ ((k >> 31) & 0x7fffffff)); }

template <int KSEL>
__device__ __forceinline__ void topk_insert(float (&top)[KSEL], float v) {
#pragma unroll
    for (int s = 0; s < KSEL; ++s) {
        const float hi = fmaxf(top[s], v);
        v = fminf(top[s], v);
        top[s] = hi;
    }
}

// The 20-slot network as ONE asm block: min into the spare register, max IN PLACE, the carried value alternating
// between two registers.  (Left to itself hipcc writes each max one register further and shifts the whole file back
// with 20 v_mov per insertion; slot-wise asm statements get an s_nop each from the hazard recogniser.)
__device__ __forceinline__ void topk_insert_key(int (&top)[20], int v) {
    int t;
    asm(
        "v_min_i32 %[t], %[k0], %[v]\n\tv_max_i32 %[k0], %[k0], %[v]\n\t"
        "v_min_i32 %[v], %[k1], %[t]\n\tv_max_i32 %[k1], %[k1], %[t]\n\t"
        "v_min_i32 %[t], %[k2], %[v]\n\tv_max_i32 %[k2], %[k2], %[v]\n\t"
        "v_min_i32 %[v], %[k3], %[t]\n\tv_max_i32 %[k3], %[k3], %[t]\n\t"
        "v_min_i32 %[t], %[k4], %[v]\n\tv_max_i32 %[k4], %[k4], %[v]\n\t"
        "v_min_i32 %[v], %[k5], %[t]\n\tv_max_i32 %[k5], %[k5], %[t]\n\t"
        "v_min_i32 %[t], %[k6], %[v]\n\tv_max_i32 %[k6], %[k6], %[v]\n\t"
        "v_min_i32 %[v], %[k7], %[t]\n\tv_max_i32 %[k7], %[k7], %[t]\n\t"
        "v_min_i32 %[t], %[k8], %[v]\n\tv_max_i32 %[k8], %[k8], %[v]\n\t"
        "v_min_i32 %[v], %[k9], %[t]\n\tv_max_i32 %[k9], %[k9], %[t]\n\t"
        "v_min_i32 %[t], %[k10], %[v]\n\tv_max_i32 %[k10], %[k10], %[v]\n\t"
        "v_min_i32 %[v], %[k11], %[t]\n\tv_max_i32 %[k11], %[k11], %[t]\n\t"
        "v_min_i32 %[t], %[k12], %[v]\n\tv_max_i32 %[k12], %[k12], %[v]\n\t"
        "v_min_i32 %[v], %[k13], %[t]\n\tv_max_i32 %[k13], %[k13], %[t]\n\t"
        "v_min_i32 %[t], %[k14], %[v]\n\tv_max_i32 %[k14], %[k14], %[v]\n\t"
        "v_min_i32 %[v], %[k15], %[t]\n\tv_max_i32 %[k15], %[k15], %[t]\n\t"
        "v_min_i32 %[t], %[k16], %[v]\n\tv_max_i32 %[k16], %[k16], %[v]\n\t"
        "v_min_i32 %[v], %[k17], %[t]\n\tv_max_i32 %[k17], %[k17], %[t]\n\t"
        "v_min_i32 %[t], %[k18], %[v]\n\tv_max_i32 %[k18], %[k18], %[v]\n\t"
        "v_min_i32 %[v], %[k19], %[t]\n\tv_max_i32 %[k19], %[k19], %[t]\n\t"
        : [v] "+v"(v), [t] "=&v"(t), [k0] "+v"(top[0]), [k1] "+v"(top[1]), [k2] "+v"(top[2]), [k3] "+v"(top[3]), [k4] "+v"(top[4]), [k5] "+v"(top[5]), [k6] "+v"(top[6]), [k7] "+v"(top[7]), [k8] "+v"(top[8]), [k9] "+v"(top[9]), [k10] "+v"(top[10]), [k11] "+v"(top[11]), [k12] "+v"(top[12]), [k13] "+v"(top[13]), [k14] "+v"(top[14]), [k15] "+v"(top[15]), [k16] "+v"(top[16]), [k17] "+v"(top[17]), [k18] "+v"(top[18]), [k19] "+v"(top[19]));
}

// The network on the distances themselves: the list holds the 20 SMALLEST d' ascending (formerly: min in place, max carried
// on), so the current threshold is top[19] as it stands and no candidate needs an order-preserving integer key (3
// instructions per candidate of every hit batch).  v_min_f32 / v_max_f32 are single instructions in asm -- the extra
// canonicalising v_max per slot that made the float form slower belongs to the compiler's fminf / fmaxf, not to the
// hardware; no NaN can occur, d' = +0 for coincident points (never -0), and a value that is not below top[19] falls through.
__device__ __forceinline__ void topk_insert_dist(float (&top)[20], float v) {
#ifdef KNN_MINMAX_NETWORK   // the former 40-instruction form: max carried on, min in place
    float t;
    asm(
        "v_max_f32 %[t], %[k0], %[v]\n\tv_min_f32 %[k0], %[k0], %[v]\n\t"
        "v_max_f32 %[v], %[k1], %[t]\n\tv_min_f32 %[k1], %[k1], %[t]\n\t"
        "v_max_f32 %[t], %[k2], %[v]\n\tv_min_f32 %[k2], %[k2], %[v]\n\t"
        "v_max_f32 %[v], %[k3], %[t]\n\tv_min_f32 %[k3], %[k3], %[t]\n\t"
        "v_max_f32 %[t], %[k4], %[v]\n\tv_min_f32 %[k4], %[k4], %[v]\n\t"
        "v_max_f32 %[v], %[k5], %[t]\n\tv_min_f32 %[k5], %[k5], %[t]\n\t"
        "v_max_f32 %[t], %[k6], %[v]\n\tv_min_f32 %[k6], %[k6], %[v]\n\t"
        "v_max_f32 %[v], %[k7], %[t]\n\tv_min_f32 %[k7], %[k7], %[t]\n\t"
        "v_max_f32 %[t], %[k8], %[v]\n\tv_min_f32 %[k8], %[k8], %[v]\n\t"
        "v_max_f32 %[v], %[k9], %[t]\n\tv_min_f32 %[k9], %[k9], %[t]\n\t"
        "v_max_f32 %[t], %[k10], %[v]\n\tv_min_f32 %[k10], %[k10], %[v]\n\t"
        "v_max_f32 %[v], %[k11], %[t]\n\tv_min_f32 %[k11], %[k11], %[t]\n\t"
        "v_max_f32 %[t], %[k12], %[v]\n\tv_min_f32 %[k12], %[k12], %[v]\n\t"
        "v_max_f32 %[v], %[k13], %[t]\n\tv_min_f32 %[k13], %[k13], %[t]\n\t"
        "v_max_f32 %[t], %[k14], %[v]\n\tv_min_f32 %[k14], %[k14], %[v]\n\t"
        "v_max_f32 %[v], %[k15], %[t]\n\tv_min_f32 %[k15], %[k15], %[t]\n\t"
        "v_max_f32 %[t], %[k16], %[v]\n\tv_min_f32 %[k16], %[k16], %[v]\n\t"
        "v_max_f32 %[v], %[k17], %[t]\n\tv_min_f32 %[k17], %[k17], %[t]\n\t"
        "v_max_f32 %[t], %[k18], %[v]\n\tv_min_f32 %[k18], %[k18], %[v]\n\t"
        "v_max_f32 %[v], %[k19], %[t]\n\tv_min_f32 %[k19], %[k19], %[t]\n\t"
        : [v] "+v"(v), [t] "=&v"(t), [k0] "+v"(top[0]), [k1] "+v"(top[1]), [k2] "+v"(top[2]), [k3] "+v"(top[3]), [k4] "+v"(top[4]), [k5] "+v"(top[5]), [k6] "+v"(top[6]), [k7] "+v"(top[7]), [k8] "+v"(top[8]), [k9] "+v"(top[9]), [k10] "+v"(top[10]), [k11] "+v"(top[11]), [k12] "+v"(top[12]), [k13] "+v"(top[13]), [k14] "+v"(top[14]), [k15] "+v"(top[15]), [k16] "+v"(top[16]), [k17] "+v"(top[17]), [k18] "+v"(top[18]), [k19] "+v"(top[19]));
#else
    // Sorted insertion is a MEDIAN per slot: new[s] = med3(old[s-1], old[s], v) -- old[s] when v lies above it, old[s-1] when v
    // lies below that, v itself in between -- and new[0] = min(old[0], v).  Done from the top slot down, in place, each slot
    // still reads its lower neighbour's OLD value: 20 independent instructions instead of 20 dependent min / max pairs
    // (v_med3_f32 issues at the rate of v_min_f32).  Same list bit for bit: a median only ever returns one of its operands.
    asm(
        "v_med3_f32 %[k19], %[k18], %[k19], %[v]\n\t"
        "v_med3_f32 %[k18], %[k17], %[k18], %[v]\n\t"
        "v_med3_f32 %[k17], %[k16], %[k17], %[v]\n\t"
        "v_med3_f32 %[k16], %[k15], %[k16], %[v]\n\t"
        "v_med3_f32 %[k15], %[k14], %[k15], %[v]\n\t"
        "v_med3_f32 %[k14], %[k13], %[k14], %[v]\n\t"
        "v_med3_f32 %[k13], %[k12], %[k13], %[v]\n\t"
        "v_med3_f32 %[k12], %[k11], %[k12], %[v]\n\t"
        "v_med3_f32 %[k11], %[k10], %[k11], %[v]\n\t"
        "v_med3_f32 %[k10], %[k9], %[k10], %[v]\n\t"
        "v_med3_f32 %[k9], %[k8], %[k9], %[v]\n\t"
        "v_med3_f32 %[k8], %[k7], %[k8], %[v]\n\t"
        "v_med3_f32 %[k7], %[k6], %[k7], %[v]\n\t"
        "v_med3_f32 %[k6], %[k5], %[k6], %[v]\n\t"
        "v_med3_f32 %[k5], %[k4], %[k5], %[v]\n\t"
        "v_med3_f32 %[k4], %[k3], %[k4], %[v]\n\t"
        "v_med3_f32 %[k3], %[k2], %[k3], %[v]\n\t"
        "v_med3_f32 %[k2], %[k1], %[k2], %[v]\n\t"
        "v_med3_f32 %[k1], %[k0], %[k1], %[v]\n\t"
        "v_min_f32 %[k0], %[k0], %[v]\n\t"
        : [k0] "+v"(top[0]), [k1] "+v"(top[1]), [k2] "+v"(top[2]), [k3] "+v"(top[3]), [k4] "+v"(top[4]), [k5] "+v"(top[5]), [k6] "+v"(top[6]), [k7] "+v"(top[7]), [k8] "+v"(top[8]), [k9] "+v"(top[9]), [k10] "+v"(top[10]), [k11] "+v"(top[11]), [k12] "+v"(top[12]), [k13] "+v"(top[13]), [k14] "+v"(top[14]), [k15] "+v"(top[15]), [k16] "+v"(top[16]), [k17] "+v"(top[17]), [k18] "+v"(top[18]), [k19] "+v"(top[19])
        : [v] "v"(v));
#endif
}

// wave vote straight from the compare's lane mask (HIP's __any goes through an integer predicate: a v_cndmask and a second
// compare per vote)
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

#ifndef KNN_BATCH
#define KNN_BATCH 8
#endif
// hit-mask words per wave: KNN_LDS_MAX_N / KNN_BATCH bits
#define KNN_MASK_WORDS (KNN_LDS_MAX_N / KNN_BATCH / 32)
static_assert(32 % (KNN_CT / KNN_BATCH) == 0, "a tile's batch bits must not straddle a mask word");

#ifdef KNN_STATS  // tuning builds only (scripts/tune_knn.sh): wave-level event counters
__device__ unsigned long long g_knn_stats[8];
#define KSTAT(i) do { if ((threadIdx.x & 63) == 0) atomicAdd(&g_knn_stats[i], 1ull); } while (0)
extern "C" int epc_debug_knn_stats(unsigned long long* host_out, int reset) {
    if (hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_knn_stats), sizeof(g_knn_stats)) != hipSuccess) return -3;
    if (reset) {
        unsigned long long z[8] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_knn_stats), z, sizeof(z)) != hipSuccess) return -3;
    }
    return 0;
}
#else
#define KSTAT(i) do { } while (0)
#endif

// CONV1: the workgroup also produces conv1 (3 -> 64, models/epc-net.py:66-69) of its own KNN_THREADS points from the
// cloud image it has just put in LDS -- 16 lanes x 4 channels per point, whole rows per store like conv1_kernel, ~1.5 %
// more VALU work for this kernel instead of a separate 15-us launch that re-reads the cloud.
// BATCH: candidates evaluated per group vote.  8 for grids of up to one workgroup per CU (fewest votes); 4 for larger grids
// (EPC-Net-L at batch 256: 1024 workgroups): 48 instead of 68 registers, so that TWO 1024-thread workgroups share a CU
// (8 waves per SIMD; their two 73-KB LDS images fit) and cover each other's LDS and branch latency -- the kernel issues a
// vector instruction in under 40 % of the SIMD's cycles at 4 waves.  0.60 -> 0.51 ms per 256 clouds; at batch 64 (256
// workgroups: nothing to share a CU with) the 8-wide form stays faster (0.19 vs 0.20 ms).  Same lists either way.
template <int KSEL, bool CONV1, int BATCH>
__global__ __launch_bounds__(KNN_THREADS, BATCH == 4 ? 8 : 1) void knn_topk_culled_kernel(const float* __restrict__ xyz, int n, int cap,
                                                                      int32_t* __restrict__ idx,
                                                                      int32_t* __restrict__ cnt,
                                                                      float* __restrict__ kth_out,
                                                                      const float* __restrict__ conv1_pack,
                                                                      float* __restrict__ x32,
                                                                      unsigned short* __restrict__ x16, int idx_u16,
                                                                      int32_t* __restrict__ status) {
    extern __shared__ __attribute__((aligned(16))) float4 cand[];  // [npad] points, then 2 float4 per tile (lo, hi)
    const int ntiles = (n + KNN_CT - 1) / KNN_CT;
    const int npad = ntiles * KNN_CT;
    float4* bb = cand + npad;                  // [2*ntiles] boxes, then one float4 holding the margin
    float* s_margin = reinterpret_cast<float*>(bb + 2 * ntiles);
    // per wave: one bit per (tile, 8-candidate batch) that produced a hit in pass 1 (KNN_CT / BATCH bits per tile)
    constexpr int BPT = KNN_CT / BATCH;
    constexpr int MASKW = KNN_LDS_MAX_N / BATCH / 32;   // hit-mask words per wave
    static_assert(32 % BPT == 0, "a tile's batch bits must not straddle a mask word");
    unsigned int* hitmask = reinterpret_cast<unsigned int*>(s_margin + 4) + (threadIdx.x >> 6) * MASKW;
    const int cloud = blockIdx.y;
#ifdef KNN_WAVE_VECTOR   // (tuning: the wave index as the compiler sees it from threadIdx, a vector value)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
#else
    // the wave index is uniform, which the compiler cannot tell from threadIdx: through readfirstlane the tile numbers of the walk and
    // the tile addresses are scalar values
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#endif
    const float* pc = xyz + (size_t)cloud * n * 3;

    bool bad = false;   // a NaN / Inf coordinate (or one whose square overflows): |p|^2 is not a finite number
    for (int j = tid; j < npad; j += KNN_THREADS) {
        float4 v = make_float4(0.f, 0.f, 0.f, INFINITY);
        if (j < n) {
            v.x = pc[3 * j + 0];
            v.y = pc[3 * j + 1];
            v.z = pc[3 * j + 2];
            v.w = sq3(v.x, v.y, v.z);
            bad |= !(v.w <= 3.4028234664e38f);
        }
        cand[j] = v;
    }
    // every workgroup of the cloud has seen the whole cloud: the first one reports (EPC_STATUS_NONFINITE_INPUT)
    if (status && blockIdx.x == 0 && wave_any(bad) && lane == 0) atomicOr(status + cloud, EPC_STATUS_NONFINITE_INPUT);
    __syncthreads();
    if constexpr (CONV1) {
        const int q = tid & 15;   // KNN_THREADS is a multiple of 16: a thread keeps its channel quad
        const float4 w0 = *reinterpret_cast<const float4*>(conv1_pack + 4 * q);
        const float4 w1 = *reinterpret_cast<const float4*>(conv1_pack + 64 + 4 * q);
        const float4 w2 = *reinterpret_cast<const float4*>(conv1_pack + 128 + 4 * q);
        const float4 b = *reinterpret_cast<const float4*>(conv1_pack + 192 + 4 * q);
        const int p0 = blockIdx.x * KNN_THREADS;
        const int np = min(KNN_THREADS, n - p0);
        bool ovf = false;   // a conv1 output that does not fit the fp16 row (outputs are >= 0 after the ReLU)
        for (int t = tid; t < np * 16; t += KNN_THREADS) {
            const int g = p0 + (t >> 4);
            const float4 pt = cand[g];
            const float4 y = conv1_quad(pt.x, pt.y, pt.z, w0, w1, w2, b);
            const size_t row = (size_t)cloud * n + g;
            if (x32) *reinterpret_cast<float4*>(x32 + row * 64 + 4 * q) = y;
            if (x16) {
                reinterpret_cast<uint2*>(x16)[row * 16 + q] = pack_half4(y);
                ovf |= fmaxf(fmaxf(y.x, y.y), fmaxf(y.z, y.w)) > 65504.0f;
            }
        }
        if (status && x16 && wave_any(ovf) && lane == 0) atomicOr(status + cloud, EPC_STATUS_FP16_RANGE);
    }
    // tile bounding boxes: KNN_CT lanes per tile, shuffle min/max
    for (int t = wave * (64 / KNN_CT) + lane / KNN_CT; t < ntiles; t += KNN_WAVES * (64 / KNN_CT)) {
        const float4 v = cand[t * KNN_CT + (lane % KNN_CT)];
        const bool ok = v.w != INFINITY;
        float lx = ok ? v.x : INFINITY, ly = ok ? v.y : INFINITY, lz = ok ? v.z : INFINITY;
        float hx = ok ? v.x : -INFINITY, hy = ok ? v.y : -INFINITY, hz = ok ? v.z : -INFINITY;
#pragma unroll
        for (int off = KNN_CT / 2; off >= 1; off >>= 1) {
            lx = fminf(lx, __shfl_xor(lx, off));
            ly = fminf(ly, __shfl_xor(ly, off));
            lz = fminf(lz, __shfl_xor(lz, off));
            hx = fmaxf(hx, __shfl_xor(hx, off));
            hy = fmaxf(hy, __shfl_xor(hy, off));
            hz = fmaxf(hz, __shfl_xor(hz, off));
        }
        if ((lane % KNN_CT) == 0) {
            bb[2 * t] = make_float4(lx, ly, lz, 0.f);
            bb[2 * t + 1] = make_float4(hx, hy, hz, 0.f);
        }
    }
    __syncthreads();
    if (wave == 0) {
        // margin >= |computed(-a_ij) - true d^2| + rounding of the box bound: 256 ulp of the largest |p|^2 bound
        float m = 0.f;
        for (int t = lane; t < ntiles; t += 64) {
            const float4 lo = bb[2 * t], hi = bb[2 * t + 1];
            const float ax = fmaxf(fabsf(lo.x), fabsf(hi.x)), ay = fmaxf(fabsf(lo.y), fabsf(hi.y)),
                        az = fmaxf(fabsf(lo.z), fabsf(hi.z));
            m = fmaxf(m, ax * ax + ay * ay + az * az);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if (lane == 0) *s_margin = m * (256.0f * 5.9604645e-08f);
    }
    __syncthreads();
    const float margin = *s_margin;
    for (int o = lane; o < MASKW; o += 64) hitmask[o] = 0u;  // wave-private: no barrier needed

    const int i = blockIdx.x * KNN_THREADS + tid;
    const bool valid = i < n;
    float xi = 0.f, yi = 0.f, zi = 0.f, sqi = 0.f;
    if (valid) {
        const float4 me = cand[i];
        xi = me.x;
        yi = me.y;
        zi = me.z;
        sqi = me.w;
    }
    static_assert(KSEL == 20, "topk_insert_key is written for the 20-slot list");
    float top[KSEL];  // the KSEL smallest d' = -a_ij seen, ascending: top[KSEL-1] is the current threshold
    // (a lane past the end of the cloud starts from -inf: no bound and no candidate is ever <= its threshold, and a value pushed
    // through its network leaves the list as it is -- so the scan loops need no `valid` test)
#pragma unroll
    for (int s = 0; s < KSEL; ++s) top[s] = valid ? INFINITY : -INFINITY;

    auto lower_bound = [&](int c) {
        const float4 lo = bb[2 * c], hi = bb[2 * c + 1];
        const float dx = fmaxf(fmaxf(lo.x - xi, xi - hi.x), 0.f);
        const float dy = fmaxf(fmaxf(lo.y - yi, yi - hi.y), 0.f);
        const float dz = fmaxf(fmaxf(lo.z - zi, zi - hi.z), 0.f);
        return dx * dx + dy * dy + dz * dz - margin;
    };
    // BATCH candidates at a time: the LDS reads are issued together, one wave vote decides whether any lane has
    // anything to insert (almost never once the threshold has tightened).
    // d'_ij = (sq_i + -2*inner) + sq_j = -a_ij (negation is exact, so all comparisons are done on d').
    auto pos_sq_dist = [&](const float4& q) {
#pragma clang fp contract(off)
        const float inner = (xi * q.x + yi * q.y) + zi * q.z;
        // sq_i + (-2 * inner): the product by -2 is exact, so ONE fused instruction rounds exactly like the reference's
        // separate multiply and add (one instruction less per candidate)
        return __builtin_fmaf(-2.0f, inner, sqi) + q.w;
    };
#define thr top[KSEL - 1]   // current 20th smallest d'; candidates must be strictly below it
    auto scan1 = [&](int c) {
        KSTAT(0);
        if (!wave_any(lower_bound(c) <= thr)) return;
        KSTAT(1);
        const float4* tp = cand + c * KNN_CT;
#pragma unroll
        for (int k0 = 0; k0 < KNN_CT; k0 += BATCH) {
            float4 q[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) q[u] = tp[k0 + u];
            float d[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) d[u] = pos_sq_dist(q[u]);
            // one compare on the batch's smallest d' (v_min3_f32 in asm: four instructions for eight values, no canonicalising
            // v_max from fminf) instead of eight compares and eight mask ORs.  Non-strict: a batch without hits holds no
            // member of any lane's final set.  (Lanes past the cloud's end hold a list of -inf: nothing is ever <= it.)
            bool hit;
            if constexpr (BATCH == 8) {
                float dmin;
                asm("v_min3_f32 %0, %1, %2, %3\n\tv_min3_f32 %0, %0, %4, %5\n\tv_min3_f32 %0, %0, %6, %7\n\tv_min_f32 %0, %0, %8"
                    : "=&v"(dmin)
                    : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]), "v"(d[4]), "v"(d[5]), "v"(d[6]), "v"(d[7]));
                hit = dmin <= thr;
            } else if constexpr (BATCH == 4) {
                float dmin;
                asm("v_min3_f32 %0, %1, %2, %3\n\tv_min_f32 %0, %0, %4" : "=&v"(dmin) : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]));
                hit = dmin <= thr;
            } else {
                hit = false;
#pragma unroll
                for (int u = 0; u < BATCH; ++u) hit |= d[u] <= thr;
            }
            if (wave_any(hit)) {
                KSTAT(2);
                if (lane == 0) {
                    const int bit = c * BPT + k0 / BATCH;
                    hitmask[bit >> 5] |= 1u << (bit & 31);
                }
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    // wave-UNIFORM branch; every lane pushes its d': one that is not below the lane's threshold is >= all 20
                    // entries and falls straight through the network (min leaves each slot as it is), so no masking.
                    // (A per-lane `if` would make every slot a conditional update: +1 v_mov per slot to merge the paths.)
                    if (wave_any(d[u] < top[KSEL - 1])) {
                        KSTAT(3);
                        topk_insert_dist(top, d[u]);
                    }
                }
            }
        }
    };

    // ---- pass 1: own tiles first, then outwards ----
    const int t0 = (blockIdx.x * KNN_THREADS + wave * 64) / KNN_CT;  // wave-uniform
    constexpr int OWN = 64 / KNN_CT;  // tiles covered by the wave's own 64 queries
#pragma unroll
    for (int o = 0; o < OWN; ++o)
        if (t0 + o < ntiles) scan1(t0 + o);
    for (int d = 1; d < ntiles; ++d) {
        const int cl = t0 - d, cr = t0 + OWN - 1 + d;
        if (cl >= 0 && cl < ntiles) scan1(cl);
        if (cr < ntiles) scan1(cr);
    }
#undef thr
    const float kth = 0.0f - top[KSEL - 1];   // a = -d' (exact); +0 for coincident points

    // ---- pass 2: emit {j : a_ij >= kth} = {j : d'_ij <= -kth} ascending ----
    // The lists are 4-byte or (idx_u16, wave-uniform: the fused pipeline's format, half the list traffic of this kernel and
    // of every block kernel) 2-byte entries; the loop is instantiated for each so that no format test sits on the emit path.
    // A lane's slot address is its row start (a 32-bit byte offset from the cloud's lists, wave-uniform base) + the count.
    unsigned int count = 0;
    char* lists = reinterpret_cast<char*>(idx) + (size_t)cloud * n * cap * (idx_u16 ? 2 : 4);
    const float dk = -kth;
    const unsigned int ucap = (unsigned int)cap;
    // every member of a final set was a (non-strict) hit when pass 1 saw it (thresholds only tighten), so only the
    // batches flagged in the wave's hit mask are re-visited, in ascending order -- no bounding-box tests here
    auto pass2 = [&](auto wide) {
        constexpr int ESZ = decltype(wide)::value ? 4 : 2;
        const unsigned int row = (unsigned int)(valid ? i : 0) * ucap * ESZ;
        for (int c = 0; c < ntiles; ++c) {
            const int bit0 = c * BPT;
            const unsigned int bits = (hitmask[bit0 >> 5] >> (bit0 & 31)) & ((1u << BPT) - 1u);
            if (bits == 0u) continue;
            KSTAT(4);
            const float4* tp = cand + c * KNN_CT;
#pragma unroll
            for (int k0 = 0; k0 < KNN_CT; k0 += BATCH) {
                if (!((bits >> (k0 / BATCH)) & 1u)) continue;
                float4 q[BATCH];
#pragma unroll
                for (int u = 0; u < BATCH; ++u) q[u] = tp[k0 + u];
                float d[BATCH];
                bool hit = false;
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    d[u] = pos_sq_dist(q[u]);
                    hit |= d[u] <= dk;
                }
                if (wave_any(hit)) {
                    KSTAT(5);
#pragma unroll
                    for (int u = 0; u < BATCH; ++u)
                        if (d[u] <= dk) {
                            if (valid && count < ucap) {
                                char* slot = lists + (row + count * ESZ);
                                if (ESZ == 4)
                                    *reinterpret_cast<int32_t*>(slot) = c * KNN_CT + k0 + u;
                                else
                                    *reinterpret_cast<unsigned short*>(slot) = (unsigned short)(c * KNN_CT + k0 + u);
                            }
                            ++count;
                        }
                }
            }
        }
    };
    if (idx_u16)
        pass2(std::false_type{});
    else
        pass2(std::true_type{});
    // A row can be short only when a coordinate is NaN / Inf (a_ij >= kth is then false for the pairs involved): its unused
    // slots get the point's own index so that the 20 entries every consumer reads are valid row numbers (the cloud's
    // descriptor is NaN anyway: EPC_STATUS_NONFINITE_INPUT).
    if (valid && count < (unsigned int)KSEL) {
        const size_t row = ((size_t)cloud * n + i) * cap;
        for (unsigned int c = count; c < (unsigned int)KSEL; ++c) {
            if (idx_u16)
                reinterpret_cast<unsigned short*>(idx)[row + c] = (unsigned short)i;
            else
                idx[row + c] = i;
        }
    }
    if (valid) {
        cnt[(size_t)cloud * n + i] = (int32_t)count;
        kth_out[(size_t)cloud * n + i] = kth;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// knn_topk_quad_kernel (round 4): FOUR lanes per query.  The culled kernel above is one query per lane: a wave's time is its
// own dependent chain (LDS read -> distance -> compare -> vote -> branch, ~18 k vector instructions), whatever the grid; a
// small grid (the 18 clouds of a training tuple: 72 workgroups on 256 CUs) waits for that chain with three quarters of the
// chip idle.  Here a wave holds 16 queries x 4 lanes; lane `sub` of a query's quad evaluates the candidates 4u + sub of a
// 32-point tile (consecutive 16-byte LDS addresses per quad: conflict-free) and keeps the 20 smallest d' of ITS quarter of the
// candidates.  Same selection, bit for bit:
//   * threshold.  thr_q = the LARGEST of the four lists' fifth values (two DPP exchanges): 4 x 5 candidates lie at or below it,
//     so thr_q >= the final 20th distance kth at all times.  A candidate with d' >= thr_q is not needed: at that moment every
//     list holds five values <= thr_q, values a list only ever replaces by smaller ones, so the union keeps 20 values <= d'.  A
//     candidate with d' < kth is always pushed (kth <= thr_q) and never evicted (a list's 20th value cannot fall below kth, the
//     20th of a superset).  Hence the union of the four lists always contains the 20 smallest.  Tiles and batches are culled
//     against thr_q (non-strict, for pass 2), insertion is attempted when d' < thr_q.
//   * kth = the 20th smallest of the union: after the scan the lanes of a pair push each other's list through their own
//     network (both then hold the pair's 20 smallest), then the pairs do the same -- 40 pushes at most, cut short wave-wide as
//     soon as no lane's list would change (lists are ascending).
//   * pass 2 re-visits the flagged batches in ascending order; the four lanes' hits of one u are consecutive j, so a lane's slot
//     is the query's running count + the number of hits on lower lanes of its quad (one ballot per u).
//   * box tests in two levels, four per vote: lane `sub` tests the sub-th of the next four SUPER-tiles (boxes of four consecutive
//     tiles) of the outward walk; a super-tile that passes has its four tiles tested by one more vote (lane `sub` the sub-th tile).
//     ~20 votes per wave instead of the culled kernel's 128; a test may use a threshold that is a few scans old -- conservative
//     (thresholds only tighten).
//   * the batches that produced a hit are flagged in eight wave-uniform 64-bit words (scalar registers); pass 2 walks the set bits.
// A wave's 16 queries are neighbours on the Hilbert curve, a tighter group than 64: fewer tiles pass the vote.
// ---------------------------------------------------------------------------------------------------------------------
#define KQ_LANES 4

// DPP reads of a register need two wait states after the VALU write of it.  The compiler's hazard recogniser does not look
// inside asm blocks -- and the lists are written by one (topk_insert_dist) -- so the exchanges are asm with their own s_nop.
__device__ __forceinline__ float quad_max(float v) {
    float r;   // volatile: must not be sunk into a divergent region (a lane's partners have to be active when it reads them)
    asm volatile("s_nop 1\n\t"
        "v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\t"
        "v_max_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf"
        : "=&v"(r)
        : "v"(v));
    return r;
}
template <int STEP>
__device__ __forceinline__ float dpp_quad_swap(float v) {   // the value of lane ^ 1 (STEP 0) / lane ^ 2 (STEP 1)
    float r;
    if (STEP == 0)
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    else
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v));
    return r;
}

template <int KSEL, bool CONV1, int HMW>   // HMW: 64-bit hit-mask words, 2 bits per tile (4: up to 128 tiles, 8: up to 256)
__global__ __launch_bounds__(KNN_THREADS, 8) void knn_topk_quad_kernel(const float* __restrict__ xyz, int n, int cap,
                                                                       int32_t* __restrict__ idx, int32_t* __restrict__ cnt,
                                                                       float* __restrict__ kth_out,
                                                                       const float* __restrict__ conv1_pack,
                                                                       float* __restrict__ x32, unsigned short* __restrict__ x16,
                                                                       int idx_u16, int32_t* __restrict__ status, int rounds) {
    static_assert(KNN_CT == 32 && KSEL == 20, "written for 32-point tiles and the 20-slot list");
    extern __shared__ __attribute__((aligned(16))) float4 cand[];  // [npad] points, then 2 float4 per tile (lo, hi)
    const int ntiles = (n + KNN_CT - 1) / KNN_CT;
    const int npad = ntiles * KNN_CT;
    float4* bb = cand + npad;
    float* s_margin = reinterpret_cast<float*>(bb + 2 * ntiles);
    // one bit per (tile, 16-candidate batch) that produced a hit in pass 1: 2 x 256 tiles at most = eight wave-uniform 64-bit words
    // (scalar registers: set by scalar selects, walked by s_ff1 in pass 2 -- no LDS traffic, no loop over unflagged tiles)
    static_assert(HMW * 32 * KNN_CT <= KNN_LDS_MAX_N, "mask words beyond the largest cloud");
    const int cloud = blockIdx.y;
    // (readfirstlane: the wave index is uniform, which the compiler cannot tell from threadIdx -- tile numbers, tile addresses and
    // the hit mask then live in scalar registers)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), sub = lane & 3;
    const int nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int wg_queries = nwaves * 16 * rounds;   // the host sizes the workgroup and the number of 16-query groups per wave (launch_knn)
    const float* pc = xyz + (size_t)cloud * n * 3;

    bool bad = false;
    for (int j = tid; j < npad; j += nthreads) {
        float4 v = make_float4(0.f, 0.f, 0.f, INFINITY);
        if (j < n) {
            v.x = pc[3 * j + 0];
            v.y = pc[3 * j + 1];
            v.z = pc[3 * j + 2];
            v.w = sq3(v.x, v.y, v.z);
            bad |= !(v.w <= 3.4028234664e38f);
        }
        cand[j] = v;
    }
    if (status && blockIdx.x == 0 && wave_any(bad) && lane == 0) atomicOr(status + cloud, EPC_STATUS_NONFINITE_INPUT);
    __syncthreads();
    if constexpr (CONV1) {   // conv1 of the workgroup's own points (as in the culled kernel)
        const int q = tid & 15;
        const float4 w0 = *reinterpret_cast<const float4*>(conv1_pack + 4 * q);
        const float4 w1 = *reinterpret_cast<const float4*>(conv1_pack + 64 + 4 * q);
        const float4 w2 = *reinterpret_cast<const float4*>(conv1_pack + 128 + 4 * q);
        const float4 b = *reinterpret_cast<const float4*>(conv1_pack + 192 + 4 * q);
        const int p0 = blockIdx.x * wg_queries;
        const int np = min(wg_queries, n - p0);
        bool ovf = false;
        for (int t = tid; t < np * 16; t += nthreads) {
            const int g = p0 + (t >> 4);
            const float4 pt = cand[g];
            const float4 y = conv1_quad(pt.x, pt.y, pt.z, w0, w1, w2, b);
            const size_t row = (size_t)cloud * n + g;
            if (x32) *reinterpret_cast<float4*>(x32 + row * 64 + 4 * q) = y;
            if (x16) {
                reinterpret_cast<uint2*>(x16)[row * 16 + q] = pack_half4(y);
                ovf |= fmaxf(fmaxf(y.x, y.y), fmaxf(y.z, y.w)) > 65504.0f;
            }
        }
        if (status && x16 && wave_any(ovf) && lane == 0) atomicOr(status + cloud, EPC_STATUS_FP16_RANGE);
    }
    for (int t = wave * 2 + (lane >> 5); t < ntiles; t += nwaves * 2) {
        const float4 v = cand[t * KNN_CT + (lane & 31)];
        const bool ok = v.w != INFINITY;
        float lx = ok ? v.x : INFINITY, ly = ok ? v.y : INFINITY, lz = ok ? v.z : INFINITY;
        float hx = ok ? v.x : -INFINITY, hy = ok ? v.y : -INFINITY, hz = ok ? v.z : -INFINITY;
#pragma unroll
        for (int off = KNN_CT / 2; off >= 1; off >>= 1) {
            lx = fminf(lx, __shfl_xor(lx, off));
            ly = fminf(ly, __shfl_xor(ly, off));
            lz = fminf(lz, __shfl_xor(lz, off));
            hx = fmaxf(hx, __shfl_xor(hx, off));
            hy = fmaxf(hy, __shfl_xor(hy, off));
            hz = fmaxf(hz, __shfl_xor(hz, off));
        }
        if ((lane & 31) == 0) {
            bb[2 * t] = make_float4(lx, ly, lz, 0.f);
            bb[2 * t + 1] = make_float4(hx, hy, hz, 0.f);
        }
    }
    __syncthreads();
    // boxes of four consecutive tiles ("super-tiles"): the walk tests those first
    const int nst = (ntiles + 3) >> 2;
    float4* sbb = reinterpret_cast<float4*>(s_margin + 4);   // [2 * nst]
    for (int t = tid; t < nst; t += nthreads) {
        float4 lo = bb[8 * t], hi = bb[8 * t + 1];
        for (int r = 1; r < 4 && 4 * t + r < ntiles; ++r) {
            const float4 l = bb[8 * t + 2 * r], h = bb[8 * t + 2 * r + 1];
            lo.x = fminf(lo.x, l.x), lo.y = fminf(lo.y, l.y), lo.z = fminf(lo.z, l.z);
            hi.x = fmaxf(hi.x, h.x), hi.y = fmaxf(hi.y, h.y), hi.z = fmaxf(hi.z, h.z);
        }
        sbb[2 * t] = lo, sbb[2 * t + 1] = hi;
    }
    if (wave == 0) {
        float m = 0.f;
        for (int t = lane; t < ntiles; t += 64) {
            const float4 lo = bb[2 * t], hi = bb[2 * t + 1];
            const float ax = fmaxf(fabsf(lo.x), fabsf(hi.x)), ay = fmaxf(fabsf(lo.y), fabsf(hi.y)),
                        az = fmaxf(fabsf(lo.z), fabsf(hi.z));
            m = fmaxf(m, ax * ax + ay * ay + az * az);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if (lane == 0) *s_margin = m * (256.0f * 5.9604645e-08f);
    }
    __syncthreads();
    const float margin = *s_margin;

    // the image above is built once per workgroup; a wave then takes `rounds` groups of 16 queries, one after the other
    for (int round = 0; round < rounds; ++round) {
    const int i0 = blockIdx.x * wg_queries + (round * nwaves + wave) * 16;   // the group's first query (wave-uniform)
    if (i0 >= n) break;
    const int i = i0 + (lane >> 2);
    const bool valid = i < n;
    unsigned long long hm[HMW];
#pragma unroll
    for (int w = 0; w < HMW; ++w) hm[w] = 0ull;
    float xi = 0.f, yi = 0.f, zi = 0.f, sqi = 0.f;
    if (valid) {
        const float4 me = cand[i];
        xi = me.x, yi = me.y, zi = me.z, sqi = me.w;
    }
    float top[KSEL];
#pragma unroll
    for (int s = 0; s < KSEL; ++s) top[s] = valid ? INFINITY : -INFINITY;

    auto lower_bound = [&](const float4* boxes, int c) {
        const float4 lo = boxes[2 * c], hi = boxes[2 * c + 1];
        const float dx = fmaxf(fmaxf(lo.x - xi, xi - hi.x), 0.f);
        const float dy = fmaxf(fmaxf(lo.y - yi, yi - hi.y), 0.f);
        const float dz = fmaxf(fmaxf(lo.z - zi, zi - hi.z), 0.f);
        return dx * dx + dy * dy + dz * dz - margin;
    };
    auto pos_sq_dist = [&](const float4& q) {
#pragma clang fp contract(off)
        const float inner = (xi * q.x + yi * q.y) + zi * q.z;
        return __builtin_fmaf(-2.0f, inner, sqi) + q.w;
    };
    // An upper bound of the query's 20th distance from the four lists: the largest of their FIFTH values -- 4 x 5 candidates lie at
    // or below it.  (The smallest of their 20th values is a bound too, and a loose one: a list sees a quarter of the candidates, its
    // 20th value is about the query's 80th; with it alone a wave scanned 57 tiles instead of 17.)
    auto thr_q = [&]() { return quad_max(top[4]); };
    auto scan1 = [&](int c) {
        KSTAT(1);
        const float4* tp = cand + c * KNN_CT + sub;
#pragma unroll
        for (int ub = 0; ub < 2; ++ub) {
            float4 q[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) q[u] = tp[16 * ub + 4 * u];
            float d[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) d[u] = pos_sq_dist(q[u]);
            float dmin;
            asm("v_min3_f32 %0, %1, %2, %3\n\tv_min_f32 %0, %0, %4" : "=&v"(dmin) : "v"(d[0]), "v"(d[1]), "v"(d[2]), "v"(d[3]));
            float thr = thr_q();
            if (wave_any(dmin <= thr)) {
                KSTAT(2);
                const int bit = c * 2 + ub;
                const unsigned long long bv = 1ull << (bit & 63);
#pragma unroll
                for (int w = 0; w < HMW; ++w) hm[w] |= (bit >> 6) == w ? bv : 0ull;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (wave_any(d[u] < thr)) {
                        KSTAT(3);
                        topk_insert_dist(top, d[u]);
                        thr = thr_q();
                    }
                }
            }
        }
    };

    // ---- pass 1: outwards from the wave's own super-tile (four tiles); a vote tests four super-tiles (lane `sub` the sub-th of
    // them), a super-tile that passes has its four tiles tested by one more vote, and the tiles that pass are scanned ----
    const int t0 = i0 / KNN_CT;   // wave-uniform: the tile that holds the wave's 16 queries
    const int T0 = t0 >> 2;
    const int kmax = 2 * max(T0, nst - 1 - T0);   // visiting order k -> super-tile: T0, T0-1, T0+1, T0-2, T0+2, ...
    auto super_of = [&](int k) { return T0 + ((k & 1) ? -((k + 1) >> 1) : (k >> 1)); };
    for (int kk = 0; kk <= kmax; kk += 4) {
        const int S = super_of(kk + sub);
        const bool in_s = S >= 0 && S < nst;
        KSTAT(0);
        const float lbs = in_s ? lower_bound(sbb, S) : INFINITY;   // (a list that is not full yet has threshold +inf: test `in` too)
        const float thr_s = thr_q();
        const unsigned long long ms = __builtin_amdgcn_ballot_w64(in_s && lbs <= thr_s);
#pragma unroll 1
        for (int s = 0; s < 4; ++s) {
            if (!(ms & (0x1111111111111111ull << s))) continue;
            const int c0 = 4 * super_of(kk + s);
            const bool in = c0 + sub < ntiles;
            KSTAT(0);
            const float lb = in ? lower_bound(bb, c0 + sub) : INFINITY;
            const float thr = thr_q();
            const unsigned long long m = __builtin_amdgcn_ballot_w64(in && lb <= thr);
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (m & (0x1111111111111111ull << r)) scan1(c0 + r);
        }
    }
    // ---- the quad's four lists -> the 20th smallest of their union, in every lane ----
    // The lanes of a pair first: c[s] = min(X[s], Y[19 - s]) holds the 20 smallest of two ascending lists X, Y (the lower half of a
    // bitonic merge), rising then falling.  Forty compare-exchanges sort it: the bitonic merge network for 32 values with the twelve
    // -inf pads behind the 20 real ones folded away at compile time (an exchange with a pad is a move, i.e. a renaming).  120
    // instructions without a vote or a branch -- pushing the partner's list through the insertion network took ~450 with twenty of
    // each.  (Both lanes of a pair compute the same multiset: s <-> 19 - s.)
    {
        float c[KSEL];
#pragma unroll
        for (int s = 0; s < KSEL; ++s) {
            const float o = dpp_quad_swap<0>(top[KSEL - 1 - s]);
            asm("v_min_f32 %0, %1, %2" : "=v"(c[s]) : "v"(top[s]), "v"(o));
        }
        constexpr unsigned char CE[40][2] = {
            {0, 16}, {1, 17}, {2, 18},  {3, 19},  {16, 8},  {17, 9},  {18, 10}, {19, 11}, {4, 12},  {5, 13},
            {6, 14}, {7, 15}, {16, 4},  {17, 5},  {18, 6},  {19, 7},  {8, 12},  {9, 13},  {10, 14}, {11, 15},
            {0, 2},  {1, 3},  {16, 18}, {17, 19}, {4, 6},   {5, 7},   {8, 10},  {9, 11},  {12, 14}, {13, 15},
            {0, 1},  {2, 3},  {16, 17}, {18, 19}, {4, 5},   {6, 7},   {8, 9},   {10, 11}, {12, 13}, {14, 15}};
#pragma unroll
        for (int e = 0; e < 40; ++e) {
            float lo, hi;
            asm("v_min_f32 %0, %2, %3\n\tv_max_f32 %1, %2, %3" : "=&v"(lo), "=&v"(hi) : "v"(c[CE[e][0]]), "v"(c[CE[e][1]]));
            c[CE[e][0]] = lo, c[CE[e][1]] = hi;
        }
        constexpr unsigned char ORDER[KSEL] = {0, 1, 2, 3, 16, 17, 18, 19, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15};   // ascending
#pragma unroll
        for (int s = 0; s < KSEL; ++s) top[s] = c[ORDER[s]];
    }
    // the two pairs: only the 20th smallest of the union is needed, and for two ascending lists X, Y that is
    // max over s of min(X[s], Y[19 - s]) (the lower half of a bitonic merge) -- 20 exchanges, no votes, no network
    float top20 = -INFINITY;
#pragma unroll
    for (int s = 0; s < KSEL; ++s) {
        const float o = dpp_quad_swap<1>(top[KSEL - 1 - s]);
        float m;
        asm("v_min_f32 %0, %1, %2" : "=v"(m) : "v"(top[s]), "v"(o));
        asm("v_max_f32 %0, %1, %2" : "=v"(top20) : "v"(top20), "v"(m));
    }
    const float kth = 0.0f - top20;

    // ---- pass 2: emit {j : d'_ij <= -kth} ascending ----
    unsigned int count = 0;
    char* lists = reinterpret_cast<char*>(idx) + (size_t)cloud * n * cap * (idx_u16 ? 2 : 4);
    const float dk = -kth;
    const unsigned int ucap = (unsigned int)cap;
    const unsigned int qshift = lane & ~3u, below = (1u << sub) - 1u;
    auto pass2 = [&](auto wide) {
        constexpr int ESZ = decltype(wide)::value ? 4 : 2;
        const unsigned int row = (unsigned int)(valid ? i : 0) * ucap * ESZ;
        const int words = (2 * ntiles + 63) >> 6;
        for (int w = 0; w < words; ++w) {
            unsigned long long flagged = hm[0];
#pragma unroll
            for (int v = 1; v < HMW; ++v) flagged = w == v ? hm[v] : flagged;
            while (flagged) {
                const int bit = __builtin_ctzll(flagged);
                flagged &= flagged - 1ull;
                const int c = (w * 64 + bit) >> 1, ub = bit & 1;
                KSTAT(4);
                const float4* tp = cand + c * KNN_CT + sub;
                float4 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = tp[16 * ub + 4 * u];
                float d[4];
                bool hit = false;
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    d[u] = pos_sq_dist(q[u]);
                    hit |= d[u] <= dk;
                }
                if (wave_any(hit)) {
                    KSTAT(5);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool h = d[u] <= dk;
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(h);
                        if (m == 0ull) continue;
                        const unsigned int qb = (unsigned int)(m >> qshift) & 15u;
                        const unsigned int pos = count + __builtin_popcount(qb & below);
                        if (h && valid && pos < ucap) {
                            char* slot = lists + (row + pos * ESZ);
                            const int j = c * KNN_CT + 16 * ub + 4 * u + sub;
                            if (ESZ == 4)
                                *reinterpret_cast<int32_t*>(slot) = j;
                            else
                                *reinterpret_cast<unsigned short*>(slot) = (unsigned short)j;
                        }
                        count += __builtin_popcount(qb);
                    }
                }
            }
        }
    };
    if (idx_u16)
        pass2(std::false_type{});
    else
        pass2(std::true_type{});
    if (valid && sub == 0) {
        if (count < (unsigned int)KSEL) {   // NaN / Inf coordinates only (see the culled kernel)
            const size_t row = ((size_t)cloud * n + i) * cap;
            for (unsigned int c = count; c < (unsigned int)KSEL; ++c) {
                if (idx_u16)
                    reinterpret_cast<unsigned short*>(idx)[row + c] = (unsigned short)i;
                else
                    idx[row + c] = i;
            }
        }
        cnt[(size_t)cloud * n + i] = (int32_t)count;
        kth_out[(size_t)cloud * n + i] = kth;
    }
    }   // round
}

#ifdef KNN_COLLECT   // built and measured in round 2, NOT the default: see the verdict at the end of this comment
// ---------------------------------------------------------------------------------------------------------------------
// knn_collect_kernel (round 2): bound, collect, select.  Same inputs / outputs / bit-exact semantics as the culled kernel
// above, without the insertion network inside the candidate scan:
//   pass 0  the wave's own two tiles through the 20-slot network (all 64 lanes useful): B_i = the 20th smallest d' among
//           the query's 64 Hilbert neighbours -- an upper bound of its true 20th distance;
//   pass 1  every tile in ASCENDING order, culled by its bounding box against the FIXED bounds B_i; a candidate with
//           d' <= B_i is appended to the lane's list (u16 indices in LDS, [slot][thread]: conflict-free) -- one compare, one
//           predicated ds_write and an add instead of a 40-instruction network pass;
//           when a lane's list is nearly full the wave compacts: the exact 20th smallest of every lane's list (the network,
//           run once over the stored candidates) becomes its new, tighter B_i -- still an upper bound: the 20th smallest of
//           a subset -- and entries above it are dropped in place;
//   select  the network once more over the final lists: kth; emit {j : d' <= kth} in list order (= ascending j), count.
// A lane whose list cannot shrink below the trigger (more than that many candidates TIED at or below its 20th distance:
// duplicated points, zero padding) is "saturated"; a wave with a saturated lane falls back to the exact un-culled two-scan
// (network over all tiles, then count / emit), which is what such clouds cost in the culled kernel as well.
// One workgroup handles CHUNKS consecutive groups of THREADS queries with one copy of the cloud in LDS.
//
// MEASURED (64 Hilbert-ordered 4096-point clouds, scripts/tune_knn_collect.sh; bit-identical lists in all 64 parity / golden
// tests): 0.478 ms at 512 threads x 2 chunks x 64 slots, 0.542 at 1024 x 1 x 40, 0.511 at 512 x 1 x 64, 0.788 at 256 x 4 x 64,
// against 0.234 ms for the culled kernel above.  Per wave (KNN_STATS): 31.6 tiles scanned (26.6 adaptive), 66 hit batches,
// 3.1 compactions and 218 list entries through the network -- the bound from 64 Hilbert neighbours is loose enough that the
// 56-entry trigger fires three times per wave, and a compaction (recompute + network + filter per entry) costs what ~90 of
// the old network passes cost; with the lists in LDS only 2 waves per SIMD fit beside the 64-KB cloud image (the culled kernel
// runs 4 and is VALU-throughput-bound there).  Tightening the bound first with the adaptive network on the nearest tiles
// turns this into the culled kernel plus ~10 %.  Kept for reference behind -DKNN_COLLECT.
// ---------------------------------------------------------------------------------------------------------------------
#ifndef KC_THREADS
#define KC_THREADS 512
#endif
#ifndef KC_CHUNKS
#define KC_CHUNKS 2
#endif
#ifndef KC_CAP
#define KC_CAP 64
#endif
#define KC_TRIG (KC_CAP - KNN_BATCH)   // compaction trigger: a batch appends at most KNN_BATCH entries

template <int KSEL, bool CONV1>
__global__ __launch_bounds__(KC_THREADS) void knn_collect_kernel(const float* __restrict__ xyz, int n, int cap,
                                                                 int32_t* __restrict__ idx, int32_t* __restrict__ cnt,
                                                                 float* __restrict__ kth_out,
                                                                 const float* __restrict__ conv1_pack,
                                                                 float* __restrict__ x32, unsigned short* __restrict__ x16,
                                                                 int idx_u16, int32_t* __restrict__ status) {
    static_assert(KSEL == 20, "topk_insert_dist is written for the 20-slot list");
    extern __shared__ __attribute__((aligned(16))) float4 cand[];  // [npad] points, 2 float4 per tile, the margin, the lists
    const int ntiles = (n + KNN_CT - 1) / KNN_CT;
    const int npad = ntiles * KNN_CT;
    float4* bb = cand + npad;
    float* s_margin = reinterpret_cast<float*>(bb + 2 * ntiles);
    unsigned short* lists = reinterpret_cast<unsigned short*>(s_margin + 4);   // [KC_CAP][KC_THREADS]
    const int cloud = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int WAVES = KC_THREADS / 64;
    const float* pc = xyz + (size_t)cloud * n * 3;

    bool bad = false;
    for (int j = tid; j < npad; j += KC_THREADS) {
        float4 v = make_float4(0.f, 0.f, 0.f, INFINITY);
        if (j < n) {
            v.x = pc[3 * j + 0];
            v.y = pc[3 * j + 1];
            v.z = pc[3 * j + 2];
            v.w = sq3(v.x, v.y, v.z);
            bad |= !(v.w <= 3.4028234664e38f);
        }
        cand[j] = v;
    }
    if (status && blockIdx.x == 0 && wave_any(bad) && lane == 0) atomicOr(status + cloud, EPC_STATUS_NONFINITE_INPUT);
    __syncthreads();
    const int p0 = blockIdx.x * (KC_THREADS * KC_CHUNKS);   // first point of this workgroup
    if constexpr (CONV1) {
        const int q = tid & 15;
        const float4 w0 = *reinterpret_cast<const float4*>(conv1_pack + 4 * q);
        const float4 w1 = *reinterpret_cast<const float4*>(conv1_pack + 64 + 4 * q);
        const float4 w2 = *reinterpret_cast<const float4*>(conv1_pack + 128 + 4 * q);
        const float4 b = *reinterpret_cast<const float4*>(conv1_pack + 192 + 4 * q);
        const int np = min(KC_THREADS * KC_CHUNKS, n - p0);
        bool ovf = false;
        for (int t = tid; t < np * 16; t += KC_THREADS) {
            const int g = p0 + (t >> 4);
            const float4 pt = cand[g];
            const float4 y = conv1_quad(pt.x, pt.y, pt.z, w0, w1, w2, b);
            const size_t row = (size_t)cloud * n + g;
            if (x32) *reinterpret_cast<float4*>(x32 + row * 64 + 4 * q) = y;
            if (x16) {
                reinterpret_cast<uint2*>(x16)[row * 16 + q] = pack_half4(y);
                ovf |= fmaxf(fmaxf(y.x, y.y), fmaxf(y.z, y.w)) > 65504.0f;
            }
        }
        if (status && x16 && wave_any(ovf) && lane == 0) atomicOr(status + cloud, EPC_STATUS_FP16_RANGE);
    }
    for (int t = wave * (64 / KNN_CT) + lane / KNN_CT; t < ntiles; t += WAVES * (64 / KNN_CT)) {
        const float4 v = cand[t * KNN_CT + (lane % KNN_CT)];
        const bool ok = v.w != INFINITY;
        float lx = ok ? v.x : INFINITY, ly = ok ? v.y : INFINITY, lz = ok ? v.z : INFINITY;
        float hx = ok ? v.x : -INFINITY, hy = ok ? v.y : -INFINITY, hz = ok ? v.z : -INFINITY;
#pragma unroll
        for (int off = KNN_CT / 2; off >= 1; off >>= 1) {
            lx = fminf(lx, __shfl_xor(lx, off));
            ly = fminf(ly, __shfl_xor(ly, off));
            lz = fminf(lz, __shfl_xor(lz, off));
            hx = fmaxf(hx, __shfl_xor(hx, off));
            hy = fmaxf(hy, __shfl_xor(hy, off));
            hz = fmaxf(hz, __shfl_xor(hz, off));
        }
        if ((lane % KNN_CT) == 0) {
            bb[2 * t] = make_float4(lx, ly, lz, 0.f);
            bb[2 * t + 1] = make_float4(hx, hy, hz, 0.f);
        }
    }
    __syncthreads();
    if (wave == 0) {
        float m = 0.f;
        for (int t = lane; t < ntiles; t += 64) {
            const float4 lo = bb[2 * t], hi = bb[2 * t + 1];
            const float ax = fmaxf(fabsf(lo.x), fabsf(hi.x)), ay = fmaxf(fabsf(lo.y), fabsf(hi.y)),
                        az = fmaxf(fabsf(lo.z), fabsf(hi.z));
            m = fmaxf(m, ax * ax + ay * ay + az * az);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if (lane == 0) *s_margin = m * (256.0f * 5.9604645e-08f);
    }
    __syncthreads();
    const float margin = *s_margin;
    unsigned short* mylist = lists + tid;   // slot e at mylist[e * KC_THREADS]

    for (int chunk = 0; chunk < KC_CHUNKS; ++chunk) {
        const int wave_first = p0 + chunk * KC_THREADS + wave * 64;   // wave-uniform
        if (wave_first >= n) break;
        const int i = wave_first + lane;
        const bool valid = i < n;
        float xi = 0.f, yi = 0.f, zi = 0.f, sqi = 0.f;
        if (valid) {
            const float4 me = cand[i];
            xi = me.x, yi = me.y, zi = me.z, sqi = me.w;
        }
        auto lower_bound = [&](int c) {
            const float4 lo = bb[2 * c], hi = bb[2 * c + 1];
            const float dx = fmaxf(fmaxf(lo.x - xi, xi - hi.x), 0.f);
            const float dy = fmaxf(fmaxf(lo.y - yi, yi - hi.y), 0.f);
            const float dz = fmaxf(fmaxf(lo.z - zi, zi - hi.z), 0.f);
            return dx * dx + dy * dy + dz * dz - margin;
        };
        auto pos_sq_dist = [&](const float4& q) {
#pragma clang fp contract(off)
            const float inner = (xi * q.x + yi * q.y) + zi * q.z;
            return __builtin_fmaf(-2.0f, inner, sqi) + q.w;   // (-2 * inner is exact: one rounding, like the reference's mul + add)
        };
        float top[KSEL];
        auto reset_top = [&]() {
#pragma unroll
            for (int s = 0; s < KSEL; ++s) top[s] = INFINITY;
        };
        // a whole tile through the network (pass 0, and the exact fallback)
        auto scan_net = [&](int c) {
            const float4* tp = cand + c * KNN_CT;
#pragma unroll
            for (int k0 = 0; k0 < KNN_CT; k0 += KNN_BATCH) {
                float4 q[KNN_BATCH];
#pragma unroll
                for (int u = 0; u < KNN_BATCH; ++u) q[u] = tp[k0 + u];
                float d[KNN_BATCH];
                bool hit = false;
#pragma unroll
                for (int u = 0; u < KNN_BATCH; ++u) {
                    d[u] = pos_sq_dist(q[u]);
                    hit |= valid && d[u] < top[KSEL - 1];
                }
                if (wave_any(hit)) {
#pragma unroll
                    for (int u = 0; u < KNN_BATCH; ++u)
                        if (wave_any(d[u] < top[KSEL - 1])) topk_insert_dist(top, d[u]);
                }
            }
        };
        // the exact 20 smallest d' of every lane's list -> top[]
        auto select_from_list = [&](int count) {
            reset_top();
            for (int e = 0; wave_any(e < count); ++e) {
                KSTAT(4);
                const bool have = e < count;
                const int j = have ? (int)mylist[e * KC_THREADS] : 0;
                const float d = have ? pos_sq_dist(cand[j]) : INFINITY;
                if (wave_any(d < top[KSEL - 1])) topk_insert_dist(top, d);
            }
        };

        // ---- pass 0: own tiles ----
        reset_top();
        const int t0 = wave_first / KNN_CT;
        constexpr int OWN = 64 / KNN_CT;
#pragma unroll
        for (int o = 0; o < OWN; ++o)
            if (t0 + o < ntiles) scan_net(t0 + o);
        float B = top[KSEL - 1];        // upper bound of the query's 20th smallest d'
        int count = 0;
        bool sat = false;

        // ---- pass 1: collect, ascending ----
        for (int c = 0; c < ntiles; ++c) {
            KSTAT(0);
            if (!wave_any(valid && !sat && lower_bound(c) <= B)) continue;
            KSTAT(1);
            const float4* tp = cand + c * KNN_CT;
#pragma unroll
            for (int k0 = 0; k0 < KNN_CT; k0 += KNN_BATCH) {
                float4 q[KNN_BATCH];
#pragma unroll
                for (int u = 0; u < KNN_BATCH; ++u) q[u] = tp[k0 + u];
                float d[KNN_BATCH];
                bool hit = false;
#pragma unroll
                for (int u = 0; u < KNN_BATCH; ++u) {
                    d[u] = pos_sq_dist(q[u]);
                    hit |= d[u] <= B;
                }
                hit = hit && valid && !sat;
                if (wave_any(hit)) {
                    KSTAT(2);
#pragma unroll
                    for (int u = 0; u < KNN_BATCH; ++u)
                        if (hit && d[u] <= B) {
                            mylist[count * KC_THREADS] = (unsigned short)(c * KNN_CT + k0 + u);
                            ++count;
                        }
                    if (wave_any(count >= KC_TRIG && !sat)) {
                        // compaction: tighten every lane's bound to the 20th smallest of its list, drop what is above it
                        KSTAT(3);
                        select_from_list(count);
                        const float nb = top[KSEL - 1];
                        if (nb < B) B = nb;   // (lanes with fewer than 20 entries keep their bound: top[19] is +Inf there)
                        int kept = 0;
                        for (int e = 0; wave_any(e < count); ++e) {
                            if (e < count) {
                                const int j = (int)mylist[e * KC_THREADS];
                                if (pos_sq_dist(cand[j]) <= B) {
                                    mylist[kept * KC_THREADS] = (unsigned short)j;
                                    ++kept;
                                }
                            }
                        }
                        count = kept;
                        sat = sat || count >= KC_TRIG;   // more than that many candidates tied at / below the 20th distance
                    }
                }
            }
        }

        float kth;
        unsigned int total = 0;
        const unsigned int ucap = (unsigned int)cap;
        char* out_lists = reinterpret_cast<char*>(idx) + (size_t)cloud * n * cap * (idx_u16 ? 2 : 4);
        auto emit = [&](unsigned int slot, int j) {
            if (idx_u16)
                *reinterpret_cast<unsigned short*>(out_lists + ((size_t)i * ucap + slot) * 2) = (unsigned short)j;
            else
                *reinterpret_cast<int32_t*>(out_lists + ((size_t)i * ucap + slot) * 4) = j;
        };
        if (!wave_any(sat)) {
            // ---- select + emit from the lists ----
            select_from_list(count);
            kth = 0.0f - top[KSEL - 1];
            const float dk = top[KSEL - 1];
            for (int e = 0; wave_any(e < count); ++e) {
                if (e < count) {
                    const int j = (int)mylist[e * KC_THREADS];
                    if (pos_sq_dist(cand[j]) <= dk) {
                        if (valid && total < ucap) emit(total, j);
                        ++total;
                    }
                }
            }
        } else {
            // ---- exact fallback for the wave: un-culled two-scan ----
            KSTAT(5);
            reset_top();
            for (int c = 0; c < ntiles; ++c) scan_net(c);
            kth = 0.0f - top[KSEL - 1];
            const float dk = top[KSEL - 1];
            for (int c = 0; c < ntiles; ++c) {
                const float4* tp = cand + c * KNN_CT;
                for (int k = 0; k < KNN_CT; ++k) {
                    const float d = pos_sq_dist(tp[k]);
                    if (d <= dk) {
                        if (valid && total < ucap) emit(total, c * KNN_CT + k);
                        ++total;
                    }
                }
            }
        }
        if (valid && total < (unsigned int)KSEL) {   // NaN / Inf coordinates: pad with the point's own index (see the culled kernel)
            for (unsigned int c = total; c < (unsigned int)KSEL; ++c) emit(c, i);
        }
        if (valid) {
            cnt[(size_t)cloud * n + i] = (int32_t)total;
            kth_out[(size_t)cloud * n + i] = kth;
        }
    }
}

#endif  // KNN_COLLECT

template <int KSEL>
__global__ __launch_bounds__(KNN_THREADS) void knn_topk_stream_kernel(const float* __restrict__ xyz, int n, int cap,
                                                                      int32_t* __restrict__ idx,
                                                                      int32_t* __restrict__ cnt,
                                                                      float* __restrict__ kth_out,
                                                                      int32_t* __restrict__ status) {
    __shared__ float4 tile[KNN_TILE];
    bool bad = false;
    const int cloud = blockIdx.y;
    const int i = blockIdx.x * KNN_THREADS + threadIdx.x;
    const bool valid = i < n;
    const float* pc = xyz + (size_t)cloud * n * 3;
    float xi = 0.f, yi = 0.f, zi = 0.f;
    if (valid) {
        xi = pc[3 * i + 0];
        yi = pc[3 * i + 1];
        zi = pc[3 * i + 2];
    }
    const float sqi = sq3(xi, yi, zi);
    float top[KSEL];
#pragma unroll
    for (int s = 0; s < KSEL; ++s) top[s] = -INFINITY;

    auto load_tile = [&](int t0) {
        __syncthreads();
        for (int c = threadIdx.x; c < KNN_TILE; c += KNN_THREADS) {
            const int j = t0 + c;
            float4 v = make_float4(0.f, 0.f, 0.f, INFINITY);
            if (j < n) {
                v.x = pc[3 * j + 0];
                v.y = pc[3 * j + 1];
                v.z = pc[3 * j + 2];
                v.w = sq3(v.x, v.y, v.z);
                bad |= !(v.w <= 3.4028234664e38f);
            }
            tile[c] = v;
        }
        __syncthreads();
    };

    for (int t0 = 0; t0 < n; t0 += KNN_TILE) {
        load_tile(t0);
        const int lim = min(KNN_TILE, n - t0);
#pragma unroll 4
        for (int c = 0; c < lim; ++c) {
            const float4 q = tile[c];
            const float v = neg_sq_dist(sqi, xi, yi, zi, q.x, q.y, q.z, q.w);
            if (v > top[KSEL - 1]) topk_insert<KSEL>(top, v);
        }
    }
    const float kth = top[KSEL - 1];
    int count = 0;
    int32_t* my = idx + ((size_t)cloud * n + (valid ? i : 0)) * cap;
    for (int t0 = 0; t0 < n; t0 += KNN_TILE) {
        load_tile(t0);
        const int lim = min(KNN_TILE, n - t0);
#pragma unroll 4
        for (int c = 0; c < lim; ++c) {
            const float4 q = tile[c];
            const float v = neg_sq_dist(sqi, xi, yi, zi, q.x, q.y, q.z, q.w);
            if (v >= kth) {
                if (valid && count < cap) my[count] = t0 + c;
                ++count;
            }
        }
    }
    if (status && blockIdx.x == 0 && wave_any(bad) && (threadIdx.x & 63) == 0) atomicOr(status + cloud, EPC_STATUS_NONFINITE_INPUT);
    if (valid)   // short rows (non-finite coordinates): pad with the point's own index, see the LDS kernel
        for (int c = count; c < KSEL; ++c) my[c] = i;
    if (valid) {
        cnt[(size_t)cloud * n + i] = count;
        kth_out[(size_t)cloud * n + i] = kth;
    }
}

__global__ __launch_bounds__(256) void knn_mask_kernel(const float* __restrict__ xyz,
                                                       const float* __restrict__ kth, int n,
                                                       float* __restrict__ mask) {
    const int cloud = blockIdx.z;
    const int i = blockIdx.y;
    const int j = blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const float* pc = xyz + (size_t)cloud * n * 3;
    const float xi = pc[3 * i], yi = pc[3 * i + 1], zi = pc[3 * i + 2];
    const float xj = pc[3 * j], yj = pc[3 * j + 1], zj = pc[3 * j + 2];
    const float a = neg_sq_dist(sq3(xi, yi, zi), xi, yi, zi, xj, yj, zj, sq3(xj, yj, zj));
    mask[((size_t)cloud * n + i) * n + j] = (a >= kth[(size_t)cloud * n + i]) ? 1.0f : 0.0f;
}

// one_lane: the one-lane-per-query form of the LDS kernel instead of the four-lane one (epc_knn_topk_form: the parity tests and the
// timing scripts hold the two against each other; the product entry points always pass false)
static int launch_knn(const float* xyz, int num_clouds, int n, int cap, int32_t* idx, int32_t* cnt, float* kth,
                      const float* conv1_pack, float* x32, void* x16, int idx_u16, int32_t* status, void* stream,
                      const char* who, bool one_lane = false) {
    dim3 grid((n + KNN_THREADS - 1) / KNN_THREADS, num_clouds);
#ifdef KNN_COLLECT
    {
        // bound / collect / select form (knn_collect_kernel) wherever its LDS image fits: cloud + boxes + the per-lane lists
        const int ntiles = (n + KNN_CT - 1) / KNN_CT;
        const size_t lds_bytes = ((size_t)ntiles * KNN_CT + 2 * (size_t)ntiles + 1) * sizeof(float4) +
                                 (size_t)KC_CAP * KC_THREADS * sizeof(unsigned short);
        if (lds_bytes <= 160 * 1024 && n <= 65535) {
            const void* fn = conv1_pack ? reinterpret_cast<const void*>(knn_collect_kernel<EPC_KNN_SELECT, true>)
                                        : reinterpret_cast<const void*>(knn_collect_kernel<EPC_KNN_SELECT, false>);
            hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
            if (e != hipSuccess) {
                epc_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e));
                return EPC_EHIP;
            }
            dim3 cgrid((n + KC_THREADS * KC_CHUNKS - 1) / (KC_THREADS * KC_CHUNKS), num_clouds);
            if (conv1_pack)
                hipLaunchKernelGGL((knn_collect_kernel<EPC_KNN_SELECT, true>), cgrid, dim3(KC_THREADS), lds_bytes,
                                   (hipStream_t)stream, xyz, n, cap, idx, cnt, kth, conv1_pack, x32, (unsigned short*)x16, idx_u16,
                                   status);
            else
                hipLaunchKernelGGL((knn_collect_kernel<EPC_KNN_SELECT, false>), cgrid, dim3(KC_THREADS), lds_bytes,
                                   (hipStream_t)stream, xyz, n, cap, idx, cnt, kth, nullptr, nullptr, nullptr, 0, status);
            EPC_CHECK_LAUNCH();
            return EPC_OK;
        }
    }
#endif
    if (n <= KNN_LDS_MAX_N) {
        const int ntiles = (n + KNN_CT - 1) / KNN_CT;
        const size_t lds_bytes = ((size_t)ntiles * KNN_CT + 2 * (size_t)ntiles + 2) * sizeof(float4) +
                                 (size_t)KNN_WAVES * (KNN_LDS_MAX_N / 4 / 32) * sizeof(unsigned int);   // (sized for the 4-wide form's masks)
        // grids of more than one workgroup per CU run the 4-wide form, two workgroups to a CU (comment at the kernel)
        const int num_cus = epc_device_cu_count();
#ifdef KNN_FORCE_BATCH
        const bool wide_grid = KNN_FORCE_BATCH == 4;
#else
        const bool wide_grid = (long)grid.x * grid.y >= 2L * num_cus;
#endif
#define EPC_KNN_LAUNCH(C1, B)                                                                                                   \
    do {                                                                                                                        \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_topk_culled_kernel<EPC_KNN_SELECT, C1, B>),       \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                        \
        if (e_ != hipSuccess) {                                                                                                 \
            epc_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e_));                                           \
            return EPC_EHIP;                                                                                                    \
        }                                                                                                                       \
        hipLaunchKernelGGL((knn_topk_culled_kernel<EPC_KNN_SELECT, C1, B>), grid, dim3(KNN_THREADS), lds_bytes,                \
                           (hipStream_t)stream, xyz, n, cap, idx, cnt, kth, C1 ? conv1_pack : nullptr, C1 ? x32 : nullptr,     \
                           C1 ? (unsigned short*)x16 : nullptr, C1 ? idx_u16 : 0, status);                                      \
    } while (0)
        // four lanes per query (knn_topk_quad_kernel): measured faster than the one-lane form at every batch of 4096-point clouds
        // (1 cloud 0.047 vs 0.156 ms, 18: 0.073 vs 0.167, 64: 0.149 vs 0.189, 256: 0.483 vs 0.536)
        const bool quad = !one_lane;
        if (quad) {
            // 16-query groups per workgroup: as many as spread the grid over two workgroups per CU in ONE round (two cloud images fit
            // a CU's LDS): up to 16 waves, each taking `rounds` groups one after the other -- the LDS image (the whole cloud, its
            // boxes) is built once per workgroup, and no CU holds 32 waves while others hold 16 or none
            const int groups = (n + 15) / 16;
            const long slots = 2L * num_cus;
            int per_wg = (int)(((long)groups * num_clouds + slots - 1) / slots);   // groups per workgroup
            per_wg = min(max(per_wg, 4), groups);
            while (per_wg < groups && (long)((groups + per_wg - 1) / per_wg) * num_clouds > slots) ++per_wg;   // (rounding per cloud)
            const int rounds = (per_wg + KNN_WAVES - 1) / KNN_WAVES;
            const int g = (per_wg + rounds - 1) / rounds;
            const int wgs = (groups + g * rounds - 1) / (g * rounds);
            const dim3 qgrid(wgs, num_clouds), qblock(64 * g);
#define EPC_KNN_QLAUNCH(C1, W)                                                                                                   \
    do {                                                                                                                        \
        hipError_t e_ = hipFuncSetAttribute(reinterpret_cast<const void*>(knn_topk_quad_kernel<EPC_KNN_SELECT, C1, W>),         \
                                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);                        \
        if (e_ != hipSuccess) {                                                                                                 \
            epc_set_error("%s: hipFuncSetAttribute: %s", who, hipGetErrorString(e_));                                           \
            return EPC_EHIP;                                                                                                    \
        }                                                                                                                       \
        hipLaunchKernelGGL((knn_topk_quad_kernel<EPC_KNN_SELECT, C1, W>), qgrid, qblock, lds_bytes,                             \
                           (hipStream_t)stream, xyz, n, cap, idx, cnt, kth, C1 ? conv1_pack : nullptr, C1 ? x32 : nullptr,     \
                           C1 ? (unsigned short*)x16 : nullptr, C1 ? idx_u16 : 0, status, rounds);                              \
    } while (0)
            const bool small = ntiles <= 128;
            if (conv1_pack) {
                if (small) EPC_KNN_QLAUNCH(true, 4);
                else EPC_KNN_QLAUNCH(true, 8);
            } else {
                if (small) EPC_KNN_QLAUNCH(false, 4);
                else EPC_KNN_QLAUNCH(false, 8);
            }
#undef EPC_KNN_QLAUNCH
        } else if (conv1_pack) {
            if (wide_grid) EPC_KNN_LAUNCH(true, 4);
            else EPC_KNN_LAUNCH(true, 8);
        } else {
            if (wide_grid) EPC_KNN_LAUNCH(false, 4);
            else EPC_KNN_LAUNCH(false, 8);
        }
#undef EPC_KNN_LAUNCH
    } else {
        hipLaunchKernelGGL(knn_topk_stream_kernel<EPC_KNN_SELECT>, grid, dim3(KNN_THREADS), 0, (hipStream_t)stream,
                           xyz, n, cap, idx, cnt, kth, status);
    }
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_knn_topk(const float* xyz, int num_clouds, int n, int cap, int32_t* idx, int32_t* cnt,
                            float* kth, void* stream) {
    EPC_CHECK_ARG(xyz && idx && cnt && kth, "null pointer");
    EPC_CHECK_ARG(num_clouds >= 0 && num_clouds <= 65535 && n >= EPC_KNN_SELECT,
                  "need num_points >= 20 (tf.nn.top_k k=20)");
    EPC_CHECK_ARG(cap >= EPC_KNN_SELECT, "list capacity must be >= 20");
    if (num_clouds == 0) return EPC_OK;
    return launch_knn(xyz, num_clouds, n, cap, idx, cnt, kth, nullptr, nullptr, nullptr, 0, nullptr, stream, __func__);
}

static int knn_topk_conv1_impl(const float* xyz, int num_clouds, int n, int cap, void* idx, int idx_u16, int32_t* cnt, float* kth,
                               const void* packed_conv1, float* x, void* x16, int32_t* status, void* stream, bool one_lane);

// Test / tuning entry points: epc_knn_topk and epc_knn_topk_conv1 with the LDS kernel's form chosen by the caller (form 0 = one lane
// per query, the kernel of rounds 1-3; 1 = four lanes per query, what the product entry points run).  An explicit argument: the
// library reads no environment and holds no global state.
extern "C" int epc_knn_topk_form(const float* xyz, int num_clouds, int n, int cap, int32_t* idx, int32_t* cnt, float* kth, int form,
                                 void* stream) {
    EPC_CHECK_ARG(xyz && idx && cnt && kth, "null pointer");
    EPC_CHECK_ARG(num_clouds >= 0 && num_clouds <= 65535 && n >= EPC_KNN_SELECT && cap >= EPC_KNN_SELECT && (form == 0 || form == 1),
                  "need num_points >= 20, capacity >= 20, form 0 or 1");
    if (num_clouds == 0) return EPC_OK;
    return launch_knn(xyz, num_clouds, n, cap, idx, cnt, kth, nullptr, nullptr, nullptr, 0, nullptr, stream, __func__, form == 0);
}

extern "C" int epc_knn_topk_conv1_form(const float* xyz, int num_clouds, int n, int cap, void* idx, int idx_u16, int32_t* cnt,
                                       float* kth, const void* packed_conv1, float* x, void* x16, int32_t* status, int form,
                                       void* stream) {
    EPC_CHECK_ARG(form == 0 || form == 1, "form 0 or 1");
    return knn_topk_conv1_impl(xyz, num_clouds, n, cap, idx, idx_u16, cnt, kth, packed_conv1, x, x16, status, stream, form == 0);
}

extern "C" int epc_knn_topk_conv1(const float* xyz, int num_clouds, int n, int cap, void* idx, int idx_u16, int32_t* cnt,
                                  float* kth, const void* packed_conv1, float* x, void* x16, int32_t* status,
                                  void* stream) {
    return knn_topk_conv1_impl(xyz, num_clouds, n, cap, idx, idx_u16, cnt, kth, packed_conv1, x, x16, status, stream, false);
}

static int knn_topk_conv1_impl(const float* xyz, int num_clouds, int n, int cap, void* idx, int idx_u16, int32_t* cnt, float* kth,
                               const void* packed_conv1, float* x, void* x16, int32_t* status, void* stream, bool one_lane) {
    EPC_CHECK_ARG(xyz && idx && cnt && kth && packed_conv1 && (x || x16), "null pointer");
    EPC_CHECK_ARG(!idx_u16 || (n <= 65535 && n <= KNN_LDS_MAX_N), "2-byte lists need num_points <= 8192 (the LDS kernel)");
    EPC_CHECK_ARG(num_clouds >= 0 && num_clouds <= 65535 && n >= EPC_KNN_SELECT,
                  "need num_points >= 20 (tf.nn.top_k k=20)");
    EPC_CHECK_ARG(cap >= EPC_KNN_SELECT, "list capacity must be >= 20");
    if (num_clouds == 0) return EPC_OK;
    if (n > KNN_LDS_MAX_N) {   // the streaming kNN kernel keeps no cloud image: two launches
        int rc = launch_knn(xyz, num_clouds, n, cap, (int32_t*)idx, cnt, kth, nullptr, nullptr, nullptr, 0, status, stream, __func__);
        if (rc != EPC_OK) return rc;
        return epc_conv1_launch(xyz, packed_conv1, num_clouds * n, x, x16, status, n, stream);   // (f32 rows: EPC-Net-L / F32 precision run at this size)
    }
    return launch_knn(xyz, num_clouds, n, cap, (int32_t*)idx, cnt, kth, (const float*)packed_conv1, x, x16, idx_u16, status,
                      stream, __func__, one_lane);
}

extern "C" int epc_knn_mask(const float* xyz, const float* kth, int num_clouds, int n, float* mask,
                            void* stream) {
    EPC_CHECK_ARG(xyz && kth && mask, "null pointer");
    EPC_CHECK_ARG(num_clouds >= 0 && num_clouds <= 65535 && n > 0 && n <= 65535, "bad shape");
    if (num_clouds == 0) return EPC_OK;
    dim3 grid((n + 255) / 256, n, num_clouds);
    hipLaunchKernelGGL(knn_mask_kernel, grid, dim3(256), 0, (hipStream_t)stream, xyz, kth, n, mask);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}
