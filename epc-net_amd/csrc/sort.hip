// Per-cloud spatial sort of the points (along the Hilbert curve; the entry point keeps its first name, epc_morton_sort)
// -- a pure re-ordering stage in front of the pipeline.
//
// Why it is legal: every stage of EPC-Net / EPC-Net-L is equivariant to a permutation of the points and the final
// pooling (VLAD sums, EPC-Net-L max) is invariant, so descriptors of the sorted cloud equal those of the original
// up to fp32 summation order (tests/test_gpu_parity.py::test_permutation_invariance).  Why it pays: consecutive
// points become spatial neighbours, so (i) the kNN kernel's tile bounding boxes are tight and most candidate
// tiles are culled, (ii) the ProxyConv gathers hit rows that sit close together in L2/L1.
//
// One workgroup (1024 threads) per cloud: cloud bounding box -> 10 bits per axis -> 30-bit Hilbert index; 64-bit keys
// (index << 32 | original point index: a total order, so the result is deterministic) are sorted in LDS.
#include "common.h"

#define SORT_THREADS 1024
#define SORT_MAX_N 16384  // 128 KB of 8-byte keys

__device__ __forceinline__ unsigned int spread10(unsigned int v) {
    v &= 0x3ffu;
    v = (v | (v << 16)) & 0x030000ffu;
    v = (v | (v << 8)) & 0x0300f00fu;
    v = (v | (v << 4)) & 0x030c30c3u;
    v = (v | (v << 2)) & 0x09249249u;
    return v;
}

// Index of the cell (x, y, z), BITS bits per axis, along the 3-D Hilbert curve (Skilling's transpose algorithm, "Programming
// the Hilbert curve", 2004): consecutive indices are always face-adjacent cells, whereas the Z-order (plain bit interleave)
// jumps across the cube at every octant boundary.  On 4096-point clouds the 32-point tiles of the Hilbert order have 20 %
// shorter bounding-box diagonals and the kNN kernel scans 26 instead of 38 of the 128 tiles per wave.  The index of a
// prefix of the coordinate bits is the prefix of the index, so the 4-byte keys compute only the 7 levels they keep.
template <int BITS>
__device__ __forceinline__ unsigned int hilbert_index(unsigned int x, unsigned int y, unsigned int z) {
    unsigned int X[3] = {x, y, z};
#pragma unroll
    for (unsigned int Q = 1u << (BITS - 1); Q > 1; Q >>= 1) {
        const unsigned int P = Q - 1;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) {
                X[0] ^= P;
            } else {
                const unsigned int t = (X[0] ^ X[i]) & P;
                X[0] ^= t;
                X[i] ^= t;
            }
        }
    }
    X[1] ^= X[0];
    X[2] ^= X[1];
    unsigned int t = 0;
#pragma unroll
    for (unsigned int Q = 1u << (BITS - 1); Q > 1; Q >>= 1)
        if (X[2] & Q) t ^= Q - 1;
    X[0] ^= t, X[1] ^= t, X[2] ^= t;
    return (spread10(X[0]) << 2) | (spread10(X[1]) << 1) | spread10(X[2]);   // X[0] carries the most significant bits
}

// Bitonic sort of EPT * 1024 keys with EPT consecutive keys per thread held in registers: compare-exchange partners at
// distance < EPT are in the same thread, at distance < 64 * EPT in the same wave (lane exchange, no barrier); only the
// longer distances go through LDS.  For 4096 keys that is 10 LDS stages instead of 78 barrier-separated passes.
template <int EPT, typename KEY>
__device__ __forceinline__ void bitonic_registers(KEY* __restrict__ keys, int tid) {
    constexpr int NP = EPT * SORT_THREADS;
    KEY v[EPT];
#pragma unroll
    for (int r = 0; r < EPT; ++r) v[r] = keys[EPT * tid + r];
    for (int k = 2; k <= NP; k <<= 1)
        for (int s = k >> 1; s > 0; s >>= 1) {
            if (s < EPT) {
#pragma unroll
                for (int r = 0; r < EPT; ++r) {
                    if (r & s) continue;
                    const bool up = (((EPT * tid + r) & k) == 0);
                    const KEY a = v[r], b = v[r | s];
                    const bool swap = (a > b) == up;
                    v[r] = swap ? b : a;
                    v[r | s] = swap ? a : b;
                }
            } else {
                KEY w[EPT];
                if (s < 64 * EPT) {
#pragma unroll
                    for (int r = 0; r < EPT; ++r) w[r] = __shfl_xor(v[r], s / EPT);
                } else {
                    __syncthreads();   // every thread has finished reading the previous exchange
#pragma unroll
                    for (int r = 0; r < EPT; ++r) keys[EPT * tid + r] = v[r];
                    __syncthreads();
                    const int tp = tid ^ (s / EPT);
#pragma unroll
                    for (int r = 0; r < EPT; ++r) w[r] = keys[EPT * tp + r];
                }
                const bool lower = ((EPT * tid) & s) == 0;
#pragma unroll
                for (int r = 0; r < EPT; ++r) {
                    const bool up = (((EPT * tid + r) & k) == 0);
                    const bool keep_min = lower == up;
                    const KEY a = v[r], b = w[r];
                    v[r] = ((a < b) == keep_min) ? a : b;
                }
            }
        }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < EPT; ++r) keys[EPT * tid + r] = v[r];
    __syncthreads();
}

// n <= 4096: two-level sort of the (unique: the low 12 bits are the point index) 32-bit keys in place of the 78-stage
// network.  Level 1 is a counting sort by the top 12 bits (4096 buckets, LDS atomics: the order INSIDE a bucket is
// whatever the atomics made it); level 2 ranks every key inside its bucket by counting the smaller keys there (a bucket
// holds ~1 key on volumetric clouds, ~16 on planes).  Unique keys make the result independent of the atomics' order, i.e.
// identical to the network's.  Returns false, with keysA untouched, when a bucket holds more than SORT_BUCKET_MAX keys
// (degenerate clouds: many points in one cell) -- the caller then runs the network.  Keys of the padding (index >= n)
// are not sorted: they are ~0u and follow the n real keys.
#define SORT_BUCKET_MAX 64
__device__ __forceinline__ bool bucket_sort_4096(unsigned int* __restrict__ keysA, unsigned int* __restrict__ keysB,
                                                 unsigned int* __restrict__ hist /* 4097 */, int n, int tid) {
    __shared__ unsigned int wave_tot[SORT_THREADS / 64];
    __shared__ unsigned int s_max;
    const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) hist[tid + i * SORT_THREADS] = 0u;
    if (tid == 0) s_max = 0u;
    __syncthreads();
    unsigned int k[4], slot[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = tid + i * SORT_THREADS;
        k[i] = keysA[j];
        slot[i] = j < n ? atomicAdd(&hist[k[i] >> 20], 1u) : 0u;
    }
    __syncthreads();
    // exclusive scan of the 4096 counts: thread t owns buckets 4t .. 4t+3
    unsigned int c[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) c[q] = hist[4 * tid + q];
    const unsigned int local = (c[0] + c[1]) + (c[2] + c[3]);
    unsigned int mx = max(max(c[0], c[1]), max(c[2], c[3]));
    unsigned int incl = local;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned int t = __shfl_up(incl, off);
        if (lane >= off) incl += t;
        mx = max(mx, (unsigned int)__shfl_xor(mx, off));
    }
    if (lane == 63) wave_tot[wave] = incl;
    if (lane == 0) atomicMax(&s_max, mx);
    __syncthreads();
    if (s_max > SORT_BUCKET_MAX) return false;   // the same answer in every thread
    unsigned int base = incl - local;
    for (int w = 0; w < wave; ++w) base += wave_tot[w];
    hist[4 * tid] = base;
    hist[4 * tid + 1] = base + c[0];
    hist[4 * tid + 2] = base + c[0] + c[1];
    hist[4 * tid + 3] = base + c[0] + c[1] + c[2];
    if (tid == SORT_THREADS - 1) hist[4 * SORT_THREADS] = base + local;   // = n
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i)
        if (tid + i * SORT_THREADS < n) keysB[hist[k[i] >> 20] + slot[i]] = k[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int j = tid + i * SORT_THREADS;
        if (j < n) {
            const unsigned int b = k[i] >> 20, lo = hist[b], hi = hist[b + 1];
            unsigned int rank = 0;
            for (unsigned int q = lo; q < hi; ++q) rank += keysB[q] < k[i] ? 1u : 0u;
            keysA[lo + rank] = k[i];
        } else {
            keysA[j] = ~0u;   // (j >= n: a padding slot stays a padding slot)
        }
    }
    __syncthreads();
    return true;
}

__global__ __launch_bounds__(SORT_THREADS) void morton_sort_kernel(const float* __restrict__ xyz, int n, int npow2,
                                                                   float* __restrict__ xyz_sorted,
                                                                   int32_t* __restrict__ perm,
                                                                   int32_t* __restrict__ status_zero) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
    // the pipeline's per-cloud status word starts every pass at 0 (this is the pass's first kernel: no memset launch)
    if (status_zero && threadIdx.x == 0) status_zero[blockIdx.x] = 0;
    __shared__ float red[6][SORT_THREADS / 64];
    __shared__ float box[6];
    const int cloud = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* pc = xyz + (size_t)cloud * n * 3;

    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    for (int j = tid; j < n; j += SORT_THREADS)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const float v = pc[3 * j + d];
            lo[d] = fminf(lo[d], v);
            hi[d] = fmaxf(hi[d], v);
        }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], off));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], off));
        }
        if (lane == 0) {
            red[d][wave] = lo[d];
            red[3 + d][wave] = hi[d];
        }
    }
    __syncthreads();
    if (tid < 6) {
        float v = red[tid][0];
        for (int w = 1; w < SORT_THREADS / 64; ++w) v = tid < 3 ? fminf(v, red[tid][w]) : fmaxf(v, red[tid][w]);
        box[tid] = v;
    }
    __syncthreads();
    float scale[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float ext = box[3 + d] - box[d];
        scale[d] = ext > 0.f ? 1023.0f / ext : 0.f;
    }
    // n <= 4096: 32-bit keys (top 20 bits of the Morton code | 12-bit index) -- half the exchange traffic; the cell
    // grid (2^20 cells for <= 4096 points) is still far finer than the point spacing.  Larger clouds: 64-bit keys.
    const bool narrow = npow2 == 4 * SORT_THREADS;
    unsigned int* keys32 = reinterpret_cast<unsigned int*>(keys);
    for (int j = tid; j < npow2; j += SORT_THREADS) {
        unsigned long long key = ~0ull;
        unsigned int key32 = ~0u;
        if (j < n) {
            unsigned int qi[3];
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                const float q = (pc[3 * j + d] - box[d]) * scale[d];
                qi[d] = (unsigned int)fminf(fmaxf(q, 0.f), 1023.f);
            }
            if (narrow) {   // the top 20 bits of the 30-bit index = the 21-bit index of the top 7 coordinate bits, less one
                key32 = ((hilbert_index<7>(qi[0] >> 3, qi[1] >> 3, qi[2] >> 3) >> 1) << 12) | (unsigned int)j;
            } else {
                key = ((unsigned long long)hilbert_index<10>(qi[0], qi[1], qi[2]) << 32) | (unsigned int)j;
            }
        }
        if (narrow)
            keys32[j] = key32;
        else
            keys[j] = key;
    }
    __syncthreads();
    if (narrow) {
        // (the LDS behind the 4-B keys holds the second key array and, past the 8-B key area, the 4097 bucket counters)
        if (!bucket_sort_4096(keys32, keys32 + npow2, reinterpret_cast<unsigned int*>(keys + npow2), n, tid))
            bitonic_registers<4, unsigned int>(keys32, tid);
    } else if (npow2 == 8 * SORT_THREADS) {
        bitonic_registers<8, unsigned long long>(keys, tid);
    } else if (npow2 == 16 * SORT_THREADS) {
        bitonic_registers<16, unsigned long long>(keys, tid);
    } else
    for (int k = 2; k <= npow2; k <<= 1)
        for (int s = k >> 1; s > 0; s >>= 1) {
            for (int t = tid; t < npow2 / 2; t += SORT_THREADS) {
                const int i0 = ((t / s) * 2 * s) + (t % s);
                const int i1 = i0 + s;
                const bool up = ((i0 & k) == 0);
                const unsigned long long a = keys[i0], b = keys[i1];
                if ((a > b) == up) {
                    keys[i0] = b;
                    keys[i1] = a;
                }
            }
            __syncthreads();
        }
    for (int r = tid; r < n; r += SORT_THREADS) {
        const int src = narrow ? (int)(keys32[r] & 0xfffu) : (int)(keys[r] & 0xffffffffu);
        float* o = xyz_sorted + ((size_t)cloud * n + r) * 3;
        o[0] = pc[3 * src + 0];
        o[1] = pc[3 * src + 1];
        o[2] = pc[3 * src + 2];
        if (perm) perm[(size_t)cloud * n + r] = src;
    }
}

extern "C" int epc_morton_sort(const float* xyz, int num_clouds, int n, float* xyz_sorted, int32_t* perm,
                               void* stream) {
    return epc_sort_launch(xyz, num_clouds, n, xyz_sorted, perm, nullptr, stream);
}

int epc_sort_launch(const float* xyz, int num_clouds, int n, float* xyz_sorted, int32_t* perm, int32_t* status_zero,
                    void* stream) {
    EPC_CHECK_ARG(xyz && xyz_sorted, "null pointer");
    EPC_CHECK_ARG(num_clouds >= 0 && n > 0 && n <= SORT_MAX_N, "num_points must be in 1..16384");
    EPC_CHECK_ARG(xyz != xyz_sorted, "in-place sort is not supported");
    if (num_clouds == 0) return EPC_OK;
    int npow2 = 2;
    while (npow2 < n) npow2 <<= 1;
    // keys (8 B each; the n <= 4096 path splits the area into two 4-B key arrays) + that path's 4097 bucket counters
    const size_t lds_bytes = (size_t)npow2 * sizeof(unsigned long long) +
                             (npow2 == 4 * SORT_THREADS ? (size_t)(npow2 + 4) * sizeof(unsigned int) : 0);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(morton_sort_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
        epc_set_error("epc_morton_sort: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return EPC_EHIP;
    }
    hipLaunchKernelGGL(morton_sort_kernel, dim3(num_clouds), dim3(SORT_THREADS), lds_bytes, (hipStream_t)stream, xyz,
                       n, npow2, xyz_sorted, perm, status_zero);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}
