// The FORWARD of the 64-channel backbone of the training step as ONE persistent launch (models/epc-net.py:66-134, utils/tf_util.py:454-519
// in training mode) -- the forward launches of train_chain.hip with their kernel boundaries replaced by grid-wide barriers.  (The
// backward was built the same way -- 16 barriers, the gradient handed from layer to layer in registers -- was parity-green and SLOWER
// than its launches, 2.25 -> 2.41 ms per 22-cloud step: in a replayed graph a kernel boundary + the pooled prologue cost what a barrier
// costs, ~8 us, so only traffic saved pays, and the backward's gather runs 61 us on 12 waves against 44 on 16.  DESIGN.md 4 has the
// numbers; the commit before its removal has the code.)
//
// Why.  Every training-mode BatchNorm is a grid-wide dependency (its batch moments need every row), and train_chain.hip pays one
// kernel boundary per dependency: 33 launches whose 14-20 us are mostly launch gap (4.2 us), the pooling of 256 moment partials
// by every workgroup (5 us) and the re-read of a (rows, 64) tensor the previous launch has just written.  Here ONE workgroup per CU
// keeps its rows for the whole pass: a wave owns one 32-row tile, the tile handed to the next layer stays in the wave's LDS tile,
// and a dependency costs a barrier (2.3 us) with the moments reduced in two levels on the way through it.
//
// The barrier (scripts/probe/grid_barrier_probe{,2}.hip have the measurements this design follows).  The textbook form -- agent-scope
// release fence, atomic, poll, acquire fence -- costs 6-10 us bare and 24-35 us behind 72 KB of freshly written rows per workgroup
// on this chip: every workgroup's `buffer_wbl2` walks its XCD's whole L2, and 256 pollers of one word queue behind the arrivals at
// the memory side.  This one needs NO cache maintenance:
//   * whatever ANOTHER workgroup will read inside the launch (the next block's z0 rows, the backward's s rows, every partial) is
//     stored WRITE-THROUGH (`global_store ... sc1`, agent scope): once the store has completed (s_waitcnt vmcnt(0)) it is where
//     every XCD's L2 misses go;
//   * every such location is written exactly ONCE per launch, before its first read: no L1 or L2 can hold a stale copy of it (a
//     launch starts with both invalidated), so readers use plain loads and the neighbour gathers keep their L2 hit rate;
//   * arrival in two levels -- eight group counters (blockIdx & 7: under round-robin dispatch the workgroups of an XCD), the last
//     arrival of a group bumps the global counter -- and release in two: only that last arrival polls the global counter and then
//     publishes the epoch in its group's flag, the other 31 poll the flag.  Sync words are touched by atomics and sc1 accesses only.
//   * the moments travel WITH the barrier: the last arrival of a group merges its group's partials (<= 32, ascending rows, double
//     precision) into one group partial before it bumps the global counter; after the release every workgroup merges the eight
//     group partials.  196 KB of partials per workgroup and boundary (50 MB through the L2s) become 6 KB for eight of them + 16 KB.
// Nothing depends on WHICH workgroup arrives last: the merge order is by row range, so results are bit-reproducible.
//
// Safety.  The grid is at most the CU count and the host checks with the occupancy query that every workgroup is co-resident; every
// spin is bounded by a wall-clock budget (s_memrealtime), a time-out sets the sticky error word and every workgroup leaves at its next
// poll; a launch that finds the error word set leaves at once.  Either way the launch leaves NaN in the concat (a row per workgroup), so
// the loss is NaN; epc_chain_persist_status() reads the word, _reset() clears it.
#include "train_chain_common.h"

#define PST_WAVES 12                 // one 32-row tile per wave: at most 384 rows per workgroup (98 304 rows on 256 CUs)
#define PST_TILE_BYTES (CH_STG_FLOATS * 4)
// sync words (unsigned index): the launch sequence number, the exit counter, the error word -- each on its own line
#define PST_W_SEQ 0
#define PST_W_EXIT 2048
#define PST_W_ERR 2112
#define PST_SYNC_WORDS 4096
// the workspace: [sync words][workgroup partials: PST_MAX_PHASES x PST_MAX_PARTS x 192 granules][group partials: PST_MAX_PHASES x 8 x 384
// granules][time stamps of a -DPST_STAMPS build: PST_MAX_PARTS x 64 long long]
#define PST_MAX_PHASES 20
#define PST_MAX_PARTS 512
#define PST_WS_PARTIALS (PST_SYNC_WORDS * 4)
#define PST_WS_GROUPS (PST_WS_PARTIALS + (size_t)PST_MAX_PHASES * PST_MAX_PARTS * 192 * 8)
#define PST_WS_STAMPS (PST_WS_GROUPS + (size_t)PST_MAX_PHASES * 8 * 384 * 8)
#define PST_WS_BYTES (PST_WS_STAMPS + (size_t)PST_MAX_PARTS * 64 * 8)

typedef float pst_f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 pst_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_wt4(float* p, float4 v) {
    const pst_f32x4 x = {v.x, v.y, v.z, v.w};
    asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
}
// (relaxed agent-scope atomic stores ARE `global_store ... sc1`, and leave the compiler its immediate offsets)
__device__ __forceinline__ void st_wt1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_wt_d(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_wt_u(unsigned* p, unsigned v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned ld_coh(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

// A tagged value: 32 payload bits + the 32-bit tag of (launch, barrier), stored and loaded as ONE 64-bit relaxed agent-scope atomic
// (`global_store_dwordx2 / global_load_dwordx2 ... sc1`): whoever sees the tag sees the payload, and -- the poster waited for its
// earlier write-through stores (s_waitcnt vmcnt(0)) before it stored the tag -- everything the poster wrote before.
typedef unsigned long long pst_gran;
__device__ __forceinline__ void st_gran(pst_gran* p, unsigned payload, unsigned tag) {
    __hip_atomic_store(p, ((pst_gran)tag << 32) | payload, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ pst_gran ld_gran(const pst_gran* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// threadIdx.x behind an opaque move: what is derived from it inside a helper is recomputed at every call instead of being hoisted out of the
// block loop by the compiler and held (then spilled) for the whole kernel
__device__ __forceinline__ int pst_tid() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

struct PstCtx {
    unsigned* w;
    long long budget;          // spin budget in s_memrealtime ticks (100 MHz)
    unsigned nwg, seq;         // workgroups; this launch's sequence number (the tag's upper bits)
    unsigned g, gsize, ngroups;
    int lb;                    // the workgroup's logical block (its rows: lb * wg_rows ..); group g's blocks are lb0 .. lb0 + gsize - 1
    int lb0;
    bool leader;               // the group's first workgroup reduces the group's partials
};
__device__ __forceinline__ PstCtx pst_init(unsigned* w, long long budget) {
    PstCtx c;
    c.w = w, c.budget = budget, c.nwg = gridDim.x;
    c.seq = ld_coh(w + PST_W_SEQ);
    c.g = blockIdx.x & 7, c.ngroups = c.nwg < 8 ? c.nwg : 8u, c.gsize = (c.nwg + 7 - c.g) / 8;
    c.lb = xcd_contiguous_block(blockIdx.x, gridDim.x);
    c.lb0 = xcd_contiguous_block(c.g, gridDim.x);
    c.leader = blockIdx.x < 8;
    return c;
}
__device__ __forceinline__ unsigned pst_tag(const PstCtx& c, int phase) { return c.seq * 64u + (unsigned)phase + 1u; }
// first block and block count of group k (its rows: lb0 * wg_rows .. (lb0 + size) * wg_rows, clipped)
__device__ __forceinline__ void pst_group_range(const PstCtx& c, int k, int& lb0, int& size) {
    lb0 = xcd_contiguous_block(k, c.nwg);
    size = (c.nwg + 7 - k) / 8;
}

// Polls `count` granules starting at `src` (granule e of the list goes to thread e % blockDim, CH per thread and round trip) until
// every one carries `tag`, and leaves the payloads in LDS (dst[e]).  Returns false -- to every thread, through a barrier -- when the
// wait was abandoned (spin budget, or the error word set by another workgroup).  Contains barriers: call from every thread.
template <int CH>
__device__ __forceinline__ bool pst_poll(const PstCtx& c, const pst_gran* src, int count, unsigned tag, unsigned* dst) {
    const int tid = pst_tid(), nt = blockDim.x;
    int fail = 0;
    const long long t0 = wall_clock64();
    for (int e0 = tid; e0 < count && !fail; e0 += nt * CH) {
        unsigned it = 0;
        for (;;) {
            pst_gran v[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) v[u] = ld_gran(src + min(e0 + nt * u, count - 1));
            bool ok = true;
#pragma unroll
            for (int u = 0; u < CH; ++u) ok = ok && (unsigned)(v[u] >> 32) == tag;
            if (ok) {
#pragma unroll
                for (int u = 0; u < CH; ++u)
                    if (e0 + nt * u < count) dst[e0 + nt * u] = (unsigned)v[u];
                break;
            }
            if ((it++ & 31u) == 0u) {   // (the clock from the first miss on; the error word of the others every 32 misses)
                if (it > 1u && ld_coh(c.w + PST_W_ERR) != 0u) { fail = 1; break; }
                if (wall_clock64() - t0 > c.budget) { st_wt_u(c.w + PST_W_ERR, 1u); fail = 1; break; }
            }
            __builtin_amdgcn_s_sleep(1);
        }
    }
    return __syncthreads_or(fail) == 0;
}

// ---- moments through the barrier ------------------------------------------------------------------------------------------
// Workgroup partials: granules [parts][3][64] (sum (v - p), sum (v - p)^2, pivot p of the product WITHOUT the bias: ch_store_stats);
// group partials: granules [8][3][64][2] -- A = sum (v - P), B = sum (v - P)^2, P = the pivot of the group's first block as doubles
// in two halves.   sum (v - p0) = sum (v - pw) + n (pw - p0);   sum (v - p0)^2 = sum (v - pw)^2 + 2 (pw - p0) sum (v - pw) + n (pw - p0)^2
// The group's first workgroup merges its group's partials in ascending row order; every workgroup then merges the group partials in
// group order -- double precision, fixed order: bit-reproducible, the same bits in every workgroup.
// scratch: PST_SCRATCH_BYTES of LDS.  Returns false when the wait was abandoned.
#define PST_SCRATCH_BYTES 24576
template <int K>
__device__ __forceinline__ bool pst_group_reduce(const PstCtx& c, const pst_gran* partials, pst_gran* gp_all, unsigned tag, int rows, int wg_rows,
                                                 void* scratch) {
    // the group's partials -> LDS as floats [member][K][64] (member m of the group is block lb0 + m: contiguous granules)
    unsigned* raw = reinterpret_cast<unsigned*>(scratch);
    const int count = (int)c.gsize * K * 64;
    if (!pst_poll<8>(c, partials + (size_t)c.lb0 * K * 64, count, tag, raw)) return false;   // (32 x 192 granules on 768 threads: one round trip)
    const float* pf = reinterpret_cast<const float*>(raw);
    const int tid = pst_tid(), col = tid & 63, sl = tid >> 6;
    // four slices of consecutive members (threads 0..255), each merged in ascending order around its first member's pivot; then
    // thread `col` of slice 0 merges the slices in order: a dependent chain of 8 + 4 instead of 32
    const int nsl = min(4, (int)blockDim.x >> 6), per = ((int)c.gsize + nsl - 1) / nsl;
    double acc[K];
#pragma unroll
    for (int k = 0; k < K; ++k) acc[k] = 0.0;
    double n_sl = 0.0;
    if (sl < nsl) {
        const int m0 = sl * per, m1 = min((int)c.gsize, m0 + per);
        if constexpr (K == 3) {
            if (m0 < m1) acc[2] = (double)pf[(m0 * 3 + 2) * 64 + col];
            for (int m = m0; m < m1; ++m) {
                const double nt = (double)min(wg_rows, rows - (c.lb0 + m) * wg_rows);
                const double s1 = (double)pf[(m * 3 + 0) * 64 + col], s2 = (double)pf[(m * 3 + 1) * 64 + col], d = (double)pf[(m * 3 + 2) * 64 + col] - acc[2];
                acc[0] += s1 + nt * d, acc[1] += s2 + 2.0 * d * s1 + nt * d * d, n_sl += nt;
            }
        } else {
            for (int m = m0; m < m1; ++m)
#pragma unroll
                for (int k = 0; k < K; ++k) acc[k] += (double)pf[(m * K + k) * 64 + col];
        }
    }
    __syncthreads();   // raw is read
    double* part = reinterpret_cast<double*>(scratch);   // [slice][K + 1][64]
    if (sl < nsl) {
#pragma unroll
        for (int k = 0; k < K; ++k) part[(sl * (K + 1) + k) * 64 + col] = acc[k];
        part[(sl * (K + 1) + K) * 64 + col] = n_sl;
    }
    __syncthreads();
    if (tid < 64) {
        pst_gran* out = gp_all + (size_t)c.g * K * 64 * 2;
        double q[K];
#pragma unroll
        for (int k = 0; k < K; ++k) q[k] = part[k * 64 + tid];
        for (int s2 = 1; s2 < nsl; ++s2) {
            if constexpr (K == 3) {
                const double n = part[(s2 * 4 + 3) * 64 + tid];
                if (n > 0.0) {
                    const double a = part[(s2 * 4 + 0) * 64 + tid], b = part[(s2 * 4 + 1) * 64 + tid], d = part[(s2 * 4 + 2) * 64 + tid] - q[2];
                    q[0] += a + n * d, q[1] += b + 2.0 * d * a + n * d * d;
                }
            } else {
#pragma unroll
                for (int k = 0; k < K; ++k) q[k] += part[(s2 * (K + 1) + k) * 64 + tid];
            }
        }
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const unsigned long long bits = (unsigned long long)__double_as_longlong(q[k]);
            st_gran(out + (k * 64 + tid) * 2 + 0, (unsigned)bits, tag), st_gran(out + (k * 64 + tid) * 2 + 1, (unsigned)(bits >> 32), tag);
        }
    }
    __syncthreads();   // (the slices are read: the next poll may overwrite them)
    return true;
}
// Every workgroup: the group partials -> LDS doubles gd[group][K][64] (scratch).  Returns false when the wait was abandoned.
template <int K>
__device__ __forceinline__ bool pst_gather_groups(const PstCtx& c, const pst_gran* gp_all, unsigned tag, void* scratch) {
    return pst_poll<4>(c, gp_all, (int)c.ngroups * K * 64 * 2, tag, reinterpret_cast<unsigned*>(scratch));
}
// the moments of the whole batch from gd (pst_gather_groups<3>): threads 0..63 merge the groups in order; mean / population variance go
// to s_mean / s_var (LDS), to mean_out / var_out (global, workgroup 0) and as the BatchNorm's s, t to coef[0..1][64].  Ends with a barrier.
__device__ __forceinline__ void pst_final_moments(const PstCtx& c, const void* scratch, int rows, int wg_rows, const float* bias, const float* gamma,
                                                  const float* beta, float eps, float* mean_out, float* var_out, float* s_mean, float* s_var,
                                                  float (*coef)[64]) {
    const int tid = pst_tid();
    if (tid < 64) {
        const double* gd = reinterpret_cast<const double*>(scratch);
        const float ga = gamma[tid], be = beta[tid], bi = bias ? bias[tid] : 0.f;
        double A = gd[0 * 64 + tid], B = gd[1 * 64 + tid];
        const double p0 = gd[2 * 64 + tid];
        for (int k = 1; k < (int)c.ngroups; ++k) {
            int lb0, size;
            pst_group_range(c, k, lb0, size);
            const double n = (double)(min(rows, (lb0 + size) * wg_rows) - lb0 * wg_rows);
            const double a = gd[(k * 3 + 0) * 64 + tid], b = gd[(k * 3 + 1) * 64 + tid], d = gd[(k * 3 + 2) * 64 + tid] - p0;
            A += a + n * d, B += b + 2.0 * d * a + n * d * d;
        }
        const double m1 = A / (double)rows;
        const float mean = (float)(p0 + m1 + (double)bi);
        const float var = (float)fmax(B / (double)rows - m1 * m1, 0.0);
        if (blockIdx.x == 0) mean_out[tid] = mean, var_out[tid] = var;
        s_mean[tid] = mean, s_var[tid] = var;
        const ChBnAffine a2 = ch_bn_affine(mean, var, ga, be, eps);
        coef[0][tid] = a2.s, coef[1][tid] = a2.t;
    }
    __syncthreads();
}

// Last thing a workgroup does: the last one out advances the launch sequence number and zeroes the exit counter.
__device__ __forceinline__ void pst_exit(PstCtx& c) {
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned old = __hip_atomic_fetch_add(c.w + PST_W_EXIT, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old + 1 == c.nwg) {
            st_wt_u(c.w + PST_W_SEQ, c.seq + 1u);
            st_wt_u(c.w + PST_W_EXIT, 0u);
        }
    }
}

// A workgroup's moment partial from its waves' (sum, sum of squares, pivot, rows) -- ch_store_stats -- POSTED: every wave first waits
// for its own stores of the phase (the rows other workgroups will read are write-through: complete = visible), then wave 0 merges
// the waves in wave order and stores the tagged granules.  sred: [waves][3][64] floats, snrows: [waves].  Contains a barrier.
__device__ __forceinline__ void pst_post_stats(float (&s1)[2], float (&s2)[2], const float (&piv)[2], int my_rows, float (*sred)[3][64],
                                               int* snrows, pst_gran* out, unsigned tag) {
    const int tid = pst_tid();
    const int lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const int i = lane & 31, h = lane >> 5;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        s1[nt] += __shfl_xor(s1[nt], 32);
        s2[nt] += __shfl_xor(s2[nt], 32);
    }
    if (wave > 0 && h == 0) {
        sred[wave][0][i] = s1[0], sred[wave][0][32 + i] = s1[1];
        sred[wave][1][i] = s2[0], sred[wave][1][32 + i] = s2[1];
        sred[wave][2][i] = piv[0], sred[wave][2][32 + i] = piv[1];
    }
    if (lane == 0) snrows[wave] = my_rows;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave == 0 && h == 0) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int c = 32 * nt + i;
            float t1 = s1[nt], t2 = s2[nt];
            for (int w = 1; w < nw; ++w) {
                const float n_w = (float)snrows[w];
                if (n_w > 0.f) {
                    const float dp = sred[w][2][c] - piv[nt];
                    t1 += sred[w][0][c] + n_w * dp;
                    t2 += sred[w][1][c] + (2.0f * dp) * sred[w][0][c] + n_w * dp * dp;
                }
            }
            st_gran(out + 0 * 64 + c, __float_as_uint(t1), tag), st_gran(out + 1 * 64 + c, __float_as_uint(t2), tag);
            st_gran(out + 2 * 64 + c, __float_as_uint(piv[nt]), tag);
        }
    }
}

// forward B fragments of a (64, 64) weight: lane (n = 32 nt + i, k group h) of k-step s holds W[16 s + 8 h .. + 7][n]
template <int PF>
__device__ __forceinline__ void pst_stage_fwd_weights(const float* W, u32x4 (*Wf)[4][PF][64]) {
    for (int f = pst_tid(); f < 2 * 4 * 64; f += blockDim.x) {
        const int l = f & 63, s4 = (f >> 6) & 3, nt = f >> 8;
        const float* src = W + (size_t)(16 * s4 + 8 * (l >> 5)) * 64 + 32 * nt + (l & 31);
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = src[(size_t)u * 64];
        bf16x8 p[PF];
        ch_split<PF>(v, p);
#pragma unroll
        for (int pc = 0; pc < PF; ++pc) Wf[nt][s4][pc][l] = __builtin_bit_cast(u32x4, p[pc]);
    }
}

// ----------------------------------------------------------------------------------------------------------------
// FORWARD.  Per block b:   (models/epc-net.py:70-83)
//   G  x = relu(bn0(z0)) formed as rows are gathered;  xm = mean over the selected neighbours;  d = xm - x;  za = d Wa + ba
//   M  zb = relu(bna(za)) Wb + bb                                   (za from the wave's tile)
//   H  out = relu(bnb(zb)) + xm -> cat slice (+ its bf16 copy);  z0' = out W0' + b0' for the next block    (zb from the tile)
// with one barrier (and the moments of the tensor just formed) between consecutive phases; before the first G the moments of the
// input z0 (conv1's pre-activation).  d, za, zb are written for the backward, z0' for the other workgroups' gathers (write-through).
// ----------------------------------------------------------------------------------------------------------------
// kernel argument: the public descriptor + the launch geometry
struct PstFwdArgs {
    epc_chain_fwd_args a;
    int rows, wg_rows, width;
    float kdiv;
    long long budget;
};

// -DPST_STAMPS (scripts/time_chain_persist.py STAMPS=1): thread 0 of every workgroup leaves s_memrealtime stamps behind the group
// partials of gstats ([workgroup][64] long long) -- where a phase's microseconds go and how far apart the workgroups arrive.
#ifdef PST_STAMPS
#define PST_STAMP() do { if (threadIdx.x == 0 && sidx < 64) stamps[sidx++] = wall_clock64(); } while (0)
#else
#define PST_STAMP() do { } while (0)
#endif

template <int PF>
__global__ __launch_bounds__(64 * PST_WAVES) void chain_fwd_persist_kernel(PstFwdArgs g) {
    extern __shared__ __attribute__((aligned(16))) float tiles_all[];   // one staging tile per wave
    __shared__ u32x4 Wf[2][4][PF][64];
    __shared__ __attribute__((aligned(16))) double scratch[PST_SCRATCH_BYTES / 8];   // polls' landing zone; the waves' moment merge; phase S
    __shared__ __attribute__((aligned(16))) float coef[2][64];
    __shared__ __attribute__((aligned(16))) float s_mean[64], s_var[64];
    __shared__ int snrows[PST_WAVES];
    float (*sred)[3][64] = reinterpret_cast<float (*)[3][64]>(scratch);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (the wave index as a scalar)
    const int q = lane & 15;
    const int rows = g.rows;
    char* ws = reinterpret_cast<char*>(g.a.workspace);
    unsigned* sync = reinterpret_cast<unsigned*>(ws);
    if (ld_coh(sync + PST_W_ERR) != 0u) {         // an earlier launch was abandoned and not reset: NaN out (as an abandoned launch does)
        const int lb_ = xcd_contiguous_block(blockIdx.x, gridDim.x);
        for (int o = threadIdx.x; o < g.width; o += blockDim.x) g.a.cat[(size_t)lb_ * g.wg_rows * g.width + o] = __int_as_float(0x7fc00000);
        return;
    }
    PstCtx cx = pst_init(sync, g.budget);
#ifdef PST_STAMPS
    long long* stamps = reinterpret_cast<long long*>(ws + PST_WS_STAMPS) + (size_t)blockIdx.x * 64;
    int sidx = 0;
#endif
    PST_STAMP();
    const int wg0 = cx.lb * g.wg_rows, tiles = min(g.wg_rows, rows - wg0 + 31) / 32;
    const int wg_rows_here = min(g.wg_rows, rows - wg0);
    float* tile = tiles_all + wave * CH_STG_FLOATS;
    const int base = wg0 + wave * 32;   // the wave's tile (wave < tiles)
    const bool have = wave < tiles;
    int phase = 0;
    // an abandoned launch leaves NaN in the first row of the workgroup's slice of the concat: whatever consumes the result sees it
    // (the loss is NaN) even when nobody asks epc_chain_persist_status
    auto poison = [&]() {
        for (int o = threadIdx.x; o < g.width; o += blockDim.x) g.a.cat[(size_t)wg0 * g.width + o] = __int_as_float(0x7fc00000);
    };
    // The wave's tile as it stands -> its rows of a dense (rows, 64) tensor, 16 lanes x float4 per row.  The tensors only LATER launches
    // read (za, zb for the backward; the concat) leave this way AFTER the phase's partial is posted: their stores drain under the
    // barrier's round trips instead of in front of the post (a wave's stores and the loads of its polls retire in order).
    auto store_tile_rows = [&](float* dst) {
        const int ln = pst_tid() & 63, p4_ = ln >> 4, q_ = ln & 15;
#pragma unroll
        for (int r8 = 0; r8 < 8; ++r8) {
            const int rl = 4 * r8 + p4_, pt = base + rl;
            if (pt < rows) *reinterpret_cast<float4*>(dst + (size_t)pt * 64 + 4 * q_) = *reinterpret_cast<const float4*>(tile + rl * CH_STG_STRIDE + 4 * q_);
        }
    };
    auto stats_of = [&](int ph) { return reinterpret_cast<pst_gran*>(ws + PST_WS_PARTIALS) + (size_t)ph * PST_MAX_PARTS * 192; };
    auto gstats_of = [&](int ph) { return reinterpret_cast<pst_gran*>(ws + PST_WS_GROUPS) + (size_t)ph * 8 * 384; };
    // one barrier: (the caller has posted its partial) -> the group's first workgroup reduces its group -> everyone merges the groups
    auto barrier_moments = [&](const float* bias, const float* gamma, const float* beta, float* mean_out, float* var_out) -> bool {
        const unsigned tag = pst_tag(cx, phase);
        __syncthreads();   // (wave 0 is done with the waves' merge in `scratch`: the polls land there)
        if (cx.leader && !pst_group_reduce<3>(cx, stats_of(phase), gstats_of(phase), tag, rows, g.wg_rows, scratch)) return false;
        PST_STAMP();
        if (!pst_gather_groups<3>(cx, gstats_of(phase), tag, scratch)) return false;
        PST_STAMP();
        pst_final_moments(cx, scratch, rows, g.wg_rows, bias, gamma, beta, g.a.eps, mean_out, var_out, s_mean, s_var, coef);
        phase += 1;
        return true;
    };

    // ---- phase S: moment partial of the workgroup's rows of z01 (pivot: its first row); 16 lanes x float4 per row ----
    {
        const float* z = g.a.blk[0].z0;
        const int slot = tid >> 4, nslots = blockDim.x >> 4;
        const float4 pv = *reinterpret_cast<const float4*>(z + (size_t)wg0 * 64 + 4 * q);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        for (int r = wg0 + slot; r < wg0 + wg_rows_here; r += nslots) {
            const float4 v = *reinterpret_cast<const float4*>(z + (size_t)r * 64 + 4 * q);
            const float d0 = v.x - pv.x, d1 = v.y - pv.y, d2 = v.z - pv.z, d3 = v.w - pv.w;
            a.x += d0, a.y += d1, a.z += d2, a.w += d3;
            b.x += d0 * d0, b.y += d1 * d1, b.z += d2 * d2, b.w += d3 * d3;
        }
        float* red = reinterpret_cast<float*>(scratch);   // [2][48][64]
        *reinterpret_cast<float4*>(red + (0 * 48 + slot) * 64 + 4 * q) = a;
        *reinterpret_cast<float4*>(red + (1 * 48 + slot) * 64 + 4 * q) = b;
        __syncthreads();
        pst_gran* out = stats_of(phase) + (size_t)cx.lb * 192;
        const unsigned tag = pst_tag(cx, phase);
        for (int o = tid; o < 192; o += blockDim.x) {
            const int k = o >> 6, c = o & 63;
            float t;
            if (k < 2) {
                t = 0.f;
                for (int s = 0; s < nslots; ++s) t += red[(k * 48 + s) * 64 + c];
            } else {
                t = z[(size_t)wg0 * 64 + c];
            }
            st_gran(out + o, __float_as_uint(t), tag);
        }
        __syncthreads();   // (red is scratch: the polls may overwrite it)
    }
    pst_stage_fwd_weights<PF>(g.a.blk[0].Wa, Wf);
    PST_STAMP();

#pragma unroll 1
    for (int b = 0; b < g.a.nblocks; ++b) {
        const epc_chain_fwd_block& B = g.a.blk[b];
        const int lane = pst_tid() & 63;   // (lane indices re-derived per block: not held across the whole kernel)
        const int i = lane & 31, h = lane >> 5;
        const int p4 = lane >> 4, q = lane & 15;
        // ================= barrier: z0's moments =================
        if (!barrier_moments(B.in_bias, B.gamma0, B.beta0, B.mean0, B.var0)) { poison(); return; }
        PST_STAMP();
        // ================= phase G =================
        float s1[2] = {0.f, 0.f}, s2[2] = {0.f, 0.f}, piv[2] = {0.f, 0.f};
        int my_rows = 0;
        // the neighbour means of the wave's tile stay in REGISTERS until phase H adds them back (lane (p4, q): points 4 r8 + p4, channels
        // 4 q ..): as a tensor they were 23 MB written and read per block, and the read stood in front of a barrier's polls
        float4 xmr[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) xmr[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (have) {
            const float4 cs = *reinterpret_cast<const float4*>(&coef[0][4 * q]), ct = *reinterpret_cast<const float4*>(&coef[1][4 * q]);
            auto act = [&](const float4& v) {   // relu(bn0(.)) of the lane's four channels: the forward's own expression
                return make_float4(fmaxf(v.x * cs.x + ct.x, 0.f), fmaxf(v.y * cs.y + ct.y, 0.f), fmaxf(v.z * cs.z + ct.z, 0.f),
                                   fmaxf(v.w * cs.w + ct.w, 0.f));
            };
            const float4* z4 = reinterpret_cast<const float4*>(B.z0);
            // (Round 6: the next iteration's count and first 16 entries requested under this iteration's rows -- what took chain_bwd_gather
            // from 37 to 27 us -- measured 291 us against 280 here: 4 registers over the 168 of twelve waves, and this loop's three round
            // trips per iteration are already hidden by the other eleven waves.  Not kept.)
#pragma unroll 1
            for (int r8 = 0; r8 < 8; ++r8) {
                const int pt = base + 4 * r8 + p4;
                float4 dd = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pt < rows) {
                    const int cloud_base = (pt / g.a.n) * g.a.n;
                    const int c = g.a.cnt[pt];
                    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
                    auto add = [&](const float4& v) {
                        const float4 y = act(v);
                        acc.x += y.x, acc.y += y.y, acc.z += y.z, acc.w += y.w;
                    };
                    if (c <= g.a.cap) {
                        int m = 0;
                        if (c >= 20 && g.a.cap % 4 == 0) {
                            const int4* il = reinterpret_cast<const int4*>(g.a.idx + (size_t)pt * g.a.cap);
                            int nb[20];
#pragma unroll
                            for (int m4 = 0; m4 < 5; ++m4) {
                                const int4 tq = il[m4];
                                nb[4 * m4] = tq.x, nb[4 * m4 + 1] = tq.y, nb[4 * m4 + 2] = tq.z, nb[4 * m4 + 3] = tq.w;
                            }
                            float4 v[20];
#pragma unroll
                            for (int u = 0; u < 20; ++u) v[u] = z4[(unsigned)(cloud_base + nb[u]) * 16u + (unsigned)q];   // (32-bit offsets: rows * 16 < 2^32)
#pragma unroll
                            for (int u = 0; u < 20; ++u) add(v[u]);
                            m = 20;
                        }
                        for (; m < c; ++m) add(z4[(size_t)(cloud_base + g.a.idx[(size_t)pt * g.a.cap + m]) * 16 + q]);
                    } else {
                        const float* pc = g.a.xyz + (size_t)cloud_base * 3;
                        const int ii = pt - cloud_base;
                        const float xi = pc[3 * ii], yi = pc[3 * ii + 1], zi = pc[3 * ii + 2];
                        const float sqi = sq3(xi, yi, zi), kv = g.a.kth[pt];
                        for (int j = 0; j < g.a.n; ++j) {
                            const float xj = pc[3 * j], yj = pc[3 * j + 1], zj = pc[3 * j + 2];
                            if (neg_sq_dist(sqi, xi, yi, zi, xj, yj, zj, sq3(xj, yj, zj)) >= kv) add(z4[(size_t)(cloud_base + j) * 16 + q]);
                        }
                    }
                    acc.x /= g.kdiv, acc.y /= g.kdiv, acc.z /= g.kdiv, acc.w /= g.kdiv;
                    const float4 own = act(z4[(size_t)pt * 16 + q]);
                    dd = make_float4(acc.x - own.x, acc.y - own.y, acc.z - own.z, acc.w - own.w);
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        if (r8 == k) xmr[k] = acc;   // (r8 is wave-uniform: selects, no dynamic register index)
                    reinterpret_cast<float4*>(B.d)[(size_t)pt * 16 + q] = dd;
                }
                *reinterpret_cast<float4*>(tile + (4 * r8 + p4) * CH_STG_STRIDE + 4 * q) = dd;   // (rows past the end: zeros)
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave: its own tile writes have landed before its reads
            bf16x8 a[4][PF];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                float v[8];
                ch_ld8(tile + i * CH_STG_STRIDE + 16 * s4 + 8 * h, v);
                ch_split<PF>(v, a[s4]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // in registers: the tile takes za next
            const float b0 = B.ba ? B.ba[i] : 0.f, b1 = B.ba ? B.ba[32 + i] : 0.f;
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
                piv[nt] = __shfl(acc[0], i);   // row `base`
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl4 = (r & 3) + 8 * (r >> 2), rl = rl4 + 4 * h, rr = base + rl;
                    const float v = acc[r];
                    if (rr < rows) {
                        const float dlt = v - piv[nt];
                        s1[nt] += dlt;
                        s2[nt] += dlt * dlt;
                    }
                    tile[rl * CH_STG_STRIDE + 32 * nt + i] = v + bv;
                }
            }
            my_rows = min(32, rows - base);
        }
        PST_STAMP();
        pst_post_stats(s1, s2, piv, my_rows, sred, snrows, stats_of(phase) + (size_t)cx.lb * 192, pst_tag(cx, phase));
        PST_STAMP();
        if (have) store_tile_rows(B.za);       // (after the post: under the barrier)
        pst_stage_fwd_weights<PF>(B.Wb, Wf);   // (every wave is past its products: pst_post_stats' barrier) -- under the wait
        // ================= barrier: za's moments =================
        if (!barrier_moments(B.ba, B.gamma_a, B.beta_a, B.mean_a, B.var_a)) { poison(); return; }
        PST_STAMP();
        // ================= phase M =================
        s1[0] = s1[1] = s2[0] = s2[1] = piv[0] = piv[1] = 0.f;
        if (have) {
            bf16x8 a[4][PF];
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                const int c0 = 16 * s4 + 8 * h;
                float cs[8], ct[8], zr[8], v[8];
                ch_ld8(&coef[0][c0], cs), ch_ld8(&coef[1][c0], ct);
                ch_ld8(tile + i * CH_STG_STRIDE + c0, zr);
                const bool ok = base + i < rows;
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = ok ? fmaxf(zr[u] * cs[u] + ct[u], 0.f) : 0.f;
                ch_split<PF>(v, a[s4]);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // in registers: the tile takes zb next
            const float b0 = B.bb ? B.bb[i] : 0.f, b1 = B.bb ? B.bb[32 + i] : 0.f;
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
                piv[nt] = __shfl(acc[0], i);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int rl4 = (r & 3) + 8 * (r >> 2), rl = rl4 + 4 * h, rr = base + rl;
                    const float v = acc[r];
                    if (rr < rows) {
                        const float dlt = v - piv[nt];
                        s1[nt] += dlt;
                        s2[nt] += dlt * dlt;
                    }
                    tile[rl * CH_STG_STRIDE + 32 * nt + i] = v + bv;
                }
            }
        }
        PST_STAMP();
        pst_post_stats(s1, s2, piv, my_rows, sred, snrows, stats_of(phase) + (size_t)cx.lb * 192, pst_tag(cx, phase));
        PST_STAMP();
        if (have) store_tile_rows(B.zb);
        if (B.W0_next) pst_stage_fwd_weights<PF>(B.W0_next, Wf);
        // ================= barrier: zb's moments =================
        if (!barrier_moments(B.bb, B.gamma_b, B.beta_b, B.mean_b, B.var_b)) { poison(); return; }
        PST_STAMP();
        // ================= phase H =================
        s1[0] = s1[1] = s2[0] = s2[1] = piv[0] = piv[1] = 0.f;
        if (have) {
            const float4 cs = *reinterpret_cast<const float4*>(&coef[0][4 * q]), ct = *reinterpret_cast<const float4*>(&coef[1][4 * q]);
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int rl = 4 * r8 + p4, pt = base + rl;
                const float4 z = *reinterpret_cast<const float4*>(tile + rl * CH_STG_STRIDE + 4 * q);
                float4 v;
                v.x = fmaxf(z.x * cs.x + ct.x, 0.f), v.y = fmaxf(z.y * cs.y + ct.y, 0.f), v.z = fmaxf(z.z * cs.z + ct.z, 0.f),
                v.w = fmaxf(z.w * cs.w + ct.w, 0.f);
                v.x += xmr[r8].x, v.y += xmr[r8].y, v.z += xmr[r8].z, v.w += xmr[r8].w;
                if (pt >= rows) v = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(tile + rl * CH_STG_STRIDE + 4 * q) = v;   // (the concat's slice leaves from here after the post)
            }
            if (B.W0_next) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                bf16x8 a[4][PF];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    float v[8];
                    ch_ld8(tile + i * CH_STG_STRIDE + 16 * s4 + 8 * h, v);
                    ch_split<PF>(v, a[s4]);
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const float b0 = B.b0_next ? B.b0_next[i] : 0.f, b1 = B.b0_next ? B.b0_next[32 + i] : 0.f;
                float* zlane = B.z0_next + (size_t)(base + 4 * h) * 64 + i;
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
                    piv[nt] = __shfl(acc[0], i);
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int rl4 = (r & 3) + 8 * (r >> 2), rr = base + rl4 + 4 * h;
                        if (rr < rows) {
                            const float v = acc[r];
                            const float dlt = v - piv[nt];
                            s1[nt] += dlt;
                            s2[nt] += dlt * dlt;
                            st_wt1(zlane + rl4 * 64 + 32 * nt, v + bv);   // gathered by OTHER workgroups after the barrier
                        }
                    }
                }
            }
        }
        PST_STAMP();
        if (B.W0_next) {
            pst_post_stats(s1, s2, piv, my_rows, sred, snrows, stats_of(phase) + (size_t)cx.lb * 192, pst_tag(cx, phase));
            PST_STAMP();
        }
        if (have) {   // the block's slice of the concat (and its bf16 copy) from the tile: after the post, under the barrier
            const int ln = pst_tid() & 63, p4_ = ln >> 4, q_ = ln & 15;
            float* cat_b = g.a.cat + 64 * b;
            unsigned short* cat16_b = g.a.cat_bf16 ? (unsigned short*)g.a.cat_bf16 + 64 * b : nullptr;
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) {
                const int rl = 4 * r8 + p4_, pt = base + rl;
                if (pt < rows) {
                    const float4 v = *reinterpret_cast<const float4*>(tile + rl * CH_STG_STRIDE + 4 * q_);
                    *reinterpret_cast<float4*>(cat_b + (size_t)pt * g.width + 4 * q_) = v;
                    if (cat16_b) {
                        const pst_bf16x4 pk = {(__bf16)v.x, (__bf16)v.y, (__bf16)v.z, (__bf16)v.w};
                        *reinterpret_cast<uint2*>(cat16_b + (size_t)pt * g.width + 4 * q_) = __builtin_bit_cast(uint2, pk);
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the tile is read: the next block's gather stages d in it
        }
        if (B.W0_next) pst_stage_fwd_weights<PF>(g.a.blk[b + 1].Wa, Wf);
    }
    pst_exit(cx);
}

// ---- C ABI ------------------------------------------------------------------------------------------------------
static bool pst_aligned16(const void* p) { return (reinterpret_cast<size_t>(p) & 15) == 0; }
#define PST_DEFAULT_BUDGET 25000000ll   // a quarter second of the 100-MHz s_memrealtime

// rows per workgroup: the chain's own geometry (epc_chain_parts workgroups, the 32-row tiles spread evenly over the CUs)
static int pst_wg_rows(int rows) {
    const int parts = epc_chain_parts(rows);
    if (parts <= 0) return 0;
    const int tiles = (rows + 31) / 32, per = (tiles + parts - 1) / parts;
    return 32 * per;
}

// co-resident workgroups per CU of the two kernels at the geometry of `rows`, by the occupancy query (cached per block size)
template <typename K>
static int pst_occupancy(K kernel, int threads, size_t lds) {
    int occ = 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, threads, lds) != hipSuccess) return 0;
    return occ;
}

extern "C" int epc_chain_persist_ok(int rows) {
    if (rows <= 0) return 0;
    const int wg_rows = pst_wg_rows(rows), nw = wg_rows / 32, parts = epc_chain_parts(rows);
    if (nw < 1 || nw > PST_WAVES) return 0;
    if ((long)parts * wg_rows < rows) return 0;
    const size_t lds = (size_t)nw * PST_TILE_BYTES;
    const int cus = epc_device_cu_count();
    static int cached[PST_WAVES + 1] = {0};   // 0: unknown, 1: fits, 2: does not (per wave count; a racing first call writes the same value)
    if (cached[nw] == 0) {
        const int o1 = pst_occupancy(chain_fwd_persist_kernel<1>, 64 * nw, lds), o3 = pst_occupancy(chain_fwd_persist_kernel<3>, 64 * nw, lds);
        cached[nw] = (o1 >= 1 && o3 >= 1) ? 1 : 2;
    }
    return cached[nw] == 1 && parts <= cus;
}

extern "C" size_t epc_chain_persist_workspace_bytes(void) { return PST_WS_BYTES; }

extern "C" int epc_chain_fwd_persist(const epc_chain_fwd_args* a, int pieces, void* stream) {
    EPC_CHECK_ARG(a && a->nblocks >= 1 && a->nblocks <= EPC_CHAIN_MAX_BLOCKS, "null descriptor / 1 .. 4 blocks");
    EPC_CHECK_ARG(pieces == 3 || pieces == 1, "pieces: 3 or 1");
    EPC_CHECK_ARG(a->xyz && a->idx && a->cnt && a->kth && a->cat && a->workspace, "null pointer");
    EPC_CHECK_ARG(a->num_clouds > 0 && a->n > 0 && a->knn > 0 && a->cap >= EPC_KNN_SELECT, "bad shape");
    EPC_CHECK_ARG((long)a->num_clouds * a->n < (1L << 31) / 64, "too many rows");
    const int rows = a->num_clouds * a->n;
    EPC_CHECK_ARG(epc_chain_persist_ok(rows), "rows not covered by the persistent chain (epc_chain_persist_ok)");
    EPC_CHECK_ARG(pst_aligned16(a->cat) && pst_aligned16(a->cat_bf16) && pst_aligned16(a->idx) && pst_aligned16(a->workspace),
                  "tensors must be 16-byte aligned");
    EPC_CHECK_ARG(epc_chain_parts(rows) <= PST_MAX_PARTS && 1 + 3 * a->nblocks <= PST_MAX_PHASES, "grid / phase count beyond the workspace layout");
    for (int b = 0; b < a->nblocks; ++b) {
        const epc_chain_fwd_block& B = a->blk[b];
        const bool last = b + 1 == a->nblocks;
        EPC_CHECK_ARG(B.gamma0 && B.beta0 && B.Wa && B.gamma_a && B.beta_a && B.Wb && B.gamma_b && B.beta_b && B.z0 && B.mean0 && B.var0 &&
                          B.mean_a && B.var_a && B.mean_b && B.var_b && B.d && B.za && B.zb,
                      "null pointer in a block");
        EPC_CHECK_ARG(last ? (!B.W0_next && !B.z0_next) : (B.W0_next && B.z0_next), "W0_next / z0_next: both in every block but the last");
        EPC_CHECK_ARG(b == 0 || B.z0 == a->blk[b - 1].z0_next, "a block's z0 is the previous block's z0_next");
        EPC_CHECK_ARG(pst_aligned16(B.z0) && pst_aligned16(B.d) && pst_aligned16(B.za) && pst_aligned16(B.zb) && pst_aligned16(B.z0_next),
                      "tensors must be 16-byte aligned");
    }
    PstFwdArgs g;
    g.a = *a;
    g.rows = rows, g.wg_rows = pst_wg_rows(rows), g.width = 64 * a->nblocks, g.kdiv = (float)a->knn;
    g.budget = a->spin_ticks > 0 ? a->spin_ticks : PST_DEFAULT_BUDGET;
    const int nw = g.wg_rows / 32;
    const dim3 grid(epc_chain_parts(rows)), block(64 * nw);
    const size_t lds = (size_t)nw * PST_TILE_BYTES;
    if (pieces == 3) hipLaunchKernelGGL(chain_fwd_persist_kernel<3>, grid, block, lds, (hipStream_t)stream, g);
    else hipLaunchKernelGGL(chain_fwd_persist_kernel<1>, grid, block, lds, (hipStream_t)stream, g);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_chain_persist_init(void* workspace, void* stream) {
    EPC_CHECK_ARG(workspace && pst_aligned16(workspace), "null / unaligned workspace");
    if (hipMemsetAsync(workspace, 0, PST_WS_BYTES, (hipStream_t)stream) != hipSuccess) {
        epc_set_error("%s: hipMemsetAsync failed", __func__);
        return EPC_EHIP;
    }
    return EPC_OK;
}

extern "C" int epc_chain_persist_status(const void* workspace, void* stream) {
    EPC_CHECK_ARG(workspace, "null pointer");
    unsigned err = 0;
    if (hipMemcpyAsync(&err, reinterpret_cast<const unsigned*>(workspace) + PST_W_ERR, sizeof(err), hipMemcpyDeviceToHost, (hipStream_t)stream) != hipSuccess ||
        hipStreamSynchronize((hipStream_t)stream) != hipSuccess) {
        epc_set_error("%s: reading the error word failed", __func__);
        return EPC_EHIP;
    }
    if (err != 0) {
        epc_set_error("%s: a persistent chain launch was abandoned (a grid barrier ran out of its spin budget: some workgroup was not resident)", __func__);
        return EPC_EHIP;
    }
    return EPC_OK;
}

// after an abandoned launch: a NEW sequence number (the abandoned launch's tags must not be taken for the next launch's), the exit
// counter and the error word cleared
__global__ void pst_reset_kernel(unsigned* w) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        st_wt_u(w + PST_W_SEQ, ld_coh(w + PST_W_SEQ) + 1u);
        st_wt_u(w + PST_W_EXIT, 0u);
        st_wt_u(w + PST_W_ERR, 0u);
    }
}
extern "C" int epc_chain_persist_reset(void* workspace, void* stream) {
    EPC_CHECK_ARG(workspace, "null pointer");
    hipLaunchKernelGGL(pst_reset_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, reinterpret_cast<unsigned*>(workspace));
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}
