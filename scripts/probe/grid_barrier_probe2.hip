// Grid-wide barrier probe, second pass (MI355X).  The first pass (grid_barrier_probe.hip) priced the textbook barrier -- agent-scope
// release fence, atomic add, poll, acquire fence, as cooperative groups do it -- at 6-10 us bare and 24-35 us with 72 KB of rows
// written per workgroup and round: every workgroup's `buffer_wbl2 sc1` walks its XCD's whole L2, and 256 pollers of one word
// serialise behind the arrivals at the memory side.  A kernel boundary costs 4.2 us, so that barrier loses.
// This pass prices protocols that need NO cache maintenance:
//   * data that another workgroup will read is stored WRITE-THROUGH (`global_store ... sc1`: agent scope; the store's completion,
//     s_waitcnt vmcnt(0), means it has reached the level every XCD's L2 misses go to);
//   * every such location is written exactly ONCE per launch, before its first read (so no L1 / L2 anywhere can hold a stale copy:
//     the launch itself starts with both invalidated) -- readers use plain loads and keep their L2 hit rates;
//   * arrival / release through words that are only ever touched by atomics and sc1 accesses.
// Protocols (template S):
//   1  one counter; everyone polls it
//   2  8 group counters (blockIdx & 7) -> the last of a group bumps the global counter; everyone polls the global counter
//   3  as 2, but only the group's last arrival polls the global counter and then publishes the epoch in its group's flag; the
//      others poll that flag (8 different addresses)
//   4  no read-modify-write at all: workgroup w stores epoch into flag[w]; everyone's wave 0 reads all flags (1 KB, one load per lane)
//   5  flags in two levels: flag[w]; the leader of a group of 32 (w & 31 == 0 ... by blockIdx >> 5) polls its group's 32 flags (one
//      128-byte line) and publishes gflag[g]; everyone polls the 8 gflags (one 32-byte read)
// Every spin is bounded (0.2 s of s_memrealtime); a time-out sets the error word and every workgroup leaves.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define SPIN_BUDGET_TICKS 20000000ll
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void st_wt(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_wt1(float* p, float v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void st_wt_u(unsigned* p, unsigned v) { asm volatile("global_store_dword %0, %1, off sc1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ unsigned ld_coh(const unsigned* p) {
    unsigned v;
    asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}
__device__ __forceinline__ u32x4 ld_coh4(const unsigned* p) {
    u32x4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    return v;
}

struct Bar {
    unsigned* w;      // [0] global counter; [64 * (1 + g)] group counters; [1024 + 64 g] group flags; [2048 + wg] per-workgroup flags; [4096 + 64 g] gflags
    unsigned* err;
    int sleep;        // s_sleep argument between polls is fixed at compile time; this selects 0: none, 1: s_sleep 1, 2: s_sleep 4, 3: s_sleep 16
};
__device__ __forceinline__ void nap(int k) {
    if (k == 1) __builtin_amdgcn_s_sleep(1);
    else if (k == 2) __builtin_amdgcn_s_sleep(4);
    else if (k == 3) __builtin_amdgcn_s_sleep(16);
}

template <int S>
__device__ __forceinline__ bool grid_barrier(const Bar& b, unsigned nwg, unsigned& epoch) {
    __shared__ int ok_s;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's write-through stores have completed
    __syncthreads();
    epoch += 1;
    const unsigned wg = blockIdx.x;
    if (S == 4) {
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            if (lane == 0) st_wt_u(b.w + 2048 + wg, epoch);
            int ok = 1;
            const long long t0 = wall_clock64();
            unsigned it = 0;
            for (;;) {
                bool all = true;
                if (4u * lane < nwg) {
                    const u32x4 f = ld_coh4(b.w + 2048 + 4 * lane);
                    all = f[0] >= epoch && (4u * lane + 1 >= nwg || f[1] >= epoch) && (4u * lane + 2 >= nwg || f[2] >= epoch) &&
                          (4u * lane + 3 >= nwg || f[3] >= epoch);
                }
                if (__all(all)) break;
                if ((++it & 63u) == 0u) {
                    if (ld_coh(b.err) != 0u) { ok = 0; break; }
                    if (wall_clock64() - t0 > SPIN_BUDGET_TICKS) { st_wt_u(b.err, 1u); ok = 0; break; }
                }
                nap(b.sleep);
            }
            if (lane == 0) ok_s = ok;
        }
        __syncthreads();
        return ok_s != 0;
    }
    if (S == 5) {
        if (threadIdx.x < 64) {
            const int lane = threadIdx.x;
            const unsigned g = wg >> 5, ng = (nwg + 31) >> 5;
            if (lane == 0) st_wt_u(b.w + 2048 + wg, epoch);
            int ok = 1;
            const long long t0 = wall_clock64();
            unsigned it = 0;
            if ((wg & 31) == 0) {   // the group's leader: its 32 flags are one 128-byte line
                for (;;) {
                    bool mine = true;
                    if (lane < 8) {
                        const u32x4 f = ld_coh4(b.w + 2048 + 32 * g + 4 * lane);
                        const unsigned base = 32 * g + 4 * lane;
                        mine = (base >= nwg || f[0] >= epoch) && (base + 1 >= nwg || f[1] >= epoch) && (base + 2 >= nwg || f[2] >= epoch) &&
                               (base + 3 >= nwg || f[3] >= epoch);
                    }
                    if (__all(mine)) break;
                    if ((++it & 63u) == 0u) {
                        if (ld_coh(b.err) != 0u) { ok = 0; break; }
                        if (wall_clock64() - t0 > SPIN_BUDGET_TICKS) { st_wt_u(b.err, 1u); ok = 0; break; }
                    }
                    nap(b.sleep);
                }
                if (lane == 0 && ok) st_wt_u(b.w + 4096 + g, epoch);
            }
            while (ok) {
                bool all = true;
                if (lane < 2) {
                    const u32x4 f = ld_coh4(b.w + 4096 + 4 * lane);
                    const unsigned base = 4 * lane;
                    all = (base >= ng || f[0] >= epoch) && (base + 1 >= ng || f[1] >= epoch) && (base + 2 >= ng || f[2] >= epoch) &&
                          (base + 3 >= ng || f[3] >= epoch);
                }
                if (__all(all)) break;
                if ((++it & 63u) == 0u) {
                    if (ld_coh(b.err) != 0u) { ok = 0; break; }
                    if (wall_clock64() - t0 > SPIN_BUDGET_TICKS) { st_wt_u(b.err, 1u); ok = 0; break; }
                }
                nap(b.sleep);
            }
            if (lane == 0) ok_s = ok;
        }
        __syncthreads();
        return ok_s != 0;
    }
    if (threadIdx.x == 0) {
        int ok = 1;
        bool leader = false;
        const unsigned g = wg & 7, ng = nwg < 8 ? nwg : 8u;
        if (S == 1) {
            __hip_atomic_fetch_add(b.w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const unsigned per = (nwg + 7 - g) / 8;
            const unsigned old = __hip_atomic_fetch_add(b.w + 64 * (1 + g), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == per * epoch) {
                leader = true;
                __hip_atomic_fetch_add(b.w, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        const unsigned target = (S == 1 ? nwg : ng) * epoch;
        const long long t0 = wall_clock64();
            unsigned it = 0;
        if (S != 3 || leader) {
            while (ld_coh(b.w) < target) {
                if ((++it & 63u) == 0u) {
                    if (ld_coh(b.err) != 0u) { ok = 0; break; }
                    if (wall_clock64() - t0 > SPIN_BUDGET_TICKS) { st_wt_u(b.err, 1u); ok = 0; break; }
                }
                nap(b.sleep);
            }
            if (S == 3 && ok) st_wt_u(b.w + 1024 + 64 * g, epoch);
        } else {
            while (ld_coh(b.w + 1024 + 64 * g) < epoch) {
                if ((++it & 63u) == 0u) {
                    if (ld_coh(b.err) != 0u) { ok = 0; break; }
                    if (wall_clock64() - t0 > SPIN_BUDGET_TICKS) { st_wt_u(b.err, 1u); ok = 0; break; }
                }
                nap(b.sleep);
            }
        }
        ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
}

template <int S>
__global__ __launch_bounds__(768) void bar_only(Bar b, int nbar, unsigned* out) {
    unsigned epoch = 0;
    for (int i = 0; i < nbar; ++i)
        if (!grid_barrier<S>(b, gridDim.x, epoch)) return;
    if (threadIdx.x == 0) out[blockIdx.x] = epoch;
}

// per round (round i uses ITS OWN partial and row buffers: write-once): every workgroup writes `row_floats` floats of rows and a
// 768-byte partial (write-through), barrier, pools all partials and checks 24 rows of three other workgroups with PLAIN loads.
template <int S>
__global__ __launch_bounds__(768) void bar_pool(Bar b, int nbar, float* partials, float* rows, int row_floats, unsigned* out) {
    __shared__ float red[16][64];
    unsigned epoch = 0;
    const int tid = threadIdx.x, nwg = gridDim.x;
    unsigned bad = 0, badrow = 0;
    for (int i = 0; i < nbar; ++i) {
        float* pbuf = partials + (size_t)i * nwg * 192;
        float* rbuf = rows + (size_t)i * nwg * row_floats;
        if (tid < 192) st_wt1(pbuf + (size_t)blockIdx.x * 192 + tid, (float)(i + 1));
        for (int o = tid * 4; o < row_floats; o += 768 * 4) {
            const f32x4 v = {1.f, (float)blockIdx.x, 3.f, (float)i};
            st_wt(rbuf + (size_t)blockIdx.x * row_floats + o, v);
        }
        if (!grid_barrier<S>(b, nwg, epoch)) return;
        const int cq = tid & 15, ps = tid >> 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ps < 16) {
            for (int t = ps; t < nwg; t += 16) {
                const float4 v = *reinterpret_cast<const float4*>(pbuf + (size_t)t * 192 + 4 * cq);
                const float4 w = *reinterpret_cast<const float4*>(pbuf + (size_t)t * 192 + 64 + 4 * cq);
                const float4 u = *reinterpret_cast<const float4*>(pbuf + (size_t)t * 192 + 128 + 4 * cq);
                acc.x += v.x + w.x + u.x, acc.y += v.y + w.y + u.y, acc.z += v.z + w.z + u.z, acc.w += v.w + w.w + u.w;
            }
            *reinterpret_cast<float4*>(&red[ps][4 * cq]) = acc;
        }
        if (row_floats > 0 && tid < 3 * 64) {   // rows of workgroups +1, +37, +128 (other XCDs): one float4 per thread
            const int which = tid >> 6, other = (blockIdx.x + (which == 0 ? 1 : which == 1 ? 37 : 128)) % nwg;
            const int o = ((tid & 63) * 292 * 4) % row_floats & ~3;
            const float4 v = *reinterpret_cast<const float4*>(rbuf + (size_t)other * row_floats + o);
            if (v.x != 1.f || v.y != (float)other || v.z != 3.f || v.w != (float)i) badrow += 1;
        }
        __syncthreads();
        if (tid < 64) {
            float t = 0.f;
            for (int s = 0; s < 16; ++s) t += red[s][tid];
            if (t != 3.f * (float)(i + 1) * (float)nwg) bad += 1;
        }
        __syncthreads();
    }
    if (bad) atomicAdd(out + 1024, bad);
    if (badrow) atomicAdd(out + 1025, badrow);
    if (tid == 0) out[blockIdx.x] = epoch;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int S>
static void run_all(int grid, Bar b, unsigned* out, float* partials, float* rows, int row_floats) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](int nbar, auto launch, const char* what) {
        float best = 1e30f;
        for (int rep = 0; rep < 4; ++rep) {
            CK(hipMemsetAsync(b.w, 0, 32768, 0)); CK(hipMemsetAsync(out, 0, 8192, 0));
            CK(hipEventRecord(e0, 0));
            launch(nbar);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        std::vector<unsigned> h(2048);
        CK(hipMemcpy(h.data(), out, 8192, hipMemcpyDeviceToHost));
        unsigned herr; CK(hipMemcpy(&herr, b.err, 4, hipMemcpyDeviceToHost));
        if (herr || h[0] != (unsigned)nbar || h[1024] || h[1025])
            printf("    !! %s nbar %d: err %u epoch %u bad partial sums %u bad rows %u\n", what, nbar, herr, h[0], h[1024], h[1025]);
        CK(hipMemset(b.err, 0, 64));
        return best * 1000.f;
    };
    for (int sl = 0; sl < 4; ++sl) {
        b.sleep = sl;
        auto l0 = [&](int nbar) { hipLaunchKernelGGL(bar_only<S>, dim3(grid), dim3(768), 0, 0, b, nbar, out); };
        auto l1 = [&](int nbar) { hipLaunchKernelGGL(bar_pool<S>, dim3(grid), dim3(768), 0, 0, b, nbar, partials, rows, 0, out); };
        auto l2 = [&](int nbar) { hipLaunchKernelGGL(bar_pool<S>, dim3(grid), dim3(768), 0, 0, b, nbar, partials, rows, row_floats, out); };
        const float a0 = timeit(8, l0, "bare"), c0 = timeit(48, l0, "bare");
        const float a1 = timeit(8, l1, "pool"), c1 = timeit(48, l1, "pool");
        const float a2 = timeit(8, l2, "rows"), c2 = timeit(48, l2, "rows");
        printf("protocol %d grid %3d sleep %d: bare %6.2f us   + partial + pool %6.2f us   + 72 KB rows %6.2f us\n", S, grid, sl, (c0 - a0) / 40.f,
               (c1 - a1) / 40.f, (c2 - a2) / 40.f);
    }
}

int main() {
    int dev = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, bar_pool<1>, 768, 0));
    printf("CUs %d, workgroups of 768 threads per CU (occupancy query) %d\n", cus, occ);
    if (occ < 1) return 1;
    unsigned *w, *err, *out;
    float *partials, *rows;
    const int row_floats = 288 * 64, maxbar = 48;
    CK(hipMalloc(&w, 32768)); CK(hipMalloc(&err, 64)); CK(hipMalloc(&out, 8192));
    CK(hipMalloc(&partials, (size_t)maxbar * cus * 192 * 4)); CK(hipMalloc(&rows, (size_t)maxbar * cus * row_floats * 4));
    CK(hipMemset(err, 0, 64));
    Bar b{w, err, 1};
    for (int grid : {cus, cus / 2}) {
        run_all<1>(grid, b, out, partials, rows, row_floats);
        run_all<2>(grid, b, out, partials, rows, row_floats);
        run_all<3>(grid, b, out, partials, rows, row_floats);
        run_all<4>(grid, b, out, partials, rows, row_floats);
        run_all<5>(grid, b, out, partials, rows, row_floats);
    }
    return 0;
}
