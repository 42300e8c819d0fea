// Grid-wide barrier probe (MI355X): what does one barrier over ONE workgroup per CU cost inside a plain launch, and what does a
// "producer partial -> barrier -> every workgroup pools all partials" round cost -- the dependency a training-mode BatchNorm puts
// between two layers of the chain (csrc/train_chain.hip)?
// Build + run on the GPU box:  hipcc -O3 --offload-arch=gfx950 scripts/probe/grid_barrier_probe.hip -o /tmp/gb && /tmp/gb
//
// Safety: every spin is bounded by a wall-clock budget (s_memrealtime, 100 MHz); a workgroup that runs out sets the error word and
// every workgroup leaves.  The grid never exceeds what the occupancy query says is co-resident.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define SPIN_BUDGET_TICKS 20000000ll   // 0.2 s at 100 MHz

struct Bar {
    unsigned* count;   // arrivals, monotonic within a launch (zeroed by the host before it)
    unsigned* err;     // sticky error word
};

// mode 0: one counter, thread 0 of each workgroup arrives and polls
// mode 1: per-XCD counters (blockIdx & 7), the last arrival of an XCD bumps the global one; everyone polls the global one
template <int MODE>
__device__ __forceinline__ bool grid_barrier(const Bar& b, unsigned nwg, unsigned& epoch) {
    __shared__ int ok_s;
    __syncthreads();
    epoch += 1;
    if (threadIdx.x == 0) {
        int ok = 1;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        if (MODE == 0) {
            __hip_atomic_fetch_add(b.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            const unsigned x = blockIdx.x & 7, per = (nwg + 7 - x) / 8;   // workgroups with this residue
            const unsigned old = __hip_atomic_fetch_add(b.count + 16 * (1 + x), 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (old + 1 == per * epoch) __hip_atomic_fetch_add(b.count, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const unsigned target = (MODE == 0 ? nwg : (nwg < 8 ? nwg : 8u)) * epoch;
        const long long t0 = wall_clock64();
        while (__hip_atomic_load(b.count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            if (__hip_atomic_load(b.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) { ok = 0; break; }
            if (wall_clock64() - t0 > SPIN_BUDGET_TICKS) {
                __hip_atomic_store(b.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = 0;
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        ok_s = ok;
    }
    __syncthreads();
    return ok_s != 0;
}

// nbar barriers and nothing else
template <int MODE>
__global__ __launch_bounds__(768) void bar_only(Bar b, int nbar, unsigned* out) {
    unsigned epoch = 0;
    for (int i = 0; i < nbar; ++i)
        if (!grid_barrier<MODE>(b, gridDim.x, epoch)) return;
    if (threadIdx.x == 0) out[blockIdx.x] = epoch;
}

// per round: every workgroup writes `rows_bytes` of rows and a 768-byte partial, barrier, pools all partials (16 lanes x float4 x 3 per
// partial, 16 slices -- the chain's ch_bn_begin), checks the sum.
template <int MODE>
__global__ __launch_bounds__(768) void bar_pool(Bar b, int nbar, float* partials, float* rows, int row_floats, unsigned* out) {
    __shared__ float red[16][64];
    unsigned epoch = 0;
    const int tid = threadIdx.x, nwg = gridDim.x;
    unsigned bad = 0;
    for (int i = 0; i < nbar; ++i) {
        float* mine = partials + ((size_t)(i & 1) * nwg + blockIdx.x) * 192;
        if (tid < 192) mine[tid] = (float)(i + 1);
        for (int o = tid * 4; o < row_floats; o += 768 * 4)
            *reinterpret_cast<float4*>(rows + (size_t)blockIdx.x * row_floats + o) = make_float4(1.f, 2.f, 3.f, (float)i);
        if (!grid_barrier<MODE>(b, nwg, epoch)) return;
        const int cq = tid & 15, ps = tid >> 4;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ps < 16) {
            const float* base = partials + (size_t)(i & 1) * nwg * 192;
            for (int t = ps; t < nwg; t += 16) {
                const float4 v = *reinterpret_cast<const float4*>(base + (size_t)t * 192 + 4 * cq);
                const float4 w = *reinterpret_cast<const float4*>(base + (size_t)t * 192 + 64 + 4 * cq);
                const float4 u = *reinterpret_cast<const float4*>(base + (size_t)t * 192 + 128 + 4 * cq);
                acc.x += v.x + w.x + u.x, acc.y += v.y + w.y + u.y, acc.z += v.z + w.z + u.z, acc.w += v.w + w.w + u.w;
            }
            *reinterpret_cast<float4*>(&red[ps][4 * cq]) = acc;
        }
        __syncthreads();
        if (tid < 64) {
            float t = 0.f;
            for (int s = 0; s < 16; ++s) t += red[s][tid];
            if (t != 3.f * (float)(i + 1) * (float)nwg) bad += 1;
        }
        __syncthreads();
    }
    if (tid < 64 && bad) atomicAdd(out + 1024, bad);
    if (tid == 0) out[blockIdx.x] = epoch;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

int main() {
    int dev = 0, cus = 0;
    CK(hipGetDevice(&dev));
    CK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, bar_pool<0>, 768, 0));
    printf("CUs %d, workgroups of 768 threads per CU (occupancy query) %d\n", cus, occ);
    if (occ < 1) return 1;
    unsigned *count, *err, *out;
    float *partials, *rows;
    const int row_floats = 288 * 64;   // 72 KB per workgroup (an 18 x 4096 tuple's share)
    CK(hipMalloc(&count, 4096)); CK(hipMalloc(&err, 64)); CK(hipMalloc(&out, 8192));
    CK(hipMalloc(&partials, (size_t)2 * cus * 192 * 4)); CK(hipMalloc(&rows, (size_t)cus * row_floats * 4));
    CK(hipMemset(err, 0, 64));
    Bar b{count, err};
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char* name, int nbar, auto launch) {
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            CK(hipMemsetAsync(count, 0, 4096, 0)); CK(hipMemsetAsync(out, 0, 8192, 0));
            CK(hipEventRecord(e0, 0));
            launch(nbar);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (ms < best) best = ms;
        }
        std::vector<unsigned> h(2048);
        CK(hipMemcpy(h.data(), out, 8192, hipMemcpyDeviceToHost));
        unsigned herr; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
        printf("%-44s nbar %4d: %8.1f us total, err %u, epoch[0] %u, bad %u\n", name, nbar, best * 1000.f, herr, h[0], h[1024]);
        return best * 1000.f;
    };
    for (int grid : {cus, cus / 2, 32}) {
        printf("--- grid %d ---\n", grid);
        for (int mode = 0; mode < 2; ++mode) {
            auto l_only = [&](int nbar) {
                if (mode == 0) hipLaunchKernelGGL(bar_only<0>, dim3(grid), dim3(768), 0, 0, b, nbar, out);
                else hipLaunchKernelGGL(bar_only<1>, dim3(grid), dim3(768), 0, 0, b, nbar, out);
            };
            const float a = timeit(mode ? "barrier only, per-XCD counters" : "barrier only, one counter", 10, l_only);
            const float c = timeit(mode ? "barrier only, per-XCD counters" : "barrier only, one counter", 110, l_only);
            printf("    => %.2f us per barrier\n", (c - a) / 100.f);
            for (int rf : {0, row_floats}) {
                auto l_pool = [&](int nbar) {
                    if (mode == 0) hipLaunchKernelGGL(bar_pool<0>, dim3(grid), dim3(768), 0, 0, b, nbar, partials, rows, rf, out);
                    else hipLaunchKernelGGL(bar_pool<1>, dim3(grid), dim3(768), 0, 0, b, nbar, partials, rows, rf, out);
                };
                const float a2 = timeit(rf ? "partial + 72 KB rows + barrier + pool" : "partial + barrier + pool", 10, l_pool);
                const float c2 = timeit(rf ? "partial + 72 KB rows + barrier + pool" : "partial + barrier + pool", 110, l_pool);
                printf("    => %.2f us per round\n", (c2 - a2) / 100.f);
            }
        }
    }
    // an empty launch pair for scale
    {
        CK(hipEventRecord(e0, 0));
        for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(bar_only<0>, dim3(cus), dim3(768), 0, 0, b, 0, out);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("empty launches back to back: %.2f us each\n", ms * 10.f);
    }
    return 0;
}
