// Descriptor retrieval -- replaces sklearn KDTree(database).query(q, k=25) of evaluate.py:463,481.
//
// Two forms with identical results (exact k nearest rows by Euclidean distance, ties -> lower database index):
//   epc_pairwise_topk      no workspace: one wave per query, distances of the whole database in LDS (<= ~40 k rows);
//   epc_pairwise_topk_ws   the pairwise matrix as a tiled f32-MFMA GEMM (d2 = |q|^2 + |d|^2 - 2 q.d into the caller's
//                          workspace, any database size), per query the k + 8 smallest of its row, those candidates re-ranked
//                          by the EXACT sum_c (q_c - d_c)^2 the first form uses, and a proof per query that no other row can
//                          belong to the answer (the k-th exact distance lies below the last candidate's GEMM value by more
//                          than the GEMM's rounding bound); the rare query without proof (many near-ties) is redone exactly.
//
// First form: one wave per query.  Exact Euclidean distances sum_c (q_c - d_c)^2 in f32 (no ||q||^2+||d||^2-2q.d cancellation),
// kept in LDS; then k rounds of wave-wide arg-min (ties -> lower database index), so the result is the sorted
// k-nearest list.  Database sizes on this path are 10^2..10^4 rows (Oxford runs hold ~400 submaps each).
#include "common.h"

__global__ __launch_bounds__(64) void pairwise_topk_kernel(const float* __restrict__ db, int num_db,
                                                           const float* __restrict__ queries, int dim, int k,
                                                           int32_t* __restrict__ idx, float* __restrict__ dist) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* qv = lds;          // dim
    float* d2 = lds + dim;    // num_db
    const int lane = threadIdx.x;
    const int qi = blockIdx.x;
    for (int c = lane; c < dim; c += 64) qv[c] = queries[(size_t)qi * dim + c];
    __syncthreads();
    for (int d = lane; d < num_db; d += 64) {
        const float* row = db + (size_t)d * dim;
        float acc = 0.f;
        for (int c = 0; c < dim; c += 4) {
            const float4 v = *reinterpret_cast<const float4*>(row + c);
            const float e0 = qv[c] - v.x, e1 = qv[c + 1] - v.y, e2 = qv[c + 2] - v.z, e3 = qv[c + 3] - v.w;
            acc += e0 * e0;
            acc += e1 * e1;
            acc += e2 * e2;
            acc += e3 * e3;
        }
        d2[d] = acc <= 3.0e38f ? acc : INFINITY;   // NaN / overflow: never selected
    }
    __syncthreads();
    for (int r = 0; r < k; ++r) {
        float best = INFINITY;
        int bi = 0x7fffffff;
        for (int d = lane; d < num_db; d += 64) {
            const float v = d2[d];
            if (v < best) {  // ascending d within a lane: strict < keeps the lower index on ties
                best = v;
                bi = d;
            }
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const float ov = __shfl_xor(best, off);
            const int oi = __shfl_xor(bi, off);
            if (ov < best || (ov == best && oi < bi)) {
                best = ov;
                bi = oi;
            }
        }
        if (lane == 0) {
            // fewer than k rows at a finite distance (NaN / Inf query or database rows -- the descriptors of flagged clouds):
            // no such neighbour, index -1 at distance +Inf
            idx[(size_t)qi * k + r] = bi < num_db ? bi : -1;
            dist[(size_t)qi * k + r] = sqrtf(best);
            if (bi < num_db) d2[bi] = INFINITY;
        }
        __syncthreads();
    }
}

extern "C" int epc_pairwise_topk(const float* database, int num_db, const float* queries, int num_q, int dim,
                                 int k, int32_t* idx, float* dist, void* stream) {
    EPC_CHECK_ARG(database && queries && idx && dist, "null pointer");
    EPC_CHECK_ARG(dim > 0 && dim % 4 == 0, "descriptor dim must be a multiple of 4");
    EPC_CHECK_ARG(k > 0 && k <= num_db, "need 0 < k <= num_db");
    EPC_CHECK_ARG(num_q >= 0, "bad shape");
    const size_t lds_bytes = ((size_t)dim + num_db) * sizeof(float);
    EPC_CHECK_ARG(lds_bytes <= 160 * 1024, "database shard too large for one LDS-resident pass (shard it)");
    if (num_q == 0) return EPC_OK;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pairwise_topk_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) {
        epc_set_error("epc_pairwise_topk: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return EPC_EHIP;
    }
    hipLaunchKernelGGL(pairwise_topk_kernel, dim3(num_q), dim3(64), lds_bytes, (hipStream_t)stream, database,
                       num_db, queries, dim, k, idx, dist);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}


// ---------------------------------------------------------------------------------------------------------------------
// Workspace form.
// ---------------------------------------------------------------------------------------------------------------------
#define RT_EXTRA 8          // candidates beyond k whose GEMM distances bracket the answer
#define RT_MAXC 64          // k + RT_EXTRA <= one lane per candidate

__global__ __launch_bounds__(256) void rownorm2_kernel(const float* __restrict__ x, int rows, int dim,
                                                       float* __restrict__ out) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= rows) return;
    float s = 0.f;
    for (int c = lane * 4; c < dim; c += 256) {
        const float4 v = *reinterpret_cast<const float4*>(x + (size_t)r * dim + c);
        s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[r] = s;
}

// S[q][d] = max(|q|^2 + |d|^2 - 2 q.d, 0): 128 x 128 tile per workgroup, 64 x 64 per wave (four 32x32 f32-MFMA accumulators).
// Both operands are row-major (rows, dim) with the row on the lane -- exactly the A[i][k] / B[k][j] lane layouts of
// v_mfma_f32_32x32x2_f32 when a lane reads a float4 along K (k = 8b + 4 (lane >> 5) + c for component c of block b), so the
// fragments come straight from global memory (L2): no LDS, 4 float4 loads per 16 MFMAs.
__global__ __launch_bounds__(256) void pairdist_gemm_kernel(const float* __restrict__ Q, const float* __restrict__ D,
                                                            const float* __restrict__ qn, const float* __restrict__ dn,
                                                            int nq, int nd, int dim, float* __restrict__ S) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, h = lane >> 5;
    const int q0 = blockIdx.y * 128 + (wave >> 1) * 64, d0 = blockIdx.x * 128 + (wave & 1) * 64;
    if (q0 >= nq || d0 >= nd) return;
    const float* a0 = Q + (size_t)min(q0 + j, nq - 1) * dim + 4 * h;
    const float* a1 = Q + (size_t)min(q0 + 32 + j, nq - 1) * dim + 4 * h;
    const float* b0 = D + (size_t)min(d0 + j, nd - 1) * dim + 4 * h;
    const float* b1 = D + (size_t)min(d0 + 32 + j, nd - 1) * dim + 4 * h;
    f32x16 acc[2][2];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][0][r] = acc[0][1][r] = acc[1][0][r] = acc[1][1][r] = 0.f;
    for (int k = 0; k < dim; k += 8) {
        const float4 x0 = *reinterpret_cast<const float4*>(a0 + k), x1 = *reinterpret_cast<const float4*>(a1 + k);
        const float4 y0 = *reinterpret_cast<const float4*>(b0 + k), y1 = *reinterpret_cast<const float4*>(b1 + k);
        const float xa[4] = {x0.x, x0.y, x0.z, x0.w}, xb[4] = {x1.x, x1.y, x1.z, x1.w};
        const float ya[4] = {y0.x, y0.y, y0.z, y0.w}, yb[4] = {y1.x, y1.y, y1.z, y1.w};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc[0][0] = mfma32(xa[c], ya[c], acc[0][0]);
            acc[0][1] = mfma32(xa[c], yb[c], acc[0][1]);
            acc[1][0] = mfma32(xb[c], ya[c], acc[1][0]);
            acc[1][1] = mfma32(xb[c], yb[c], acc[1][1]);
        }
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int d = d0 + 32 * tj + j;
            if (d >= nd) continue;
            const float dnv = dn[d];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int q = q0 + 32 * ti + mfma_row(r, h);
                if (q < nq) {
                    // rounding can leave a tiny negative value for coincident rows: 0; a NaN / Inf descriptor (a flagged cloud):
                    // +Inf, never selected -- as in the exact form, where a NaN distance never wins a comparison
                    const float v = (qn[q] + dnv) - 2.0f * acc[ti][tj][r];
                    S[(size_t)q * nd + d] = v >= 0.f ? (v <= 3.0e38f ? v : INFINITY) : (v < 0.f ? 0.f : INFINITY);
                }
            }
        }
}

// exact squared distance, the arithmetic of pairwise_topk_kernel (sequential over c, one rounding per operation)
__device__ __forceinline__ float exact_d2(const float* __restrict__ q, const float* __restrict__ row, int dim) {
    float acc = 0.f;
    for (int c = 0; c < dim; c += 4) {
        const float4 v = *reinterpret_cast<const float4*>(row + c);
        const float4 u = *reinterpret_cast<const float4*>(q + c);
        const float e0 = u.x - v.x, e1 = u.y - v.y, e2 = u.z - v.z, e3 = u.w - v.w;
        acc += e0 * e0;
        acc += e1 * e1;
        acc += e2 * e2;
        acc += e3 * e3;
    }
    return acc;
}

// lexicographic (value, index) minimum over the wave
__device__ __forceinline__ void wave_argmin(float& best, int& bi) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const float ov = __shfl_xor(best, off);
        const int oi = __shfl_xor(bi, off);
        if (ov < best || (ov == best && oi < bi)) {
            best = ov;
            bi = oi;
        }
    }
}

// One wave per query: the kc = min(k + RT_EXTRA, nd) smallest entries of its row of S in (value, index) order, their exact
// distances, the k best of those; flag[q] = 1 when the bracket does not prove the answer.
__global__ __launch_bounds__(64) void select_rerank_kernel(const float* __restrict__ S, const float* __restrict__ Q,
                                                           const float* __restrict__ D, const float* __restrict__ qn,
                                                           const float* __restrict__ dn_max, int nd, int dim, int k,
                                                           int rows_in_lds, int32_t* __restrict__ idx,
                                                           float* __restrict__ dist, int32_t* __restrict__ flag) {
    extern __shared__ __attribute__((aligned(16))) float srow[];
    const int lane = threadIdx.x, qi = blockIdx.x;
    const float* grow = S + (size_t)qi * nd;
    const float* row = grow;
    if (rows_in_lds) {
        for (int d = lane; d < nd; d += 64) srow[d] = grow[d];
        __syncthreads();
        row = srow;
    }
    const int kc = min(k + RT_EXTRA, nd);
    float lv = -1.f;      // last selected (value, index): values are >= 0
    int li = -1;
    int my_cand = -1;
    float my_approx = 0.f;
    // ---- fast selection: a bound, one filtering scan, a rank over <= 64 survivors (instead of kc full scans) ----------
    // (1) every lane keeps the 4 smallest VALUES of its stride of the row (a 4-slot median network: 4 instructions per entry);
    // (2) U = the kc-th smallest of those 256 values -- the kc-th smallest of a SUBSET of the row, hence an upper bound of the
    //     row's kc-th smallest value;  (3) one more scan appends every finite (value, index) with value <= U to an LDS list:
    //     it holds the row's kc smallest entries, usually kc of them or a few more;  (4) with at most 64 survivors a rank by
    //     (value, index) puts the kc smallest on lanes 0 .. kc-1 in exactly the order the kc scans below would have found
    //     them.  More than 64 survivors (masses of ties) or fewer than kc finite entries: the scans below take over.
    bool fast_ok = false;
    {
        float* sv = srow + (rows_in_lds ? ((nd + 3) & ~3) : 0);          // [64] survivor values, [64] indices, [64] + [64] ranked
        int* si = reinterpret_cast<int*>(sv + 64);
        float* cv = sv + 128;
        int* ci = reinterpret_cast<int*>(sv + 192);
        float t0 = INFINITY, t1 = INFINITY, t2 = INFINITY, t3 = INFINITY;
        for (int d = lane; d < nd; d += 64) {
            const float v = row[d];
            t3 = __builtin_amdgcn_fmed3f(t2, t3, v);
            t2 = __builtin_amdgcn_fmed3f(t1, t2, v);
            t1 = __builtin_amdgcn_fmed3f(t0, t1, v);
            t0 = fminf(t0, v);
        }
        float U = INFINITY;
        for (int r = 0; r < kc; ++r) {
            float m = t0;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) m = fminf(m, __shfl_xor(m, off));
            U = m;
            if (!(m < INFINITY)) break;                                   // fewer than kc finite values among the candidates
            const unsigned long long holders = __builtin_amdgcn_ballot_w64(t0 == m);
            if (lane == __builtin_ctzll(holders)) t0 = t1, t1 = t2, t2 = t3, t3 = INFINITY;   // pop ONE holder per round
        }
        int total = 0;
        if (U < INFINITY) {
            for (int d0 = 0; d0 < nd; d0 += 64) {
                const int d = d0 + lane;
                const float v = d < nd ? row[d] : INFINITY;
                const bool hit = v <= U && v < INFINITY;
                const unsigned long long mask = __builtin_amdgcn_ballot_w64(hit);
                if (mask) {
                    const int pos = total + __builtin_popcountll(mask & ((1ull << lane) - 1ull));
                    if (hit && pos < 64) sv[pos] = v, si[pos] = d;
                    total += __builtin_popcountll(mask);
                }
            }
        }
        if (total >= kc && total <= 64) {
            __syncthreads();
            const float v = lane < total ? sv[lane] : INFINITY;
            const int d = lane < total ? si[lane] : 0x7fffffff;
            int rank = 0;
            for (int c = 0; c < total; ++c) {
                const float ov = sv[c];
                const int od = si[c];
                if (ov < v || (ov == v && od < d)) ++rank;
            }
            if (lane < total) cv[rank] = v, ci[rank] = d;
            __syncthreads();
            if (lane < kc) my_cand = ci[lane], my_approx = cv[lane];
            fast_ok = true;
        }
    }
    for (int r = 0; r < (fast_ok ? 0 : kc); ++r) {
        float best = INFINITY;
        int bi = 0x7fffffff;
        for (int d = lane; d < nd; d += 64) {
            const float v = row[d];
            const bool after = v > lv || (v == lv && d > li);
            if (after && v < best) {   // ascending d within a lane: strict < keeps the lower index on ties
                best = v;
                bi = d;
            }
        }
        wave_argmin(best, bi);
        lv = best;
        li = bi;
        if (lane == r) {
            my_cand = bi < nd ? bi : 0x7fffff00 + lane;   // fewer than kc finite rows: a unique invalid index that sorts last
            my_approx = best;
        }
    }
    // exact distances of the candidates, one per lane
    const float* qrow = Q + (size_t)qi * dim;
    float e = INFINITY;
    if (lane < kc && my_cand < nd) e = exact_d2(qrow, D + (size_t)my_cand * dim, dim);
    if (!(e <= 3.0e38f)) e = INFINITY;   // NaN distance (NaN query): sorts last, like the comparisons of the exact form
    // rank by (exact, index) among the candidates
    int rank = 0;
    for (int c = 0; c < kc; ++c) {
        const float oe = __shfl(e, c);
        const int oi = __shfl(my_cand, c);
        if (lane < kc && (oe < e || (oe == e && oi < my_cand))) ++rank;
    }
    if (lane < kc && rank < k) {
        idx[(size_t)qi * k + rank] = my_cand < nd ? my_cand : -1;
        dist[(size_t)qi * k + rank] = sqrtf(e);
    }
    // Proof that no row r outside the candidates belongs to the answer.  Its GEMM value g_r is >= the last candidate's (m).
    // With u = 2^-24 and c = (dim + 3) u (first order; the constant below carries slack):
    //   |g_r - d2_r| <= c (sqrt(qn) + sqrt(dn_r))^2 =: E_r   (norms: <= dim u each; dot product: dim sequential f32
    //                   accumulations, |err| <= dim u sum|q_c d_c| <= dim u sqrt(qn dn_r); the two final additions: 2 u),
    //   exact form:     e_r >= d2_r (1 - c)                  (one rounding per difference, square and addition).
    // E_r needs a bound that does not know r.  (a) dn_r <= dn_max, the largest finite squared norm of the database:
    // E_r <= c (sqrt(qn) + sqrt(dn_max))^2.  (b) norm-free: if sqrt(dn_r) <= 3 sqrt(qn), E_r <= 16 c qn; otherwise
    // sqrt(qn) + sqrt(dn_r) < 2 (sqrt(dn_r) - sqrt(qn)) <= 2 sqrt(d2_r), so E_r <= 4 c d2_r and d2_r >= m (1 - 4c):
    // E_r <= c (16 qn + 4 m) either way.  Both hold, so the smaller applies:
    //   e_r >= m - eps,   eps = c (min((sqrt(qn) + sqrt(dn_max))^2, 16 qn + 4 m) + m).
    // The answer stands when the k-th exact distance is STRICTLY below m - eps (an equal distance at a lower index would
    // win the tie).  A row whose GEMM value is +Inf (NaN / Inf descriptor) has no finite exact distance either and is
    // never part of an answer: m = +Inf means every finite row is a candidate, nothing to prove.
    float kth_exact = (lane < kc && rank == k - 1) ? e : -INFINITY;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) kth_exact = fmaxf(kth_exact, __shfl_xor(kth_exact, off));
    const float m = __shfl(my_approx, kc - 1);
    if (lane == 0) {
        int f = 0;
        if (kc < nd && m < INFINITY) {
            const float c = 1.05f * (float)(dim + 8) * 5.9604645e-8f;
            const float qq = qn[qi], sq = sqrtf(qq) + sqrtf(*dn_max);
            const float eps = c * (fminf(sq * sq, 16.0f * qq + 4.0f * m) + m) + 1e-37f;
            f = !(kth_exact < m - eps);
        }
        flag[qi] = f;
    }
}

// largest FINITE squared norm of the database rows (one workgroup; nd is 10^2 .. 10^5)
__global__ __launch_bounds__(1024) void finite_max_kernel(const float* __restrict__ x, int n, float* __restrict__ out) {
    __shared__ float part[16];
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) {
        const float v = x[i];
        if (v <= 3.0e38f) m = fmaxf(m, v);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = fmaxf(m, part[w]);
        *out = m;
    }
}

// flagged queries only: exact distances of the whole database into the query's row of S, then k rounds of arg-min
__global__ __launch_bounds__(64) void exact_fallback_kernel(float* __restrict__ S, const float* __restrict__ Q,
                                                            const float* __restrict__ D, int nd, int dim, int k,
                                                            const int32_t* __restrict__ flag, int32_t* __restrict__ idx,
                                                            float* __restrict__ dist) {
    const int lane = threadIdx.x, qi = blockIdx.x;
    if (!flag[qi]) return;
    float* row = S + (size_t)qi * nd;
    const float* qrow = Q + (size_t)qi * dim;
    for (int d = lane; d < nd; d += 64) {
        const float v = exact_d2(qrow, D + (size_t)d * dim, dim);
        row[d] = v <= 3.0e38f ? v : INFINITY;      // NaN / overflow: never selected (as in select_rerank_kernel)
    }
    __syncthreads();
    for (int r = 0; r < k; ++r) {
        float best = INFINITY;
        int bi = 0x7fffffff;
        for (int d = lane; d < nd; d += 64) {
            const float v = row[d];
            if (v < best) {
                best = v;
                bi = d;
            }
        }
        wave_argmin(best, bi);
        if (lane == 0) {
            // fewer than k rows at a finite distance: no such neighbour (-1, +Inf) -- bi is NOT a row then, never store through it
            idx[(size_t)qi * k + r] = bi < nd ? bi : -1;
            dist[(size_t)qi * k + r] = sqrtf(best);
            if (bi < nd) row[bi] = INFINITY;
        }
        __syncthreads();
    }
}

static inline size_t rt_align(size_t v) { return (v + 255) / 256 * 256; }
// queries per pass: the pairwise matrix of one pass stays below 1 GiB
static int rt_chunk(int num_db, int num_q) {
    long per = (1L << 28) / (num_db > 0 ? num_db : 1);
    if (per < 128) per = 128;
    return (int)(per < num_q ? per : num_q);
}

extern "C" size_t epc_pairwise_topk_workspace_bytes(int num_db, int num_q) {
    if (num_db <= 0 || num_q <= 0) return 0;
    const int ch = rt_chunk(num_db, num_q);
    return rt_align((size_t)ch * num_db * 4) + rt_align((size_t)num_db * 4) + rt_align((size_t)num_q * 4) + rt_align((size_t)num_q * 4) +
           rt_align(4);
}

extern "C" int epc_pairwise_topk_ws(const float* database, int num_db, const float* queries, int num_q, int dim, int k,
                                    int32_t* idx, float* dist, void* workspace, size_t workspace_bytes, void* stream) {
    EPC_CHECK_ARG(database && queries && idx && dist && workspace, "null pointer");
    EPC_CHECK_ARG(dim > 0 && dim % 8 == 0, "descriptor dim must be a multiple of 8");
    EPC_CHECK_ARG(k > 0 && k <= num_db && k + RT_EXTRA <= RT_MAXC, "need 0 < k <= min(num_db, 56)");
    EPC_CHECK_ARG(num_q >= 0, "bad shape");
    if (num_q == 0) return EPC_OK;
    if (workspace_bytes < epc_pairwise_topk_workspace_bytes(num_db, num_q)) {
        epc_set_error("epc_pairwise_topk_ws: workspace too small");
        return EPC_ENOMEM;
    }
    const int ch = rt_chunk(num_db, num_q);
    char* w = (char*)workspace;
    float* S = (float*)w;
    w += rt_align((size_t)ch * num_db * 4);
    float* dn = (float*)w;
    w += rt_align((size_t)num_db * 4);
    float* qn = (float*)w;
    w += rt_align((size_t)num_q * 4);
    int32_t* flag = (int32_t*)w;
    w += rt_align((size_t)num_q * 4);
    float* dn_max = (float*)w;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(rownorm2_kernel, dim3((num_db + 3) / 4), dim3(256), 0, st, database, num_db, dim, dn);
    hipLaunchKernelGGL(rownorm2_kernel, dim3((num_q + 3) / 4), dim3(256), 0, st, queries, num_q, dim, qn);
    hipLaunchKernelGGL(finite_max_kernel, dim3(1), dim3(1024), 0, st, dn, num_db, dn_max);
    EPC_CHECK_LAUNCH();
    const int in_lds = (size_t)num_db * 4 <= 149 * 1024;
    const size_t lds_bytes = (in_lds ? (((size_t)num_db + 3) & ~(size_t)3) * 4 : 0) + 4 * 64 * 4;   // the row (if it fits) + the selection's 4 x 64 words
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(select_rerank_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(150 * 1024));
    if (e != hipSuccess) {
        epc_set_error("epc_pairwise_topk_ws: hipFuncSetAttribute: %s", hipGetErrorString(e));
        return EPC_EHIP;
    }
    for (int q0 = 0; q0 < num_q; q0 += ch) {
        const int nq = (num_q - q0) < ch ? (num_q - q0) : ch;
        const float* Qc = queries + (size_t)q0 * dim;
        hipLaunchKernelGGL(pairdist_gemm_kernel, dim3((num_db + 127) / 128, (nq + 127) / 128), dim3(256), 0, st, Qc, database,
                           qn + q0, dn, nq, num_db, dim, S);
        hipLaunchKernelGGL(select_rerank_kernel, dim3(nq), dim3(64), lds_bytes, st, S, Qc, database, qn + q0, dn_max, num_db, dim, k,
                           in_lds, idx + (size_t)q0 * k, dist + (size_t)q0 * k, flag + q0);
        hipLaunchKernelGGL(exact_fallback_kernel, dim3(nq), dim3(64), 0, st, S, Qc, database, num_db, dim, k, flag + q0,
                           idx + (size_t)q0 * k, dist + (size_t)q0 * k);
        EPC_CHECK_LAUNCH();
    }
    return EPC_OK;
}
