// Whole-path orchestration: xyz -> descriptors.  Replaces MODEL.forward(...) executed by sess.run with
// is_training=False (train.py:254, evaluate.py:250-251).  Launch sequence per micro-batch (one HIP stream):
//   sort -> knn+conv1 -> block x4 (x2 for EPC-Net-L) -> conv5+assign -> aggregate -> head      (EPC-Net)
//                                               -> conv5+maxpool -> fc head               (EPC-Net-L)
#include "common.h"

static inline size_t al(size_t v) { return (v + 255) / 256 * 256; }

static int micro_batch(const epc_cfg* c, int num_clouds) {
    int mb = c->micro_batch > 0 ? c->micro_batch : (c->arch == EPC_ARCH_EPC_NET ? 64 : 256);
    return num_clouds < mb ? num_clouds : mb;
}

// EPC-Net in EPC_PRECISION_FAST: fp16 rows / fragments between the stages; otherwise f32 tensors, split-bf16 kernels
static bool fast_path(const epc_cfg* c) { return c->arch == EPC_ARCH_EPC_NET && c->precision == EPC_PRECISION_FAST; }

struct WsLayout {
    size_t status, sorted, idx, cnt, kth, xa, xb, xa16, xb16, cat, feat, rnorm, assign, afrag, vlad, colss, apart, head, pooled, total;
};

static WsLayout ws_layout(const epc_cfg* c, int mb) {
    WsLayout w;
    const size_t M = (size_t)mb * c->num_points;
    size_t o = 0;
    auto take = [&](size_t bytes) {
        size_t at = o;
        o += al(bytes);
        return at;
    };
    w.status = take((size_t)mb * 4);   // per-cloud EPC_STATUS_* words: offset 0 (epc_net_last_status)
    w.sorted = take(M * 3 * 4);
    w.idx = take(M * EPC_KNN_CAP * 4);
    w.cnt = take(M * 4);
    w.kth = take(M * 4);
    // EPC-Net: the block chain's tensors are fp16 rows (block.hip); EPC-Net-L: f32 rows
    w.xa = w.xb = w.xa16 = w.xb16 = 0;
    const bool fast = fast_path(c);
    if (fast) {
        w.xa16 = take(M * 64 * 2);
        w.xb16 = take(M * 64 * 2);
        w.cat = take(M * 256 * 2);
    } else {
        w.xa = take(M * 64 * 4);
        w.xb = take(M * 64 * 4);
        w.cat = take(M * (c->arch == EPC_ARCH_EPC_NET ? 256 : 128) * 4);
    }
    w.feat = w.rnorm = w.assign = w.afrag = w.vlad = w.colss = w.apart = w.head = w.pooled = 0;
    if (c->arch == EPC_ARCH_EPC_NET) {
        w.feat = take(M * 1024 * (fast ? 2 : 3));   // fp16 / 3-byte fragments
        w.rnorm = take(M * 4);
        w.afrag = take(M * 64 * (fast ? 2 : 4));    // fp16 / bf16 hi + lo fragments
        w.vlad = take((size_t)mb * 65536 * 4);
        w.colss = take((size_t)mb * 32 * 64 * 4);
        w.apart = take(M / 32 * 64 * 4);
        w.head = take(epc_vlad_head_workspace_bytes(mb, c->groups));
    } else {
        w.pooled = take((size_t)mb * 1024 * 4);
    }
    w.total = o;
    return w;
}

extern "C" size_t epc_net_workspace_bytes(const epc_cfg* cfg, int num_clouds) {
    if (epc_net_packed_bytes(cfg) == 0 || num_clouds <= 0) return 0;
    return ws_layout(cfg, micro_batch(cfg, num_clouds)).total;
}

#define TRY(expr)                        \
    do {                                 \
        int rc__ = (expr);               \
        if (rc__ != EPC_OK) return rc__; \
    } while (0)

// ---- stage profile: HIP events recorded on the caller's stream at the stage boundaries of ONE pass -----------
struct epc_profile {
    hipEvent_t ev[EPC_NUM_STAGES + 1];
    int recorded[EPC_NUM_STAGES + 1];
};

extern "C" int epc_profile_create(epc_profile** prof) {
    EPC_CHECK_ARG(prof, "null pointer");
    epc_profile* p = new epc_profile();
    for (int i = 0; i <= EPC_NUM_STAGES; ++i) {
        p->recorded[i] = 0;
        hipError_t e = hipEventCreate(&p->ev[i]);
        if (e != hipSuccess) {
            epc_set_error("epc_profile_create: hipEventCreate: %s", hipGetErrorString(e));
            for (int k = 0; k < i; ++k) (void)hipEventDestroy(p->ev[k]);
            delete p;
            return EPC_EHIP;
        }
    }
    *prof = p;
    return EPC_OK;
}

extern "C" int epc_profile_destroy(epc_profile* prof) {
    if (!prof) return EPC_OK;
    for (int i = 0; i <= EPC_NUM_STAGES; ++i) (void)hipEventDestroy(prof->ev[i]);
    delete prof;
    return EPC_OK;
}

extern "C" int epc_profile_elapsed_ms(epc_profile* prof, float* stage_ms) {
    EPC_CHECK_ARG(prof && stage_ms, "null pointer");
    int last = -1;
    for (int i = 0; i <= EPC_NUM_STAGES; ++i) {
        if (i < EPC_NUM_STAGES) stage_ms[i] = 0.f;
        if (!prof->recorded[i]) continue;
        if (last >= 0) {
            float ms = 0.f;
            hipError_t e = hipEventElapsedTime(&ms, prof->ev[last], prof->ev[i]);
            if (e != hipSuccess) {
                epc_set_error("epc_profile_elapsed_ms: %s (stream not synchronised?)", hipGetErrorString(e));
                return EPC_EHIP;
            }
            stage_ms[last] = ms;  // boundary `last` opens stage `last`; the next recorded boundary closes it
        }
        last = i;
    }
    return EPC_OK;
}

static int mark(epc_profile* prof, int boundary, void* stream) {
    if (!prof) return EPC_OK;
    hipError_t e = hipEventRecord(prof->ev[boundary], (hipStream_t)stream);
    if (e != hipSuccess) {
        epc_set_error("epc_net_forward_profiled: hipEventRecord: %s", hipGetErrorString(e));
        return EPC_EHIP;
    }
    prof->recorded[boundary] = 1;
    return EPC_OK;
}

// One pass: xyz of `nc` <= micro_batch clouds -> descriptors, every launch on `stream`, every intermediate in `ws`.
static int forward_pass(const epc_cfg* cfg, const char* pk, const float* pc, int nc, float* o, char* ws,
                        const WsLayout& w, void* stream, epc_profile* prof) {
    const int n = cfg->num_points;
    const int nblocks = cfg->arch == EPC_ARCH_EPC_NET ? 4 : 2;
    const int ccat = 64 * nblocks;
    int32_t* idx = (int32_t*)(ws + w.idx);
    int32_t* cnt = (int32_t*)(ws + w.cnt);
    float* kth = (float*)(ws + w.kth);
    float* xs[2] = {(float*)(ws + w.xa), (float*)(ws + w.xb)};
    const bool f16 = fast_path(cfg);
    int32_t* status = (int32_t*)(ws + w.status);
    void* xs16[2] = {f16 ? (void*)(ws + w.xa16) : nullptr, f16 ? (void*)(ws + w.xb16) : nullptr};
    if (f16) xs[0] = xs[1] = nullptr;
    float* cat = (float*)(ws + w.cat);

    TRY(mark(prof, EPC_STAGE_SORT, stream));
    if (n <= 16384) {  // descriptors are permutation-invariant: run the whole pipeline on the Z-ordered cloud
        float* sorted = (float*)(ws + w.sorted);
        TRY(epc_sort_launch(pc, nc, n, sorted, nullptr, status, stream));   // (also zeroes the status words)
        pc = sorted;
    } else {
        hipError_t e = hipMemsetAsync(status, 0, (size_t)nc * sizeof(int32_t), (hipStream_t)stream);
        if (e != hipSuccess) {
            epc_set_error("epc_net_forward: hipMemsetAsync: %s", hipGetErrorString(e));
            return EPC_EHIP;
        }
    }
    TRY(mark(prof, EPC_STAGE_KNN, stream));
    // kNN graph + conv1 in one launch (the kNN workgroup already holds the cloud in LDS); a stage profile therefore
    // reports conv1 inside the kNN stage
#ifdef PIPE_NO_U16
    const int idx_u16 = 0;
#else
    const int idx_u16 = n <= 8192;   // 2-byte neighbour lists wherever the LDS kNN kernel runs
#endif
    TRY(epc_knn_topk_conv1(pc, nc, n, EPC_KNN_CAP, idx, idx_u16, cnt, kth, pk + epc_net_packed_offset(cfg, 0), xs[0], xs16[0],
                           status, stream));
    for (int b = 1; b <= nblocks; ++b) {
        TRY(mark(prof, EPC_STAGE_BLOCK1 + b - 1, stream));
        const int has_next = b < nblocks;
        TRY(epc_proxyconv_block_fwd(xs[(b - 1) & 1], xs16[(b - 1) & 1], pc, idx, idx_u16, cnt, kth, EPC_KNN_CAP,
                                    pk + epc_net_packed_offset(cfg, b), has_next, nc, n, cfg->knn,
                                    f16 ? nullptr : cat, f16 ? (void*)cat : nullptr, ccat, 64 * (b - 1), xs[b & 1],
                                    xs16[b & 1], status, stream));
    }
    if (cfg->arch == EPC_ARCH_EPC_NET) {
        float* feat = (float*)(ws + w.feat);
        float* rnorm = (float*)(ws + w.rnorm);
        float* vlad = (float*)(ws + w.vlad);
        float* colss = (float*)(ws + w.colss);
        float* apart = (float*)(ws + w.apart);
        TRY(mark(prof, EPC_STAGE_CONV5, stream));
        float* afrag = (float*)(ws + w.afrag);
        const char* head_pack = pk + epc_net_packed_offset(cfg, 6);   // starts with the cluster centres
        if (f16) {
            TRY(epc_conv5_assign_fwd(cat, 1, ccat, pk + epc_net_packed_offset(cfg, 5), nc * n, n, feat, rnorm, nullptr, afrag,
                                     apart, status, stream));
            TRY(mark(prof, EPC_STAGE_AGGREGATE, stream));
            TRY(epc_vlad_aggregate_fwd(feat, afrag, rnorm, apart, (const float*)head_pack, nc, n, vlad, colss, stream));
        } else {
            TRY(epc_conv5_assign_f32_fwd(cat, ccat, pk + epc_net_packed_offset(cfg, 5), nc * n, feat, rnorm, nullptr, afrag,
                                         apart, stream));
            TRY(mark(prof, EPC_STAGE_AGGREGATE, stream));
            TRY(epc_vlad_aggregate_f32_fwd(feat, afrag, rnorm, apart, (const float*)head_pack, nc, n, vlad, colss, stream));
        }
        TRY(mark(prof, EPC_STAGE_HEAD, stream));
        TRY(epc_vlad_head_fwd(vlad, colss, head_pack, cfg->groups, nc, o, status, ws + w.head, w.total - w.head, stream));
    } else {
        float* pooled = (float*)(ws + w.pooled);
        TRY(mark(prof, EPC_STAGE_CONV5, stream));
        TRY(epc_conv5_maxpool_fwd(cat, ccat, pk + epc_net_packed_offset(cfg, 5), nc, n, pooled, stream));
        TRY(mark(prof, EPC_STAGE_HEAD, stream));
        TRY(epc_fc_head_fwd(pooled, pk + epc_net_packed_offset(cfg, 6), nc, o, status, stream));
    }
    return mark(prof, EPC_NUM_STAGES, stream);
}

extern "C" int epc_net_last_status(const epc_cfg* cfg, const void* workspace, int num_clouds, int32_t* status_host,
                                   void* stream) {
    EPC_CHECK_ARG(epc_net_packed_bytes(cfg) != 0, "unsupported configuration");
    EPC_CHECK_ARG(workspace && status_host && num_clouds >= 0, "null pointer / bad shape");
    if (num_clouds == 0) return EPC_OK;
    const int mb = micro_batch(cfg, num_clouds);
    const int last = num_clouds % mb ? num_clouds % mb : mb;   // clouds of the last pass
    const WsLayout w = ws_layout(cfg, mb);
    hipError_t e = hipMemcpyAsync(status_host, (const char*)workspace + w.status, (size_t)last * sizeof(int32_t),
                                  hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) {
        epc_set_error("epc_net_last_status: %s", hipGetErrorString(e));
        return EPC_EHIP;
    }
    return EPC_OK;
}

extern "C" int epc_net_forward(const epc_cfg* cfg, const void* packed, const float* xyz, int num_clouds, float* out,
                               void* workspace, size_t workspace_bytes, void* stream) {
    return epc_net_forward_profiled(cfg, packed, xyz, num_clouds, out, workspace, workspace_bytes, stream, nullptr);
}

extern "C" int epc_net_forward_profiled(const epc_cfg* cfg, const void* packed, const float* xyz, int num_clouds,
                                        float* out, void* workspace, size_t workspace_bytes, void* stream,
                                        epc_profile* prof) {
    EPC_CHECK_ARG(epc_net_packed_bytes(cfg) != 0, "unsupported configuration");
    EPC_CHECK_ARG(packed && xyz && out, "null pointer");
    EPC_CHECK_ARG(num_clouds >= 0, "bad shape");
    if (num_clouds == 0) return EPC_OK;
    const int mb = micro_batch(cfg, num_clouds);
    EPC_CHECK_ARG(!prof || num_clouds <= mb, "a stage profile covers one pass: num_clouds must be <= micro_batch");
    if (prof)
        for (int i = 0; i <= EPC_NUM_STAGES; ++i) prof->recorded[i] = 0;
    const WsLayout w = ws_layout(cfg, mb);
    if (!workspace || workspace_bytes < w.total) {
        epc_set_error("epc_net_forward: workspace too small (%zu < %zu)", workspace_bytes, w.total);
        return EPC_ENOMEM;
    }
    const int n = cfg->num_points;
    for (int c0 = 0; c0 < num_clouds; c0 += mb) {
        const int nc = (num_clouds - c0) < mb ? (num_clouds - c0) : mb;
        TRY(forward_pass(cfg, (const char*)packed, xyz + (size_t)c0 * n * 3, nc, out + (size_t)c0 * cfg->output_dim,
                         (char*)workspace, w, stream, prof));
    }
    return EPC_OK;
}

// Successive passes dealt round-robin over `stream` and the caller's auxiliary streams, each lane with its own
// workspace slice.  The stages of one pass are bound by different units (kNN: VALU issue, conv5: MFMA, aggregate:
// HBM, head/sort: latency of tiny grids), so two passes in flight fill each other's idle units: 0.93 -> 0.83 ms per
// 64-cloud pass on MI355X (scripts/time_two_streams.py).  Stream-ordered like epc_net_forward: the auxiliary
// streams wait for `stream` (inputs ready) and `stream` waits for them before the call's work counts as done.
extern "C" int epc_net_forward_overlapped(const epc_cfg* cfg, const void* packed, const float* xyz, int num_clouds,
                                          float* out, void* workspace, size_t workspace_bytes, void* stream,
                                          void* const* aux_streams, int num_aux) {
    EPC_CHECK_ARG(epc_net_packed_bytes(cfg) != 0, "unsupported configuration");
    EPC_CHECK_ARG(packed && xyz && out, "null pointer");
    EPC_CHECK_ARG(num_clouds >= 0 && num_aux >= 0 && num_aux <= 7 && (num_aux == 0 || aux_streams), "bad shape");
    if (num_clouds == 0) return EPC_OK;
    const int mb = micro_batch(cfg, num_clouds);
    const int passes = (num_clouds + mb - 1) / mb;
    const int lanes = passes < 1 + num_aux ? passes : 1 + num_aux;
    const WsLayout w = ws_layout(cfg, mb);
    if (!workspace || workspace_bytes < w.total * lanes) {
        epc_set_error("epc_net_forward_overlapped: workspace too small (%zu < %zu = %d lanes x %zu)", workspace_bytes,
                      w.total * lanes, lanes, w.total);
        return EPC_ENOMEM;
    }
    hipEvent_t ev = nullptr;
    auto hip_ok = [&](hipError_t e, const char* what) {
        if (e == hipSuccess) return true;
        epc_set_error("epc_net_forward_overlapped: %s: %s", what, hipGetErrorString(e));
        if (ev) (void)hipEventDestroy(ev);
        return false;
    };
    if (lanes > 1) {
        if (!hip_ok(hipEventCreateWithFlags(&ev, hipEventDisableTiming), "hipEventCreate")) return EPC_EHIP;
        if (!hip_ok(hipEventRecord(ev, (hipStream_t)stream), "hipEventRecord")) return EPC_EHIP;
        for (int l = 1; l < lanes; ++l)
            if (!hip_ok(hipStreamWaitEvent((hipStream_t)aux_streams[l - 1], ev, 0), "hipStreamWaitEvent")) return EPC_EHIP;
    }
    const int n = cfg->num_points;
    int rc = EPC_OK;
    for (int p = 0; p < passes && rc == EPC_OK; ++p) {
        const int c0 = p * mb, nc = (num_clouds - c0) < mb ? (num_clouds - c0) : mb, l = p % lanes;
        rc = forward_pass(cfg, (const char*)packed, xyz + (size_t)c0 * n * 3, nc, out + (size_t)c0 * cfg->output_dim,
                          (char*)workspace + (size_t)l * w.total, w, l == 0 ? stream : aux_streams[l - 1], nullptr);
    }
    // join even after a failed launch: whatever was enqueued on the auxiliary streams stays ordered before `stream`
    for (int l = 1; l < lanes; ++l) {
        if (!hip_ok(hipEventRecord(ev, (hipStream_t)aux_streams[l - 1]), "hipEventRecord")) return EPC_EHIP;
        if (!hip_ok(hipStreamWaitEvent((hipStream_t)stream, ev, 0), "hipStreamWaitEvent")) return EPC_EHIP;
    }
    if (ev) (void)hipEventDestroy(ev);
    return rc;
}
