/*
 * epcnet_forward.c -- the C ABI of libepcnet_hip.so used from plain C (no Python, no torch): what a non-Python caller
 * (or a TensorFlow custom op wrapping the whole `forward`) does.
 *
 *   1. build the variable table of EPC-Net (names relative to "query_triplets/", shapes of models/epc-net.py:62-149 and
 *      loupe.py:219-231) with seeded values, upload the tensors;
 *   2. epc_net_pack_weights   once per weight load;
 *   3. epc_net_forward        (num_clouds, 4096, 3) -> (num_clouds, 256), timed with HIP events;
 *   4. check: finite, unit L2 norm (models/epc-net.py:153), identical descriptors for a repeated cloud.
 *
 * Build:  make -C examples          Run:  examples/epcnet_forward [num_clouds]
 */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "epcnet.h"

#define CHECK_HIP(x)                                                                   \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "%s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
            return 2;                                                                  \
        }                                                                              \
    } while (0)
#define CHECK_EPC(x)                                                                      \
    do {                                                                                  \
        int rc_ = (x);                                                                    \
        if (rc_ != EPC_OK) {                                                              \
            fprintf(stderr, "%s:%d: status %d: %s\n", __FILE__, __LINE__, rc_, epc_last_error()); \
            return 3;                                                                     \
        }                                                                                 \
    } while (0)

#define MAX_VARS 256
static char g_names[MAX_VARS][160];
static const char* g_name_ptrs[MAX_VARS];
static float* g_dev[MAX_VARS];
static int g_nvars = 0;

static unsigned int g_seed = 12345u;
static float frand(void) { /* uniform in [0, 1) */
    g_seed = g_seed * 1664525u + 1013904223u;
    return (float)(g_seed >> 8) * (1.0f / 16777216.0f);
}

/* one variable: uniform(lo, hi) values, uploaded to the device */
static int add_var(const char* name, size_t count, float lo, float hi) {
    float* host = (float*)malloc(count * sizeof(float));
    size_t i;
    if (!host || g_nvars >= MAX_VARS) return 1;
    for (i = 0; i < count; ++i) host[i] = lo + (hi - lo) * frand();
    snprintf(g_names[g_nvars], sizeof(g_names[0]), "%s", name);
    g_name_ptrs[g_nvars] = g_names[g_nvars];
    if (hipMalloc((void**)&g_dev[g_nvars], count * sizeof(float)) != hipSuccess) return 1;
    if (hipMemcpy(g_dev[g_nvars], host, count * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return 1;
    free(host);
    ++g_nvars;
    return 0;
}

/* tf_util.conv1d variables (utils/tf_util.py:52-107, 454-491): weights [1,cin,cout], biases, bn/beta, bn/gamma and the
 * two moving-average shadows whose names embed the outer scope a second time */
static int add_conv(const char* scope, int cin, int cout) {
    char n[160];
    const float a = sqrtf(6.0f / (float)(cin + cout)); /* xavier */
    int bad = 0;
    snprintf(n, sizeof n, "%s/weights", scope);
    bad |= add_var(n, (size_t)cin * cout, -a, a);
    snprintf(n, sizeof n, "%s/biases", scope);
    bad |= add_var(n, cout, -0.05f, 0.05f);
    snprintf(n, sizeof n, "%s/bn/beta", scope);
    bad |= add_var(n, cout, -0.1f, 0.1f);
    snprintf(n, sizeof n, "%s/bn/gamma", scope);
    bad |= add_var(n, cout, 0.5f, 1.5f);
    snprintf(n, sizeof n, "%s/bn/query_triplets/%s/bn/moments/Squeeze/ExponentialMovingAverage", scope, scope);
    bad |= add_var(n, cout, -0.1f, 0.1f);
    snprintf(n, sizeof n, "%s/bn/query_triplets/%s/bn/moments/Squeeze_1/ExponentialMovingAverage", scope, scope);
    bad |= add_var(n, cout, 0.5f, 1.5f);
    return bad;
}

static int add_slim_bn(const char* scope, int n_ch) {
    char n[160];
    int bad = 0;
    snprintf(n, sizeof n, "%s/beta", scope);
    bad |= add_var(n, n_ch, -0.1f, 0.1f);
    snprintf(n, sizeof n, "%s/gamma", scope);
    bad |= add_var(n, n_ch, 0.5f, 1.5f);
    snprintf(n, sizeof n, "%s/moving_mean", scope);
    bad |= add_var(n, n_ch, -0.1f, 0.1f);
    snprintf(n, sizeof n, "%s/moving_variance", scope);
    bad |= add_var(n, n_ch, 0.5f, 1.5f);
    return bad;
}

int main(int argc, char** argv) {
    const int num_clouds = argc > 1 ? atoi(argv[1]) : 8;
    const int N = 4096;
    epc_cfg cfg;
    char scope[64];
    int b, i, bad = 0, iters = 20;
    size_t packed_bytes, ws_bytes, k;
    void *packed, *ws;
    float *h_xyz, *d_xyz, *d_out, *h_out, ms = 0.f, worst = 0.f, same = 0.f;
    hipStream_t stream;
    hipEvent_t e0, e1;

    memset(&cfg, 0, sizeof cfg);
    cfg.arch = EPC_ARCH_EPC_NET;
    cfg.num_points = N;
    cfg.input_dim = 3;
    cfg.knn = 20;
    cfg.cluster_size = 64;
    cfg.output_dim = 256;
    cfg.groups = 4;
    cfg.precision = argc > 2 && strcmp(argv[2], "fast") == 0 ? EPC_PRECISION_FAST : EPC_PRECISION_F32;
    if (num_clouds < 2) {
        fprintf(stderr, "need at least 2 clouds\n");
        return 1;
    }
    CHECK_HIP(hipStreamCreate(&stream));

    /* 1. variables (names relative to the outer scope "query_triplets/", train.py:251) */
    bad |= add_conv("fastdgcnn/conv1", 3, 64);
    for (b = 1; b <= 4; ++b) {
        if (b > 1) {
            snprintf(scope, sizeof scope, "fastdgcnn/conv%d", b);
            bad |= add_conv(scope, 64, 64);
        }
        snprintf(scope, sizeof scope, "fastdgcnn/conv%d_a", b);
        bad |= add_conv(scope, 64, 64);
        snprintf(scope, sizeof scope, "fastdgcnn/conv%d_b", b);
        bad |= add_conv(scope, 64, 64);
    }
    bad |= add_conv("fastdgcnn/conv5", 256, 1024);
    bad |= add_var("VLAD/cluster_weights", 1024 * 64, -0.054f, 0.054f);   /* ~N(0, 1/sqrt(1024)), loupe.py:252 */
    bad |= add_var("VLAD/cluster_weights2", 1024 * 64, -0.054f, 0.054f);  /* loupe.py:281 */
    bad |= add_var("VLAD/hidden1_weights", 16384 * 256, -0.2f, 0.2f);     /* loupe.py:316 */
    bad |= add_var("VLAD/gating_weights", 256 * 256, -0.1f, 0.1f);        /* loupe.py:78 */
    bad |= add_slim_bn("VLAD/cluster_bn", 64);
    bad |= add_slim_bn("VLAD/bn", 256);
    bad |= add_slim_bn("VLAD/gating_bn", 256);
    if (bad) {
        fprintf(stderr, "building the variable table failed\n");
        return 1;
    }

    /* 2. pack once */
    packed_bytes = epc_net_packed_bytes(&cfg);
    CHECK_HIP(hipMalloc(&packed, packed_bytes));
    CHECK_EPC(epc_net_pack_weights(&cfg, g_name_ptrs, (const float* const*)g_dev, g_nvars, packed, packed_bytes, stream));

    /* 3. forward; cloud 1 is a copy of cloud 0 */
    h_xyz = (float*)malloc((size_t)num_clouds * N * 3 * sizeof(float));
    h_out = (float*)malloc((size_t)num_clouds * 256 * sizeof(float));
    for (k = 0; k < (size_t)num_clouds * N * 3; ++k) h_xyz[k] = 2.0f * frand() - 1.0f;
    memcpy(h_xyz + (size_t)N * 3, h_xyz, (size_t)N * 3 * sizeof(float));
    CHECK_HIP(hipMalloc((void**)&d_xyz, (size_t)num_clouds * N * 3 * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&d_out, (size_t)num_clouds * 256 * sizeof(float)));
    CHECK_HIP(hipMemcpy(d_xyz, h_xyz, (size_t)num_clouds * N * 3 * sizeof(float), hipMemcpyHostToDevice));
    ws_bytes = epc_net_workspace_bytes(&cfg, num_clouds);
    CHECK_HIP(hipMalloc(&ws, ws_bytes));
    CHECK_EPC(epc_net_forward(&cfg, packed, d_xyz, num_clouds, d_out, ws, ws_bytes, stream)); /* warm-up */
    CHECK_HIP(hipEventCreate(&e0));
    CHECK_HIP(hipEventCreate(&e1));
    CHECK_HIP(hipEventRecord(e0, stream));
    for (i = 0; i < iters; ++i) CHECK_EPC(epc_net_forward(&cfg, packed, d_xyz, num_clouds, d_out, ws, ws_bytes, stream));
    CHECK_HIP(hipEventRecord(e1, stream));
    CHECK_HIP(hipStreamSynchronize(stream));
    CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    CHECK_HIP(hipMemcpy(h_out, d_out, (size_t)num_clouds * 256 * sizeof(float), hipMemcpyDeviceToHost));

    /* 4. checks */
    for (i = 0; i < num_clouds; ++i) {
        double ss = 0.0;
        for (k = 0; k < 256; ++k) {
            const float v = h_out[(size_t)i * 256 + k];
            if (!isfinite(v)) {
                fprintf(stderr, "non-finite descriptor value\n");
                return 4;
            }
            ss += (double)v * v;
        }
        if (fabs(sqrt(ss) - 1.0) > worst) worst = (float)fabs(sqrt(ss) - 1.0);
    }
    for (k = 0; k < 256; ++k) same = fmaxf(same, fabsf(h_out[k] - h_out[256 + k]));
    printf("epc_net_forward: %d clouds x %d points -> %d x 256 in %.3f ms (%.0f clouds/s); | |d| - 1 | <= %.2e; repeated cloud "
           "differs by %.2e; packed weights %zu B, workspace %zu B, library version %d\n",
           num_clouds, N, num_clouds, ms / iters, num_clouds * iters / (ms * 1e-3), worst, same, packed_bytes, ws_bytes,
           epc_version());
    if (worst > 1e-3f || same != 0.0f) {
        fprintf(stderr, "descriptor check failed\n");
        return 5;
    }
    return 0;
}
