/*
 * epcnet.h -- C ABI of libepcnet_hip.so: the MI355X (gfx950) EPC-Net hot path.
 *
 * The reference (fpthink/EPC-Net) has no FFI: its "operator API" is Python graph-building functions executed
 * by TensorFlow.  This header declares the boundary a maintainer would bind in their place
 * (ctypes, a TF custom op, cgo, JNI ... see INTEGRATION.md).  Each entry cites the reference code it replaces
 * (paths relative to the reference tree).
 *
 * Conventions
 *   - plain C: pointers + sizes, no C++/torch/HIP types.  `stream` is a hipStream_t passed as void* (NULL =
 *     default stream).  All pointers are DEVICE pointers unless a parameter is documented as host.
 *   - the caller owns every buffer; the library allocates nothing and keeps no mutable global state
 *     (except the thread-local last-error string).  Calls are asynchronous on `stream` and re-entrant
 *     across streams as long as workspaces are distinct.
 *   - tensors are contiguous row-major float32 (indices int32) unless a parameter is documented as fp16 (the
 *     activation rows and MFMA fragments that travel between EPC-Net's stages; type void*).
 *   - return value: EPC_OK (0) or a negative epc_status.
 */
#ifndef EPCNET_H
#define EPCNET_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum epc_status {
    EPC_OK = 0,
    EPC_EINVAL = -1, /* bad shape / alignment / unsupported configuration */
    EPC_ENOMEM = -2, /* workspace or packed-weight buffer too small       */
    EPC_EHIP = -3,   /* a HIP runtime call failed: see epc_last_error()   */
    EPC_ENOTFOUND = -4, /* a required named tensor is missing             */
    EPC_ERANGE = -5    /* EPC_PRECISION_FAST: a folded weight leaves fp16's range (use EPC_PRECISION_F32) */
} epc_status;

#define EPC_ARCH_EPC_NET 0   /* models/epc-net.py   */
#define EPC_ARCH_EPC_NET_L 1 /* models/epc-net-l.py */

/* Arithmetic of the inference path (epc_cfg.precision).  The reference computes in float32 (models/epc-net.py:24-26
 * tf.placeholder(tf.float32 ...), every op of utils/tf_util.py:52-107).
 *   EPC_PRECISION_F32  f32-equivalent (the default; EPC-Net-L always).  Every dense contraction is a SPLIT product on the
 *                      16-bit matrix pipe with f32 accumulation -- three MFMA products (lo*hi + hi*lo + hi*hi) per algorithmic one:
 *                        - the conv layers (the 64 -> 64 layers of the ProxyConv blocks, conv5 of both models): SCALED split-fp16
 *                          ("f16x3").  Every row of the activation operand and every column of the BN-folded weight operand is
 *                          multiplied by a power of two that brings its largest magnitude into [2^14, 2^15) (exact), then split
 *                          hi = fp16(v), lo = fp16(v - hi): 22 significant bits, no range restriction beyond float32's; the
 *                          accumulator is un-scaled by inv_row * inv_col in the epilogue.  2^-21 relative per product;
 *                        - the soft assignment (feat . cluster_weights) and the VLAD aggregate (assign^T . feat): split-bf16
 *                          ("bf16x3": hi = bf16(v), lo = bf16(v - hi); bf16 keeps float32's range, which the un-normalised
 *                          features need).  2^-16 relative per product, on operands the later normalisations average;
 *                        - conv1 (3 -> 64), the kNN distances, every normalisation, sum and the head: float32 VALU arithmetic.
 *                      Tensors that cross HBM between stages are float32, with ONE exception inside the pipeline: `feat`, the
 *                      (points, 1024) conv5 map between epc_conv5_assign_f32_fwd and epc_vlad_aggregate_f32_fwd, is stored as
 *                      3-byte values (the upper 24 bits of the float32, rounded) -- the 16 significant bits its only reader
 *                      keeps when it splits into bf16 hi + lo.  Measured 2-3e-7 from the float32 oracle on ordinary weights,
 *                      <= 1e-4 on the adversarial set (tests/test_gpu_adversarial.py).
 *   EPC_PRECISION_FAST EPC-Net only: one fp16 value per activation, weights fp16 hi + MX-fp6 lo, fp16 tensors in HBM
 *                      (DESIGN.md 2).  Folded weights must satisfy |W' * 256| <= 65504 (checked by
 *                      epc_net_pack_weights: EPC_ERANGE) and activations must stay inside fp16's range (checked by the
 *                      kernels per cloud: EPC_STATUS_FP16_RANGE, the cloud's descriptor is returned as NaN, never as a
 *                      wrong finite vector). */
#define EPC_PRECISION_F32 0
#define EPC_PRECISION_FAST 1

/* Per-cloud status bits (int32 per cloud; epc_net_forward keeps them in its workspace and epc_net_last_status copies
 * them out).  A cloud with a non-zero status gets a NaN descriptor: the reference returns NaN for a cloud with a NaN/Inf
 * coordinate as well (every a_ij of utils/tf_util.py:651-656 involving the point is NaN). */
#define EPC_STATUS_NONFINITE_INPUT 1 /* a coordinate of the cloud is NaN or +-Inf                                 */
#define EPC_STATUS_FP16_RANGE 2      /* EPC_PRECISION_FAST: an activation left fp16's range (|v| > 65504)         */

#define EPC_KNN_SELECT 20 /* utils/tf_util.py:660: tf.nn.top_k(a, k=20), hard-coded */
#define EPC_KNN_CAP 32    /* neighbour-list slots per point; rows with more ties take the exact scan path */

/* Model configuration = the YAML keys `forward` consumes (models/epc-net.py:34-40,144; configs/epc-net.yaml). */
typedef struct epc_cfg {
    int32_t arch;          /* EPC_ARCH_*                                   */
    int32_t num_points;    /* NUM_POINTS (N), multiple of 32               */
    int32_t input_dim;     /* INPUT_DIM, must be 3                         */
    int32_t knn;           /* KNN: the DIVISOR of the neighbour mean       */
    int32_t cluster_size;  /* CLUSTER_SIZE, must be 64   (EPC-Net only)    */
    int32_t output_dim;    /* FEATURE_OUTPUT_DIM, must be 256              */
    int32_t groups;        /* GROUPS, must divide 1024   (EPC-Net only)    */
    int32_t micro_batch;   /* clouds processed per internal pass (<= 0: library default) */
    int32_t precision;     /* EPC_PRECISION_* (EPC-Net-L: ignored, always f32-equivalent) */
} epc_cfg;

const char* epc_last_error(void); /* thread-local, valid until the next failing call on this thread */
int epc_version(void);
/* Host utility: CRC-32C as stored by TensorFlow's checkpoint bundles (tf.train.Saver, train.py:611: per-tensor and
 * per-table-block checksums).  crc = epc_crc32c(0, data, n); chainable over pieces. */
uint32_t epc_crc32c(uint32_t crc, const void* data, size_t n);

/* ------------------------------------------------------------------------------------------------------ */
/* Whole-path entry points: replace `MODEL.forward(...)` inside `sess.run` for is_training=False           */
/* (train.py:254, evaluate.py:250-251 -> models/epc-net.py:29-157, models/epc-net-l.py:29-102).             */
/* ------------------------------------------------------------------------------------------------------ */

/* Bytes of the packed inference-weight buffer (BN folded, MFMA operand order; depends on cfg->precision). */
size_t epc_net_packed_bytes(const epc_cfg* cfg);

/* Fold + pack the reference's variables.  `names[i]` are the checkpoint names relative to the outer
 * `query_triplets/` scope (e.g. "fastdgcnn/conv1/weights", "VLAD/cluster_bn/moving_mean" -- table in
 * tests/golden/ckpt_tables.json); `tensors[i]` the matching float32 DEVICE buffers in the reference's shapes.
 * `names`/`tensors` are HOST arrays.  EPC_ENOTFOUND if a variable the architecture needs is absent.
 * EPC_PRECISION_FAST: the call additionally checks that every folded weight and bias fits fp16 after the 2^8 packing
 * scale (a small moving variance or a large gamma can make |W'| exceed 255.9) and returns EPC_ERANGE otherwise -- it
 * never packs an Inf.  That check reads one word back, so in FAST precision the call synchronises `stream`; callers that
 * want automatic selection pack FAST first and fall back to EPC_PRECISION_F32 on EPC_ERANGE (epc-net_amd/engine.py). */
int epc_net_pack_weights(const epc_cfg* cfg, const char* const* names, const float* const* tensors, int n,
                         void* packed, size_t packed_bytes, void* stream);

size_t epc_net_workspace_bytes(const epc_cfg* cfg, int num_clouds);

/* xyz (num_clouds, N, 3) -> out (num_clouds, output_dim) unit-norm descriptors ("last_output",
 * models/epc-net.py:153-155).  num_clouds = B*P of the reference's (B,P,N,3) placeholder. */
int epc_net_forward(const epc_cfg* cfg, const void* packed, const float* xyz, int num_clouds, float* out,
                    void* workspace, size_t workspace_bytes, void* stream);

/* Per-cloud status words (EPC_STATUS_*) of the LAST pass that ran in `workspace` (the last min(num_clouds, micro_batch)
 * clouds of the call), copied to the HOST array status_host; synchronises `stream`.  A non-zero word means that cloud's
 * descriptor was returned as NaN. */
int epc_net_last_status(const epc_cfg* cfg, const void* workspace, int num_clouds, int32_t* status_host, void* stream);

/* Throughput form of epc_net_forward for num_clouds > micro_batch: successive passes (micro_batch clouds each) are
 * dealt round-robin over `stream` and `num_aux` (0..7) auxiliary streams of the caller, so that the passes' stages --
 * bound by different units of the chip: kNN by VALU issue, conv5 by the MFMA pipe, the VLAD aggregate by HBM --
 * overlap.  Same results as epc_net_forward (passes are independent: evaluate.py:351-452 extracts one cloud per
 * sess.run).  Stream-ordered on `stream`: the auxiliary streams are made to wait for it before the first launch and
 * it waits for them at the end.  workspace_bytes >= min(passes, 1 + num_aux) * epc_net_workspace_bytes(cfg, n). */
int epc_net_forward_overlapped(const epc_cfg* cfg, const void* packed, const float* xyz, int num_clouds, float* out,
                               void* workspace, size_t workspace_bytes, void* stream, void* const* aux_streams,
                               int num_aux);

/* Stage profile: same launches as epc_net_forward, plus HIP events recorded on `stream` at the stage boundaries
 * (measurement only; SURVEY.md 5 "tracing": the reference wraps sess.run in RunOptions(FULL_TRACE),
 * evaluate.py:275,385-390).  One profile covers one pass (num_clouds <= micro_batch).  Read the per-stage
 * milliseconds with epc_profile_elapsed_ms after the stream has been synchronised. */
#define EPC_STAGE_SORT 0
#define EPC_STAGE_KNN 1
#define EPC_STAGE_CONV1 2
#define EPC_STAGE_BLOCK1 3 /* .. EPC_STAGE_BLOCK1+3 */
#define EPC_STAGE_CONV5 7  /* conv5 + L2 + assignment (EPC-Net) / conv5 + max-pool (EPC-Net-L) */
#define EPC_STAGE_AGGREGATE 8
#define EPC_STAGE_HEAD 9
#define EPC_NUM_STAGES 10
typedef struct epc_profile epc_profile;
int epc_profile_create(epc_profile** prof);
int epc_profile_destroy(epc_profile* prof);
int epc_net_forward_profiled(const epc_cfg* cfg, const void* packed, const float* xyz, int num_clouds, float* out,
                             void* workspace, size_t workspace_bytes, void* stream, epc_profile* prof);
int epc_profile_elapsed_ms(epc_profile* prof, float* stage_ms /* host, EPC_NUM_STAGES floats */);

/* ------------------------------------------------------------------------------------------------------ */
/* Stage entry points (what epc_net_forward chains; exported for parity tests and for op-level callers).    */
/* ------------------------------------------------------------------------------------------------------ */

/* Spatial sort of each cloud's points along the 3-D Hilbert curve (10-bit cells of the cloud's bounding box; ties by
 * point index; the entry point keeps its first name): xyz (num_clouds,N,3) -> xyz_sorted, optional perm (num_clouds,N)
 * with xyz_sorted[r] = xyz[perm[r]].  No reference counterpart: descriptors are permutation-invariant, the
 * pipeline sorts first so that kNN tiles are spatially tight and gathers are cache-local.  N <= 16384. */
int epc_morton_sort(const float* xyz, int num_clouds, int n, float* xyz_sorted, int32_t* perm, void* stream);

/* utils/tf_util.py:647-666 pairwise_distance_mask, in index form.  For every point i of every cloud:
 *   kth[i]  = 20th largest a_ij (with multiplicity), a_ij = -((|p_i|^2 + -2 p_i.p_j) + |p_j|^2) in fp32,
 *             products/sums rounded individually (no FMA), inner = (x x' + y y') + z z';
 *   cnt[i]  = #{j : a_ij >= kth[i]}  (>= 20; ties and zero-padded clouds make it larger);
 *   idx[i]  = the first min(cnt, cap) such j in ascending order (cap slots per point).
 * The dense mask the reference materialises is mask[i][j] = (a_ij >= kth[i]).
 * n <= 8192: the cloud sits in LDS and four lanes share a query.  The older one-lane-per-query kernel (the same kth, cnt and idx
 * bit for bit) is reachable through epc_knn_topk_form(..., form = 0) only: the library reads no environment. */
int epc_knn_topk(const float* xyz, int num_clouds, int n, int cap, int32_t* idx, int32_t* cnt, float* kth,
                 void* stream);

/* Dense (num_clouds,N,N) 0/1 float mask from kth -- API parity with pairwise_distance_mask's return value
 * (64 MB per 4096-point cloud: for tests / op-level callers only; the fused path never builds it). */
int epc_knn_mask(const float* xyz, const float* kth, int num_clouds, int n, float* mask, void* stream);

/* models/epc-net.py:66-69 conv1 (3->64) + folded BN + ReLU.  Writes x (M,64) f32 and/or x16 (M,64) fp16 (either may
 * be NULL): EPC-Net's blocks consume the fp16 rows, EPC-Net-L's the f32 rows. */
int epc_conv1_fwd(const float* xyz, const void* packed_conv1, int num_points_total, float* x, void* x16, void* stream);
/* epc_knn_topk and epc_conv1_fwd of the same (sorted) clouds in ONE launch: the kNN workgroup holds the cloud in LDS, so
 * conv1 costs it ~1.5 % more work instead of a launch of its own.  Bit-identical to the two separate calls.
 * idx_u16 != 0: the lists are written as uint16 entries (cap slots per point; n <= 8192) -- the pipeline's format, half
 * the list bytes for this kernel and for every epc_proxyconv_block_fwd that reads them.
 * status (num_clouds int32, may be NULL): OVERWRITTEN with EPC_STATUS_NONFINITE_INPUT / 0 per cloud, and
 * EPC_STATUS_FP16_RANGE is added when a conv1 output does not fit the fp16 row (x16).  Rows of points with a non-finite
 * coordinate have fewer than 20 selected entries; their unused list slots hold the point's own index, so every list
 * entry any consumer reads is a valid row number. */
int epc_knn_topk_conv1(const float* xyz, int num_clouds, int n, int cap, void* idx, int idx_u16, int32_t* cnt, float* kth,
                       const void* packed_conv1, float* x, void* x16, int32_t* status, void* stream);
/* Test / tuning entry points: epc_knn_topk and epc_knn_topk_conv1 with the form of the LDS kernel (num_points <= 8192) chosen by the
 * caller -- form 0 = one lane per query, 1 = four lanes per query (what the two entry points above run).  Both forms leave bit-identical
 * lists, counts and thresholds (tests/test_gpu_knn_forms.py); an explicit argument, because the library reads no environment. */
int epc_knn_topk_form(const float* xyz, int num_clouds, int n, int cap, int32_t* idx, int32_t* cnt, float* kth, int form, void* stream);
int epc_knn_topk_conv1_form(const float* xyz, int num_clouds, int n, int cap, void* idx, int idx_u16, int32_t* cnt, float* kth,
                            const void* packed_conv1, float* x, void* x16, int32_t* status, int form, void* stream);

/* models/epc-net.py:70-83 (and :87-100, :104-117, :121-132): one ProxyConv block after its leading conv:
 *   xm = (sum_{j in nbr(i)} x_j) / knn ; t = xm - x ; t = conv_a(t) ; t = conv_b(t) ; out = t + xm ;
 *   x_next = conv_{b+1}(out) (when has_next).
 * `out` is written with row stride `out_stride` elements at column offset `out_off` (the concat buffer of
 * models/epc-net.py:134).  Two forms, selected by x16 and matching the arch the weights were packed for:
 *   x16 == NULL (f32-equivalent pack of either model): f32 rows x -> out, x_next (f32); scaled split-fp16 (f16x3) MFMA
 *     layers, f32-accurate (EPC_PRECISION_F32 above).
 *   x16 != NULL (EPC-Net pack):   fp16 rows x16 -> out16, x_next16 (fp16; x, out, x_next are ignored): every tensor that
 *     crosses HBM is fp16 and every MFMA operand is one fp16 value per activation against fp16 hi+lo weights; sums,
 *     mean, xm - x, t + xm and the accumulators are f32.  The roundings are independent per point and channel and
 *     average out in the VLAD aggregation (descriptor effect 6e-7). */
/* idx: the neighbour lists of epc_knn_topk (int32 entries, idx_u16 = 0) or of epc_knn_topk_conv1 with idx_u16 = 1
 * (uint16 entries, cap slots per point either way).
 * status (num_clouds int32, may be NULL): the fp16 form ORs EPC_STATUS_FP16_RANGE into a cloud's word when one of the
 * values it rounds to fp16 (xm - x, the conv_a activations, out, x_next) is outside fp16's range. */
int epc_proxyconv_block_fwd(const float* x, const void* x16, const float* xyz, const void* idx, int idx_u16,
                            const int32_t* cnt, const float* kth, int cap, const void* packed_block, int has_next, int num_clouds,
                            int n, int knn, float* out, void* out16, int out_stride, int out_off, float* x_next,
                            void* x_next16, int32_t* status, void* stream);

/* models/epc-net.py:136-139,147-148 + loupe.py:249-272: conv5 (+BN+ReLU), per-point L2 normalisation and the soft
 * assignment, in split fp16 MFMA arithmetic with f32 accumulation (one fp16 value per activation, weights as fp16
 * hi + lo: DESIGN.md 2).  cat (M, cin): f32 when cat_fp16 == 0, fp16 (the blocks' out16) when cat_fp16 == 1 -- the
 * kernel rounds f32 input to fp16 on load, so both give the same result ->
 *   feat_frag   (M/32, 32 chunks, 2 halves s, 64 lanes, 8 fp16) = 2 bytes per value: the UN-normalised conv5 output
 *               rounded to fp16, in the kernel's accumulator-fragment order: lane l of (tile g, chunk c, half s) holds
 *               point 32g + (l&31), element q = channel 32c + 16s + 8(q>>2) + 4(l>>5) + (q&3).  Range: values must be
 *               below 65504 (a BN+ReLU output);
 *   rnorm (M)   rsqrt(max(|feat|^2, 1e-12)) from the f32 values;
 *   assign (M,64)  softmax(cluster_bn((feat*rnorm) @ cluster_weights)), f32, point-major; OPTIONAL (may be NULL: the
 *               aggregate consumes assign_frag);
 *   assign_frag (M/32, 2 cluster tiles, 2 k-steps, 64 lanes, 8 fp16): assign * 2^14 as B fragments (lane l of
 *               (tile g, t, s): cluster 32t + (l&31) at points 32g + 16s + 8(l>>5) + 0..7);
 *   apart (M/32, 64)  per-tile sums of assign over its 32 points (a_sum partials, loupe.py:276).
 * status (M/n int32, may be NULL; n = points per cloud): EPC_STATUS_FP16_RANGE is ORed into a cloud's word when a
 * conv5 output of one of its points does not fit fp16 (detected on the f32 row norm: |feat|^2 > 65504^2). */
int epc_conv5_assign_fwd(const void* cat, int cat_fp16, int cin, const void* packed_conv5, int num_points_total, int n,
                         void* feat_frag, float* rnorm, float* assign, void* assign_frag, float* apart, int32_t* status,
                         void* stream);

/* The same stage in EPC_PRECISION_F32 (weights packed for that precision): cat (M, 256) f32 ->
 *   feat_frag   (M/32, 32 chunks, 3 pieces, 64 lanes, 16 bytes): the un-normalised conv5 output as 3-BYTE values -- the upper
 *               24 bits of the float32 (sign, exponent, 15 fraction bits), rounded to nearest: the 16 significant bits the
 *               aggregate's bf16 hi + lo split keeps.  Accumulator order of the 16x16x32 MFMA (csrc/conv5_f32.hip): the 48 bytes
 *               of lane l of (tile g, chunk c), pieces concatenated, are 16 values, little-endian; with li = l & 15, q = l >> 4,
 *               value 4t + r (t = 2 g2 + p) = point 32g + 16p + li, channel 32c + 16 g2 + 4q + r.  3 KB per point (M * 3072 bytes);
 *   rnorm, assign, apart as above;
 *   assign_frag (M/32, 2 cluster tiles t, 2 k-steps s, 2 parts (hi, lo), 64 lanes, 8 bf16): assign split as
 *               hi = bf16(a), lo = bf16(a - hi), B fragments (lane l: cluster 32t + (l&31), points 32g + 16s + 8(l>>5) + 0..7). */
int epc_conv5_assign_f32_fwd(const float* cat, int cin, const void* packed_conv5, int num_points_total, void* feat_frag,
                             float* rnorm, float* assign, void* assign_frag, float* apart, void* stream);

/* loupe.py:276-292: V[f][k] = sum_n (feat[n][f]*rnorm[n]) * assign[n][k] - a_sum[k] * centres[f][k] from the
 * fragment-ordered operands (fp16 MFMA, f32 accumulate; the 2^14 of assign_frag is removed), a_sum = the sum of the
 * n/32 per-tile partials `apart` of epc_conv5_assign_fwd, centres = cluster_weights2 (1024, 64) (the first 65536 floats of
 * the packed head stage).  Outputs V (num_clouds, 1024, 64) and colss (num_clouds, 32, 64): per cluster the sums of
 * V^2 over each chunk of 32 features (what the intra-normalisation of :295 needs). */
int epc_vlad_aggregate_fwd(const void* feat_frag, const void* assign_frag, const float* rnorm, const float* apart,
                           const float* centres, int num_clouds, int n, float* V, float* colss, void* stream);
/* EPC_PRECISION_F32 form: operands in the layouts of epc_conv5_assign_f32_fwd; (feat * rnorm) is split into bf16 hi + lo
 * and multiplied with the hi + lo assignments in three products (f32-equivalent). */
int epc_vlad_aggregate_f32_fwd(const void* feat_frag, const void* assign_frag, const float* rnorm, const float* apart,
                               const float* centres, int num_clouds, int n, float* V, float* colss, void* stream);

/* loupe.py:295-331 + models/epc-net.py:153: intra-normalisation, flatten + L2, grouped hidden projection with the
 * shared weight (+BN, summed over groups), context gating, final L2. */
size_t epc_vlad_head_workspace_bytes(int num_clouds, int groups);
/* status (num_clouds int32, may be NULL): a cloud whose word is non-zero gets a NaN descriptor (see EPC_STATUS_*). */
int epc_vlad_head_fwd(const float* V, const float* colss, const void* packed_head, int groups, int num_clouds,
                      float* out, const int32_t* status, void* workspace, size_t workspace_bytes, void* stream);

/* models/epc-net-l.py:84-98: conv5 (128->1024)+BN+ReLU, global max over N, fc1 (1024->256)+BN+ReLU, L2. */
int epc_conv5_maxpool_fwd(const float* cat, int cin, const void* packed_conv5, int num_clouds, int n,
                          float* pooled, void* stream);
int epc_fc_head_fwd(const float* pooled, const void* packed_fc, int num_clouds, float* out, const int32_t* status,
                    void* stream);

/* evaluate.py:463,481: exact k nearest database descriptors per query by Euclidean distance (replaces
 * sklearn KDTree.query(k=25)); ties -> lower database index first.  idx (Q,k) int32, dist (Q,k) float32. */
int epc_pairwise_topk(const float* database, int num_db, const float* queries, int num_q, int dim, int k,
                      int32_t* idx, float* dist, void* stream);
/* The same result for any database size, at matrix-pipe speed: the pairwise matrix |q|^2 + |d|^2 - 2 q.d as a tiled f32-MFMA
 * GEMM into the caller's workspace (queries are processed in passes of at most 2^28 / num_db rows), per query the k + 8
 * smallest entries of its row, re-ranked by the exact sum (q_c - d_c)^2 of epc_pairwise_topk; a query whose candidates do not
 * PROVE the answer (the k-th exact distance is not below the last candidate's GEMM value by more than the GEMM's rounding
 * bound: dozens of near-tied rows) is redone exactly over the whole database.  idx / dist are bit-identical to
 * epc_pairwise_topk's.  k <= 56, dim a multiple of 8. */
size_t epc_pairwise_topk_workspace_bytes(int num_db, int num_q);
int epc_pairwise_topk_ws(const float* database, int num_db, const float* queries, int num_q, int dim, int k, int32_t* idx,
                         float* dist, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------------ */
/* Training-step operators (config c3; train.py:251-277).  Per-layer forward/backward pairs: training-mode    */
/* BatchNorm needs batch statistics between layers (utils/tf_util.py:454-491), so nothing is folded here.     */
/* ------------------------------------------------------------------------------------------------------ */

/* C (M,N; row stride ldc) = op(A) op(B) (+ bias[N]) at f32 accuracy: bf16 MFMA on three bf16 pieces per f32 operand,
 * six products, f32 accumulation (shapes with a side below 64: the f32 MFMA).  A(m,k) = A[m*sAm + k*sAk], B(k,n) =
 * B[k*sBk + n*sBn]: y = x W (tf.nn.conv1d k=1 / tf.matmul, utils/tf_util.py:94,336; loupe.py:255,290,322), dx = dy W^T,
 * dW = x^T dy (split-K: f32 atomics into a zeroed C).  `batch` strided problems (bA, bB, bC elements apart). */
int epc_gemm_f32(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long sAm, long sAk,
                 long sBk, long sBn, int ldc, int batch, long bA, long bB, long bC, int splitk, int accumulate,
                 void* stream);
/* Same interface, two bf16 pieces per operand instead of three (half the matrix-pipe work, 2^-16 relative per product):
 * for GEMMs that are linear in a gradient (the backward of the training step). */
int epc_gemm_f32_fast(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long sAm, long sAk,
                 long sBk, long sBn, int ldc, int batch, long bA, long bB, long bC, int splitk, int accumulate,
                 void* stream);

/* y = x W + b (the f32-accurate three-piece arithmetic of epc_gemm_f32, one problem, no split-K) TOGETHER with the batch
 * statistics a training-mode BatchNorm on y needs (utils/tf_util.py:472 tf.nn.moments over the rows): mean[N] and POPULATION
 * var[N].  The GEMM's epilogue leaves per row tile and column a pivot p (the product's value in the tile's first row) and the
 * sums of (y - p) and (y - p)^2 in `stats` (epc_gemm_stats_tiles(M) * 3 * N floats); a second tiny launch merges the tiles'
 * (count, mean, M2) in tile order in double precision (Chan's pairwise update) -- no pass of epc_col_moments over y, and no
 * cancellation for columns whose |mean| is far above their spread.  M, N >= 64, K >= 32. */
int epc_gemm_stats_tiles(int M);
int epc_gemm_f32_stats(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long sAm, long sAk,
                       long sBk, long sBn, int ldc, float* stats, size_t stats_floats, float* mean, float* var, void* stream);
/* epc_gemm_f32_stats with ONE bf16 value per operand (one product, f32 accumulate): the "bf16" training arithmetic. */
int epc_gemm_bf16_stats(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long sAm, long sAk,
                        long sBk, long sBn, int ldc, float* stats, size_t stats_floats, float* mean, float* var, void* stream);
/* The same product in the split-fp16 three-product arithmetic (2^-22 per product at half the matrix work of the six-product
 * form): A * 2^a_scale_log2 and B * 2^b_scale_log2 are split into fp16 hi + lo as they are staged (magnitudes beyond fp16's range
 * are clamped to it) and the product is un-scaled exactly.  For operands that are bounded by construction (BatchNorm / l2-normalised
 * activations against weights: conv5 of models/epc-net.py:136, the assignment of loupe.py:255); choose the exponents so that typical
 * magnitudes land high in fp16's range. */
int epc_gemm_f16x3_stats(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long sAm, long sAk,
                         long sBk, long sBn, int ldc, int a_scale_log2, int b_scale_log2, float* stats, size_t stats_floats,
                         float* mean, float* var, void* stream);


/* Same interface, operands rounded to ONE bf16 value each (f32 data in memory, f32 accumulation, one product): the
 * "bf16" training configuration of BASELINE.json configs[2].  2^-9 relative per operand. */
int epc_gemm_bf16(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long sAm, long sAk,
                 long sBk, long sBn, int ldc, int batch, long bA, long bB, long bC, int splitk, int accumulate,
                 void* stream);

/* Linear layers with 1 <= cin <= 4 (conv1 of the training step: 3 -> 64 on the coordinates, models/epc-net.py:66-69): the forward
 * z = x W + b as three FMAs per output (cout a multiple of 4), and dW = x^T dy for cout == 64 as per-workgroup partials added in
 * ascending order (`partials`: epc_linear_smallk_dw_partial_floats(rows, cin) floats of scratch) -- the same bits on every run. */
int epc_linear_smallk_fwd(const float* x, const float* W, const float* bias, int rows, int cin, int cout, float* z, void* stream);
size_t epc_linear_smallk_dw_partial_floats(int rows, int cin);
int epc_linear_smallk_dw(const float* x, const float* dy, int rows, int cin, int cout, float* dW, float* partials,
                         size_t partial_floats, void* stream);

/* y = x W + b for a 64 -> 64 layer (x, y: (rows, 64) row-major, 16-byte aligned; W: (64 in, 64 out)) TOGETHER with the batch
 * moments of y (mean, population variance: tf.nn.moments): one pass over the rows in the f32-accurate six-product
 * arithmetic with per-workgroup pivot-shifted column sums (as epc_gemm_f32_stats), and a small second launch that pools
 * them in a fixed order in double precision.
 * `workspace`: the column-reduction workspace (epc_colreduce_workspace_bytes(rows, 64)).
 * Replaces tf.nn.conv1d + bias_add + tf.nn.moments of utils/tf_util.py:94-99, 472 for the thin layers. */
int epc_linear_stats64(const float* x, const float* W, const float* bias, int rows, float* z, float* mean, float* var,
                       void* workspace, size_t workspace_bytes, void* stream);
/* The same with the layer's input formed as it is loaded: x_in = relu(batch_norm(x_pre)) from the batch moments and affine
 * parameters of the layer that produced x_pre (conv*_a -> conv*_b, models/epc-net.py:77-85: the activation between the two
 * layers is never written). */
int epc_linear_stats64_bn(const float* x_pre, const float* in_mean, const float* in_var, const float* in_gamma,
                          const float* in_beta, float eps, const float* W, const float* bias, int rows, float* z, float* mean,
                          float* var, void* workspace, size_t workspace_bytes, void* stream);

/* Backward of a 64 -> 64 layer followed by a training-mode BatchNorm (+ReLU) (utils/tf_util.py:94-106 from the gradient side:
 * the thin layers conv*_a / conv*_b / conv2..4 of models/epc-net.py:66-132) in three launches instead of four GEMM-sized ones:
 * the BatchNorm column sums (dbeta, dgamma), ONE pass over the rows that forms dz in registers and produces dx = dz W^T
 * (dx may be NULL) and the per-workgroup partials of dW = x^T dz (dz is never written), and an ORDERED sum of the partials:
 * dW is the same bits on every run.  dy, z, x: (rows, 64) row-major; W: (64 in, 64 out); `workspace`: the column-reduction
 * workspace (epc_colreduce_workspace_bytes(rows, 64): workspace contract below); `dw_partials`:
 * epc_linear_bn_bwd64_partial_floats(rows) floats of scratch.  Two bf16 pieces per operand (epc_gemm_f32_fast's arithmetic). */
size_t epc_linear_bn_bwd64_partial_floats(int rows);
int epc_linear_bn_bwd64(const float* dy, const float* z, const float* x, const float* W, const float* mean, const float* var,
                        const float* gamma, const float* beta, float eps, int relu, int rows, float* dx, float* dW,
                        float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, float* dw_partials,
                        size_t dw_partial_floats, void* stream);
/* The same for a layer inside a fused block.  in_* (all four or none): the layer's input was relu(batch_norm(x)) of the previous
 * layer's pre-activation x (epc_linear_stats64_bn) and is re-formed as x is loaded.  dx_addend (optional, (rows, 64)): dx leaves
 * as dz W^T + dx_addend -- the gradient reaching the same tensor by the block's residual path (models/epc-net.py:86). */
int epc_linear_bn_bwd64_ex(const float* dy, const float* z, const float* x, const float* in_mean, const float* in_var,
                           const float* in_gamma, const float* in_beta, const float* W, const float* mean, const float* var,
                           const float* gamma, const float* beta, float eps, int relu, int rows, float* dx, const float* dx_addend,
                           float* dW, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, float* dw_partials,
                           size_t dw_partial_floats, void* stream);

/* Split-K WITHOUT atomics (the forward products of the training step: the same bits on every run).  Every K slice stores its
 * partial product in `workspace` (batch * splitk * M * N floats) and a second launch adds the slices in ascending order (+ bias,
 * + C when accumulate).  pieces: 3 = the arithmetic of epc_gemm_f32, 2 = epc_gemm_f32_fast, 1 = epc_gemm_bf16.  N, ldc and bC
 * must be multiples of 4.  Replaces the same tf.matmul sites as epc_gemm_f32 (loupe.py:286-291, 306; utils/tf_util.py:336). */
int epc_gemm_splitk_det(const float* A, const float* B, float* C, const float* bias, int M, int N, int K, long sAm, long sAk,
                        long sBk, long sBn, int ldc, int batch, long bA, long bB, long bC, int splitk, int accumulate,
                        int pieces, float* workspace, size_t workspace_floats, void* stream);

/* Column reductions over the rows of (rows, C) tensors, C a multiple of 4 (at most 4096): per-panel partial sums, then a small
 * second launch that adds them in a fixed order (deterministic).
 * WORKSPACE CONTRACT: epc_colreduce_workspace_bytes(rows, C) bytes, 16-byte aligned, contents arbitrary (the completion
 * counters of earlier versions are gone); reuse it for any number of calls on ONE stream (calls sharing a workspace must be
 * stream-ordered).
 * epc_col_moments: tf.nn.moments -- mean and POPULATION variance (utils/tf_util.py:472), one pass over x. */
size_t epc_colreduce_workspace_bytes(int rows, int C);
int epc_col_moments(const float* x, int rows, int C, float* mean, float* var, void* workspace, size_t workspace_bytes,
                    void* stream);
int epc_col_sum(const float* x, int rows, int C, float* out, void* workspace, size_t workspace_bytes, void* stream);

/* tf.nn.batch_normalization (+ReLU) with given statistics, and its training-mode backward (statistics are functions
 * of z): dz, dgamma, dbeta.  utils/tf_util.py:490, loupe.py:257-263.  The backward recomputes the ReLU mask from z
 * with the forward's own expression, so the forward output is not an input.  Workspace: contract above. */
int epc_bn_apply_fwd(const float* z, const float* mean, const float* var, const float* gamma, const float* beta,
                     float eps, int relu, int rows, int C, float* y, void* stream);
/* y = act(batch_norm(z)) + addend: the block's residual, t + x1 (models/epc-net.py:86), in the same pass. */
int epc_bn_apply_add_fwd(const float* z, const float* mean, const float* var, const float* gamma, const float* beta,
                         float eps, int relu, int rows, int C, const float* addend, float* y, void* stream);
int epc_bn_apply_bwd(const float* dy, const float* z, const float* mean, const float* var, const float* gamma,
                     const float* beta, float eps, int relu, int rows, int C, float* dz, float* dgamma, float* dbeta,
                     void* workspace, size_t workspace_bytes, void* stream);

/* conv5's tail in one pass (models/epc-net.py:136-148), C == 1024: f = l2_normalize(relu(batch_norm(z))) over the
 * channels and rn (rows) = the reciprocal row norm; the BatchNorm output is not materialised.  Backward: epc_rownorm_bwd
 * on (df, f, rn), then epc_bn_apply_bwd with relu = 1 (it recomputes the mask from z). */
int epc_bn_relu_rownorm_fwd(const float* z, const float* mean, const float* var, const float* gamma, const float* beta,
                            float eps, int rows, int C, float* f, float* rn, void* stream);
/* Its backward in two passes over (df, z) (f is recomputed from z, rn; the row-norm's input gradient is never written):
 * dz (rows, C); dbeta_dgamma (2, C): row 0 = dbeta, row 1 = dgamma; rowdot: rows floats of scratch (sum_c df f per row);
 * partials: caller-owned scratch of epc_bn_relu_rownorm_bwd_partial_floats(rows).  Sums added in a fixed order. */
size_t epc_bn_relu_rownorm_bwd_partial_floats(int rows);
int epc_bn_relu_rownorm_bwd(const float* df, const float* z, const float* rn, const float* mean, const float* var,
                            const float* gamma, const float* beta, float eps, int rows, int C, float* dz, float* dbeta_dgamma,
                            float* rowdot, float* partials, size_t partial_floats, void* stream);

/* VLAD normalisations in one launch (loupe.py:284,292-298): v = raw - a_sum (x) w2; intra-normalisation over the F axis
 * per (cloud, cluster); L2 normalisation of the flattened (F*C) vector per cloud.  raw, out: (num_clouds, F, C) with
 * C == 64; a_sum: (num_clouds, C); w2: (F, C); r1: (num_clouds, C) and r2: (num_clouds) receive the two reciprocal
 * norms (inputs of the backward).  Backward: d raw and d a_sum; the cluster_weights2 gradient crosses clouds
 * (-sum_b a_sum[b][c] * draw[b][f][c]) and is left to the caller. */
int epc_vlad_normalize_fwd(const float* raw, const float* a_sum, const float* w2, int num_clouds, int F, int C,
                           float* out, float* r1, float* r2, void* stream);
int epc_vlad_normalize_bwd(const float* dout, const float* out, const float* r1, const float* r2, const float* w2,
                           int num_clouds, int F, int C, float* draw, float* da_sum, void* stream);

/* lazy_quadruplet_loss (models/epc-net.py:269-284 with best_pos_distance :160-167) on descriptors q (B,1,D), pos (B,P,D),
 * neg (B,Nn,D), other (B,1,D): loss[0] = mean_b max_n max(m1 + best_b - |neg_n - q|^2, 0) + mean_b max_n max(m2 + best_b -
 * |neg_n - other|^2, 0).  sel (B,3) int32 receives the selected positive and the two selected negatives (-1: hinge
 * inactive) for the backward, which writes all four gradients scaled by dloss[0] (a device scalar). */
int epc_lazy_quadruplet_loss_fwd(const float* q, const float* pos, const float* neg, const float* other, int B, int P,
                                 int Nn, int D, float m1, float m2, float* loss, int32_t* sel, void* stream);
int epc_lazy_quadruplet_loss_bwd(const float* q, const float* pos, const float* neg, const float* other,
                                 const int32_t* sel, const float* dloss, int B, int P, int Nn, int D, float* dq,
                                 float* dpos, float* dneg, float* dother, void* stream);

/* xm = (mask @ x) / knn in index form (models/epc-net.py:70-71) for 64-channel x (its transpose: epc_neighbour_mean_bwd_gather). */
int epc_neighbour_mean_fwd(const float* x, const float* xyz, const int32_t* idx, const int32_t* cnt, const float* kth,
                           int cap, int num_clouds, int n, int knn, float* xm, void* stream);

/* The two together as the blocks use them (models/epc-net.py:70-72): xm and diff = xm - x in one launch; backward
 * dx = mask^T (dxm + ddiff) / knn - ddiff over the transposed graph (below), dx overwritten. */
int epc_neighbour_mean_diff_fwd(const float* x, const float* xyz, const int32_t* idx, const int32_t* cnt,
                                const float* kth, int cap, int num_clouds, int n, int knn, float* xm, float* diff,
                                void* stream);
int epc_neighbour_mean_diff_bwd_gather(const float* dxm, const float* ddiff, const float* xyz, const int32_t* cnt,
                                       const float* kth, int cap, const int32_t* rdeg, const int32_t* roff,
                                       const int32_t* rlist, int num_clouds, int n, int knn, float* dx, void* stream);
/* The same from the SUM s = dxm + ddiff (epc_linear_bn_bwd64_ex with dx_addend = dxm writes it): one gathered tensor instead
 * of two; dx = mask^T s / knn - (s - dxm). */
int epc_neighbour_mean_diff_bwd_gather_sum(const float* s, const float* dxm, const float* xyz, const int32_t* cnt,
                                           const float* kth, int cap, const int32_t* rdeg, const int32_t* roff,
                                           const int32_t* rlist, int num_clouds, int n, int knn, float* dx, void* stream);

/* The kNN graph transposed: for every point j the points i whose neighbour list holds j (rows with more than `cap`
 * entries are not listed).  rdeg, roff, cursor: (num_clouds*n) int32; rlist: (num_clouds*n*cap) int32 holding absolute
 * row numbers, cloud c's lists inside [c*n*cap, (c+1)*n*cap).  Built once per step: the graph is the same for every block. */
int epc_knn_transpose(const int32_t* idx, const int32_t* cnt, int cap, int num_clouds, int n, int32_t* rdeg,
                      int32_t* roff, int32_t* cursor, int32_t* rlist, void* stream);
/* Backward of the neighbour mean as a GATHER over the transposed graph (dx is overwritten; rows with cnt > cap are
 * added by an exact scan) -- no float atomics: bit-reproducible. */
int epc_neighbour_mean_bwd_gather(const float* dxm, const float* xyz, const int32_t* cnt, const float* kth, int cap,
                                  const int32_t* rdeg, const int32_t* roff, const int32_t* rlist, int num_clouds, int n,
                                  int knn, float* dx, void* stream);

/* tf.nn.l2_normalize over the last axis (models/epc-net.py:148): y = x*rn, rn = rsqrt(max(sum x^2, 1e-12)). */
int epc_rownorm_fwd(const float* x, int rows, int C, float* y, float* rn, void* stream);
int epc_rownorm_bwd(const float* dy, const float* y, const float* rn, int rows, int C, float* dx, void* stream);

/* tf.nn.softmax over 64 clusters (loupe.py:272). */
int epc_softmax64_fwd(const float* x, int rows, float* y, void* stream);
int epc_softmax64_bwd(const float* dy, const float* y, int rows, float* dx, void* stream);
/* The same with the incoming gradient dy[row] + dsum[row / n_points] (dsum (rows/n_points, 64)): the gradient of
 * a_sum = reduce_sum(activation, -2) (loupe.py:276) reaches every point of its cloud, added here without being expanded. */
int epc_softmax64_bwd_bcast(const float* dy, const float* dsum, int n_points, const float* y, int rows, float* dx,
                            void* stream);

/* scratch floats of the per-cloud column sums (a_sum of loupe.py:276) inside epc_assign_softmax_fwd */
size_t epc_cloud_colsum64_partial_floats(int num_clouds);

/* The soft assignment behind its product, one pass each way (loupe.py:255-276; what epc_bn_apply_fwd + epc_softmax64_fwd +
 * a per-cloud column sum, and epc_softmax64_bwd_bcast + epc_bn_apply_bwd, do in five and four launches):
 *   fwd: a (num_clouds * n_points, 64) = softmax(batch_norm(z; mean, var, gamma, beta, eps)) and a_sum (num_clouds, 64) = the sum of
 *        a over each cloud's points, added in a fixed order; partials = epc_cloud_colsum64_partial_floats(num_clouds) floats.
 *   bwd: from da (gradient of a), dsum (num_clouds, 64) (gradient of a_sum, may be NULL), a and z:  dz (gradient of z; also used
 *        as scratch for the softmax's input gradient), dgamma, dbeta (64 each).  workspace: epc_colreduce_workspace_bytes(rows, 64),
 *        16-byte aligned.  rowdot (rows floats, may be NULL): sum_k (a[r][k] da[r][k] + dz[r][k] z[r][k]) -- with da = f dvlad and
 *        z = f Wc this IS sum_c df[r][c] f[r][c] of the feature gradient df = a dvlad^T + dz Wc^T, the row dot product the l2-norm
 *        backward of the features needs, before df exists (epc_vlad_df_tail). */
int epc_assign_softmax_fwd(const float* z, const float* mean, const float* var, const float* gamma, const float* beta, float eps,
                           int num_clouds, int n_points, float* a, float* a_sum, float* partials, size_t partial_floats,
                           void* stream);
int epc_assign_softmax_bwd(const float* da, const float* dsum, const float* a, const float* z, const float* mean, const float* var,
                           const float* gamma, const float* beta, float eps, int num_clouds, int n_points, float* dz, float* dgamma,
                           float* dbeta, float* rowdot, void* workspace, size_t workspace_bytes, void* stream);

/* Context gating's product (loupe.py:99-100): out = y * sigmoid(g); bwd: dy = dout * s, dg = dout * y * s * (1 - s). */
int epc_gate_fwd(const float* y, const float* g, long n, float* out, void* stream);
int epc_gate_bwd(const float* dout, const float* y, const float* g, long n, float* dy, float* dg, void* stream);

/* The VLAD tail behind the hidden projection as ONE launch each way (loupe.py:323-331 and :61-101): from h (B G, O), the projection's
 * output,   y = batch_norm(h) over the B G rows (training mode; mean1 / var1 = batch mean / population variance);   v[b] = the sum of
 * the G group rows of y (:326-328);   gl = v Wg (gating_weights, (O, O));   out = v * sigmoid(batch_norm(gl)) over the B rows
 * (mean2 / var2); var1u / var2u = var x bessel1 / bessel2, what the fused slim op feeds its moving variance (rows / (rows - 1)).
 * One workgroup; reductions meet in a fixed order (bit-reproducible); the (B, O) x (O, O) products on the matrix pipe with the per-op
 * GEMMs' f32-accurate arithmetic (forward three bf16 pieces per operand: six products; backward two: three) in both arithmetics of the
 * step: products with at most 32 rows stay float32 under "bf16" as well.  Shapes: epc_hidden_tail_ok(B, G, O): B <= 32, O in {64, 128, 256}.
 * The backward takes the forward's h, mean1, var1, v, gl, mean2, var2 and returns dh, dgamma1, dbeta1, dWg, dgamma2, dbeta2. */
int epc_hidden_tail_ok(int B, int G, int O);
int epc_hidden_tail_fwd(const float* h, int B, int G, int O, const float* gamma1, const float* beta1, const float* Wg,
                        const float* gamma2, const float* beta2, float eps, float bessel1, float bessel2, float* mean1,
                        float* var1, float* var1u, float* v, float* gl, float* mean2, float* var2, float* var2u, float* out, void* stream);
int epc_hidden_tail_bwd(const float* dout, const float* h, int B, int G, int O, const float* gamma1, const float* mean1, const float* var1,
                        const float* v, const float* gl, const float* Wg, const float* gamma2, const float* beta2, const float* mean2,
                        const float* var2, float eps, float* dh, float* dgamma1, float* dbeta1, float* dWg,
                        float* dgamma2, float* dbeta2, void* stream);

/* The grouped hidden projection itself in the training step (loupe.py:302-322; csrc/train_hidden.hip): Y (M, 256) = X (M, K) W (K, 256)
 * with M = B G rows and K = C F / G, and its gradients dX (M, K) = dY W^T, dW (K, 256) = X^T dY -- three skinny products around a 16-MB
 * weight matrix, each ONE pass over it by 256 workgroups (the tile GEMM's split-K forms took 34 + 10, 17 and 16 us at 18 clouds where the
 * bytes are 3 us each).  pieces: bf16 pieces per operand -- 1 (the "bf16" step), 2 (three products: the default step's backward products),
 * 3 (six products: its forward products); f32 accumulation, sums met in a fixed order (bit-reproducible).  Shapes:
 * epc_hidden_proj_ok(M, K, N): 64 <= M <= 128, M % 4 == 0, N == 256, K a positive multiple of 256; other shapes stay on the tile GEMMs (epc_gemm_f32 and its siblings).
 * dX or dW may be NULL (not computed).  scratch: the forward's K / 256 slice partials. */
int epc_hidden_proj_ok(int M, int K, int N);
size_t epc_hidden_proj_scratch_bytes(int M, int K);
int epc_hidden_proj_fwd(const float* X, const float* W, int M, int K, int pieces, float* Y, void* scratch, size_t scratch_bytes, void* stream);
int epc_hidden_proj_bwd(const float* X, const float* W, const float* dY, int M, int K, int pieces, float* dX, float* dW, void* stream);

/* ---- The 64-channel backbone of the training step as a chain of fused launches (csrc/train_chain.hip) -----------------------
 * models/epc-net.py:66-134 in training mode (utils/tf_util.py:52-107, 454-519): every 64 -> 64 layer is followed by a training-mode
 * BatchNorm + ReLU whose batch statistics need all rows.  A producer leaves per-workgroup PARTIALS -- epc_chain_parts(rows) of
 * them, one per 256 rows: moment partials [parts][3][64] (sums of (v - p), (v - p)^2 and the pivot p of its pre-activation WITHOUT
 * the bias) or BatchNorm-backward sum partials [parts][2][64] (sum dy [mask], sum dy [mask] zhat) -- and the CONSUMER pools them in
 * its prologue (double precision, fixed order; workgroup 0 writes the pooled mean / var, resp. dbeta / dgamma).  A BatchNorm + ReLU
 * is applied to an operand as it is loaded.  (rows, 64) tensors are dense f32 unless a stride (in floats) is given; `pieces`:
 * forward 3 = six bf16 products per f32 product (f32-accurate) or 1 = one bf16 value per operand; backward 2 or 1. */
int epc_chain_parts(int rows);
/* moment partials of z (rows, 64) as it stands (the first block's z0 = conv1's output, models/epc-net.py:66) */
int epc_chain_stats(const float* z, int rows, float* stats, void* stream);
/* a = relu(bn(zin)) (+ resid); a -> a_out (row stride a_stride; NULL: not written) and, a_out_bf16 != NULL, the same values rounded
 * to bf16 at the same element stride (the concat buffer as the bf16 head's left operand: epc_h16_conv5_fwd / _dw); W != NULL:
 * z_out = a W + bias, stats_out = its moment partials.  in_stats != NULL: the BatchNorm's batch moments are pooled from it (+ in_bias)
 * and written to in_mean / in_var; NULL: in_mean / in_var are read.  conv_b of a block (:78-79); a block's tail + the next block's
 * leading conv (:81-83). */
int epc_chain_fwd_linear(const float* zin, const float* in_stats, const float* in_bias, float* in_mean, float* in_var,
                         const float* in_gamma, const float* in_beta, float eps, const float* resid, float* a_out, int a_stride,
                         void* a_out_bf16, const float* W, const float* bias, float* z_out, float* stats_out, int rows, int pieces,
                         void* stream);
/* models/epc-net.py:70-76: x = relu(bn(z0)) formed as the rows are gathered; xm = mask x / knn over the kNN lists of
 * epc_knn_topk (int32, cap slots; rows with cnt > cap take the exact scan); d = xm - x; z_out = d W + bias + moment partials. */
int epc_chain_fwd_gather(const float* z0, const float* in_stats, const float* in_bias, float* in_mean, float* in_var,
                         const float* in_gamma, const float* in_beta, float eps, const float* xyz, const int32_t* idx,
                         const int32_t* cnt, const float* kth, int cap, int num_clouds, int n, int knn, const float* W,
                         const float* bias, float* xm, float* d, float* z_out, float* stats_out, int pieces, void* stream);
/* Backward of y = relu(bn(z)), z = in W + b for one 64 -> 64 layer: the BatchNorm's sums are pooled from `sums` (-> dgamma, dbeta);
 * dz = gamma rstd (dy [mask] - dbeta / rows - zhat dgamma / rows); dx = dz W^T (+ dx_addend); dw_partials[parts][64][64] += in^T dz
 * with in = x, or relu(bn_x(x)) when x_mean .. x_beta are given; zp != NULL: psums = the sum partials of (dx + addend) against the
 * BatchNorm (p_mean .. p_beta) of pre-activation zp -- what the next backward layer pools. */
int epc_chain_bwd_linear(const float* dy, int dy_stride, const float* z, const float* mean, const float* var, const float* gamma,
                         const float* beta, float eps, const float* sums, float* dgamma, float* dbeta, const float* W,
                         const float* x, int x_stride, const float* x_mean, const float* x_var, const float* x_gamma,
                         const float* x_beta, float* dx, const float* dx_addend, int addend_stride, float* dw_partials,
                         const float* zp, const float* p_mean, const float* p_var, const float* p_gamma, const float* p_beta,
                         float* psums, int rows, int pieces, void* stream);
/* Backward of the gather layer: dx[j] = (sum over the points i that select j of s[i]) / knn - (s[j] - dout[j]) over the transposed
 * graph of epc_knn_transpose plus the overflow lists of epc_knn_overflow_lists (no atomics); psums = the sum partials of dx against
 * the BatchNorm (mean .. beta) of z0. */
int epc_chain_bwd_gather(const float* s, const float* dout, int dout_stride, const int32_t* rdeg, const int32_t* roff,
                         const int32_t* rlist, const int32_t* ovf_cnt, const int32_t* ovf_list, const float* xyz, const float* kth,
                         int num_clouds, int n, int knn, const float* z0, const float* mean, const float* var, const float* gamma,
                         const float* beta, float eps, float* psums, float* dx, void* stream);
/* sum partials of (dy, z) for the chain's last layer; dz of a BatchNorm + ReLU alone with pooled sums (the first block's leading
 * BatchNorm); dW[l] = the sum of layer l's workgroup partials, ascending, for up to 16 layers in one launch. */
int epc_chain_sums(const float* dy, int dy_stride, const float* z, const float* mean, const float* var, const float* gamma,
                   const float* beta, float eps, int rows, float* psums, void* stream);
int epc_chain_bn_bwd(const float* dy, const float* z, const float* mean, const float* var, const float* gamma, const float* beta,
                     float eps, const float* sums, float* dgamma, float* dbeta, int rows, float* dz, void* stream);
int epc_chain_dw_sum(int layers, const float* const* partials, float* const* dW, int rows, void* stream);
/* ---- The same backbone's FORWARD as ONE persistent launch (csrc/train_chain_persist.hip) -----------------------------------------
 * One workgroup per CU keeps its rows for the whole pass (a wave owns one 32-row tile; the tile handed to the next layer stays in
 * LDS) and every training-mode BatchNorm costs a grid-wide BARRIER instead of a kernel boundary; the batch moments are reduced in
 * two levels on the way through it.  Same arithmetic, operands and outputs as the launches above (the statistics' summation order
 * differs: group partials by row range).  Usable when epc_chain_persist_ok(rows) != 0: at most 12 tiles per workgroup (rows <=
 * 384 x CUs) and every workgroup co-resident by the occupancy query.  Nothing else may need the CUs these workgroups wait on: do not
 * run it beside a kernel that waits for IT.  Every spin is bounded (spin_ticks of the 100-MHz s_memrealtime, 0 = a quarter second): on
 * a time-out the sticky error word of the workspace is set, every workgroup leaves, the concat holds NaN rows (so the loss is NaN) and every later launch on
 * the same workspace returns at once -- epc_chain_persist_status() (synchronises the stream) then returns EPC_EHIP until
 * epc_chain_persist_reset().  workspace: epc_chain_persist_workspace_bytes() bytes, zeroed ONCE with epc_chain_persist_init() and then
 * left to the library (launch sequence number, the barriers' tagged partials); one launch at a time per workspace. */
#define EPC_CHAIN_MAX_BLOCKS 4
typedef struct epc_chain_fwd_block {
    const float *gamma0, *beta0;   /* the block's leading BatchNorm (models/epc-net.py:66-69, 83, 99, 115) */
    const float* in_bias;          /* the bias z0's moment partials lack: NULL for the first block (z0 = conv1's output as it stands) */
    const float *Wa, *ba, *gamma_a, *beta_a, *Wb, *bb, *gamma_b, *beta_b;
    const float *W0_next, *b0_next; /* the NEXT block's leading conv; NULL in the last block */
    const float* z0;               /* the leading pre-activation: the input for the first block, the previous block's z0_next after */
    float *mean0, *var0, *mean_a, *var_a, *mean_b, *var_b;   /* out: batch moments */
    float *d, *za, *zb;            /* out (rows, 64): xm - x and the two pre-activations (the neighbour mean xm itself stays in registers) */
    float* z0_next;                /* out (rows, 64); NULL in the last block */
} epc_chain_fwd_block;
typedef struct epc_chain_fwd_args {
    epc_chain_fwd_block blk[EPC_CHAIN_MAX_BLOCKS];
    int nblocks;
    const float* xyz;              /* the kNN graph of epc_knn_topk: int32 lists, cap slots */
    const int32_t* idx;
    const int32_t* cnt;
    const float* kth;
    int cap, num_clouds, n, knn;
    float* cat;                    /* out (rows, 64 nblocks): the concat (models/epc-net.py:134) */
    void* cat_bf16;                /* optional: the same rounded to bf16 */
    float eps;
    void* workspace;
    long long spin_ticks;
} epc_chain_fwd_args;
int epc_chain_persist_ok(int rows);
size_t epc_chain_persist_workspace_bytes(void);
int epc_chain_persist_init(void* workspace, void* stream);
int epc_chain_fwd_persist(const epc_chain_fwd_args* a, int pieces, void* stream);
int epc_chain_persist_status(const void* workspace, void* stream);
int epc_chain_persist_reset(void* workspace, void* stream);
/* per cloud the points whose neighbour list overflowed (cnt > cap), ascending: ovf_cnt (num_clouds), ovf_list (num_clouds, n) */
int epc_knn_overflow_lists(const int32_t* cnt, int cap, int num_clouds, int n, int32_t* ovf_cnt, int32_t* ovf_list, void* stream);

/* Backward of loupe.py:255-291 with respect to the point features: df[b][n][f] = sum_k a[b][n][k] dvlad[b][f][k] + sum_k dz[b][n][k] Wc[f][k]
 * (a, dz: (num_clouds n_points, 64) -- the soft assignment and the gradient of its pre-BatchNorm logits; dvlad (num_clouds, F, 64);
 * Wc = cluster_weights (F, 64); df (num_clouds n_points, F)).  One pass: the rows' operand resident in registers, the cloud's right
 * operand packed into `packed` (epc_vlad_df_packed_bytes, caller-owned scratch) and streamed through LDS.  pieces: 2 = two bf16 pieces
 * per operand, three products (the backward arithmetic of epc_gemm_f32_fast); 1 = one bf16 value per operand. */
size_t epc_vlad_df_packed_bytes(int num_clouds, int F);
int epc_vlad_df(const float* a, const float* dz, const float* dvlad, const float* Wc, int num_clouds, int n_points, int F, int pieces,
                void* packed, size_t packed_bytes, float* df, void* stream);
/* epc_vlad_df for conv5's features (F = 1024) continued through the backward of f = l2_normalize(relu(batch_norm(z5)))
 * (models/epc-net.py:136-148) while the product is in registers: writes du (rows, 1024), the gradient of the BatchNorm OUTPUT
 * ([f > 0] rn (df - f trow)), instead of df, and dbeta_dgamma (2 x 1024: sum du, sum du zhat) -- what epc_bn_relu_rownorm_bwd's
 * first pass over (df, z5) computes.  trow: the rows' dot products df . f (epc_assign_softmax_bwd's rowdot); rn: the forward's
 * reciprocal row norms; mean .. beta, eps: conv5's BatchNorm.  Finish with epc_bn_apply_bwd_given(du, z5, ...) (in place).
 * n_points a multiple of 32; partials: epc_vlad_df_tail_partial_floats floats of scratch. */
size_t epc_vlad_df_tail_partial_floats(int num_clouds, int n_points);
int epc_vlad_df_tail(const float* a, const float* dz, const float* dvlad, const float* Wc, int num_clouds, int n_points, int pieces,
                     void* packed, size_t packed_bytes, const float* z5, const float* rn, const float* trow, const float* mean,
                     const float* var, const float* gamma, const float* beta, float eps, float* du, float* dbeta_dgamma,
                     float* partials, size_t partial_floats, void* stream);
/* dz = gamma rstd (dy - dbeta / rows - zhat dgamma / rows): the last step of a training-mode BatchNorm backward whose column sums
 * are known (utils/tf_util.py:454-519 seen from the gradient side); dy carries its ReLU mask already; dz may be dy. */
int epc_bn_apply_bwd_given(const float* dy, const float* z, const float* mean, const float* var, const float* gamma, const float* beta,
                           const float* dbeta, const float* dgamma, float eps, int rows, int C, float* dz, void* stream);

/* ---- The head of the training step on bf16-stored tensors (csrc/train_head16.hip) -------------------------------------------------
 * conv5 + per-point l2 norm + VLAD soft assignment + aggregation (models/epc-net.py:136-148, loupe.py:249-291 in training mode),
 * forward and backward, for params["TRAIN_PRECISION"] = "bf16" (BASELINE.json configs[2]): the (rows, 1024) tensors -- conv5's
 * pre-activation z5, the gradient du of its BatchNorm output, dz5 -- are bf16 in HBM (`void*` below: rows x 1024 x 2 bytes, row-major,
 * 16-byte aligned); accumulators, batch statistics, column sums, rn and every (rows, 64) tensor are f32; every product rounds each
 * operand to ONE bf16 value.  The feature map f = l2_normalize(relu(batch_norm(z5))) is never written: BatchNorm + ReLU are applied to
 * z5 as it is loaded (mean5 .. beta5, eps = conv5's BatchNorm with its BATCH moments), the row factor rn in the epilogue or on the
 * other operand.  rows = num_clouds * n_points, n_points a multiple of 32, rows * 1024 < 2^32.  scratch: caller-owned, 16-byte aligned,
 * of the size the matching *_scratch_bytes function returns; deterministic (fixed summation orders, no atomics).
 *
 * epc_h16_conv5_fwd: z5 = bf16(cat W5 + b5) (cat: (rows, 256) f32, or bf16 when cat_is_bf16; W5 (256, 1024), b5 (1024) f32) and the batch
 *   mean / population variance of conv5 (utils/tf_util.py:472-476) from the f32 accumulators.
 * epc_h16_assign: out (rows, 64) = rn (u B), u = relu(bn(z5)), rn = rsqrt(max(sum_c u^2, 1e-12)) (tf.nn.l2_normalize, :147-148).
 *   per_cloud_operand = 0: B = cluster_weights (1024, 64): the assignment's logits (loupe.py:255); rn_out (rows) and the batch moments
 *   mean_out / var_out (64 each; slim.batch_norm's, :257-263) are written when not NULL.  per_cloud_operand = 1: B = dvlad
 *   (num_clouds, 1024, 64): da = f dvlad[cloud], the assignment's gradient through the aggregation.
 * epc_h16_colgemm: out = u^T (rn C), C (rows, 64) f32.  per_cloud = 1: out (num_clouds, 1024, 64) = f[b]^T C[b] -- the aggregation
 *   vlad = f^T a (loupe.py:286-291); per_cloud = 0: out (1024, 64) over all rows -- dWc = f^T dz.
 * epc_h16_df_tail: epc_vlad_df_tail on these tensors: du (rows, 1024) bf16 = [f > 0] rn ([a | dz] [dvlad^T ; Wc^T] - f trow) and
 *   dbeta_dgamma (2, 1024) = (sum du, sum du zhat) from the f32 values.
 * epc_h16_bn_bwd_apply: dz5 = gamma rstd (du - dbeta / rows - zhat dgamma / rows), bf16 in, bf16 out; dz5 may be du.
 * epc_h16_conv5_dx: dcat (rows, 256) f32 = dz5 W5^T -- the product alone: the step uses the fused form below; this one stays as the second
 *   implementation the fused form is tested against (tests/test_gpu_head_stream.py) and timed beside (scripts/time_head_kernels.py).
 * epc_h16_conv5_dx_bn: epc_h16_bn_bwd_apply and epc_h16_conv5_dx in one pass -- dz5 is formed from du and z5 as they stream, written once
 *   (dz5 may be du) for epc_h16_conv5_dw, and multiplied with W5^T from registers; the same values, one read of a (rows, 1024) tensor
 *   and one launch fewer (utils/tf_util.py:94-106 seen from the gradient side, models/epc-net.py:136).
 * epc_h16_conv5_dw: dW5 (256, 1024) f32 = cat^T dz5 (cat f32 or bf16), row slices added in a fixed order.
 * epc_h16_expand: y (rows, 1024) f32 = the bf16 values (mean5 NULL) or relu(bn(z5)) rn (rn NULL: no row factor): the feature map
 *   for callers that need it materialised (the distillation variants' second output, models/kd_epc-net.py:158) and for tests. */
size_t epc_h16_conv5_fwd_scratch_bytes(int rows);
int epc_h16_conv5_fwd(const void* cat, int cat_is_bf16, const float* W5, const float* b5, int rows, void* z5, float* mean, float* var,
                      void* scratch, size_t scratch_bytes, void* stream);
size_t epc_h16_assign_scratch_bytes(int num_clouds, int n_points, int per_cloud_operand);
int epc_h16_assign(const void* z5, const float* mean5, const float* var5, const float* gamma5, const float* beta5, float eps,
                   const float* B, int per_cloud_operand, int num_clouds, int n_points, float* out, float* rn_out, float* mean_out,
                   float* var_out, void* scratch, size_t scratch_bytes, void* stream);
size_t epc_h16_colgemm_scratch_bytes(int num_clouds, int n_points);
int epc_h16_colgemm(const void* z5, const float* mean5, const float* var5, const float* gamma5, const float* beta5, float eps,
                    const float* C, const float* rn, int num_clouds, int n_points, int per_cloud, float* out, void* scratch,
                    size_t scratch_bytes, void* stream);
size_t epc_h16_df_tail_scratch_bytes(int num_clouds, int n_points);
int epc_h16_df_tail(const float* a, const float* dz, const float* dvlad, const float* Wc, int num_clouds, int n_points, const void* z5,
                    const float* rn, const float* trow, const float* mean5, const float* var5, const float* gamma5, const float* beta5,
                    float eps, void* du, float* dbeta_dgamma, void* scratch, size_t scratch_bytes, void* stream);
int epc_h16_bn_bwd_apply(const void* du, const void* z5, const float* mean5, const float* var5, const float* gamma5, const float* beta5,
                         float eps, const float* dbeta, const float* dgamma, int rows, void* dz5, void* stream);
size_t epc_h16_dx_scratch_bytes(void);
int epc_h16_conv5_dx(const void* dz5, const float* W5, int rows, float* dcat, void* scratch, size_t scratch_bytes, void* stream);
int epc_h16_conv5_dx_bn(const void* du, const void* z5, const float* mean5, const float* var5, const float* gamma5, float eps,
                        const float* dbeta, const float* dgamma, const float* W5, int rows, void* dz5, float* dcat, void* scratch,
                        size_t scratch_bytes, void* stream);
size_t epc_h16_conv5_dw_scratch_bytes(int rows);
int epc_h16_conv5_dw(const void* cat, int cat_is_bf16, const void* dz5, int rows, float* dW5, void* scratch, size_t scratch_bytes,
                     void* stream);
int epc_h16_expand(const void* z, const float* mean5, const float* var5, const float* gamma5, const float* beta5, float eps,
                   const float* rn, int rows, float* y, void* stream);
/* ---- The same head on f32 tensors in the f32-accurate arithmetic (csrc/train_head32.hip): the default training step ---------------
 * The streaming kernels above with f32 rows (z5, du, dz5 are `float*`): one pass over the (rows, 1024) tensor per product, no feature map.
 * Arithmetic: epc_h32_conv5_fwd scaled split-fp16 (every row of cat and every column of W5 brought into [2^14, 2^15) by a power of two,
 * hi + lo fp16, three products: 2^-21 per product, no range restriction); epc_h32_assign, epc_h32_colgemm and epc_h32_conv5_dx two bf16
 * pieces per operand, three products (epc_gemm_f32_fast's: 2^-16 per product).  Arguments as the bf16 head's entry points of the same name.  The rest of the head's backward on f32 tensors:
 * epc_vlad_df_tail, epc_bn_apply_bwd_given; dW5 = cat^T dz5 by epc_h32_conv5_dw (round 6: epc_h16_conv5_dw's shape on f32 rows, two bf16
 * pieces per operand; the split-K tile product epc_gemm_splitk_det stays the second implementation it is tested against). */
size_t epc_h32_conv5_fwd_scratch_bytes(int rows);
int epc_h32_conv5_fwd(const float* cat, const float* W5, const float* b5, int rows, float* z5, float* mean, float* var, void* scratch,
                      size_t scratch_bytes, void* stream);
size_t epc_h32_assign_scratch_bytes(int num_clouds, int n_points, int per_cloud_operand);
int epc_h32_assign(const float* z5, const float* mean5, const float* var5, const float* gamma5, const float* beta5, float eps,
                   const float* B, int per_cloud_operand, int num_clouds, int n_points, float* out, float* rn_out, float* mean_out,
                   float* var_out, void* scratch, size_t scratch_bytes, void* stream);
size_t epc_h32_colgemm_scratch_bytes(int num_clouds, int n_points);
int epc_h32_colgemm(const float* z5, const float* mean5, const float* var5, const float* gamma5, const float* beta5, float eps,
                    const float* C, const float* rn, int num_clouds, int n_points, int per_cloud, float* out, void* scratch,
                    size_t scratch_bytes, void* stream);
size_t epc_h32_dx_scratch_bytes(void);
int epc_h32_conv5_dx(const float* dz5, const float* W5, int rows, float* dcat, void* scratch, size_t scratch_bytes, void* stream);
/* epc_bn_apply_bwd_given and epc_h32_conv5_dx in one pass (epc_h16_conv5_dx_bn on f32 tensors) */
int epc_h32_conv5_dx_bn(const float* du, const float* z5, const float* mean5, const float* var5, const float* gamma5, float eps,
                        const float* dbeta, const float* dgamma, const float* W5, int rows, float* dz5, float* dcat, void* scratch,
                        size_t scratch_bytes, void* stream);
/* dW5 (256, 1024) f32 = cat^T dz5: cat (rows, 256), dz5 (rows, 1024) f32; rows any positive count (rows * 1024 < 2^32); row slices added
 * in a fixed order (the same bits every run).  Replaces on this path: models/epc-net.py:136's weight gradient. */
size_t epc_h32_conv5_dw_scratch_bytes(int rows);
int epc_h32_conv5_dw(const float* cat, const float* dz5, int rows, float* dW5, void* scratch, size_t scratch_bytes, void* stream);
/* epc_gemm_splitk_det with the RIGHT operand stored as bf16 (strides and batch stride in elements; every side of the product at least
 * 64, K at least 32).  pieces: 1 = A rounded to one bf16 value, 2 = A in two bf16 pieces (the bf16 operand is exact either way). */
int epc_gemm_splitk_det_b16(const float* A, const void* B16, float* C, const float* bias, int M, int N, int K, long sAm, long sAk,
                            long sBk, long sBn, int ldc, int batch, long bA, long bB, long bC, int splitk, int accumulate, int pieces,
                            float* workspace, size_t workspace_floats, void* stream);

/* cluster_weights2's gradient (loupe.py:284,292; the part of the VLAD normalisation's backward that crosses clouds):
 * dw2 (F, 64) = - sum over the clouds, in ascending order, of draw (num_clouds, F, 64) x a_sum (num_clouds, 64); 16-byte aligned. */
int epc_vlad_w2_grad(const float* draw, const float* a_sum, int num_clouds, int F, int C, float* dw2, void* stream);
/* The sum over the G group rows behind G_VLAD's shared hidden projection (loupe.py:326-328): y (rows_out, O) = sum_g x (rows_out G, O);
 * bwd: dx (rows_out G, O) = dy (rows_out, O) repeated over the group. */
int epc_group_sum_fwd(const float* x, int rows_out, int G, int O, float* y, void* stream);
int epc_group_sum_bwd(const float* dy, int rows_out, int G, int O, float* dx, void* stream);

/* EPC-Net-L's global max-pool over a cloud's points in training mode (models/epc-net-l.py:88-92; utils/tf_util.py:349-372 with the
 * kernel covering all N points): out (num_clouds, C) = max over n of x (num_clouds, n, C), arg = the row holding it (the first on ties --
 * where tf.nn.max_pool's gradient goes); bwd: dx = dy at that row, zero elsewhere (dx is cleared here).  NaN propagates. */
int epc_maxpool_points_fwd(const float* x, int num_clouds, int n, int C, float* out, int32_t* arg, void* stream);
int epc_maxpool_points_bwd(const float* dy, const int32_t* arg, int num_clouds, int n, int C, float* dx, void* stream);

/* Distillation terms of kd_train.py:330-340, 376-383 (square_error_sum / square_error_mean between the student's and the
 * teacher's soft labels or point features): loss[0] = sum (a - b)^2 (mean != 0: divided by n), one read of both tensors, partials
 * added in a fixed order (bit-reproducible).  bwd: da = 2 (a - b) dloss[0] (/ n); b -- the teacher's output, fed through a
 * placeholder in the reference -- has no gradient.  a, b, da 16-byte aligned; partials: epc_sq_err_partial_floats(n) floats. */
size_t epc_sq_err_partial_floats(long n);
int epc_sq_err_fwd(const float* a, const float* b, long n, int mean, float* loss, float* partials, size_t partial_floats,
                   void* stream);
int epc_sq_err_bwd(const float* a, const float* b, long n, int mean, const float* dloss, float* da, void* stream);

/* tf.train.AdamOptimizer.apply (train.py:273): t = 1-based step count for the bias correction. */
int epc_adam_step(float* w, float* m, float* v, const float* g, long n, float lr, float beta1, float beta2, float eps,
                  int t, void* stream);
/* The same update with lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t) read from device memory (one float), so that a captured
 * HIP graph of the whole training step can be replayed with new schedule values. */
int epc_adam_step_dev(float* w, float* m, float* v, const float* g, long n, const float* lr_t_dev, float beta1,
                      float beta2, float eps, void* stream);

/* BatchNorm moving-average update (utils/tf_util.py:474-487, slim assign_moving_average):
 * shadow -= (1 - decay) * (shadow - value); decay is read from decay_dev (device, one float) when that is not NULL. */
int epc_ema_update(float* shadow, const float* value, long n, float decay, const float* decay_dev, void* stream);

/* Many small tensors, one launch (host arrays of `count` device pointers / element counts; the table travels in the
 * kernel arguments, so the launch is HIP-graph capturable): epc_adam_step(_dev) over all trainables of a step, and
 * epc_ema_update over all BatchNorm statistics (scheduled[k] != 0: statistic k follows sched_decay -- read from
 * sched_decay_dev when not NULL --, otherwise fixed_decay). */
int epc_adam_multi(int count, float* const* w, float* const* m, float* const* v, const float* const* g, const long* n,
                   float lr, float beta1, float beta2, float eps, int t, const float* lr_t_dev, void* stream);
int epc_ema_multi(int count, float* const* shadow, const float* const* value, const long* n, const int* scheduled,
                  float fixed_decay, float sched_decay, const float* sched_decay_dev, void* stream);

/* Offsets (in bytes) of the per-stage sub-buffers inside the packed weight buffer, for the stage entry
 * points above.  stage: 0 conv1, 1..4 block b, 5 conv5(+assign), 6 head. */
size_t epc_net_packed_offset(const epc_cfg* cfg, int stage);

#ifdef __cplusplus
}
#endif
#endif /* EPCNET_H */
