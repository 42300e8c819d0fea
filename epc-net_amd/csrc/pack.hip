// Inference weight preparation: fold every BatchNorm into its producer (tf.nn.batch_normalization:
// y = x*inv + (beta - mean*inv), inv = rsqrt(var + 1e-3)*gamma; utils/tf_util.py:490, slim default epsilon) and
// lay the matrices out in MFMA operand order (common.h).  Runs once per weight load on the device.
#include <string.h>
#include <string>
#include "common.h"

#define BN_EPS 1e-3f

static inline size_t align64(size_t v) { return (v + 63) / 64 * 64; }

struct StageLayout {
    size_t off[8];  // float offsets of stages 0..6, off[7] = total
    size_t guard;   // 64 floats after the stages: word 0 = max |packed 16-bit operand| bit pattern (fp16 range guard)
};

// EPC-Net in EPC_PRECISION_FAST packs fp16 operands (scaled by W5_SCALE); everything else packs split-bf16 operands
static bool cfg_fast(const epc_cfg* c) { return c->arch == EPC_ARCH_EPC_NET && c->precision == EPC_PRECISION_FAST; }

static bool cfg_ok(const epc_cfg* c) {
    if (!c) return false;
    if (c->arch != EPC_ARCH_EPC_NET && c->arch != EPC_ARCH_EPC_NET_L) return false;
    if (c->num_points < 32 || c->num_points % 32) return false;
    if (c->input_dim != 3 || c->output_dim != 256 || c->knn <= 0) return false;
    if (c->precision != EPC_PRECISION_F32 && c->precision != EPC_PRECISION_FAST) return false;
    if (c->arch == EPC_ARCH_EPC_NET) {
        if (c->cluster_size != 64) return false;
        if (!(c->groups == 1 || c->groups == 2 || c->groups == 4 || c->groups == 8 || c->groups == 16)) return false;
    }
    return true;
}

static StageLayout stage_layout(const epc_cfg* c) {
    StageLayout L;
    size_t o = 0;
    const int nblocks = c->arch == EPC_ARCH_EPC_NET ? 4 : 2;
    L.off[0] = o;
    o += align64(EPC_CONV1_PACK_FLOATS);
    for (int b = 1; b <= 4; ++b) {
        L.off[b] = o;
        if (b <= nblocks) o += align64(EPC_BLOCK_PACK_FLOATS_S);
    }
    L.off[5] = o;
    if (c->arch == EPC_ARCH_EPC_NET) {
        o += align64((size_t)256 * 1024 + 1024 + 1024 * 64 + 128 + 1024);   // W5, b5, Wc, cluster_bn, inverse column scales
        L.off[6] = o;
        const size_t kh = 65536 / c->groups;
        o += align64((size_t)65536 + kh * 256 + 512 + 65536 + 512);
    } else {
        o += align64((size_t)128 * 1024 + 1024 + 1024);                    // W5, b5, inverse column scales
        L.off[6] = o;
        o += align64((size_t)1024 * 256 + 256);
    }
    L.guard = o;
    o += 64;
    L.off[7] = o;
    return L;
}

extern "C" size_t epc_net_packed_bytes(const epc_cfg* cfg) {
    if (!cfg_ok(cfg)) return 0;
    return stage_layout(cfg).off[7] * sizeof(float);
}

extern "C" size_t epc_net_packed_offset(const epc_cfg* cfg, int stage) {
    if (!cfg_ok(cfg) || stage < 0 || stage > 6) return (size_t)-1;
    return stage_layout(cfg).off[stage] * sizeof(float);
}

// ---- device kernels ------------------------------------------------------------------------------------------
__device__ __forceinline__ float bn_inv(const float* gamma, const float* var, int c) {
    return (1.0f / sqrtf(var[c] + BN_EPS)) * gamma[c];
}

// fp16 range guard of EPC_PRECISION_FAST: the largest |value| that is about to be rounded to fp16, as a bit pattern
// (|f32| orders like its bit pattern; NaN sorts above Inf), one atomic per wave that holds a new maximum candidate.
__device__ __forceinline__ void guard_track(unsigned int* guard, float v) {
    const unsigned int bits = __float_as_uint(v) & 0x7fffffffu;
    if (bits > 0x47000000u) atomicMax(guard, bits);   // only values above 32768 can matter: no atomics in the normal case
}

// Row-major [cin][cout] with folded BN (conv1 3x64, fc1 1024x256).
__global__ void fold_rowmajor_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                     const float* __restrict__ gamma, const float* __restrict__ beta,
                                     const float* __restrict__ mean, const float* __restrict__ var, int cin,
                                     int cout, float* __restrict__ dstW, float* __restrict__ dstB) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o < cin * cout) dstW[o] = W[o] * bn_inv(gamma, var, o % cout);
    if (o < cout) {
        const float inv = bn_inv(gamma, var, o);
        dstB[o] = b[o] * inv + (beta[o] - mean[o] * inv);
    }
}

// Inverse column scales of the scaled split-fp16 form (common.h): tinv[c] = 2^(E - 14), E = floor(log2(max_k |W'[k][c]|)),
// W' = W * bn_inv -- so that W' / tinv lies in [2^14, 2^15) at its largest.  One thread per output channel.
__global__ void colscale_kernel(const float* __restrict__ W, const float* __restrict__ gamma, const float* __restrict__ var,
                                int cin, int cout, float* __restrict__ tinv) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= cout) return;
    const float inv = bn_inv(gamma, var, c);
    float m = 0.f;
    for (int k = 0; k < cin; ++k) m = fmaxf(m, fabsf(W[(size_t)k * cout + c] * inv));
    float s, is;
    row_scale_pow2(m, s, is);
    tinv[c] = is;
}

// conv5 weights as fragments (layout + arithmetic: common.h, conv5_vlad.hip C5Lds).  One thread per (chunk, k, channel).
// f16 = 1 (EPC-Net: conv5 feeds the VLAD aggregation): per chunk [fp16 hi of W * W5_SCALE: k-step s (16 k), lane, 8]
//         then the MX fp6 lo fragments written by pack_conv5_lo6_kernel; bias scaled by W5_SCALE.
// f16 = 0 (f32-equivalent arithmetic, conv5_f32.hip): fp16 hi and lo fragments of W' / tinv[col] for the 16x16x32 MFMA
//         (scaled split-fp16 form, common.h); bias un-scaled.
__global__ void fold_pack_conv5_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                       const float* __restrict__ mean, const float* __restrict__ var, int cin, int f16,
                                       unsigned short* __restrict__ dstW, float* __restrict__ dstB,
                                       unsigned int* __restrict__ guard, const float* __restrict__ tinv) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    const float scale = f16 ? W5_SCALE : 1.0f;
    if (o < cin * 1024) {
        const int j = o & 7, lane = (o >> 3) & 63, rest = o >> 9;
        if (f16) {
            // 32x32x16 fragments: k-step s = 16 input channels, lane l = (column l & 31, k-octet l >> 5)
            const int steps = cin / 16;
            const int s = rest % steps, c = rest / steps;
            const int k = 16 * s + 8 * (lane >> 5) + j, col = 32 * c + (lane & 31);
            const float w = W[(size_t)k * 1024 + col] * bn_inv(gamma, var, col) * scale;
            const _Float16 h = (_Float16)w;
            guard_track(guard, w);
            const size_t chunk_halfs = (size_t)48 * cin;                       // 96*cin bytes per chunk
            dstW[(size_t)c * chunk_halfs + (size_t)s * 512 + lane * 8 + j] = __builtin_bit_cast(unsigned short, h);
        } else {
            // 16x16x32 fragments (conv5_f32.hip): [chunk c][channel group g (16)][k-step s (32 input channels)][part][lane][8],
            // lane l = (column l & 15, k-octet l >> 4); a (chunk, group) is contiguous: one LDS stage of the max-pool kernel
            const int steps = cin / 32;
            const int s = rest % steps, g = (rest / steps) & 1, c = rest / (2 * steps);
            const int k = 32 * s + 8 * (lane >> 4) + j, col = 32 * c + 16 * g + (lane & 15);
            const float w = W[(size_t)k * 1024 + col] * bn_inv(gamma, var, col);
            const float ws = w * (1.0f / tinv[col]);   // exact: a power of two
            const _Float16 hh = (_Float16)ws;
            const _Float16 ll = (_Float16)(ws - (float)hh);
            const size_t base = ((size_t)((c * 2 + g) * steps + s) * 2) * 512 + lane * 8 + j;
            dstW[base] = __builtin_bit_cast(unsigned short, hh);
            dstW[base + 512] = __builtin_bit_cast(unsigned short, ll);
        }
    }
    if (o < 1024) {
        const float inv = bn_inv(gamma, var, o);
        dstB[o] = (b[o] * inv + (beta[o] - mean[o] * inv)) * scale;
        if (f16) guard_track(guard, dstB[o] * (1.0f / W5_SCALE));   // the bias enters an fp16 activation un-scaled
    }
}

// lo parts of the EPC-Net conv5 weights as MX fp6 (common.h): one thread per (chunk c, 64-wide k-step ks, lane l) = the 32
// weights W[64ks + 32(l>>5) + j][32c + (l&31)], j < 32, that lane l feeds to the lo-term MFMA of that k-step.  Inside a
// chunk (byte offsets from the start of the lo area, 64*cin bytes into the chunk):
//   k-step ks: [lane][16 B: dwords 0-3] at 1536 ks, [lane][8 B: dwords 4-5] at 1536 ks + 1024;
//   block scales: [lane][4 B: byte ks = 127 + e] at 1536 * (cin / 64).
__global__ void pack_conv5_lo6_kernel(const float* __restrict__ W, const float* __restrict__ gamma,
                                      const float* __restrict__ var, int cin, unsigned short* __restrict__ dstW) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    const int nks = cin / 64;
    if (o >= 32 * nks * 64) return;
    const int lane = o & 63, ks = (o >> 6) % nks, c = (o >> 6) / nks;
    const int col = 32 * c + (lane & 31), k0 = 64 * ks + 32 * (lane >> 5);
    const float inv = bn_inv(gamma, var, col) * W5_SCALE;
    float lo[32];
    float m = 0.f;
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        const float w = W[(size_t)(k0 + j) * 1024 + col] * inv;   // the same expression fold_pack_conv5_kernel rounds to hi
        lo[j] = w - (float)(_Float16)w;
        m = fmaxf(m, fabsf(lo[j]));
    }
    const int e = fp6_block_exponent(m, -100, 20);
    const float down = exp2i(-e);
    f16x32 q;
#pragma unroll
    for (int j = 0; j < 32; ++j) q[j] = (_Float16)(lo[j] * down);   // |q| <= 7.5: exact scaling, then fp16 -> e2m3 below
    const i32x6 bits = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(q, 1.0f);
    char* chunk = reinterpret_cast<char*>(dstW) + (size_t)c * 96 * cin + (size_t)64 * cin;   // past the 64*cin bytes of hi fragments
    int* p0 = reinterpret_cast<int*>(chunk + 1536 * ks + 16 * lane);
    int* p1 = reinterpret_cast<int*>(chunk + 1536 * ks + 1024 + 8 * lane);
    p0[0] = bits[0], p0[1] = bits[1], p0[2] = bits[2], p0[3] = bits[3];
    p1[0] = bits[4], p1[1] = bits[5];
    reinterpret_cast<unsigned char*>(chunk + 1536 * nks + 4 * lane)[ks] = (unsigned char)(127 + e);
}

// 64x64 block layer as hi + lo fragments: Wp[tile t (2)][k-step s (4)][part (hi,lo)][lane][8 x 16 bit] + bias.
// f16 = 0: fp16 parts of W' / tinv[col] for the scaled split-fp16 form (f32-equivalent arithmetic, common.h), bias un-scaled;
// f16 = 1: fp16 parts of W * W5_SCALE, bias scaled alike (EPC-Net fast: fp16 activations, two MFMAs per product).
// mode PACK_SPLIT: k-step s covers input channels 16s + 8h + q (B operand read from a staged [pt][ch] row);
// mode PACK_ACC  : k-step s = 2*tin + s' covers 32tin + 16s' + 8(q>>2) + 4h + (q&3) (B = accumulators of the previous layer).
__global__ void fold_pack_block_kernel(const float* __restrict__ W, const float* __restrict__ b,
                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                       const float* __restrict__ mean, const float* __restrict__ var, int mode, int f16,
                                       unsigned short* __restrict__ dstW, float* __restrict__ dstB,
                                       unsigned int* __restrict__ guard, const float* __restrict__ tinv) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    const float scale = f16 ? W5_SCALE : 1.0f;
    if (o < 4096) {
        const int q = o & 7, lane = (o >> 3) & 63, st = (o >> 9) & 3, t = o >> 11;
        const int h = lane >> 5;
        const int k = mode == PACK_SPLIT ? 16 * st + 8 * h + q : 32 * (st >> 1) + 16 * (st & 1) + 8 * (q >> 2) + 4 * h + (q & 3);
        const int col = 32 * t + (lane & 31);
        const float w = W[(size_t)k * 64 + col] * bn_inv(gamma, var, col) * scale;
        unsigned short hi, lo;
        if (f16) {
            guard_track(guard, w);
            const _Float16 hh = (_Float16)w;
            const _Float16 ll = (_Float16)(w - (float)hh);
            hi = __builtin_bit_cast(unsigned short, hh);
            lo = __builtin_bit_cast(unsigned short, ll);
        } else {
            const float ws = w * (1.0f / tinv[col]);   // exact: a power of two
            const _Float16 hh = (_Float16)ws;
            const _Float16 ll = (_Float16)(ws - (float)hh);
            hi = __builtin_bit_cast(unsigned short, hh);
            lo = __builtin_bit_cast(unsigned short, ll);
        }
        const size_t base = ((size_t)(t * 4 + st) * 2) * 512 + lane * 8 + q;
        dstW[base] = hi;
        dstW[base + 512] = lo;
    }
    if (o < 64) {
        const float inv = bn_inv(gamma, var, o);
        dstB[o] = (b[o] * inv + (beta[o] - mean[o] * inv)) * scale;
        if (f16) guard_track(guard, dstB[o] * (1.0f / W5_SCALE));
    }
}

// cluster weights for the assignment GEMM: ONE fp16 per weight (x W5_SCALE), fragment order
// [chunk c][k-step sp][cluster tile t][lane][8]: element j of lane l = Wc[32c + 16sp + 8(j>>2) + 4(l>>5) + (j&3)][32t + (l&31)]
__global__ void pack_wc_f16_kernel(const float* __restrict__ Wc, unsigned short* __restrict__ dst,
                                   unsigned int* __restrict__ guard) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= 1024 * 64) return;
    const int j = o & 7, lane = (o >> 3) & 63, t = (o >> 9) & 1, sp = (o >> 10) & 1, c = o >> 11;
    const int ch = 32 * c + 16 * sp + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
    const float w = Wc[(size_t)ch * 64 + 32 * t + (lane & 31)] * W5_SCALE;
    guard_track(guard, w);
    const _Float16 h = (_Float16)w;
    dst[o] = __builtin_bit_cast(unsigned short, h);
}

// EPC_PRECISION_F32: bf16 hi + lo, un-scaled, as A fragments of v_mfma_f32_16x16x32_bf16 (conv5_f32.hip):
// [chunk c][cluster group cg (4 x 16)][part (hi, lo)][lane][8]: element j of lane l (q = l >> 4) =
// Wc[32c + 16(j >> 2) + 4q + (j & 3)][16 cg + (l & 15)] -- the k order in which a lane's conv5 accumulators
// (acc[g = j >> 2][.][r = j & 3] = channel 16 g + 4 q + r) form the B operand.
__global__ void pack_wc_bf16x2_kernel(const float* __restrict__ Wc, unsigned short* __restrict__ dst) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= 1024 * 64) return;
    const int j = o & 7, lane = (o >> 3) & 63, cg = (o >> 9) & 3, c = o >> 11;
    const int ch = 32 * c + 16 * (j >> 2) + 4 * (lane >> 4) + (j & 3);
    const float w = Wc[(size_t)ch * 64 + 16 * cg + (lane & 15)];
    const unsigned short hi = bf16_bits_rne(w);
    const unsigned short lo = bf16_bits_rne(w - bf16_bits_to_float(hi));
    const size_t base = ((size_t)(c * 4 + cg) * 2) * 512 + lane * 8 + j;
    dst[base] = hi;
    dst[base + 512] = lo;
}

__global__ void bn_affine_kernel(const float* __restrict__ gamma, const float* __restrict__ beta,
                                 const float* __restrict__ mean, const float* __restrict__ var, int n,
                                 float* __restrict__ s, float* __restrict__ t) {
    const int o = blockIdx.x * 256 + threadIdx.x;
    if (o >= n) return;
    const float inv = bn_inv(gamma, var, o);
    s[o] = inv;
    t[o] = beta[o] - mean[o] * inv;
}

// ---- host side ---------------------------------------------------------------------------------------------
struct NameTable {
    const char* const* names;
    const float* const* tensors;
    int n;
    const float* find(const std::string& name) const {
        for (int i = 0; i < n; ++i)
            if (name == names[i]) return tensors[i];
        return nullptr;
    }
    // EMA shadows embed the caller's outer scope in the middle of the name
    // ("<scope>/bn/<outer>/<scope>/bn/moments/Squeeze/ExponentialMovingAverage", utils/tf_util.py:474-487).
    const float* find_ema(const std::string& scope, bool variance) const {
        const std::string prefix = scope + "/bn/";
        const std::string suffix = variance ? "/bn/moments/Squeeze_1/ExponentialMovingAverage"
                                            : "/bn/moments/Squeeze/ExponentialMovingAverage";
        for (int i = 0; i < n; ++i) {
            const std::string s(names[i]);
            if (s.size() > prefix.size() + suffix.size() && s.compare(0, prefix.size(), prefix) == 0 &&
                s.compare(s.size() - suffix.size(), suffix.size(), suffix) == 0)
                return tensors[i];
        }
        return nullptr;
    }
};

struct ConvVars {
    const float *W, *b, *gamma, *beta, *mean, *var;
};

static int get_conv(const NameTable& T, const std::string& scope, ConvVars* v) {
    v->W = T.find(scope + "/weights");
    v->b = T.find(scope + "/biases");
    v->gamma = T.find(scope + "/bn/gamma");
    v->beta = T.find(scope + "/bn/beta");
    v->mean = T.find_ema(scope, false);
    v->var = T.find_ema(scope, true);
    if (!(v->W && v->b && v->gamma && v->beta && v->mean && v->var)) {
        epc_set_error("epc_net_pack_weights: variables of scope '%s' are incomplete", scope.c_str());
        return EPC_ENOTFOUND;
    }
    return EPC_OK;
}

static int get_slim_bn(const NameTable& T, const std::string& scope, const float** g, const float** b,
                       const float** m, const float** v) {
    *g = T.find(scope + "/gamma");
    *b = T.find(scope + "/beta");
    *m = T.find(scope + "/moving_mean");
    *v = T.find(scope + "/moving_variance");
    if (!(*g && *b && *m && *v)) {
        epc_set_error("epc_net_pack_weights: variables of scope '%s' are incomplete", scope.c_str());
        return EPC_ENOTFOUND;
    }
    return EPC_OK;
}

#define PACK_TRY(expr)              \
    do {                            \
        int rc__ = (expr);          \
        if (rc__ != EPC_OK) return rc__; \
    } while (0)

static int launch_layer(const ConvVars& v, int cin, int cout, int mode, int f16, float* dst, unsigned int* guard,
                        float* tinv, hipStream_t st) {
    EPC_CHECK_ARG(cin == 64 && cout == 64, "block layers are 64x64");
    if (!f16) {
        hipLaunchKernelGGL(colscale_kernel, dim3(1), dim3(256), 0, st, v.W, v.gamma, v.var, 64, 64, tinv);
        EPC_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(fold_pack_block_kernel, dim3(16), dim3(256), 0, st, v.W, v.b, v.gamma, v.beta, v.mean, v.var,
                       mode, f16, (unsigned short*)dst, dst + 4096, guard, tinv);
    EPC_CHECK_LAUNCH();
    return EPC_OK;
}

extern "C" int epc_net_pack_weights(const epc_cfg* cfg, const char* const* names, const float* const* tensors,
                                    int n, void* packed, size_t packed_bytes, void* stream) {
    EPC_CHECK_ARG(cfg_ok(cfg), "unsupported configuration");
    EPC_CHECK_ARG(names && tensors && packed && n > 0, "null pointer");
    if (packed_bytes < epc_net_packed_bytes(cfg)) {
        epc_set_error("epc_net_pack_weights: packed buffer too small");
        return EPC_ENOMEM;
    }
    const NameTable T{names, tensors, n};
    const StageLayout L = stage_layout(cfg);
    float* P = (float*)packed;
    hipStream_t st = (hipStream_t)stream;
    const int nblocks = cfg->arch == EPC_ARCH_EPC_NET ? 4 : 2;
    const int f16 = cfg_fast(cfg) ? 1 : 0;  // EPC-Net FAST: fp16 activations (common.h); otherwise split-bf16
    unsigned int* guard = (unsigned int*)(P + L.guard);
    {
        hipError_t e = hipMemsetAsync(guard, 0, 64 * sizeof(float), st);
        if (e != hipSuccess) {
            epc_set_error("epc_net_pack_weights: hipMemsetAsync: %s", hipGetErrorString(e));
            return EPC_EHIP;
        }
    }
    ConvVars v;

    PACK_TRY(get_conv(T, "fastdgcnn/conv1", &v));
    hipLaunchKernelGGL(fold_rowmajor_kernel, dim3(1), dim3(256), 0, st, v.W, v.b, v.gamma, v.beta, v.mean, v.var, 3,
                       64, P + L.off[0], P + L.off[0] + 192);
    EPC_CHECK_LAUNCH();

    for (int b = 1; b <= nblocks; ++b) {
        float* dst = P + L.off[b];
        const std::string base = "fastdgcnn/conv" + std::to_string(b);
        PACK_TRY(get_conv(T, base + "_a", &v));
        PACK_TRY(launch_layer(v, 64, 64, PACK_SPLIT, f16, dst, guard, dst + EPC_BLOCK_PACK_FLOATS, st));
        PACK_TRY(get_conv(T, base + "_b", &v));
        PACK_TRY(launch_layer(v, 64, 64, PACK_ACC, f16, dst + 4160, guard, dst + EPC_BLOCK_PACK_FLOATS + 64, st));
        if (b < nblocks) {
            PACK_TRY(get_conv(T, "fastdgcnn/conv" + std::to_string(b + 1), &v));
            PACK_TRY(launch_layer(v, 64, 64, PACK_SPLIT, f16, dst + 8320, guard, dst + EPC_BLOCK_PACK_FLOATS + 128, st));
        } else {
            hipError_t e = hipMemsetAsync(dst + 8320, 0, 4160 * sizeof(float), st);
            if (e == hipSuccess) e = hipMemsetAsync(dst + EPC_BLOCK_PACK_FLOATS + 128, 0, 64 * sizeof(float), st);
            if (e != hipSuccess) {
                epc_set_error("epc_net_pack_weights: hipMemsetAsync: %s", hipGetErrorString(e));
                return EPC_EHIP;
            }
        }
    }

    PACK_TRY(get_conv(T, "fastdgcnn/conv5", &v));
    const int c5in = 64 * nblocks;
    // inverse column scales of conv5 (scaled split-fp16 form): the last 1024 floats of the stage
    float* t5inv = P + L.off[6] - 1024;
    if (!f16) {
        hipLaunchKernelGGL(colscale_kernel, dim3(4), dim3(256), 0, st, v.W, v.gamma, v.var, c5in, 1024, t5inv);
        EPC_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(fold_pack_conv5_kernel, dim3(c5in * 1024 / 256), dim3(256), 0, st, v.W, v.b, v.gamma, v.beta,
                       v.mean, v.var, c5in, f16, (unsigned short*)(P + L.off[5]),
                       P + L.off[5] + (size_t)c5in * 1024, guard, t5inv);
    EPC_CHECK_LAUNCH();
    if (f16) {
        hipLaunchKernelGGL(pack_conv5_lo6_kernel, dim3((32 * (c5in / 64) * 64 + 255) / 256), dim3(256), 0, st, v.W, v.gamma,
                           v.var, c5in, (unsigned short*)(P + L.off[5]));
        EPC_CHECK_LAUNCH();
    }

    if (cfg->arch == EPC_ARCH_EPC_NET) {
        float* s5 = P + L.off[5] + (size_t)c5in * 1024 + 1024;
        const float* Wc = T.find("VLAD/cluster_weights");
        const float* C2 = T.find("VLAD/cluster_weights2");
        const float* H = T.find("VLAD/hidden1_weights");
        const float* Wg = T.find("VLAD/gating_weights");
        if (!(Wc && C2 && H && Wg)) {
            epc_set_error("epc_net_pack_weights: VLAD weight matrices are incomplete");
            return EPC_ENOTFOUND;
        }
        if (f16)
            hipLaunchKernelGGL(pack_wc_f16_kernel, dim3(256), dim3(256), 0, st, Wc, (unsigned short*)s5, guard);
        else
            hipLaunchKernelGGL(pack_wc_bf16x2_kernel, dim3(256), dim3(256), 0, st, Wc, (unsigned short*)s5);
        EPC_CHECK_LAUNCH();
        const float *g, *b, *m, *vv;
        PACK_TRY(get_slim_bn(T, "VLAD/cluster_bn", &g, &b, &m, &vv));
        float* cbn = s5 + (f16 ? 32768 : 65536);   // past the cluster-weight fragments (conv5_vlad.hip: gcbn)
        hipLaunchKernelGGL(bn_affine_kernel, dim3(1), dim3(256), 0, st, g, b, m, vv, 64, cbn, cbn + 64);
        EPC_CHECK_LAUNCH();

        float* h = P + L.off[6];
        const size_t kh = 65536 / cfg->groups;
        hipError_t e = hipMemcpyAsync(h, C2, 65536 * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(h + 65536, H, kh * 256 * sizeof(float), hipMemcpyDeviceToDevice, st);
        float* tail = h + 65536 + kh * 256;
        if (e == hipSuccess) e = hipMemcpyAsync(tail + 512, Wg, 65536 * sizeof(float), hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) {
            epc_set_error("epc_net_pack_weights: hipMemcpyAsync: %s", hipGetErrorString(e));
            return EPC_EHIP;
        }
        PACK_TRY(get_slim_bn(T, "VLAD/bn", &g, &b, &m, &vv));
        hipLaunchKernelGGL(bn_affine_kernel, dim3(1), dim3(256), 0, st, g, b, m, vv, 256, tail, tail + 256);
        EPC_CHECK_LAUNCH();
        PACK_TRY(get_slim_bn(T, "VLAD/gating_bn", &g, &b, &m, &vv));
        hipLaunchKernelGGL(bn_affine_kernel, dim3(1), dim3(256), 0, st, g, b, m, vv, 256, tail + 512 + 65536,
                           tail + 512 + 65536 + 256);
        EPC_CHECK_LAUNCH();
    } else {
        PACK_TRY(get_conv(T, "VLAD/fc1", &v));
        float* f = P + L.off[6];
        hipLaunchKernelGGL(fold_rowmajor_kernel, dim3(1024), dim3(256), 0, st, v.W, v.b, v.gamma, v.beta, v.mean,
                           v.var, 1024, 256, f, f + 1024 * 256);
        EPC_CHECK_LAUNCH();
    }
    if (f16) {
        // fp16 range guard: read the maximum back (this is why FAST packing synchronises the stream) and refuse to hand
        // out a buffer that holds an Inf.  65504 = the largest finite fp16; NaN / Inf patterns sort above it.
        unsigned int bits = 0;
        hipError_t e = hipMemcpyAsync(&bits, guard, sizeof(bits), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) {
            epc_set_error("epc_net_pack_weights: reading the range guard: %s", hipGetErrorString(e));
            return EPC_EHIP;
        }
        if (bits > 0x477fe000u) {
            float m;
            memcpy(&m, &bits, sizeof(m));
            epc_set_error("epc_net_pack_weights: a folded weight or bias leaves fp16's range in EPC_PRECISION_FAST "
                          "(max |W' * 256| = %g > 65504: small moving variance or large gamma); pack with EPC_PRECISION_F32", m);
            return EPC_ERANGE;
        }
    }
    return EPC_OK;
}
