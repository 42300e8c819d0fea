"""CPU numerics experiment: what narrower storage formats would cost in descriptor accuracy.

Everything is computed in float64 from the oracle's restatement, and ONE storage point at a time is rounded the way a kernel would
round it; the figure reported is the L2 distance between the descriptor with that rounding and the exact float64 descriptor, per
cloud of the adversarial set (tests/test_gpu_adversarial.py: 8 cloud kinds x {benign, mild, hard} weights).  So each number is the
error the FORMAT ALONE contributes, to be read against the 1e-4 bar and the 7.7e-5 the current arithmetic already uses of it.

  feat (conv5 output, written by conv5, read by the aggregate):
     f24   : the current format -- the f32 value rounded to its upper 24 bits (16 significant bits), 3 bytes per value
     se16  : shared exponent -- per (point, 32-channel chunk) one exponent, 16-bit unsigned mantissas (feat >= 0 after the ReLU):
             2 bytes + 1/32 byte per value
     se16x : the same with the chunk's values first multiplied by the point's inverse norm (no difference: one scale per row)
  gathered rows x (block inputs, 64 channels, read 20 times per point by the neighbour mean):
     f32   : the current format
     h16l8 : fp16 hi + 8-bit lo (lo = the residual in units of hi's ulp / 256): 3 bytes per value, ~19 significant bits
     sh16l8: the same behind a per-row power-of-two scale that brings the row's largest magnitude into [2^14, 2^15) (as the f16x3
             products scale their operands): no fp16 overflow on heavy-tailed weights; 3 bytes + 1/64 of a 4-byte scale per value
     b24   : the f32 value rounded to its upper 24 bits: 3 bytes, 16 significant bits
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import epcnet_oracle as O

N = int(os.environ.get("N", "4096"))
KINDS = ["uniform", "lidar", "repeat25", "repeat50", "repeat75", "zeropad25", "zeropad50", "zeros"]
LEVELS = {"benign": None, "mild": dict(seed=8, gamma_range=(0.3, 3.0), floor_frac=0.0),
          "hard": dict(seed=7, gamma_range=(0.1, 10.0), floor_frac=0.05)}


def round_sig(x, bits):
    """x (float64) rounded to `bits` significant bits (round to nearest)."""
    m, e = np.frexp(x)
    return np.ldexp(np.round(m * (1 << bits)) / (1 << bits), e)


def q_f24(x):
    return round_sig(x.astype(np.float32).astype(np.float64), 16)


def q_se16(x, chunk=32):
    rows, C = x.shape
    v = x.reshape(rows, C // chunk, chunk)
    mx = v.max(axis=2, keepdims=True)
    _, e = np.frexp(np.maximum(mx, 1e-300))            # mx < 2^e
    scale = np.ldexp(1.0, 16 - e)                      # mantissa = round(v * 2^(16 - e)) in [0, 65535]
    q = np.minimum(np.round(v * scale), 65535.0) / scale
    return q.reshape(rows, C)


def q_h16l8(x):
    hi = x.astype(np.float16).astype(np.float64)
    _, e = np.frexp(np.where(hi == 0, 1e-30, hi))
    ulp = np.ldexp(1.0, e - 11)                        # fp16 spacing at hi (normal range)
    lo = np.clip(np.round((x - hi) / ulp * 256.0), -128, 127) * ulp / 256.0
    return hi + lo


def q_sh16l8(x):
    mx = np.abs(x).max(axis=1, keepdims=True)
    _, e = np.frexp(np.maximum(mx, 1e-300))
    scale = np.ldexp(1.0, 15 - e)
    return q_h16l8(x * scale) / scale


def tail(st, x5, n_points, p, feat_q=None):
    """conv5 output (rows, 1024) -> descriptor; feat_q rounds what the aggregate reads (the assignment uses the exact rows, as the
    kernel computes it from its accumulators)."""
    w = st.w
    rn = 1.0 / np.maximum(np.sqrt((x5 * x5).sum(1, keepdims=True)), 1e-12)
    net = x5 * rn
    act = net @ w["VLAD/cluster_weights"]
    act = O.slim_batch_norm(st, act, "VLAD/cluster_bn", False, fused=False)
    act = np.exp(act - act.max(1, keepdims=True))
    act /= act.sum(1, keepdims=True)
    B = x5.shape[0] // n_points
    act3 = act.reshape(B, n_points, -1)
    feats = (feat_q(x5) if feat_q else x5) * rn
    vlad = np.matmul(np.transpose(act3, (0, 2, 1)), feats.reshape(B, n_points, -1))
    vlad = np.transpose(vlad, (0, 2, 1)) - act3.sum(1, keepdims=True) * w["VLAD/cluster_weights2"]
    vlad = O.l2_normalize(vlad, 1).reshape(B, -1)
    vlad = O.l2_normalize(vlad, 1)
    G = p["GROUPS"]
    vlad = vlad.reshape(-1, vlad.shape[1] // G) @ w["VLAD/hidden1_weights"]
    vlad = O.slim_batch_norm(st, vlad, "VLAD/bn", False, fused=True)
    vlad = vlad.reshape(B, G, -1).sum(1)
    vlad = O.context_gating(st, vlad, False, "VLAD")
    return O.l2_normalize(vlad, 1)


def forward64(pc, w, row_q=None, feat_q=None):
    """The oracle's forward (models/epc-net.py:62-155) in float64 with the two storage points exposed."""
    p = dict(O.DEFAULT_PARAMS)
    st = O.State(w, np.float64, "query_triplets")
    _, lists = O.knn_lists(pc.astype(np.float32))
    inp = pc.astype(np.float64)
    outs = []
    for b in range(1, 5):
        x = O.conv1d(st, inp, "fastdgcnn/conv%d" % b, False, None)
        xs = row_q(x.reshape(-1, 64)).reshape(x.shape) if row_q else x      # what the gather reads; the own row stays exact
        xm = O.neighbour_mean(xs, lists=lists, k=20)
        t = xm - x
        t = O.conv1d(st, t, "fastdgcnn/conv%d_a" % b, False, None)
        t = O.conv1d(st, t, "fastdgcnn/conv%d_b" % b, False, None)
        inp = t + xm
        outs.append(inp)
    x5 = O.conv1d(st, np.concatenate(outs, -1), "fastdgcnn/conv5", False, None).reshape(-1, 1024)
    return tail(st, x5, pc.shape[1], p, feat_q)


def main():
    pc = np.concatenate([O.synthetic_clouds(1, N, 30 + i, k) for i, k in enumerate(KINDS)], 0)
    print("descriptor error of ONE rounded storage point against the exact float64 descriptor; %d-point clouds: %s" % (N, " ".join(KINDS)))
    for level, kw in LEVELS.items():
        w = O.seeded_weights("epc-net", 11) if kw is None else O.adversarial_weights("epc-net", calibrate_on=O.synthetic_clouds(2, N, 4242), **kw)
        t0 = time.time()
        with np.errstate(all="ignore"):
            ref = forward64(pc, w)
            rows = {}
            rows["feat f24 (current)"] = forward64(pc, w, feat_q=q_f24)
            rows["feat se16"] = forward64(pc, w, feat_q=q_se16)
            rows["rows h16l8"] = forward64(pc, w, row_q=q_h16l8)
            rows["rows sh16l8"] = forward64(pc, w, row_q=q_sh16l8)
            rows["rows b24"] = forward64(pc, w, row_q=q_f24)
        for name, out in rows.items():
            e = np.linalg.norm(out - ref, axis=1)
            print("%-7s %-20s %s   max %.1e" % (level, name, " ".join("%.1e" % v for v in e), e.max()), flush=True)
        print("        (%.0f s)" % (time.time() - t0), flush=True)


if __name__ == "__main__":
    main()
