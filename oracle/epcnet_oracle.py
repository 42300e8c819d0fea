"""CPU ORACLE for the EPC-Net hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.
The shipped path (``epc-net_amd/``) never imports it and has no CPU fallback.

PARITY UNPINNED.  The reference is 100 % TensorFlow-1.12 graph code; TensorFlow is not installable in this
image, the reference has no tests / golden vectors, and its trained weight blobs are absent
(``/root/reference/.MISSING_LARGE_BLOBS``).  This file is therefore a line-by-line numpy restatement of the
reference's Python sources (citations below are ``file:line`` under /root/reference).  What IS pinned against the
reference's own artefacts: variable names / shapes / parameter counts (``tests/golden/ckpt_tables.json``, generated
from the shipped ``*.ckpt.index`` files) and the graph constants decoded from the shipped ``*.ckpt.meta``
(SURVEY.md 8c): top-k = 20 sorted, BN eps = 1e-3, l2_normalize eps = 1e-12, slim momentum 0.999, loss margins.

Numerics.  Every function takes ``dtype`` (np.float32 = the reference's precision, np.float64 = shadow used to
measure the tolerance budget).  The kNN mask is ALWAYS computed in float32 with the association the graph uses
(``(sq_i + (-2*inner_ij)) + sq_j`` then negated, ``utils/tf_util.py:651-656``) and with
``inner = (x_i*x_j + y_i*y_j) + z_i*z_j`` evaluated left to right with one rounding per operation (no FMA) --
TensorFlow's own matmul accumulation order for K = 3 cannot be known here; this is the order the HIP kernel
implements bit-exactly so that neighbour SETS (including ties) are identical.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import numpy as np

BN_EPS = 1e-3          # utils/tf_util.py:490 ; slim.batch_norm default epsilon (loupe.py:82-87, 257-263, 323)
L2_EPS = 1e-12         # tf.nn.l2_normalize default epsilon (models/epc-net.py:148,153 ; loupe.py:295,298)
SLIM_DECAY = 0.999     # slim.batch_norm default decay
TOPK = 20              # utils/tf_util.py:660  (hard-coded, ignores the k argument)

EPC_NET_CONVS = ["conv1", "conv1_a", "conv1_b", "conv2", "conv2_a", "conv2_b",
                 "conv3", "conv3_a", "conv3_b", "conv4", "conv4_a", "conv4_b", "conv5"]
EPC_NET_L_CONVS = ["conv1", "conv1_a", "conv1_b", "conv2", "conv2_a", "conv2_b", "conv5"]

DEFAULT_PARAMS = {"CLUSTER_SIZE": 64, "FEATURE_OUTPUT_DIM": 256, "KNN": 20, "INPUT_DIM": 3, "GROUPS": 4,
                  "NUM_POINTS": 4096}


# --------------------------------------------------------------------------------------------------------------
# variable tables + seeded weights
# --------------------------------------------------------------------------------------------------------------
def ema_names(scope: str, outer: str = "query_triplets") -> Tuple[str, str]:
    """Shadow-variable names created by tf.train.ExponentialMovingAverage inside batch_norm_template
    (utils/tf_util.py:474-487).  The full scope is repeated -- see tests/golden/ckpt_tables.json."""
    full = (outer + "/" if outer else "") + scope + "/bn"
    rel = scope + "/bn/" + full + "/moments/"
    return rel + "Squeeze/ExponentialMovingAverage", rel + "Squeeze_1/ExponentialMovingAverage"


def variable_table(arch: str = "epc-net", params: Optional[dict] = None, outer: str = "query_triplets"
                   ) -> "OrderedDict[str, Tuple[Tuple[int, ...], str]]":
    """{name relative to the outer scope: (shape, kind)}; kind in {weight,bias,gamma,beta,mean,var}.
    Trainable = weight/bias/gamma/beta.  Names: models/epc-net.py:62-149, loupe.py:75-79,249-316."""
    p = dict(DEFAULT_PARAMS)
    p.update(params or {})
    D_in, C, O, G = p["INPUT_DIM"], p["CLUSTER_SIZE"], p["FEATURE_OUTPUT_DIM"], p["GROUPS"]
    F = 1024
    t: "OrderedDict[str, Tuple[Tuple[int, ...], str]]" = OrderedDict()

    def conv(scope, cin, cout, rank3=True):
        t[scope + "/weights"] = (((1, cin, cout) if rank3 else (cin, cout)), "weight")
        t[scope + "/biases"] = ((cout,), "bias")
        t[scope + "/bn/beta"] = ((cout,), "beta")
        t[scope + "/bn/gamma"] = ((cout,), "gamma")
        m, v = ema_names(scope, outer)
        t[m] = ((cout,), "mean")
        t[v] = ((cout,), "var")

    nblocks = 4 if arch == "epc-net" else 2
    conv("fastdgcnn/conv1", D_in, 64)
    for b in range(1, nblocks + 1):
        if b > 1:
            conv("fastdgcnn/conv%d" % b, 64, 64)
        conv("fastdgcnn/conv%d_a" % b, 64, 64)
        conv("fastdgcnn/conv%d_b" % b, 64, 64)
    conv("fastdgcnn/conv5", 64 * nblocks, F)
    if arch == "epc-net":
        t["VLAD/cluster_weights"] = ((F, C), "weight")
        for s, n in (("cluster_bn", C),):
            t["VLAD/%s/beta" % s] = ((n,), "beta")
            t["VLAD/%s/gamma" % s] = ((n,), "gamma")
            t["VLAD/%s/moving_mean" % s] = ((n,), "mean")
            t["VLAD/%s/moving_variance" % s] = ((n,), "var")
        t["VLAD/cluster_weights2"] = ((1, F, C), "weight")
        t["VLAD/hidden1_weights"] = ((C * F // G, O), "weight")
        for s in ("bn", "gating_bn"):
            t["VLAD/%s/beta" % s] = ((O,), "beta")
            t["VLAD/%s/gamma" % s] = ((O,), "gamma")
            t["VLAD/%s/moving_mean" % s] = ((O,), "mean")
            t["VLAD/%s/moving_variance" % s] = ((O,), "var")
        t["VLAD/gating_weights"] = ((O, O), "weight")
    elif arch == "epc-net-l":
        conv("VLAD/fc1", F, O, rank3=False)
    else:
        raise ValueError(arch)
    return t


def count_trainable(table) -> int:
    return sum(int(np.prod(s)) for s, k in table.values() if k in ("weight", "bias", "gamma", "beta"))


def seeded_weights(arch: str = "epc-net", seed: int = 0, params: Optional[dict] = None, mode: str = "trained",
                   outer: str = "query_triplets") -> "OrderedDict[str, np.ndarray]":
    """Seeded float32 weights keyed by the checkpoint names (relative to ``outer``).

    mode="init": the reference's initialisers -- Xavier-uniform conv/fc weights (utils/tf_util.py:42), zero
    biases (:98), gamma 1 / beta 0 (:467-470), zero EMA shadows, slim moving_mean 0 / moving_variance 1,
    N(0, 1/sqrt(F)) cluster weights (loupe.py:252-253, 281-282), N(0, 1/sqrt(C)) hidden1 (:316),
    N(0, 1/sqrt(O)) gating (:78-79).
    mode="trained": same weight distributions but non-trivial biases / BN affine / moving statistics, so every
    term of the folded-BN arithmetic is exercised.
    """
    rng = np.random.RandomState(seed)
    table = variable_table(arch, params, outer)
    p = dict(DEFAULT_PARAMS)
    p.update(params or {})
    w: "OrderedDict[str, np.ndarray]" = OrderedDict()
    for name, (shape, kind) in table.items():
        if kind == "weight":
            if name.startswith("VLAD/cluster_weights"):
                std = 1.0 / math.sqrt(1024)
                arr = rng.randn(*shape) * std
            elif name == "VLAD/hidden1_weights":
                arr = rng.randn(*shape) * (1.0 / math.sqrt(p["CLUSTER_SIZE"]))
            elif name == "VLAD/gating_weights":
                arr = rng.randn(*shape) * (1.0 / math.sqrt(shape[0]))
            else:  # xavier uniform, fan_in/fan_out = receptive(1) * channels
                cin, cout = shape[-2], shape[-1]
                lim = math.sqrt(6.0 / (cin + cout))
                arr = rng.uniform(-lim, lim, size=shape)
        elif mode == "init":
            arr = {"bias": np.zeros, "beta": np.zeros, "gamma": np.ones}.get(kind, np.zeros)(shape)
            if kind == "var" and "moving_variance" in name:
                arr = np.ones(shape)
        elif kind == "bias":
            arr = rng.randn(*shape) * 0.05
        elif kind == "beta":
            arr = rng.randn(*shape) * 0.1
        elif kind == "gamma":
            arr = rng.uniform(0.5, 1.5, size=shape)
        elif kind == "mean":
            arr = rng.randn(*shape) * 0.1
        elif kind == "var":
            arr = rng.uniform(0.5, 1.5, size=shape)
        else:
            raise AssertionError(kind)
        w[name] = np.asarray(arr, dtype=np.float32)
    return w


def adversarial_weights(arch: str = "epc-net", seed: int = 0, calibrate_on: Optional[np.ndarray] = None,
                        params: Optional[dict] = None, outer: str = "query_triplets", gamma_range=(0.1, 10.0),
                        floor_frac: float = 0.05, outlier: float = 50.0) -> "OrderedDict[str, np.ndarray]":
    """Weights that stress reduced-precision arithmetic, for the adversarial parity set (tests/test_gpu_adversarial.py):

    * conv / fc / VLAD matrices: Student-t (3 degrees of freedom: heavy tails) at the standard deviation of the reference's
      initialiser, with 1 % of the OUTPUT channels (at least one) multiplied by ``outlier`` = 50 (outlier channels);
    * BatchNorm gamma log-uniform in ``gamma_range`` ([0.1, 10]), beta N(0, 0.3), biases N(0, 0.1);
    * moving statistics: ``calibrate_on`` = None draws the variance log-uniform in [1e-4, 1.5] and the mean N(0, 0.1)
      (un-trained-like: the folded weights W * gamma / sqrt(var + 1e-3) reach the hundreds and grow the activations layer by
      layer); ``calibrate_on`` = clouds (B, N, 3) sets every moving mean / variance to the float64 batch statistics of that
      layer on those clouds (what a trained checkpoint holds: normalised activations whatever the weights' scale), then
      floors a random ``floor_frac`` (5 %) of each layer's variances at 1e-4 * (1 + their value) and leaves the others --
      near-constant channels with a large 1 / sqrt(var).
    """
    rng = np.random.RandomState(100003 + seed)
    w = seeded_weights(arch, seed, params, "trained", outer)
    table = variable_table(arch, params, outer)
    for name, (shape, kind) in table.items():
        if kind == "weight":
            std = float(np.std(w[name])) or 1.0
            t = rng.standard_t(3, size=shape) / math.sqrt(3.0) * std
            cout = shape[-1]
            hot = rng.choice(cout, size=max(1, cout // 100), replace=False)
            t[..., hot] *= outlier
            w[name] = t.astype(np.float32)
        elif kind == "gamma":
            w[name] = np.exp(rng.uniform(math.log(gamma_range[0]), math.log(gamma_range[1]), size=shape)).astype(np.float32)
        elif kind == "beta":
            w[name] = (rng.randn(*shape) * 0.3).astype(np.float32)
        elif kind == "bias":
            w[name] = (rng.randn(*shape) * 0.1).astype(np.float32)
        elif kind == "var":
            w[name] = np.exp(rng.uniform(math.log(1e-4), math.log(1.5), size=shape)).astype(np.float32)
        elif kind == "mean":
            w[name] = (rng.randn(*shape) * 0.1).astype(np.float32)
    if calibrate_on is not None:
        pc = np.asarray(calibrate_on, dtype=np.float32)
        _, st = forward(pc[:, None], w, is_training=True, bn_decay=0.0, arch=arch, dtype=np.float64, params=params,
                        outer=outer)
        for name, val in st.new_stats.items():          # decay 0: the shadow becomes the batch statistic
            w[name] = np.asarray(val, dtype=np.float32)
        for scope, (mean, var) in st.batch_stats.items():   # slim layers (decay fixed at 0.999): take the statistics
            if scope + "/moving_mean" in w:
                w[scope + "/moving_mean"] = np.asarray(mean, dtype=np.float32)
                w[scope + "/moving_variance"] = np.asarray(var, dtype=np.float32)
        for name, (shape, kind) in table.items():
            if kind == "var" and floor_frac > 0:
                v = w[name]
                low = rng.choice(v.size, size=max(1, int(v.size * floor_frac)), replace=False)
                v[low] = 1e-4 * (1.0 + v[low])
                w[name] = v
    return w


# --------------------------------------------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------------------------------------------
def synthetic_clouds(batch: int, n: int, seed: int, kind: str = "uniform") -> np.ndarray:
    rng = np.random.RandomState(seed)
    if kind == "uniform":
        pc = rng.uniform(-1.0, 1.0, size=(batch, n, 3))
    elif kind == "lidar":  # ground plane + vertical planes + noise: many near-ties
        pc = np.empty((batch, n, 3))
        for b in range(batch):
            n_g = int(0.6 * n)
            n_noise = max(int(0.05 * n), 1)
            n_w = n - n_g - n_noise
            g = np.stack([rng.uniform(-1, 1, n_g), rng.uniform(-1, 1, n_g), -0.5 + 0.01 * rng.randn(n_g)], 1)
            walls = []
            for wi in range(4):
                m = n_w // 4 + (1 if wi < n_w % 4 else 0)
                u, h = rng.uniform(-1, 1, m), rng.uniform(-0.5, 0.8, m)
                c = (-0.8, 0.8)[wi % 2] + 0.01 * rng.randn(m)
                walls.append(np.stack([c, u, h], 1) if wi < 2 else np.stack([u, c, h], 1))
            noise = rng.uniform(-1, 1, size=(n_noise, 3))
            allp = np.concatenate([g] + walls + [noise], 0)
            pc[b] = allp[rng.permutation(n)]
    elif kind == "lattice":  # exact ties everywhere
        side = int(round(n ** (1.0 / 3.0)))
        while side ** 3 < n:
            side += 1
        g = np.stack(np.meshgrid(*[np.arange(side)] * 3, indexing="ij"), -1).reshape(-1, 3)[:n]
        pc = np.tile((g / max(side - 1, 1) * 2.0 - 1.0)[None], (batch, 1, 1))
    elif kind == "dup":  # duplicated points
        base = rng.uniform(-1.0, 1.0, size=(batch, n // 2, 3))
        pc = np.concatenate([base, base[:, : n - n // 2]], 1)
    elif kind == "zeros":  # evaluate.py:425-430 / train.py:834-844 padding clouds
        pc = np.zeros((batch, n, 3))
    elif kind.startswith("repeat"):  # repeatNN: NN % of the points are copies of ONE point of the cloud (a clump of
        # identical rows: nothing averages over them), the rest uniform; shuffled
        frac = int(kind[len("repeat"):]) / 100.0
        pc = rng.uniform(-1.0, 1.0, size=(batch, n, 3))
        m = int(round(frac * n))
        for b in range(batch):
            pc[b, :m] = pc[b, n - 1]
            pc[b] = pc[b][rng.permutation(n)]
    elif kind.startswith("zeropad"):  # zeropadNN: the last NN % of the points are (0,0,0) (a partially padded cloud)
        frac = int(kind[len("zeropad"):]) / 100.0
        pc = rng.uniform(-1.0, 1.0, size=(batch, n, 3))
        pc[:, n - int(round(frac * n)):] = 0.0
    else:
        raise ValueError(kind)
    return np.ascontiguousarray(pc, dtype=np.float32)


# --------------------------------------------------------------------------------------------------------------
# kNN mask: utils/tf_util.py:647-666
# --------------------------------------------------------------------------------------------------------------
def neg_sq_dist(pc: np.ndarray) -> np.ndarray:
    """a[b,i,j] = -((sq_i + (-2 * inner_ij)) + sq_j) in float32, one rounding per op (tf_util.py:650-656)."""
    pc = np.asarray(pc, dtype=np.float32)
    assert pc.shape[-1] == 3, "the HIP path implements the INPUT_DIM == 3 configuration"
    x, y, z = pc[..., 0], pc[..., 1], pc[..., 2]
    sq = (x * x + y * y) + z * z                                     # reduce_sum(square(pc), -1)
    inner = (x[:, :, None] * x[:, None, :] + y[:, :, None] * y[:, None, :]) + z[:, :, None] * z[:, None, :]
    inner = np.float32(-2.0) * inner                                 # pc_inner = -2 * pc_inner
    a = -((sq[:, :, None] + inner) + sq[:, None, :])
    assert a.dtype == np.float32
    return a


def kth_largest(a: np.ndarray, k: int = TOPK) -> np.ndarray:
    """min over the k largest values per row, with multiplicity (tf.nn.top_k + reduce_min, tf_util.py:660-663)."""
    n = a.shape[-1]
    part = np.partition(a, n - k, axis=-1)
    return part[..., n - k]


def pairwise_distance_mask(pc: np.ndarray, k: int = TOPK) -> np.ndarray:
    """(B,N,N) float32 0/1 mask; ``k`` is ignored exactly as the reference ignores it (tf_util.py:660)."""
    a = neg_sq_dist(pc)
    kth = kth_largest(a, TOPK)[..., None]
    return (a >= kth).astype(np.float32)                             # greater_equal + cast, tf_util.py:664-665


def knn_lists(pc: np.ndarray) -> Tuple[np.ndarray, List[List[np.ndarray]]]:
    """kth (B,N) float32 and, per cloud and row, the ascending index list {j : a_ij >= kth_i}."""
    a = neg_sq_dist(pc)
    kth = kth_largest(a, TOPK)
    lists = [[np.nonzero(a[b, i] >= kth[b, i])[0].astype(np.int32) for i in range(a.shape[1])]
             for b in range(a.shape[0])]
    return kth, lists


# --------------------------------------------------------------------------------------------------------------
# layers: utils/tf_util.py
# --------------------------------------------------------------------------------------------------------------
def l2_normalize(x: np.ndarray, axis: int) -> np.ndarray:
    """x * rsqrt(max(sum(x^2), 1e-12))  (graph: Square, Sum, Maximum, Rsqrt, Mul)."""
    ss = np.sum(x * x, axis=axis, keepdims=True)
    return x * (1.0 / np.sqrt(np.maximum(ss, x.dtype.type(L2_EPS))))


def batch_normalization(x, mean, var, beta, gamma, eps):
    """tf.nn.batch_normalization: inv = rsqrt(var+eps)*gamma ; x*inv + (beta - mean*inv)."""
    dt = x.dtype.type
    inv = (dt(1.0) / np.sqrt(var + dt(eps))) * gamma
    return x * inv + (beta - mean * inv)


class State:
    """Weights + (training) EMA updates + named intermediates of one forward call."""

    def __init__(self, weights: Dict[str, np.ndarray], dtype, outer: str = "query_triplets"):
        self.dtype = dtype
        self.outer = outer
        self.w = {k: np.asarray(v, dtype=dtype) for k, v in weights.items()}
        self.new_stats: Dict[str, np.ndarray] = {}
        self.batch_stats: Dict[str, Tuple[np.ndarray, np.ndarray]] = {}
        self.taps: Dict[str, np.ndarray] = {}


def batch_norm_template(st: State, x, scope, axes, is_training, bn_decay):
    """utils/tf_util.py:454-491.  Population variance; EMA shadow = shadow - (1-decay)*(shadow - stat)."""
    dt = st.dtype
    beta, gamma = st.w[scope + "/beta"], st.w[scope + "/gamma"]
    mname, vname = ema_names(scope[:-3], st.outer)
    if is_training:
        mean = np.mean(x, axis=axes, dtype=dt)
        var = np.mean((x - mean) ** 2, axis=axes, dtype=dt)
        decay = dt(0.9 if bn_decay is None else bn_decay)
        st.new_stats[mname] = st.w[mname] - (dt(1.0) - decay) * (st.w[mname] - mean)
        st.new_stats[vname] = st.w[vname] - (dt(1.0) - decay) * (st.w[vname] - var)
        st.batch_stats[scope] = (mean, var)
    else:
        mean, var = st.w[mname], st.w[vname]
    return batch_normalization(x, mean, var, beta, gamma, BN_EPS)


def conv1d(st: State, x, scope, is_training, bn_decay, bn=True, relu=True):
    """utils/tf_util.py:52-107 with kernel_size 1: per-point matmul + bias + BN(axes 0,1) + ReLU."""
    W = st.w[scope + "/weights"]
    W = W.reshape(W.shape[-2], W.shape[-1])
    y = np.matmul(x, W) + st.w[scope + "/biases"]
    if bn:
        y = batch_norm_template(st, y, scope + "/bn", (0, 1), is_training, bn_decay)
    if relu:
        y = np.maximum(y, st.dtype(0))
    st.taps[scope] = y
    return y


def fully_connected(st: State, x, scope, is_training, bn_decay, bn=True, relu=True):
    """utils/tf_util.py:310-346: matmul + bias + BN(axis 0) + ReLU (default activation_fn)."""
    y = np.matmul(x, st.w[scope + "/weights"]) + st.w[scope + "/biases"]
    if bn:
        y = batch_norm_template(st, y, scope + "/bn", (0,), is_training, bn_decay)
    if relu:
        y = np.maximum(y, st.dtype(0))
    st.taps[scope] = y
    return y


def slim_batch_norm(st: State, x, scope, is_training, fused: bool):
    """slim.batch_norm / tf.contrib.layers.batch_norm on (rows, C) (loupe.py:82-87, 257-263, 323).
    Normalisation uses the population variance; the FUSED op feeds the Bessel-corrected variance to the
    moving-variance update (FusedBatchNorm semantics), the unfused form feeds the population variance."""
    dt = st.dtype
    beta, gamma = st.w[scope + "/beta"], st.w[scope + "/gamma"]
    mm, mv = st.w[scope + "/moving_mean"], st.w[scope + "/moving_variance"]
    if is_training:
        rows = x.shape[0]
        mean = np.mean(x, axis=0, dtype=dt)
        var = np.mean((x - mean) ** 2, axis=0, dtype=dt)
        var_upd = var * dt(rows / max(rows - 1, 1)) if fused else var
        one_m = dt(1.0 - SLIM_DECAY)
        st.new_stats[scope + "/moving_mean"] = mm - (mm - mean) * one_m
        st.new_stats[scope + "/moving_variance"] = mv - (mv - var_upd) * one_m
        st.batch_stats[scope] = (mean, var)
    else:
        mean, var = mm, mv
    return batch_normalization(x, mean, var, beta, gamma, BN_EPS)


# --------------------------------------------------------------------------------------------------------------
# loupe.py
# --------------------------------------------------------------------------------------------------------------
def context_gating(st: State, x, is_training, scope="VLAD"):
    """loupe.py:61-101: gates = sigmoid(BN(x @ Wg)); out = x * gates."""
    gates = np.matmul(x, st.w[scope + "/gating_weights"])
    gates = slim_batch_norm(st, gates, scope + "/gating_bn", is_training, fused=True)
    gates = 1.0 / (1.0 + np.exp(-gates))
    return x * gates


def _vlad_core(st: State, feats, n_points, is_training, scope="VLAD", add_batch_norm=True):
    """Shared by NetVLAD.forward (loupe.py:121-193) and G_VLAD.forward (loupe.py:233-298)."""
    C = st.w[scope + "/cluster_weights"].shape[1]
    F = feats.shape[1]
    act = np.matmul(feats, st.w[scope + "/cluster_weights"])                       # loupe.py:255
    if add_batch_norm:
        act = slim_batch_norm(st, act, scope + "/cluster_bn", is_training, fused=False)  # :257-263
    else:
        act = act + st.w[scope + "/cluster_biases"]                                  # :264-270
    act = act - np.max(act, axis=1, keepdims=True)                                 # softmax :272
    act = np.exp(act)
    act = act / np.sum(act, axis=1, keepdims=True)
    st.taps["vlad_assign"] = act
    act = act.reshape(-1, n_points, C)                                             # :274
    a_sum = np.sum(act, axis=-2, keepdims=True)                                    # :276  (B,1,C)
    a = a_sum * st.w[scope + "/cluster_weights2"]                                  # :284  (B,F,C)
    act_t = np.transpose(act, (0, 2, 1))                                           # :286
    x3 = feats.reshape(-1, n_points, F)                                            # :288
    vlad = np.matmul(act_t, x3)                                                    # :290  (B,C,F)
    vlad = np.transpose(vlad, (0, 2, 1))                                           # :291  (B,F,C)
    vlad = vlad - a                                                                # :292
    st.taps["vlad_raw"] = vlad
    vlad = l2_normalize(vlad, 1)                                                   # :295 intra-norm over F
    vlad = vlad.reshape(-1, C * F)                                                 # :297 feature-major flatten
    vlad = l2_normalize(vlad, 1)                                                   # :298
    st.taps["vlad_flat"] = vlad
    return vlad


def g_vlad_forward(st: State, feats, n_points, groups, is_training, gating=True, scope="VLAD", add_batch_norm=True):
    """loupe.py:233-333."""
    vlad = _vlad_core(st, feats, n_points, is_training, scope, add_batch_norm)
    O = st.w[scope + "/hidden1_weights"].shape[1]
    vlad = vlad.reshape(-1, vlad.shape[1] // groups)                               # :302
    vlad = np.matmul(vlad, st.w[scope + "/hidden1_weights"])                       # :322
    vlad = slim_batch_norm(st, vlad, scope + "/bn", is_training, fused=True)       # :323
    vlad = vlad.reshape(-1, groups, O)                                             # :326
    vlad = np.sum(vlad, axis=-2)                                                   # :328
    st.taps["vlad_hidden"] = vlad
    if gating:
        vlad = context_gating(st, vlad, is_training, scope)                        # :330-331
    return vlad


def netvlad_forward(st: State, feats, n_points, is_training, gating=True, scope="VLAD", add_batch_norm=True):
    """loupe.py:121-214 (ungrouped; hidden1_weights is (C*F, O))."""
    vlad = _vlad_core(st, feats, n_points, is_training, scope, add_batch_norm)
    vlad = np.matmul(vlad, st.w[scope + "/hidden1_weights"])
    vlad = slim_batch_norm(st, vlad, scope + "/bn", is_training, fused=True)
    if gating:
        vlad = context_gating(st, vlad, is_training, scope)
    return vlad


# --------------------------------------------------------------------------------------------------------------
# models/epc-net.py:29-157 and models/epc-net-l.py:29-102
# --------------------------------------------------------------------------------------------------------------
def neighbour_mean(x, mask=None, lists=None, k=20):
    """x1 = matmul(dpist, x) / float(k)  (models/epc-net.py:70-71).  ``lists`` = kNN-index formulation:
    sum of the selected rows in ascending j, then the same division."""
    dt = x.dtype.type
    if mask is not None:
        return np.matmul(mask.astype(x.dtype), x) / dt(float(k))
    out = np.empty_like(x)
    for b in range(x.shape[0]):
        xb = x[b]
        for i, idx in enumerate(lists[b]):
            acc = np.zeros(x.shape[-1], dtype=x.dtype)
            for j in idx:                       # ascending j, one rounding per add (matches the HIP kernel)
                acc = acc + xb[j]
            out[b, i] = acc / dt(float(k))
    return out


def forward(point_cloud: np.ndarray, weights: Dict[str, np.ndarray], is_training: bool = False,
            bn_decay: Optional[float] = None, params: Optional[dict] = None, arch: str = "epc-net",
            dtype=np.float32, formulation: str = "dense", mask: Optional[np.ndarray] = None,
            lists=None, outer: str = "query_triplets") -> Tuple[np.ndarray, State]:
    """(B,P,N,3) -> (B,P,FEATURE_OUTPUT_DIM).  Returns (output, State with taps / new EMA stats)."""
    p = dict(DEFAULT_PARAMS)
    p.update(params or {})
    B, P, N, D = point_cloud.shape
    assert D == p["INPUT_DIM"]
    k = p["KNN"]
    st = State(weights, dtype, outer)
    pc32 = np.ascontiguousarray(point_cloud, dtype=np.float32).reshape(B * P, N, D)   # epc-net.py:41
    pc = pc32.astype(dtype)
    if formulation == "dense":
        if mask is None:
            mask = pairwise_distance_mask(pc32, k)                                     # epc-net.py:63
        nm = lambda x: neighbour_mean(x, mask=mask, k=k)
    else:
        if lists is None:
            _, lists = knn_lists(pc32)
        nm = lambda x: neighbour_mean(x, lists=lists, k=k)
    tr, bd = is_training, bn_decay
    nblocks = 4 if arch == "epc-net" else 2
    outs = []
    inp = pc
    for b in range(1, nblocks + 1):                                                   # epc-net.py:66-132
        x = conv1d(st, inp, "fastdgcnn/conv%d" % b, tr, bd)
        xm = nm(x)
        st.taps["mean%d" % b] = xm
        t = xm - x
        t = conv1d(st, t, "fastdgcnn/conv%d_a" % b, tr, bd)
        t = conv1d(st, t, "fastdgcnn/conv%d_b" % b, tr, bd)
        inp = t + xm
        st.taps["block%d" % b] = inp
        outs.append(inp)
    x = np.concatenate(outs, axis=-1)                                                 # :134
    x = conv1d(st, x, "fastdgcnn/conv5", tr, bd)                                      # :136-139
    if arch == "epc-net":
        net = x.reshape(-1, 1024)                                                     # :147
        net = l2_normalize(net, 1)                                                    # :148
        out = g_vlad_forward(st, net, N, p["GROUPS"], tr, gating=True)                # :143-149
    else:
        net = np.max(x, axis=1)                                                       # epc-net-l.py:88-92
        st.taps["maxpool"] = net
        out = fully_connected(st, net, "VLAD/fc1", tr, bd)                            # epc-net-l.py:95
    st.taps["pre_norm"] = out
    out = l2_normalize(out, 1)                                                        # epc-net.py:153
    return out.reshape(B, P, p["FEATURE_OUTPUT_DIM"]), st                             # :155


# --------------------------------------------------------------------------------------------------------------
# losses: models/epc-net.py:160-284
# --------------------------------------------------------------------------------------------------------------
def best_pos_distance(query, pos_vecs):
    """:160-167  min_p sum((pos_p - q)^2)."""
    return np.min(np.sum((pos_vecs - query) ** 2, axis=2), axis=1)


def _hinge_terms(q_vec, pos_vecs, neg_vecs, margin, anchor):
    best = best_pos_distance(q_vec, pos_vecs)[:, None]
    d = np.sum((neg_vecs - anchor) ** 2, axis=2)
    return np.maximum(margin + best - d, 0.0)


def triplet_loss(q, pos, neg, margin):
    return np.mean(np.sum(_hinge_terms(q, pos, neg, margin, q), axis=1))               # :174-183


def lazy_triplet_loss(q, pos, neg, margin):
    return np.mean(np.max(_hinge_terms(q, pos, neg, margin, q), axis=1))               # :186-194


def _softmargin_terms(q, pos, neg):
    best = best_pos_distance(q, pos)[:, None]
    return np.log(np.exp(best - np.sum((neg - q) ** 2, axis=2)) + 1.0)


def lazy_softmargin_loss(q, pos, neg):
    return np.mean(np.max(_softmargin_terms(q, pos, neg), axis=1))                     # :207-215


def quadruplet_loss(q, pos, neg, other_neg, m1, m2):
    return triplet_loss(q, pos, neg, m1) + np.mean(np.sum(_hinge_terms(q, pos, neg, m2, other_neg), axis=1))  # :252-267


def lazy_quadruplet_loss(q, pos, neg, other_neg, m1, m2):
    return lazy_triplet_loss(q, pos, neg, m1) + np.mean(np.max(_hinge_terms(q, pos, neg, m2, other_neg), axis=1))  # :269-284


def lazy_quadruplet_loss_sm(q, pos, neg, other_neg, m2):
    return lazy_softmargin_loss(q, pos, neg) + np.mean(np.max(_hinge_terms(q, pos, neg, m2, other_neg), axis=1))  # :234-249


# --------------------------------------------------------------------------------------------------------------
# schedules: train.py:138-157
# --------------------------------------------------------------------------------------------------------------
def get_bn_decay(step: int, batch_num_queries: int = 1, bn_init_decay=0.5, decay_rate=0.5, decay_step=200000,
                 clip=0.99) -> float:
    """train.py:138-146 (+ hard-coded constants :127-130): min(0.99, 1 - 0.5*0.5^floor(step*B/200000))."""
    momentum = bn_init_decay * decay_rate ** math.floor(step * batch_num_queries / float(decay_step))
    return min(clip, 1.0 - momentum)


def get_learning_rate(epoch: int, base_lr: float = 5e-5) -> float:
    """train.py:154-157."""
    return max(base_lr * (0.9 ** (epoch // 5)), 0.00001)


# --------------------------------------------------------------------------------------------------------------
# retrieval: evaluate.py:455-537
# --------------------------------------------------------------------------------------------------------------
def knn_bruteforce(database: np.ndarray, queries: np.ndarray, k: int) -> Tuple[np.ndarray, np.ndarray]:
    """Exact Euclidean k-NN in float64 (what sklearn's KDTree.query returns, evaluate.py:463,481);
    ties broken by ascending database index."""
    d = database.astype(np.float64)
    q = queries.astype(np.float64)
    d2 = (q * q).sum(1)[:, None] - 2.0 * q @ d.T + (d * d).sum(1)[None, :]
    order = np.argsort(d2, axis=1, kind="stable")[:, :k]
    return np.sqrt(np.maximum(np.take_along_axis(d2, order, 1), 0.0)), order


def get_recall(database_output: np.ndarray, queries_output: np.ndarray, true_neighbors: List[List[int]],
               num_neighbors: int = 25, indices: Optional[np.ndarray] = None):
    """evaluate.py:455-537 for one (database run m, query run n) pair.  ``true_neighbors[i]`` =
    QUERY_SETS[n][i][m].  Returns (recall[25] in %, top1 similarity list, one_percent_recall in %)."""
    recall = [0] * num_neighbors
    top1_similarity_score = []
    one_percent_retrieved = 0
    threshold = max(int(round(len(database_output) / 100.0)), 1)                       # :470 (banker's rounding)
    num_evaluated = 0
    if indices is None:
        _, indices = knn_bruteforce(database_output, queries_output, min(num_neighbors, len(database_output)))
    for i in range(len(queries_output)):
        truth = true_neighbors[i]
        if len(truth) == 0:                                                            # :477-478
            continue
        num_evaluated += 1
        ind = indices[i]
        for j in range(len(ind)):                                                      # :512-521
            if ind[j] in truth:
                if j == 0:
                    top1_similarity_score.append(float(np.dot(queries_output[i], database_output[ind[j]])))
                recall[j] += 1
                break
        if len(set(ind[0:threshold]).intersection(set(truth))) > 0:                    # :526-527
            one_percent_retrieved += 1
    one_percent_recall = (one_percent_retrieved / float(num_evaluated)) * 100
    recall = (np.cumsum(recall) / float(num_evaluated)) * 100
    return recall, top1_similarity_score, one_percent_recall
