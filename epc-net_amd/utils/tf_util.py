"""Operator wrappers with the reference's names and argument meaning (reference ``utils/tf_util.py``).

Tensors are ``torch.Tensor`` on a ROCm device; every contraction, normalisation, gather, reduction over points, loss and
optimizer update is a call into libepcnet_hip.so and there is no torch / CPU fallback (``EpcNetError`` if no GPU is visible).
What the differentiable (``is_training=True``) path still leaves to torch, all of it glue on small or already-materialised
tensors, and only on the per-layer path (``USE_CHAIN = False`` / inference-mode ``forward_ops``): the residual ``t + x1`` of each block
and the concat of the four block outputs (models/epc-net.py:81,134 -- elementwise adds / one copy, with autograd's matching
accumulations), the ``a_sum`` reduction of loupe.py:276 outside the fused assignment node, and the ReLU of a BN-less layer.  The fused inference path has none of these.  Variables are created in the current
``variable_scope`` with the reference's names and initialisers, so a model built from these wrappers has the
same state-dict as the reference checkpoint.

The fused model path (``models/epc-net.py: forward``) does NOT go through these wrappers op by op: it runs the
fused kernels of ``epc_net_forward``.  The wrappers are what `forward(..., is_training=True)` is built from (differentiable: epc-net_amd/ops.py) and serve
op-level callers; the reference functions EPC-Net never calls (conv2d, conv3d, avg_pool*, dropout, knn, get_edge_feature, ...) are
out of scope and raise ``NotImplementedError`` naming the reference line.
"""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from .. import lib as L
from ..variables import (constant, default_store, ema_shadow_names, scoped, truncated_normal, variable_scope,
                         xavier_uniform)

relu = "relu"  # sentinel for activation_fn=tf.nn.relu (the only activation the hot path uses)


def _variable_on_cpu(name, shape, initializer, use_fp16=False):
    """utils/tf_util.py:10-22.  The '/cpu:0' pin is a TF1 idiom; variables live in HBM here."""
    if use_fp16:
        raise NotImplementedError("use_fp16 variables are not used by EPC-Net (utils/tf_util.py:20)")
    return default_store().get_variable(scoped(name), shape, initializer)


def _variable_with_weight_decay(name, shape, stddev, wd, use_xavier=True):
    """utils/tf_util.py:24-49.  wd is 0.0 everywhere on the hot path, so 'weight_loss' is identically 0."""
    init = xavier_uniform if use_xavier else truncated_normal(stddev)
    return _variable_on_cpu(name, shape, init)


def _bn_variables(scope: str, num_channels: int):
    """Variables of batch_norm_template (utils/tf_util.py:465-487) inside the CURRENT scope + ``scope``."""
    st = default_store()
    with variable_scope(scope) as full:
        beta = st.get_variable(full + "/beta", (num_channels,), constant(0.0))
        gamma = st.get_variable(full + "/gamma", (num_channels,), constant(1.0))
        mname, vname = ema_shadow_names(full)
        mean = st.get_variable(mname, (num_channels,), constant(0.0), trainable=False)
        var = st.get_variable(vname, (num_channels,), constant(0.0), trainable=False)
    return beta, gamma, mean, var


# Full names of the biases DECLARED in front of a BatchNorm (bn=True below): in training mode their gradient is exactly zero and the
# fused operators do not form it.  training.TrainStep exempts exactly these -- by declaration, not by name suffix -- from its "every
# variable below the cut received a gradient" check.
BIASES_BEFORE_BATCHNORM = set()


def declare_conv1d(scope, num_in_channels, num_output_channels, kernel_size=1, use_xavier=True, stddev=1e-3,
                   bn=True):
    """Create (or fetch) the variables ``conv1d`` owns: weights [k,Cin,Cout], biases, bn/*."""
    with variable_scope(scope):
        w = _variable_with_weight_decay("weights", [kernel_size, num_in_channels, num_output_channels], stddev,
                                        0.0, use_xavier)
        b = _variable_on_cpu("biases", [num_output_channels], constant(0.0))
        bnv = _bn_variables("bn", num_output_channels) if bn else None
        if bn:
            from ..variables import scoped
            BIASES_BEFORE_BATCHNORM.add(scoped("biases"))
    return w, b, bnv


def declare_fully_connected(scope, num_input_units, num_outputs, use_xavier=True, stddev=1e-3, bn=True):
    with variable_scope(scope):
        w = _variable_with_weight_decay("weights", [num_input_units, num_outputs], stddev, 0.0, use_xavier)
        b = _variable_on_cpu("biases", [num_outputs], constant(0.0))
        bnv = _bn_variables("bn", num_outputs) if bn else None
        if bn:
            from ..variables import scoped
            BIASES_BEFORE_BATCHNORM.add(scoped("biases"))
    return w, b, bnv


# ---- ops ---------------------------------------------------------------------------------------------------
def pairwise_distance_mask(pc: torch.Tensor, k: int = 20) -> torch.Tensor:
    """utils/tf_util.py:647-666: (B,N,3) -> (B,N,N) float 0/1 mask of every j with a_ij >= (20th largest a_i.).
    ``k`` is ignored exactly as the reference ignores it (top_k is hard-coded to 20, :660)."""
    L.require_gpu()
    if pc.dim() != 3 or pc.shape[-1] != 3:
        raise L.EpcNetError(-1, "pairwise_distance_mask expects (B,N,3)")
    pc = pc.contiguous().float()
    B, N, _ = pc.shape
    kth, _, _ = knn_index(pc)
    mask = torch.empty((B, N, N), dtype=torch.float32, device=pc.device)
    L.check(L.lib().epc_knn_mask(L.ptr(pc), L.ptr(kth), B, N, L.ptr(mask), L.current_stream()))
    return mask


def knn_index(pc: torch.Tensor, form: Optional[int] = None):
    """Index form of the same graph: (kth (B,N) f32, idx (B,N,32) int32 ascending, cnt (B,N) int32).
    ``form`` (tests / timing scripts): 0 / 1 = the one-lane / four-lane form of the kernel (epc_knn_topk_form); None = the product's."""
    L.require_gpu()
    pc = pc.contiguous().float()
    B, N, _ = pc.shape
    idx = torch.empty((B, N, L.EPC_KNN_CAP), dtype=torch.int32, device=pc.device)
    cnt = torch.empty((B, N), dtype=torch.int32, device=pc.device)
    kth = torch.empty((B, N), dtype=torch.float32, device=pc.device)
    if form is None:
        L.check(L.lib().epc_knn_topk(L.ptr(pc), B, N, L.EPC_KNN_CAP, L.ptr(idx), L.ptr(cnt), L.ptr(kth), L.current_stream()))
    else:
        L.check(L.lib().epc_knn_topk_form(L.ptr(pc), B, N, L.EPC_KNN_CAP, L.ptr(idx), L.ptr(cnt), L.ptr(kth), int(form),
                                          L.current_stream()))
    return kth, idx, cnt


# Moving-average updates can be deferred: a training step collects them (defer_ema_updates) and applies all of them in ONE
# launch at its end (flush_ema_updates) -- 34 statistics per EPC-Net step, each a few hundred bytes.
_deferred_ema = None


def defer_ema_updates() -> None:
    global _deferred_ema
    _deferred_ema = []


def flush_ema_updates(fixed_decay: float, sched_decay, scope: str = None) -> None:
    """Apply the collected updates: entries recorded with decay == fixed_decay use it, all others ``sched_decay``."""
    global _deferred_ema
    pend, _deferred_ema = _deferred_ema, None
    if pend:
        from .. import ops
        ops.ema_multi([p[0] for p in pend], [p[1] for p in pend], [p[2] for p in pend], fixed_decay, sched_decay)
        default_store().bump(scope)


def _ema_update(shadow: torch.Tensor, value: torch.Tensor, decay, scheduled: bool = True) -> None:
    """tf.train.ExponentialMovingAverage.apply / assign_moving_average: shadow -= (1 - decay) * (shadow - value), one
    kernel (epc_ema_update).  ``decay`` is a float, or a 0-d device tensor when the step is replayed as a HIP graph.
    ``scheduled``: the decay is the bn_decay schedule (tf_util BN) rather than slim's fixed 0.999."""
    if _deferred_ema is not None:
        _deferred_ema.append((shadow, value, scheduled))
        return
    value = value.detach().contiguous()
    if torch.is_tensor(decay):
        L.check(L.lib().epc_ema_update(shadow.data_ptr(), value.data_ptr(), shadow.numel(), 0.0, decay.data_ptr(),
                                       L.current_stream()))
    else:
        L.check(L.lib().epc_ema_update(shadow.data_ptr(), value.data_ptr(), shadow.numel(), float(decay), None,
                                       L.current_stream()))
    from ..variables import current_scope
    default_store().bump(current_scope() or None)


def batch_norm_template(inputs, is_training, scope, moments_dims, bn_decay, activation_relu=False):
    """utils/tf_util.py:454-491 on (rows, C) data (the caller flattens the moment axes).  Training: batch mean /
    population variance, EMA shadows updated with ``bn_decay`` (0.9 when None); inference: the EMA shadows.
    ``activation_relu`` fuses the ReLU that follows in conv1d / fully_connected into the same kernel."""
    from .. import ops
    C = int(inputs.shape[-1])
    beta, gamma, ema_mean, ema_var = _bn_variables(scope, C)
    x2 = inputs.reshape(-1, C)
    if is_training:
        y, mean, var = ops.BatchNormTrain.apply(x2, gamma, beta, 1e-3, int(activation_relu))
        decay = 0.9 if bn_decay is None else (bn_decay if torch.is_tensor(bn_decay) else float(bn_decay))
        _ema_update(ema_mean, mean, decay)
        _ema_update(ema_var, var, decay)
    else:
        y = ops.bn_inference(x2, ema_mean, ema_var, gamma, beta, 1e-3, activation_relu)
    return y.reshape(inputs.shape)


def batch_norm_for_fc(inputs, is_training, bn_decay, scope):
    """utils/tf_util.py:494-505."""
    return batch_norm_template(inputs, is_training, scope, [0, ], bn_decay)


def batch_norm_for_conv1d(inputs, is_training, bn_decay, scope):
    """utils/tf_util.py:508-519."""
    return batch_norm_template(inputs, is_training, scope, [0, 1], bn_decay)


# A call site that KNOWS its layer's operands are bounded by construction (conv5: BatchNorm'd block outputs against its weights)
# wraps the call in ``bounded_operands((activation, weight) scale exponents)``: the fused Linear + BatchNorm node then takes the
# split-fp16 three-product arithmetic (ops.F16X3_CONV5; epc_gemm_f16x3_stats, guarded against values that leave its range).  Never
# inferred from a layer's width.
_F16X3_HINT = None


class bounded_operands:
    def __init__(self, exps):
        self.exps = exps

    def __enter__(self):
        global _F16X3_HINT
        self.prev, _F16X3_HINT = _F16X3_HINT, self.exps

    def __exit__(self, *exc):
        global _F16X3_HINT
        _F16X3_HINT = self.prev


def _bn_train_fused(inputs2d, w2d, b, scope_bn, bn_decay, relu_flag, rownorm=False, f16x3=None, link=None):
    """Linear + training-mode BatchNorm (+ReLU, + conv5's row norm) as one autograd node whose batch statistics come from the
    GEMM's epilogue (ops.LinearBatchNormTrain), with the moving-average updates of batch_norm_template."""
    from .. import ops
    C = int(w2d.shape[1])
    beta, gamma, ema_mean, ema_var = _bn_variables(scope_bn, C)
    y, mean, var = ops.LinearBatchNormTrain.apply(inputs2d, w2d, b, gamma, beta, 1e-3, int(relu_flag), bool(rownorm), f16x3, link)
    if link is not None:
        y._epc_tail_link = link      # (read by loupe.G_VLAD.forward: ops.tail_link_of)
    decay = 0.9 if bn_decay is None else (bn_decay if torch.is_tensor(bn_decay) else float(bn_decay))
    _ema_update(ema_mean, mean, decay)
    _ema_update(ema_var, var, decay)
    return y


# Test hook (tests/test_gpu_train_step.py, mask-pinned gradient parity): when a dict, every ReLU'd layer records the boolean
# mask `output > 0` of its (rows, C) activations under its full variable scope.  Never set by product code; eager steps only.
RELU_MASK_TAPS = None
# Test hook of the same kind for VALUES (tests/test_gpu_train_step.py, the bf16 step): when a dict, every layer whose pre-activation
# is a rounding point of the bf16 arithmetic records it (as f32) under its scope, so that the float64 oracle can continue from the
# values the HIP path actually stored instead of from its own (which round differently near a bf16 boundary).  Eager steps only.
VALUE_TAPS = None


def _tap_relu_mask(y) -> None:
    if RELU_MASK_TAPS is not None:
        from ..variables import current_scope
        RELU_MASK_TAPS[current_scope()] = (y.detach() > 0)


def _dense(inputs2d, w2d, b, bn, scope_bn, activation_fn, bn_decay, is_training):
    y = _dense_impl(inputs2d, w2d, b, bn, scope_bn, activation_fn, bn_decay, is_training)
    if activation_fn is not None:
        _tap_relu_mask(y)
    return y


def _dense_impl(inputs2d, w2d, b, bn, scope_bn, activation_fn, bn_decay, is_training):
    from .. import ops
    want_relu = activation_fn is not None
    if activation_fn not in (None, relu):
        raise NotImplementedError("only activation_fn=tf.nn.relu / None are used by EPC-Net")
    if bn and is_training and ops.fused_linear_bn_ok(int(inputs2d.shape[0]), int(w2d.shape[0]), int(w2d.shape[1])):
        return _bn_train_fused(inputs2d, w2d, b, scope_bn, bn_decay, want_relu, f16x3=_F16X3_HINT)
    z = ops.Linear.apply(inputs2d, w2d, b, bool(bn and is_training))
    if bn:
        return batch_norm_template(z, bool(is_training), scope_bn, None, bn_decay, activation_relu=want_relu)
    return torch.relu(z) if want_relu else z


def conv1d(inputs, num_output_channels, kernel_size, scope, stride=1, padding='SAME', use_xavier=True,
           stddev=1e-3, weight_decay=0.0, activation_fn=relu, bn=False, bn_decay=None, is_training=None):
    """utils/tf_util.py:52-107 for kernel_size 1: per-point matmul + bias (+ BN over axes (0,1)) (+ ReLU) on (B,L,C)."""
    if kernel_size != 1 or stride != 1:
        raise NotImplementedError("EPC-Net only uses kernel_size=1, stride=1 (models/epc-net.py:66-139)")
    L.require_gpu()
    cin = int(inputs.shape[-1])
    w, b, _ = declare_conv1d(scope, cin, num_output_channels, kernel_size, use_xavier, stddev, bn)
    with variable_scope(scope):
        y = _dense(inputs.reshape(-1, cin), w.reshape(cin, num_output_channels), b, bn, "bn", activation_fn, bn_decay,
                   is_training)
    return y.reshape(tuple(inputs.shape[:-1]) + (num_output_channels,))


def proxyconv_tail(x, graph, k, scope_a, scope_b, bn_decay=None, is_training=None):
    """x1 = matmul(mask, x) / k; t = conv_b(conv_a(x1 - x)); return t + x1 -- models/epc-net.py:70-86 (and :88-132 for the
    other blocks) behind a block's leading conv, both convs 64 -> 64 with BatchNorm and ReLU.  Not a function of the reference's
    tf_util: a fusion point.  In training (f32-accurate arithmetic) it is ONE autograd node (ops.ProxyConvTail); otherwise the
    same graph op by op."""
    from .. import ops
    shape = tuple(x.shape)
    x2 = x.reshape(-1, 64)
    rows = int(x2.shape[0])
    if not (is_training and ops.fused_linear_bn_ok(rows, 64, 64) and ops._GEMM_PRECISION == "bf16x6"):
        xm, d = ops.NeighbourMeanDiff.apply(x2, graph, k)
        t = conv1d(d.reshape(shape), 64, 1, padding='VALID', stride=1, bn=True, is_training=is_training, scope=scope_a,
                   bn_decay=bn_decay)
        t = conv1d(t, 64, 1, padding='VALID', stride=1, bn=True, is_training=is_training, scope=scope_b, bn_decay=bn_decay)
        return t + xm.reshape(shape)
    L.require_gpu()
    wa, ba, _ = declare_conv1d(scope_a, 64, 64, 1, True, 1e-3, True)
    wb, bb, _ = declare_conv1d(scope_b, 64, 64, 1, True, 1e-3, True)
    with variable_scope(scope_a):
        beta_a, gamma_a, em_a, ev_a = _bn_variables("bn", 64)
    with variable_scope(scope_b):
        beta_b, gamma_b, em_b, ev_b = _bn_variables("bn", 64)
    out, za, ma, va, zb, mb, vb = ops.ProxyConvTail.apply(x2, graph, k, wa.reshape(64, 64), ba, gamma_a, beta_a,
                                                          wb.reshape(64, 64), bb, gamma_b, beta_b, 1e-3)
    decay = 0.9 if bn_decay is None else (bn_decay if torch.is_tensor(bn_decay) else float(bn_decay))
    for scope, z, mean, var, gamma, beta, em, ev in ((scope_a, za, ma, va, gamma_a, beta_a, em_a, ev_a),
                                                     (scope_b, zb, mb, vb, gamma_b, beta_b, em_b, ev_b)):
        with variable_scope(scope):
            _ema_update(em, mean, decay)
            _ema_update(ev, var, decay)
            if RELU_MASK_TAPS is not None:      # (test hook: the layer's activation is not materialised by the fused node)
                _tap_relu_mask(ops.bn_apply_train(z, mean, var, gamma, beta, 1e-3, True))
    return out.reshape(shape)


# False: the training backbone is built from the per-layer operators (conv1d / proxyconv_tail) instead of the fused chain --
# the second implementation the tests hold the chain to.
USE_CHAIN = True
# The backbone's output tensor of the LAST training forward (the concat of the block outputs, models/epc-net.py:134): where a
# data-parallel step cuts its backward in two, so that the exchange of the head's gradients (17.8 of EPC-Net's 18.8 MB: hidden1_weights
# and conv5) travels while the backbone's backward runs (training.TrainStep).
BACKBONE_TAP = None

def proxyconv_backbone(point_cloud, graph, k, nblocks, bn_decay=None, is_training=None, head_follows=False):
    """The ProxyConv backbone of models/epc-net.py:66-134 (four blocks) / models/epc-net-l.py:62-83 (two) up to the concat:
        for b: x = conv_b(in) [conv1 on the coordinates, conv2.. on the previous block's output]; x1 = matmul(mask, x) / k;
               t = conv_b_b(conv_b_a(x1 - x)); in = t + x1
        return concat of the blocks' outputs (B, N, 64 nblocks)
    every conv = 1x1 + BatchNorm + ReLU.  Not a function of the reference's tf_util: a fusion point.  In training it is conv1's
    product (ops.Linear, K = 3) followed by ONE autograd node on the fused chain launches (ops.ProxyConvChain); otherwise the same
    graph op by op (conv1d / proxyconv_tail).  Variables, moving-average updates and the mask-tap test hook as conv1d's.
    ``head_follows``: the caller hands the result to conv1d_l2_normalized(..., lazy=True) and nowhere else -- when that head will run in
    the bf16 arithmetic the chain writes a bf16 copy of the concat beside it and the copy travels WITH the returned tensor
    (attribute ``_epc_bf16``), not through a global."""
    import torch
    from .. import ops
    B, N, cin = (int(v) for v in point_cloud.shape)
    rows = B * N
    if not (is_training and USE_CHAIN and ops.chain_ok(rows)):
        outs, inp = [], point_cloud
        for b in range(1, nblocks + 1):
            x = conv1d(inp, 64, 1, padding='VALID', stride=1, bn=True, is_training=is_training, scope='conv%d' % b, bn_decay=bn_decay)
            inp = proxyconv_tail(x, graph, k, 'conv%d_a' % b, 'conv%d_b' % b, bn_decay=bn_decay, is_training=is_training)
            outs.append(inp)
        return torch.cat(outs, dim=-1)
    L.require_gpu()
    params, bns = [], []          # bns: (scope, gamma, beta, ema_mean, ema_var) in the node's output order
    w1 = b1 = None
    for b in range(1, nblocks + 1):
        w, bias, _ = declare_conv1d('conv%d' % b, cin if b == 1 else 64, 64, 1, True, 1e-3, True)
        with variable_scope('conv%d' % b):
            beta, gamma, em, ev = _bn_variables("bn", 64)
        if b == 1:
            w1, b1 = w.reshape(cin, 64), bias
            params += [gamma, beta]
        else:
            params += [w.reshape(64, 64), bias, gamma, beta]
        bns.append(('conv%d' % b, gamma, beta, em, ev))
        for sfx in ('_a', '_b'):
            w, bias, _ = declare_conv1d('conv%d%s' % (b, sfx), 64, 64, 1, True, 1e-3, True)
            with variable_scope('conv%d%s' % (b, sfx)):
                beta, gamma, em, ev = _bn_variables("bn", 64)
            params += [w.reshape(64, 64), bias, gamma, beta]
            bns.append(('conv%d%s' % (b, sfx), gamma, beta, em, ev))
    z01 = ops.Linear.apply(point_cloud.reshape(rows, cin), w1, b1, True)        # conv1's product (bias in front of a BatchNorm)
    pf, pb = (3, 2) if ops._GEMM_PRECISION == "bf16x6" else (1, 1)
    want16 = bool(head_follows) and ops.head_stream_mode(rows, 64 * int(nblocks), 1024, N) == "bf16"
    res = ops.ProxyConvChain.apply(z01, graph, int(k), 1e-3, int(nblocks), pf, pb, want16, *params)
    cat, rest, cat16 = res[0], list(res[1:-1]), res[-1]
    decay = 0.9 if bn_decay is None else (bn_decay if torch.is_tensor(bn_decay) else float(bn_decay))
    at = 0
    for li, (scope, gamma, beta, em, ev) in enumerate(bns):
        if li % 3 == 0 and li > 0:          # a later block's leading conv: its pre-activation is an output of the node
            z = rest[at]
            at += 1
        elif li == 0:
            z = z01
        else:
            z = None
        if li % 3 == 0:
            mean, var = rest[at], rest[at + 1]
            at += 2
        else:
            z, mean, var = rest[at], rest[at + 1], rest[at + 2]
            at += 3
        with variable_scope(scope):
            _ema_update(em, mean, decay)
            _ema_update(ev, var, decay)
            if RELU_MASK_TAPS is not None:      # (test hook: the layer's activation is not materialised by the fused node)
                _tap_relu_mask(ops.bn_apply_train(z, mean, var, gamma, beta, 1e-3, True))
            if VALUE_TAPS is not None:
                from ..variables import current_scope
                VALUE_TAPS[current_scope()] = z.detach().clone()
    assert at == len(rest)
    global BACKBONE_TAP
    BACKBONE_TAP = cat.reshape(B, N, 64 * nblocks)
    if cat16 is not None:
        BACKBONE_TAP._epc_bf16 = cat16          # (rows, 64 nblocks) bf16: conv1d_l2_normalized hands it to the streamed head
    return BACKBONE_TAP


def conv1d_l2_normalized(inputs, num_output_channels, scope, bn_decay=None, is_training=None, lazy=False):
    """conv1d(kernel 1, bn=True, relu) followed by tf.nn.l2_normalize over the channels of every point -- the pair
    models/epc-net.py:136-148 applies to conv5's output -- returned as (B*L, C).  Not a function of the reference's
    tf_util: a fusion point.  In training the BatchNorm apply, the ReLU and the row norm are one pass over the
    (rows, 1024) activations (ops.BatchNormReluRowNorm) and the un-normalised map is never written.
    ``lazy``: the caller hands the result to loupe.G_VLAD.forward and nowhere else -- in training (ops.HEAD_STREAM) the layer is
    then NOT evaluated here: an ops.LazyConv5Features carries its operands into the VLAD node (ops.Conv5VladHead), which never
    writes the feature map at all."""
    from .. import ops
    cin = int(inputs.shape[-1])
    if not (is_training and num_output_channels == 1024):
        y = conv1d(inputs, num_output_channels, 1, padding='VALID', stride=1, bn=True, is_training=is_training,
                   scope=scope, bn_decay=bn_decay)
        return ops.RowL2Normalize.apply(y.reshape(-1, num_output_channels))
    L.require_gpu()
    w, b, _ = declare_conv1d(scope, cin, num_output_channels, 1, True, 1e-3, True)
    n_points = int(inputs.shape[1]) if inputs.dim() == 3 else None      # (B, N, C): the consumer's max_samples
    with variable_scope(scope):
        x2 = inputs.reshape(-1, cin)

        def eager():
            return _conv5_l2_normalized_eager(x2, w, b, cin, num_output_channels, bn_decay)

        # ONE decision, with everything the consumer will look at (rows, widths AND the points per cloud: ADVICE r5): a shape the streamed
        # head does not cover goes through the per-layer operators right here
        if lazy and ops.head_stream_mode(int(x2.shape[0]), cin, num_output_channels, n_points) is not None:
            from ..variables import current_scope
            beta, gamma, ema_mean, ema_var = _bn_variables("bn", num_output_channels)
            decay = 0.9 if bn_decay is None else (bn_decay if torch.is_tensor(bn_decay) else float(bn_decay))
            here = current_scope()

            def on_stats(mean, var, z5, rn):
                _ema_update(ema_mean, mean, decay)
                _ema_update(ema_var, var, decay)
                if RELU_MASK_TAPS is not None:      # (test hook: the feature map is never written -- re-form the BatchNorm output from z5)
                    if z5.dtype == torch.bfloat16:
                        RELU_MASK_TAPS[here] = ops.expand16(z5, (mean, var, gamma, beta, 1e-3), None) > 0
                    else:
                        RELU_MASK_TAPS[here] = ops.bn_apply_train(z5, mean, var, gamma, beta, 1e-3, True) > 0
                if VALUE_TAPS is not None:
                    VALUE_TAPS[here] = ops.expand16(z5) if z5.dtype == torch.bfloat16 else z5.detach().clone()

            here_scope = here

            def materialize():      # (a consumer that cannot stream after all: the same layer through the per-layer operators, in this scope)
                from ..variables import absolute_scope
                with absolute_scope(here_scope):
                    return eager()

            return ops.LazyConv5Features(x2, w.reshape(cin, num_output_channels), b, gamma, beta, 1e-3, on_stats,
                                         x_bf16=getattr(inputs, "_epc_bf16", None), materialize=materialize)
        return eager()


def _conv5_l2_normalized_eager(x2, w, b, cin, num_output_channels, bn_decay):
    """conv1d_l2_normalized's layer through the per-layer operators (called inside the layer's variable scope)."""
    from .. import ops
    if ops.fused_linear_bn_ok(int(x2.shape[0]), cin, num_output_channels):
        f = _bn_train_fused(x2, w.reshape(cin, num_output_channels), b, "bn", bn_decay, True, rownorm=True,
                            f16x3=ops.F16X3_CONV5,      # conv5: BatchNorm'd block outputs against its weights
                            link=ops.TailLink() if ops.FUSE_TAIL_BACKWARD else None)
        _tap_relu_mask(f)          # (the row norm is a positive factor: f > 0 exactly where the ReLU's output is)
        return f
    z = ops.Linear.apply(x2, w.reshape(cin, num_output_channels), b, True)
    beta, gamma, ema_mean, ema_var = _bn_variables("bn", num_output_channels)
    f, mean, var = ops.BatchNormReluRowNorm.apply(z, gamma, beta, 1e-3)
    decay = 0.9 if bn_decay is None else (bn_decay if torch.is_tensor(bn_decay) else float(bn_decay))
    _ema_update(ema_mean, mean, decay)
    _ema_update(ema_var, var, decay)
    _tap_relu_mask(f)
    return f


def fully_connected(inputs, num_outputs, scope, use_xavier=True, stddev=1e-3, weight_decay=0.0,
                    activation_fn=relu, bn=False, bn_decay=None, is_training=None):
    """utils/tf_util.py:310-346: (B, Cin) -> (B, num_outputs), BN over axis 0, ReLU by default."""
    L.require_gpu()
    cin = int(inputs.shape[-1])
    w, b, _ = declare_fully_connected(scope, cin, num_outputs, use_xavier, stddev, bn)
    with variable_scope(scope):
        return _dense(inputs.reshape(-1, cin), w, b, bn, "bn", activation_fn, bn_decay, is_training)


def max_pool2d(inputs, kernel_size, scope, stride=[2, 2], padding='VALID'):
    """utils/tf_util.py:349-372.  EPC-Net-L's only use is the global max over the N points of a (B,N,1,C) map
    (models/epc-net-l.py:88-92); that is the one configuration implemented."""
    if inputs.dim() != 4 or list(kernel_size) != [int(inputs.shape[1]), int(inputs.shape[2])] or padding != 'VALID':
        raise NotImplementedError("max_pool2d is implemented for the global pool of models/epc-net-l.py:91 only")
    from .. import ops
    B, N, W, C = (int(v) for v in inputs.shape)
    return ops.MaxPoolPoints.apply(inputs.reshape(B, N * W, C)).reshape(B, 1, 1, C)


def _unused(name, where):
    def f(*a, **k):
        raise NotImplementedError("%s (reference %s) is never called by EPC-Net / EPC-Net-L and is out of the "
                                  "hot-path scope (SURVEY.md 2.1 #3)" % (name, where))
    f.__name__ = name
    return f


conv2d = _unused("conv2d", "utils/tf_util.py:111-168")
conv2d_transpose = _unused("conv2d_transpose", "utils/tf_util.py:171-248")
conv3d = _unused("conv3d", "utils/tf_util.py:251-308")
avg_pool2d = _unused("avg_pool2d", "utils/tf_util.py:374-396")
max_pool3d = _unused("max_pool3d", "utils/tf_util.py:399-422")
avg_pool3d = _unused("avg_pool3d", "utils/tf_util.py:424-447")
dropout = _unused("dropout", "utils/tf_util.py:553-574")
pairwise_distance = _unused("pairwise_distance", "utils/tf_util.py:577-596")
knn = _unused("knn", "utils/tf_util.py:599-610")
get_edge_feature = _unused("get_edge_feature", "utils/tf_util.py:613-645")
