"""NetVLAD family with the reference's class / method surface (reference ``loupe.py``).

``G_VLAD`` is the pooling EPC-Net uses (``models/epc-net.py:143-149``); ``NetVLAD`` exists in the reference but no
model instantiates it.  Constructing a class and calling ``declare_variables()`` (or ``forward``) creates the
reference's variables in the current scope: cluster_weights [F,C], cluster_bn/*, cluster_weights2 [1,F,C],
hidden1_weights [C*F/G, O], bn/*, gating_weights [O,O], gating_bn/* (``loupe.py:75-79, 249-316``).

``forward`` runs op by op on the differentiable HIP operators of ``epc-net_amd/ops.py`` (GEMMs, batch-norm, softmax,
row norms, VLAD aggregation); it is what ``models/epc-net.py: forward(is_training=True)`` uses.  The inference fast
path fuses the same computation into ``epc_conv5_assign_fwd`` / ``epc_vlad_aggregate_fwd`` / ``epc_vlad_head_fwd``.
The per-descriptor tail (B x 65536 and smaller: intra-normalisation, flatten, group reshape, sigmoid gate) is a
handful of torch tensor ops on B-row tensors.
"""
from __future__ import annotations

import math

import torch

from .variables import constant, default_store, random_normal, scoped, variable_scope

SLIM_DECAY = 0.999   # slim.batch_norm default decay
BN_EPS = 1e-3        # slim.batch_norm default epsilon


def _l2_normalize(x, dim):
    """tf.nn.l2_normalize: x * rsqrt(max(sum(x^2), 1e-12))."""
    return x * torch.rsqrt(torch.clamp((x * x).sum(dim=dim, keepdim=True), min=1e-12))


def _slim_batch_norm(x2d, scope, is_training, fused):
    """slim.batch_norm / tf.contrib.layers.batch_norm on (rows, C) (loupe.py:82-87, 257-263, 323): batch statistics
    in training (population variance normalises; the FUSED op feeds the Bessel-corrected variance to the moving
    average), moving statistics otherwise.  Moving averages are updated in place with decay 0.999 (UPDATE_OPS run
    with the train op, train.py:275-277)."""
    from . import ops
    from .utils.tf_util import _ema_update
    beta, gamma, mm, mv = _slim_bn_variables(scope, int(x2d.shape[1]))
    if is_training:
        y, mean, var = ops.BatchNormTrain.apply(x2d, gamma, beta, BN_EPS, 0)
        rows = int(x2d.shape[0])
        _ema_update(mm, mean, SLIM_DECAY, scheduled=False)
        _ema_update(mv, var * (rows / max(rows - 1, 1)) if fused else var, SLIM_DECAY, scheduled=False)
        return y
    return ops.bn_inference(x2d, mm, mv, gamma, beta, BN_EPS, False)


def _slim_bn_variables(scope: str, n: int):
    """slim.batch_norm variables (beta, gamma, moving_mean 0, moving_variance 1)."""
    st = default_store()
    with variable_scope(scope) as full:
        return (st.get_variable(full + "/beta", (n,), constant(0.0)),
                st.get_variable(full + "/gamma", (n,), constant(1.0)),
                st.get_variable(full + "/moving_mean", (n,), constant(0.0), trainable=False),
                st.get_variable(full + "/moving_variance", (n,), constant(1.0), trainable=False))


class PoolingBaseModel(object):
    """loupe.py:34-101."""

    def __init__(self, feature_size, max_samples, cluster_size, output_dim, gating=True, add_batch_norm=True,
                 is_training=True):
        self.feature_size = feature_size
        self.max_samples = max_samples
        self.output_dim = output_dim
        self.is_training = is_training
        self.gating = gating
        self.add_batch_norm = add_batch_norm
        self.cluster_size = cluster_size

    def forward(self, reshaped_input):
        raise NotImplementedError("Models should implement the forward pass.")

    def _declare_gating(self, input_dim):
        st = default_store()
        st.get_variable(scoped("gating_weights"), (input_dim, input_dim), random_normal(1 / math.sqrt(input_dim)))
        if self.add_batch_norm:
            _slim_bn_variables("gating_bn", input_dim)
        else:
            # loupe.py:88-92 builds gating_biases with `initializer=tf.random_normal(stddev=...)` -- the FUNCTION called without
            # its `shape`, not the initializer class -- so the reference raises this TypeError when the graph is built
            raise TypeError("random_normal() missing 1 required positional argument: 'shape'")

    def context_gating(self, input_layer):
        """loupe.py:61-101: input * sigmoid(BN(input @ gating_weights))."""
        from . import ops
        dim = int(input_layer.shape[1])
        self._declare_gating(dim)
        w = default_store().vars[scoped("gating_weights")]
        gates = ops.Linear.apply(input_layer, w, None)
        gates = _slim_batch_norm(gates, "gating_bn", self.is_training, fused=True)
        return ops.GateMul.apply(input_layer, gates)


class _VladBase(PoolingBaseModel):
    groups = 1

    def declare_variables(self):
        st = default_store()
        F, C, O, G = self.feature_size, self.cluster_size, self.output_dim, self.groups
        st.get_variable(scoped("cluster_weights"), (F, C), random_normal(1 / math.sqrt(F)))
        if self.add_batch_norm:
            _slim_bn_variables("cluster_bn", C)
        else:                                                                          # loupe.py:264-270
            st.get_variable(scoped("cluster_biases"), (C,), random_normal(1 / math.sqrt(F)))
        st.get_variable(scoped("cluster_weights2"), (1, F, C), random_normal(1 / math.sqrt(F)))
        st.get_variable(scoped("hidden1_weights"), (C * F // G, O), random_normal(1 / math.sqrt(C)))
        _slim_bn_variables("bn", O)
        if self.gating:
            self._declare_gating(O)

    def forward(self, reshaped_input):
        """loupe.py:121-214 (NetVLAD) / :233-333 (G_VLAD): (B*max_samples, F) -> (B, output_dim)."""
        from . import ops
        self.declare_variables()
        st = default_store().vars
        F, C, O, G, N = self.feature_size, self.cluster_size, self.output_dim, self.groups, self.max_samples
        if C != 64:
            raise NotImplementedError("cluster_size must be 64 (configs/*.yaml CLUSTER_SIZE)")
        x = reshaped_input.reshape(-1, F)
        if isinstance(x, ops.LazyConv5Features):
            # conv5 arrives un-evaluated: conv5 + l2 norm + :255-291 as ONE node of streaming kernels (ops.Conv5VladHead)
            mode = ops.head_stream_mode(x.shape[0], 256, F, N) if (self.add_batch_norm and self.is_training and F == 1024) else None
            if mode is None:      # (a consumer the producer did not foresee: the layer through the per-layer operators after all)
                return self.forward(x.materialize())
            from .utils.tf_util import _ema_update
            beta, gamma, mm, mv = _slim_bn_variables("cluster_bn", C)
            vlad, a_sum, mean5, var5, mean_c, var_c, z5, rn = ops.Conv5VladHead.apply(
                x.x, x.W, x.b, x.gamma, x.beta, x.eps, st[scoped("cluster_weights")], gamma, beta, BN_EPS, N, mode, x.x_bf16)
            x.on_stats(mean5, var5, z5, rn)
            _ema_update(mm, mean_c, SLIM_DECAY, scheduled=False)
            _ema_update(mv, var_c, SLIM_DECAY, scheduled=False)
        elif self.add_batch_norm and self.is_training:
            # :255-263, :272-274, :286-291 as one autograd node (the two gradients of x are one product: ops.VladAssignAggregate)
            from .utils.tf_util import _ema_update
            beta, gamma, mm, mv = _slim_bn_variables("cluster_bn", C)
            link = ops.tail_link_of(reshaped_input) if tuple(reshaped_input.shape) == tuple(x.shape) else None
            vlad, a_sum, mean, var = ops.VladAssignAggregate.apply(x, st[scoped("cluster_weights")], gamma, beta, BN_EPS, N,
                                                                   link)   # a_sum: :276
            _ema_update(mm, mean, SLIM_DECAY, scheduled=False)
            _ema_update(mv, var, SLIM_DECAY, scheduled=False)          # (the unfused slim op: population variance)
        else:
            if self.add_batch_norm:
                activation = ops.Linear.apply(x, st[scoped("cluster_weights")], None)                    # :255
                activation = _slim_batch_norm(activation, "cluster_bn", self.is_training, fused=False)   # :257-263
            else:
                activation = ops.Linear.apply(x, st[scoped("cluster_weights")], st[scoped("cluster_biases")])   # :264-270
            activation = ops.Softmax64.apply(activation)                                             # :272
            activation = activation.reshape(-1, N, C)                                                # :274
            a_sum = activation.sum(dim=-2, keepdim=True)                                             # :276
            vlad = ops.VladAggregate.apply(x.reshape(-1, N, F), activation)                          # :286-291 (B,F,C)
        # a = a_sum * cluster_weights2 (:284); vlad - a (:292); l2_normalize over F (:295); flatten, l2_normalize (:297-298)
        vlad = ops.VladNormalize.apply(vlad, a_sum, st[scoped("cluster_weights2")])
        vlad = vlad.reshape(-1, C * F)
        vlad = vlad.reshape(-1, C * F // G)                                                      # :302 (groups)
        if self.is_training and ops.hidden_proj_ok(int(vlad.shape[0]), C * F // G, O):
            vlad = ops.HiddenProjection.apply(vlad, st[scoped("hidden1_weights")])               # :322 (one pass over the weights)
        else:
            vlad = ops.Linear.apply(vlad, st[scoped("hidden1_weights")], None)                   # :322
        if self.is_training and self.gating and self.add_batch_norm and ops.hidden_tail_ok(int(vlad.shape[0]), G, O):
            # :323-331 (+ :61-101) as ONE node: BatchNorm, the group sum, context gating
            from .utils.tf_util import _ema_update
            beta1, gamma1, mm1, mv1 = _slim_bn_variables("bn", O)
            self._declare_gating(O)
            beta2, gamma2, mm2, mv2 = _slim_bn_variables("gating_bn", O)
            out, mean1, var1u, mean2, var2u = ops.HiddenTail.apply(vlad, gamma1, beta1, G, st[scoped("gating_weights")], gamma2, beta2,
                                                                   BN_EPS)
            for mm, mv, mean, varu in ((mm1, mv1, mean1, var1u), (mm2, mv2, mean2, var2u)):
                _ema_update(mm, mean, SLIM_DECAY, scheduled=False)
                _ema_update(mv, varu, SLIM_DECAY, scheduled=False)      # (the fused slim op: the Bessel-corrected batch variance)
            return out
        vlad = _slim_batch_norm(vlad, "bn", self.is_training, fused=True)                        # :323
        if G > 1:
            vlad = ops.GroupSum.apply(vlad, G)                                                   # :326-328
        if self.gating:
            vlad = self.context_gating(vlad)                                                     # :330-331
        return vlad


class NetVLAD(_VladBase):
    """loupe.py:103-214 (ungrouped)."""

    def __init__(self, feature_size, max_samples, cluster_size, output_dim, gating=True, add_batch_norm=True,
                 is_training=True):
        super().__init__(feature_size=feature_size, max_samples=max_samples, cluster_size=cluster_size,
                         output_dim=output_dim, gating=gating, add_batch_norm=add_batch_norm,
                         is_training=is_training)


class G_VLAD(_VladBase):
    """loupe.py:216-333 (grouped hidden projection with one shared weight)."""

    def __init__(self, feature_size, max_samples, cluster_size, output_dim, groups=4, gating=True,
                 add_batch_norm=True, is_training=True):
        super().__init__(feature_size=feature_size, max_samples=max_samples, cluster_size=cluster_size,
                         output_dim=output_dim, gating=gating, add_batch_norm=add_batch_norm,
                         is_training=is_training)
        self.groups = groups
