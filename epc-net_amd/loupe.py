"""NetVLAD family with the reference's class / method surface (reference ``loupe.py``).

``G_VLAD`` is the pooling EPC-Net uses (``models/epc-net.py:143-149``); ``NetVLAD`` exists in the reference but no
model instantiates it.  Constructing a class and calling ``declare_variables()`` (or ``forward``) creates the
reference's variables in the current scope: cluster_weights [F,C], cluster_bn/*, cluster_weights2 [1,F,C],
hidden1_weights [C*F/G, O], bn/*, gating_weights [O,O], gating_bn/* (``loupe.py:75-79, 249-316``).

Stand-alone ``forward`` on arbitrary features runs the HIP VLAD kernels (aggregate + head) -- the conv5-fused
assignment lives in ``epc_conv5_assign_fwd``; the fused model path is ``models/epc-net.py: forward``.
"""
from __future__ import annotations

import math

from .variables import constant, default_store, random_normal, scoped, variable_scope


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
            raise NotImplementedError("add_batch_norm=False (gating_biases, loupe.py:88-93) is not used by EPC-Net")

    def context_gating(self, input_layer):
        """loupe.py:61-101: input * sigmoid(BN(input @ gating_weights))."""
        self._declare_gating(int(input_layer.shape[1]))
        raise NotImplementedError("stand-alone context_gating: fused into epc_vlad_head_fwd (loupe.py:61-101)")


class _VladBase(PoolingBaseModel):
    groups = 1

    def declare_variables(self):
        if not self.add_batch_norm:
            raise NotImplementedError("add_batch_norm=False (cluster_biases, loupe.py:264-270) is not used by EPC-Net")
        st = default_store()
        F, C, O, G = self.feature_size, self.cluster_size, self.output_dim, self.groups
        st.get_variable(scoped("cluster_weights"), (F, C), random_normal(1 / math.sqrt(F)))
        _slim_bn_variables("cluster_bn", C)
        st.get_variable(scoped("cluster_weights2"), (1, F, C), random_normal(1 / math.sqrt(F)))
        st.get_variable(scoped("hidden1_weights"), (C * F // G, O), random_normal(1 / math.sqrt(C)))
        _slim_bn_variables("bn", O)
        if self.gating:
            self._declare_gating(O)

    def forward(self, reshaped_input):
        self.declare_variables()
        raise NotImplementedError(
            "stand-alone %s.forward on external features is not built yet; use models/epc-net.py forward "
            "(conv5 + assignment + aggregation are fused in libepcnet_hip.so)" % type(self).__name__)


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
