"""EPC-Net model module with the reference's surface (reference ``models/epc-net.py``): ``placeholder_inputs``,
``forward`` and the loss family.  Load it like the reference does: ``importlib.import_module`` (train.py:81).

``forward`` builds nothing symbolic: it makes sure the variables exist (reference names / shapes / initialisers)
and runs the fused MI355X pipeline of libepcnet_hip.so:
    kNN index (utils/tf_util.py:647-666) -> conv1 -> 4 x ProxyConv block (:66-132) -> conv5 + L2 + soft assignment
    (:136-148, loupe.py:249-272) -> VLAD aggregation (loupe.py:276-292) -> head (loupe.py:295-331, :153).
"""
from __future__ import annotations

from .. import loupe as lp
from ..utils import tf_util
from ..variables import variable_scope
from ._common import *  # noqa: F401,F403  (loss family, placeholder_inputs)
from ._common import LOSS_NAMES, engine_for, placeholder_inputs  # noqa: F401

ARCH = "epc-net"


def declare_variables(params, num_points, backbone_scope='fastdgcnn'):
    """Create every variable ``forward`` owns, in the reference's creation order (models/epc-net.py:62-149)."""
    input_dim = params["INPUT_DIM"]
    with variable_scope(backbone_scope):
        tf_util.declare_conv1d('conv1', input_dim, 64)
        for b in (1, 2, 3, 4):
            if b > 1:
                tf_util.declare_conv1d('conv%d' % b, 64, 64)
            tf_util.declare_conv1d('conv%d_a' % b, 64, 64)
            tf_util.declare_conv1d('conv%d_b' % b, 64, 64)
        tf_util.declare_conv1d('conv5', 256, 1024)
    with variable_scope('VLAD'):
        lp.G_VLAD(feature_size=1024, max_samples=num_points, cluster_size=params["CLUSTER_SIZE"],
                  output_dim=params["FEATURE_OUTPUT_DIM"], groups=params["GROUPS"], gating=True,
                  add_batch_norm=True, is_training=False).declare_variables()


def forward(point_cloud, is_training, bn_decay=None, params=None):
    """models/epc-net.py:29-157.
    INPUT : batch_num_queries X num_pointclouds_per_query X num_points_per_pointcloud X input_dim
    OUTPUT: batch_num_queries X num_pointclouds_per_query X output_dim ("last_output")."""
    if params is None:
        raise TypeError("forward() needs the config dict as `params` (models/epc-net.py:37-40)")
    if point_cloud.dim() != 4:
        raise ValueError("point_cloud must be (B, P, N, INPUT_DIM), got %s" % (tuple(point_cloud.shape),))
    batch_num_queries, num_pointclouds_per_query, num_points, dim = (int(s) for s in point_cloud.shape)
    CLUSTER_SIZE = params["CLUSTER_SIZE"]        # noqa: F841  default: 64
    OUTPUT_DIM = params["FEATURE_OUTPUT_DIM"]    # default: 256
    INPUT_DIM = params["INPUT_DIM"]
    if dim != INPUT_DIM:
        raise ValueError("last dimension %d != INPUT_DIM %d (reshape at models/epc-net.py:41)" % (dim, INPUT_DIM))
    declare_variables(params, num_points)
    pc = point_cloud.reshape(batch_num_queries * num_pointclouds_per_query, num_points, INPUT_DIM)
    if is_training:
        output = forward_ops(pc, True, bn_decay, params)
    else:
        output = engine_for(ARCH, params).forward(pc)
    return output.reshape(batch_num_queries, num_pointclouds_per_query, OUTPUT_DIM)


def forward_ops(point_cloud, is_training, bn_decay, params, backbone_scope='fastdgcnn', return_features=False):
    """The same graph built op by op from the differentiable operators, line for line as models/epc-net.py:62-155.
    Used for is_training=True (batch statistics, EMA updates, gradients); with is_training=False it is an
    independent (unfused) second implementation of the inference path (tests cross-check the two).
    ``backbone_scope`` / ``return_features``: the KD variants (models/kd_epc-net.py).  NOTE the rows of the returned
    point features follow the Morton order of each cloud (a per-cloud permutation of the input order); teacher and
    student of a distillation step sort the same clouds the same way, so their rows correspond."""
    import torch
    from .. import ops
    num_points = int(point_cloud.shape[1])
    k = params["KNN"]
    point_cloud = ops.morton_sort(point_cloud)           # re-ordering only (permutation-invariant network)
    with variable_scope(backbone_scope):
        dpist = ops.KnnGraph(point_cloud)                    # tf_util.pairwise_distance_mask in index form (:63)
        # conv1 .. conv4_b and the concat of the four block outputs (:66-134): one fused chain in training (tf_util.proxyconv_backbone)
        x = tf_util.proxyconv_backbone(point_cloud, dpist, k, 4, bn_decay=bn_decay, is_training=is_training,
                                       head_follows=not return_features)
        # conv5 (:136-139) and the per-point l2_normalize of :147-148 (which the reference applies inside the VLAD scope)
        net = tf_util.conv1d_l2_normalized(x, 1024, 'conv5', bn_decay=bn_decay, is_training=is_training,
                                           lazy=not return_features)     # (lazy: only G_VLAD.forward below consumes it)
    with variable_scope('VLAD'):
        NetVLAD = lp.G_VLAD(feature_size=1024, max_samples=num_points, cluster_size=params["CLUSTER_SIZE"],
                            output_dim=params["FEATURE_OUTPUT_DIM"], groups=params["GROUPS"], gating=True,
                            add_batch_norm=True, is_training=is_training)
        output = NetVLAD.forward(net)
        output = ops.RowL2Normalize.apply(output)                 # :153
    if return_features:                                      # models/kd_epc-net.py:158: (normalised point features, output)
        return net, output
    return output
