"""EPC-Net-L model module (reference ``models/epc-net-l.py``): two ProxyConv blocks, conv5 128->1024, global max
over the points, fc1 1024->256 (+BN+ReLU), L2 normalisation."""
from __future__ import annotations

from ..utils import tf_util
from ..variables import variable_scope
from ._common import *  # noqa: F401,F403
from ._common import LOSS_NAMES, engine_for, placeholder_inputs  # noqa: F401

ARCH = "epc-net-l"


def declare_variables(params, num_points, backbone_scope='fastdgcnn'):
    """models/epc-net-l.py:44-95."""
    with variable_scope(backbone_scope):
        tf_util.declare_conv1d('conv1', params["INPUT_DIM"], 64)
        for b in (1, 2):
            if b > 1:
                tf_util.declare_conv1d('conv%d' % b, 64, 64)
            tf_util.declare_conv1d('conv%d_a' % b, 64, 64)
            tf_util.declare_conv1d('conv%d_b' % b, 64, 64)
        tf_util.declare_conv1d('conv5', 128, 1024)
    with variable_scope('VLAD'):
        tf_util.declare_fully_connected('fc1', 1024, params["FEATURE_OUTPUT_DIM"])


def forward(point_cloud, is_training, bn_decay=None, params=None):
    """models/epc-net-l.py:29-102."""
    if params is None:
        raise TypeError("forward() needs the config dict as `params` (models/epc-net-l.py:37-40)")
    if point_cloud.dim() != 4:
        raise ValueError("point_cloud must be (B, P, N, INPUT_DIM), got %s" % (tuple(point_cloud.shape),))
    batch_num_queries, num_pointclouds_per_query, num_points, dim = (int(s) for s in point_cloud.shape)
    OUTPUT_DIM = params["FEATURE_OUTPUT_DIM"]
    INPUT_DIM = params["INPUT_DIM"]
    if dim != INPUT_DIM:
        raise ValueError("last dimension %d != INPUT_DIM %d" % (dim, INPUT_DIM))
    declare_variables(params, num_points)
    pc = point_cloud.reshape(batch_num_queries * num_pointclouds_per_query, num_points, INPUT_DIM)
    if is_training:
        output = forward_ops(pc, True, bn_decay, params)
    else:
        output = engine_for(ARCH, params).forward(pc)
    return output.reshape(batch_num_queries, num_pointclouds_per_query, OUTPUT_DIM)


def forward_ops(point_cloud, is_training, bn_decay, params, backbone_scope='fastdgcnn', return_features=False):
    """models/epc-net-l.py:44-98 op by op on the differentiable operators (training path / unfused cross-check)."""
    import torch
    from .. import loupe as lp
    from .. import ops
    num_points = int(point_cloud.shape[1])
    k = params["KNN"]
    point_cloud = ops.morton_sort(point_cloud)           # re-ordering only (permutation-invariant network)
    with variable_scope(backbone_scope):
        dpist = ops.KnnGraph(point_cloud)
        conv = lambda x, n, scope: tf_util.conv1d(x, n, 1, padding='VALID', stride=1, bn=True, is_training=is_training,
                                                  scope=scope, bn_decay=bn_decay)
        # conv1 .. conv2_b and the concat of the two block outputs (:62-83): one fused chain in training (tf_util.proxyconv_backbone)
        cat = tf_util.proxyconv_backbone(point_cloud, dpist, k, 2, bn_decay=bn_decay, is_training=is_training)
        with tf_util.bounded_operands(ops.F16X3_CONV5):                  # conv5: BatchNorm'd block outputs against its weights
            x = conv(cat, 1024, 'conv5')
        feats = x
        x = x.unsqueeze(2)                                               # :88
    with variable_scope('VLAD'):
        net = tf_util.max_pool2d(x, [num_points, 1], padding='VALID', scope='maxpool')
        net = net.reshape(-1, 1024)
        output = tf_util.fully_connected(net, params["FEATURE_OUTPUT_DIM"], bn=True, is_training=is_training,
                                         scope="fc1", bn_decay=bn_decay)
        output = ops.RowL2Normalize.apply(output)
    if return_features:                                                  # models/kd_epc-net-l.py:102
        return ops.RowL2Normalize.apply(feats.reshape(-1, 1024)), output
    return output
