"""EPC-Net-L model module (reference ``models/epc-net-l.py``): two ProxyConv blocks, conv5 128->1024, global max
over the points, fc1 1024->256 (+BN+ReLU), L2 normalisation."""
from __future__ import annotations

from ..utils import tf_util
from ..variables import variable_scope
from ._common import *  # noqa: F401,F403
from ._common import LOSS_NAMES, engine_for, placeholder_inputs  # noqa: F401

ARCH = "epc-net-l"


def declare_variables(params, num_points):
    """models/epc-net-l.py:44-95."""
    with variable_scope('fastdgcnn'):
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
    if is_training:
        raise NotImplementedError("is_training=True is the training-step milestone")
    pc = point_cloud.reshape(batch_num_queries * num_pointclouds_per_query, num_points, INPUT_DIM)
    output = engine_for(ARCH, params).forward(pc)
    return output.reshape(batch_num_queries, num_pointclouds_per_query, OUTPUT_DIM)
