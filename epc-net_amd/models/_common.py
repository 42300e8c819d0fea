"""Shared by models/epc-net.py and models/epc-net-l.py: the loss family (identical in both reference files,
``models/epc-net.py:160-284`` == ``models/epc-net-l.py:105-234``) and the engine cache."""
from __future__ import annotations

import torch

from ..engine import InferenceEngine
from ..variables import default_store, outer_scope

_engines = {}


def engine_for(arch: str, params: dict, backbone_scope: str = "fastdgcnn") -> InferenceEngine:
    st = default_store()
    key = (id(st), arch, outer_scope(), backbone_scope, tuple(sorted((k, v) for k, v in (params or {}).items()
                                                       if k in ("CLUSTER_SIZE", "FEATURE_OUTPUT_DIM", "KNN", "INPUT_DIM", "GROUPS", "PRECISION"))))
    eng = _engines.get(key)
    if eng is None:
        eng = InferenceEngine(arch, params, st, outer=outer_scope(), backbone_scope=backbone_scope)
        _engines[key] = eng
    return eng


def placeholder_inputs(batch_num_queries, num_pointclouds_per_query, num_point, input_dim=13):
    """models/epc-net.py:24-26.  Eager analogue of a placeholder: an (uninitialised) device tensor of that
    shape which the caller fills (``feed_dict``) before ``forward``."""
    return torch.empty((batch_num_queries, num_pointclouds_per_query, num_point, input_dim), dtype=torch.float32,
                       device=default_store().device)


# ---- losses (tiny: B x (1+P+N+1) descriptors; torch tensor ops = plumbing-sized work, K17 in SURVEY.md 2.3) ----
def best_pos_distance(query, pos_vecs):
    """:160-167."""
    num_pos = pos_vecs.shape[1]
    query_copies = query.repeat(1, int(num_pos), 1)
    return ((pos_vecs - query_copies) ** 2).sum(2).min(1).values


def _neg_terms(q_vec, pos_vecs, neg_vecs, anchor, margin):
    best_pos = best_pos_distance(q_vec, pos_vecs)
    num_neg = neg_vecs.shape[1]
    copies = anchor.repeat(1, int(num_neg), 1)
    best_pos = best_pos.reshape(-1, 1).repeat(1, int(num_neg))
    d = ((neg_vecs - copies) ** 2).sum(2)
    return torch.clamp(margin + (best_pos - d), min=0.0)


def triplet_loss(q_vec, pos_vecs, neg_vecs, margin):
    """:174-183."""
    return _neg_terms(q_vec, pos_vecs, neg_vecs, q_vec, margin).sum(1).mean()


def lazy_triplet_loss(q_vec, pos_vecs, neg_vecs, margin):
    """:186-194."""
    return _neg_terms(q_vec, pos_vecs, neg_vecs, q_vec, margin).max(1).values.mean()


def _soft_terms(q_vec, pos_vecs, neg_vecs):
    best_pos = best_pos_distance(q_vec, pos_vecs)
    num_neg = neg_vecs.shape[1]
    copies = q_vec.repeat(1, int(num_neg), 1)
    best_pos = best_pos.reshape(-1, 1).repeat(1, int(num_neg))
    return torch.log(torch.exp(best_pos - ((neg_vecs - copies) ** 2).sum(2)) + 1.0)


def softmargin_loss(q_vec, pos_vecs, neg_vecs):
    """:197-205.  The reference returns the undefined name ``soft_los`` (:205): calling it raises NameError there,
    and so it does here (error behaviour is part of the surface)."""
    soft_loss = _soft_terms(q_vec, pos_vecs, neg_vecs).sum(1).mean()  # noqa: F841
    raise NameError("name 'soft_los' is not defined")


def lazy_softmargin_loss(q_vec, pos_vecs, neg_vecs):
    """:207-215."""
    return _soft_terms(q_vec, pos_vecs, neg_vecs).max(1).values.mean()


def quadruplet_loss_sm(q_vec, pos_vecs, neg_vecs, other_neg, m2):
    """:217-232 (inherits softmargin_loss's NameError)."""
    soft_loss = softmargin_loss(q_vec, pos_vecs, neg_vecs)
    return soft_loss + _neg_terms(q_vec, pos_vecs, neg_vecs, other_neg, m2).sum(1).mean()


def lazy_quadruplet_loss_sm(q_vec, pos_vecs, neg_vecs, other_neg, m2):
    """:234-249."""
    return lazy_softmargin_loss(q_vec, pos_vecs, neg_vecs) + \
        _neg_terms(q_vec, pos_vecs, neg_vecs, other_neg, m2).max(1).values.mean()


def quadruplet_loss(q_vec, pos_vecs, neg_vecs, other_neg, m1, m2):
    """:252-267."""
    return triplet_loss(q_vec, pos_vecs, neg_vecs, m1) + \
        _neg_terms(q_vec, pos_vecs, neg_vecs, other_neg, m2).sum(1).mean()


def lazy_quadruplet_loss(q_vec, pos_vecs, neg_vecs, other_neg, m1, m2):
    """:269-284 -- the loss train.py:264 uses.  float32 descriptors on the GPU take the one-launch operator
    (ops.LazyQuadrupletLoss: the op-by-op composition below is ~60 launches forward and backward); anything else
    (float64 checks, the other loss variants) runs the composition."""
    if q_vec.is_cuda and q_vec.dtype == torch.float32 and pos_vecs.shape[1] <= 64 and neg_vecs.shape[1] <= 64:
        from .. import ops
        return ops.LazyQuadrupletLoss.apply(q_vec, pos_vecs, neg_vecs, other_neg, float(m1), float(m2))
    return lazy_triplet_loss(q_vec, pos_vecs, neg_vecs, m1) + \
        _neg_terms(q_vec, pos_vecs, neg_vecs, other_neg, m2).max(1).values.mean()


LOSS_NAMES = ["best_pos_distance", "triplet_loss", "lazy_triplet_loss", "softmargin_loss", "lazy_softmargin_loss",
              "quadruplet_loss_sm", "lazy_quadruplet_loss_sm", "quadruplet_loss", "lazy_quadruplet_loss"]
