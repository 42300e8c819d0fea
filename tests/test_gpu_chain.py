"""GPU tests of the fused backbone chain of the training step (csrc/train_chain.hip, ops.ProxyConvChain, tf_util.proxyconv_backbone;
models/epc-net.py:66-134 in training mode) against the per-layer operators it replaces and against float64 torch."""
import numpy as np
import pytest
import torch

import helpers as H
from helpers import O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _backbone(arch, w, pc, dev, use_chain, precision="bf16x6", upstream_seed=3):
    """cat, the moving statistics and every gradient of sum(cat * G) through tf_util.proxyconv_backbone."""
    V, tf_util, ops = H.pkg("variables"), H.pkg("utils.tf_util"), H.pkg("ops")
    st = H.make_store(arch, w, dev)
    nb = 4 if arch == "epc-net" else 2
    x = ops.morton_sort(torch.from_numpy(pc).to(dev))
    prev_chain, tf_util.USE_CHAIN = tf_util.USE_CHAIN, use_chain
    prev = ops.set_gemm_precision(precision)
    try:
        for v in st.vars.values():
            v.grad = None
        names = [k for k in st.trainable if "/fastdgcnn/" in k and "/conv5/" not in k]
        for k in names:
            st.vars[k].requires_grad_(True)
        with V.variable_scope(H.OUTER), V.variable_scope("fastdgcnn"):
            graph = ops.KnnGraph(x)
            cat = tf_util.proxyconv_backbone(x, graph, 20, nb, bn_decay=0.7, is_training=True)
        g = torch.Generator().manual_seed(upstream_seed)
        G = torch.randn(tuple(cat.shape), generator=g).to(dev)
        (cat * G).sum().backward()
        torch.cuda.synchronize()
        grads = {k: (st.vars[k].grad.detach().cpu().numpy().copy() if st.vars[k].grad is not None else None) for k in names}
        stats = {k: v.detach().cpu().numpy().copy() for k, v in st.vars.items() if k not in st.trainable and "/fastdgcnn/" in k}
    finally:
        tf_util.USE_CHAIN = prev_chain
        ops.set_gemm_precision(prev)
    return cat.detach().cpu().numpy(), grads, stats


@pytest.mark.parametrize("arch,ncl,n,kind", [("epc-net", 3, 256, "uniform"), ("epc-net-l", 5, 96, "uniform"),
                                            ("epc-net", 18, 4096, "uniform"), ("epc-net", 2, 256, "ties")])
def test_chain_equals_the_per_layer_operators(dev, arch, ncl, n, kind):
    """The fused chain against the per-layer operators (LinearBatchNormTrain / ProxyConvTail: the round-3 step) on the same weights,
    clouds and upstream gradient: the same products in the same arithmetic, only the pooling order of the batch statistics differs --
    outputs to 1e-5 of their scale, gradients to 1e-4 relative L2 (both implementations sit 5e-5 / 1e-5 from float64 torch after
    twelve normalised layers: scripts/debug_chain.py), moving statistics to 2e-6.  "ties": clouds with duplicated points
    and an all-zero (padding) cloud, whose lists overflow the 32 slots: the exact-scan branch of the forward gather and the overflow
    lists of its transpose."""
    w = O.seeded_weights(arch, 4)
    pc = O.synthetic_clouds(ncl, n, 11)
    if kind == "ties":
        pc[0, 40:120] = pc[0, 7]                      # 81 copies of one point: every one of them selects all the others
        pc[1] = 0.0                                   # evaluate.py:425-430 / train.py:834-844 padding cloud
    a = _backbone(arch, w, pc, dev, True)
    b = _backbone(arch, w, pc, dev, False)
    assert np.isfinite(a[0]).all()
    # (bars: full size -- 16 x the activations -- sees ReLU-mask flips between the two float32 implementations, each moving a few
    # gradient elements by per cents of a tensor's maximum (test_gpu_train_step.py holds both to float64 with the masks pinned); the
    # tie clouds' blocks sum 81 / 256 identical rows per point and normalise near-constant channels)
    bar_out, bar_grad = (1e-5, 1e-4) if (n < 4096 and kind == "uniform") else ((5e-5, 2e-2) if kind == "uniform" else (1e-4, 2e-2))
    assert np.abs(a[0] - b[0]).max() <= bar_out * max(np.abs(b[0]).max(), 1.0)
    worst = (0.0, "")
    for k, gb in b[1].items():
        ga = a[1][k]
        if gb is None or k.endswith("/biases"):
            assert ga is None or np.abs(ga).max() <= 1e-4      # exactly zero in front of a training-mode BatchNorm
            continue
        rel = np.linalg.norm(ga - gb) / max(np.linalg.norm(gb), 1e-30)
        worst = max(worst, (rel, k))
        assert rel <= bar_grad, (k, rel)
    for k, vb in b[2].items():
        assert np.abs(a[2][k] - vb).max() <= 2e-6 + (2e-6 if kind == "uniform" else 5e-5) * np.abs(vb).max(), k
    print("chain vs per-layer operators, %s %dx%d (%s): cat max diff %.2e, worst gradient rel L2 %.2e (%s)"
          % (arch, ncl, n, kind, np.abs(a[0] - b[0]).max(), worst[0], worst[1]))


def test_chain_is_bit_reproducible(dev):
    w = O.seeded_weights("epc-net", 4)
    pc = O.synthetic_clouds(6, 1024, 5)
    a = _backbone("epc-net", w, pc, dev, True)
    b = _backbone("epc-net", w, pc, dev, True)
    assert np.array_equal(a[0], b[0])
    for k in a[1]:
        assert (a[1][k] is None and b[1][k] is None) or np.array_equal(a[1][k], b[1][k]), k


@pytest.mark.parametrize("arch,ncl,n", [("epc-net-l", 4, 256), ("epc-net", 18, 4096)])
def test_chain_bf16_arithmetic_tracks_the_f32_accurate_one(dev, arch, ncl, n):
    """pieces = 1 (one bf16 value per operand; the chain's stored tensors stay float32): the same chain, every product at 2^-9 per
    operand -- outputs within bf16's error of the f32-accurate chain at one tile per workgroup and at the training tuple's nine (5.5e-3
    and 1.3e-2 relative L2 measured; the exact comparison of this arithmetic is against the oracle with the same rounding points:
    test_gpu_train_step).  The bar is also what rejected STORING the chain's activations as bf16 (round 5: 1.4e-2 and 8.6e-2 -- the
    neighbour difference xm - x cancels, and a rounded z0 goes through it -- for 0.03 ms of a 2.1-ms step: DESIGN.md)."""
    w = O.seeded_weights(arch, 4)
    pc = O.synthetic_clouds(ncl, n, 5)
    a = _backbone(arch, w, pc, dev, True)
    b = _backbone(arch, w, pc, dev, True, precision="bf16")
    rel = np.linalg.norm(a[0] - b[0]) / np.linalg.norm(a[0])
    print("bf16-operand chain vs f32-accurate chain, %s %dx%d: cat rel L2 %.2e" % (arch, ncl, n, rel))
    assert 1e-5 < rel < 3e-2, rel
    for k, va in a[2].items():
        assert np.abs(b[2][k] - va).max() <= 2e-2 * max(np.abs(va).max(), 1e-3), k
