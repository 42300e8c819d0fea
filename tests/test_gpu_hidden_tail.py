"""GPU tests of the fused VLAD tail (csrc/train_head.hip: epc_hidden_tail_fwd / _bwd, ops.HiddenTail; loupe.py:323-331 + :61-101 in training
mode) against float64 torch autograd of the same graph and against the per-op path it replaces inside G_VLAD.forward."""
import numpy as np
import pytest
import torch

import helpers as H
from helpers import O

pytestmark = pytest.mark.gpu
EPS = 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _reference(h, g1, b1, G, Wg, g2, b2, dout, round_bf16):
    """float64 restatement: slim.batch_norm (population variance) -> reduce_sum over the groups -> context gating.  round_bf16: the
    operands of the gating product and of its two backward products rounded to bf16 (a custom Function: the step's "bf16" arithmetic)."""
    rnd = (lambda t: t.to(torch.float32).to(torch.bfloat16).to(torch.float64)) if round_bf16 else (lambda t: t)

    class Prod(torch.autograd.Function):
        @staticmethod
        def forward(ctx, v, W):
            ctx.save_for_backward(v, W)
            return rnd(v) @ rnd(W)

        @staticmethod
        def backward(ctx, d):
            v, W = ctx.saved_tensors
            return rnd(d) @ rnd(W).t(), rnd(v).t() @ rnd(d)

    xs = [t.detach().double().cpu().requires_grad_(True) for t in (h, g1, b1, Wg, g2, b2)]
    hh, gg1, bb1, W, gg2, bb2 = xs
    bn = lambda z, ga, be: (z - z.mean(0)) / torch.sqrt(z.var(0, unbiased=False) + EPS) * ga + be
    y = bn(hh, gg1, bb1)
    v = y.reshape(-1, G, y.shape[1]).sum(1)
    gl = Prod.apply(v, W)
    out = v * torch.sigmoid(bn(gl, gg2, bb2))
    (out * dout.double().cpu()).sum().backward()
    stats = (hh.mean(0), hh.var(0, unbiased=True), gl.mean(0), gl.var(0, unbiased=True))
    return out.detach(), [x.grad for x in xs], [s.detach() for s in stats]


@pytest.mark.parametrize("B,G,Ow,precision", [(22, 4, 256, "bf16x6"), (18, 4, 256, "bf16x6"), (22, 4, 256, "bf16"), (5, 1, 64, "bf16x6"),
                                              (32, 2, 128, "bf16x6"), (30, 3, 64, "bf16")])
def test_hidden_tail_node_matches_float64(dev, B, G, Ow, precision):
    ops = H.pkg("ops")
    g = torch.Generator().manual_seed(B * 131 + G)
    rnd = lambda *s: torch.randn(s, generator=g)
    h = (rnd(B * G, Ow) * 0.7 + rnd(1, Ow)).to(dev)
    g1, b1 = (torch.rand(Ow, generator=g) + 0.5).to(dev), (rnd(Ow) * 0.2).to(dev)
    Wg = (rnd(Ow, Ow) / np.sqrt(Ow)).to(dev)
    g2, b2 = (torch.rand(Ow, generator=g) + 0.5).to(dev), (rnd(Ow) * 0.2).to(dev)
    dout = rnd(B, Ow).to(dev)
    prev = ops.set_gemm_precision(precision)
    try:
        xs = [t.clone().requires_grad_(True) for t in (h, g1, b1, Wg, g2, b2)]
        out, mean1, var1u, mean2, var2u = ops.HiddenTail.apply(xs[0], xs[1], xs[2], G, xs[3], xs[4], xs[5], EPS)
        (out * dout).sum().backward()
        again = ops.HiddenTail.apply(h, g1, b1, G, Wg, g2, b2, EPS)
        torch.cuda.synchronize()
    finally:
        ops.set_gemm_precision(prev)
    assert torch.equal(again[0], out)                                           # the same bits on every call
    r_out, r_grads, r_stats = _reference(h, g1, b1, G, Wg, g2, b2, dout, False)      # (at most 32 rows: float32 products in both arithmetics)
    rel = lambda a, b: float((a.detach().double().cpu() - b).norm() / b.norm().clamp(min=1e-30))
    bar = 2e-6
    names = ("h", "gamma1", "beta1", "Wg", "gamma2", "beta2")
    print("HiddenTail B=%d G=%d O=%d %s: out %.1e; gradients %s" % (B, G, Ow, precision, rel(out, r_out),
                                                                     ", ".join("%s %.1e" % (n, rel(x.grad, rg)) for n, x, rg in zip(names, xs, r_grads))))
    assert rel(out, r_out) <= bar, rel(out, r_out)
    for n, x, rg in zip(names, xs, r_grads):
        assert rel(x.grad, rg) <= 50 * bar, (n, rel(x.grad, rg))
    for got, ref in zip((mean1, var1u, mean2, var2u), r_stats):
        assert rel(got, ref) <= 10 * bar


@pytest.mark.parametrize("precision", ["bf16x6", "bf16"])
def test_g_vlad_with_the_fused_tail_equals_the_per_op_path(dev, precision):
    """loupe.G_VLAD.forward in training mode on the same weights and features with ops.HIDDEN_TAIL on and off: descriptors, every gradient and
    the four moving statistics of `bn` / `gating_bn` agree (f32-accurate arithmetic: to rounding; bf16: to the operand rounding's noise)."""
    V, ops, lp = H.pkg("variables"), H.pkg("ops"), H.pkg("loupe")
    w0 = O.seeded_weights("epc-net", 4)
    g = torch.Generator().manual_seed(5)
    feats = torch.nn.functional.normalize(torch.rand((18 * 256, 1024), generator=g), dim=1).to(dev)
    dout = torch.randn((18, 256), generator=g).to(dev)
    res = []
    for fused in (True, False):
        prev_t, ops.HIDDEN_TAIL = ops.HIDDEN_TAIL, fused
        prev = ops.set_gemm_precision(precision)
        try:
            st = H.make_store("epc-net", w0, dev)
            names = [k for k in st.trainable if "/VLAD/" in k]
            for k in names:
                st.vars[k].requires_grad_(True)
                st.vars[k].grad = None
            with V.variable_scope(H.OUTER), V.variable_scope("VLAD"):
                vl = lp.G_VLAD(feature_size=1024, max_samples=256, cluster_size=64, output_dim=256, groups=4, gating=True, add_batch_norm=True,
                               is_training=True)
                out = vl.forward(feats)
            (out * dout).sum().backward()
            torch.cuda.synchronize()
            res.append((out.detach().cpu().numpy(), {k: st.vars[k].grad.detach().cpu().numpy().copy() for k in names if st.vars[k].grad is not None},
                        {k: v.detach().cpu().numpy().copy() for k, v in st.vars.items() if "/VLAD/" in k and k not in st.trainable}))
        finally:
            ops.set_gemm_precision(prev)
            ops.HIDDEN_TAIL = prev_t
    a, b = res
    # (bf16: the node itself is float32-accurate in both; what differs upstream of it -- the 16384-wide hidden projection's dW, the VLAD
    # head's products -- sees the node's gradient through bf16-rounded operands, so its 1e-5 differences move single roundings)
    bar = 1e-5 if precision == "bf16x6" else 2e-3
    gmax = max(np.linalg.norm(v) for v in b[1].values())
    rel = lambda x, y: np.linalg.norm(x - y) / max(np.linalg.norm(y), (1e-3 if precision == "bf16x6" else 3e-2) * gmax)      # (a gradient that is noise -- cluster_bn/beta: the softmax's near-invariance -- is held against the scale of the others)
    assert rel(a[0], b[0]) <= bar
    assert set(a[1]) == set(b[1])
    print("G_VLAD fused tail vs per-op (%s): out %.1e; gradients %s" % (precision, rel(a[0], b[0]), ", ".join("%s %.1e" % (k.split("VLAD/")[1], rel(a[1][k], b[1][k])) for k in b[1])))
    # (the node itself is held to float64 at 5e-6 above; here the tensors UPSTREAM of it see its 1e-6 differences through gradients
    # formed from the differences of eighteen nearly equal descriptors -- cluster_bn/beta 1.3e-3, cluster_weights2 2.7e-4 measured in
    # the f32-accurate arithmetic: a wiring check, not a precision one)
    bar_grad = 5e-3 if precision == "bf16x6" else 1e-1
    for k in b[1]:
        assert rel(a[1][k], b[1][k]) <= bar_grad, (k, rel(a[1][k], b[1][k]))
    for k in b[2]:
        assert np.abs(a[2][k] - b[2][k]).max() <= 1e-6 + bar * np.abs(b[2][k]).max(), k


@pytest.mark.parametrize("M,K,precision", [(72, 16384, "bf16x6"), (88, 16384, "bf16x6"), (72, 16384, "bf16"), (88, 16384, "bf16"),
                                           (64, 512, "bf16x6"), (128, 1024, "bf16"), (100, 768, "bf16x6")])
def test_hidden_projection_node_matches_float64(dev, M, K, precision):
    """ops.HiddenProjection (csrc/train_hidden.hip: epc_hidden_proj_fwd / _bwd; loupe.py:322 and its two gradients) against the float64
    products -- in the "bf16" arithmetic with every product's operands rounded to bf16 first, so that what is left is accumulation
    order -- and against ops.Linear's tile GEMMs on the same inputs; bit-identical across calls; rows that do not fill a 32- or 16-row
    tile (72, 88, 100), the smallest and the largest covered shape."""
    ops, L = H.pkg("ops"), H.pkg("lib")
    assert ops.hidden_proj_ok(M, K, 256)
    g = torch.Generator().manual_seed(M + K)
    x = (torch.randn(M, K, generator=g) * torch.exp(torch.randn(M, 1, generator=g))).to(dev).requires_grad_(True)
    W = (torch.randn(K, 256, generator=g) / np.sqrt(K) * torch.exp(0.5 * torch.randn(1, 256, generator=g))).to(dev).requires_grad_(True)
    dy = torch.randn(M, 256, generator=g).to(dev)
    bf = precision == "bf16"
    rnd = (lambda t: t.to(torch.float32).to(torch.bfloat16).to(torch.float64)) if bf else (lambda t: t)
    x64, W64, d64 = x.detach().double().cpu(), W.detach().double().cpu(), dy.double().cpu()
    ref = (rnd(x64) @ rnd(W64), rnd(d64) @ rnd(W64).t(), rnd(x64).t() @ rnd(d64))
    mag = (x64.abs() @ W64.abs(), d64.abs() @ W64.abs().t(), x64.abs().t() @ d64.abs())      # the sums of |terms|
    prev = ops.set_gemm_precision(precision)
    try:
        got = []
        for _ in range(2):
            x.grad = W.grad = None
            y = ops.HiddenProjection.apply(x, W)
            y.backward(dy)
            got.append((y.detach().clone(), x.grad.clone(), W.grad.clone()))
        x.grad = W.grad = None
        y2 = ops.Linear.apply(x, W, None)
        y2.backward(dy)
        lin = (y2.detach(), x.grad, W.grad)
    finally:
        ops.set_gemm_precision(prev)
    torch.cuda.synchronize()
    for a, b in zip(got[0], got[1]):
        assert torch.equal(a, b)
    # forward: six products (2^-24 per product) or exact products of rounded operands; backward: three products (2^-16)
    bars = (2e-6, 2e-6, 2e-6) if bf else (3e-7, 3e-5, 3e-5)
    for name, a, r, m, bar, l in zip(("y", "dx", "dW"), got[0], ref, mag, bars, lin):
        err = float(((a.double().cpu() - r).abs() / m).max())
        assert err <= bar, (name, err)
        assert float((a - l).abs().max() / r.abs().max()) <= (1e-5 if bf else 2e-5), name
    # one side only; shapes that are not covered are refused
    lib = L.lib()
    dX = torch.full_like(x, float("nan"))
    L.check(lib.epc_hidden_proj_bwd(x.data_ptr(), W.data_ptr(), dy.data_ptr(), M, K, 1 if bf else 2, dX.data_ptr(), None, L.current_stream()))
    torch.cuda.synchronize()
    assert torch.equal(dX, got[0][1])
    assert not ops.hidden_proj_ok(60, K, 256) and not ops.hidden_proj_ok(132, K, 256) and not ops.hidden_proj_ok(M, K + 64, 256)
    assert not ops.hidden_proj_ok(M, K, 128) and not ops.hidden_proj_ok(66, K, 256)
    with pytest.raises(L.EpcNetError):
        L.check(lib.epc_hidden_proj_bwd(x.data_ptr(), W.data_ptr(), dy.data_ptr(), 60, K, 2, dX.data_ptr(), None, L.current_stream()))
