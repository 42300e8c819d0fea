"""GPU tests of the bf16-stored head of the training step (epc-net_amd/csrc/train_head16.hip through ops.Conv5VladHead and the C ABI)
against the float64 restatement with the same rounding points (oracle/epcnet_oracle_torch.py: _Head16; held to autograd's gradients of
the plain graph in tests/test_oracle_cpu.py).  models/epc-net.py:136-148, loupe.py:255-291 in training mode."""
import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu
EPS = 1e-3


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _inputs(B, N, seed, dev):
    g = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    R = B * N
    leaves = dict(cat=rnd(R, 256), W5=rnd(256, 1024) * 0.08, b5=rnd(1024) * 0.1, g5=1 + 0.2 * rnd(1024), bt5=0.3 * rnd(1024),
                  Wc=rnd(1024, 64) * 0.2, gc=1 + 0.2 * rnd(64), btc=0.3 * rnd(64))
    cot = (rnd(B, 1024, 64), rnd(B, 1, 64))
    return leaves, cot


def _bf16(t):
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


@pytest.mark.parametrize("B,N", [(3, 96), (5, 32), (18, 256), (2, 1024), (4, 4096)])
def test_head16_node_matches_the_rounded_restatement(dev, B, N):
    import epcnet_oracle_torch as T
    ops = H.pkg("ops")
    leaves, (wv, wa) = _inputs(B, N, 11 + N, dev)
    names = list(leaves)
    prev = ops.set_gemm_precision("bf16")
    try:
        xs = [leaves[k].float().to(dev).requires_grad_(True) for k in names]
        vlad, a_sum, mean5, var5, mean_c, var_c, z5, rn = ops.Conv5VladHead.apply(xs[0], xs[1], xs[2], xs[3], xs[4], EPS, xs[5], xs[6],
                                                                                  xs[7], EPS, N, "bf16")
        loss = (vlad * wv.float().to(dev)).sum() + (a_sum * wa.float().to(dev)).sum()
        grads = torch.autograd.grad(loss, xs, allow_unused=True)
        z5f = ops.expand16(z5).double().cpu()
        mask = (ops.expand16(z5, (mean5, var5, xs[3].detach(), xs[4].detach(), EPS), None) > 0).cpu()
        # the same call again: bit-identical results (fixed summation orders, no atomics)
        d_ = [x.detach() for x in xs]
        again = ops.Conv5VladHead.apply(d_[0], d_[1], d_[2], d_[3], d_[4], EPS, d_[5], d_[6], d_[7], EPS, N, "bf16")
        assert torch.equal(again[0], vlad) and torch.equal(again[6], z5)
    finally:
        ops.set_gemm_precision(prev)
    torch.cuda.synchronize()
    assert z5.dtype == torch.bfloat16 and tuple(z5.shape) == (B * N, 1024)

    def oracle(pin, msk):
        ys = [leaves[k].clone().requires_grad_(True) for k in names]
        out = T._Head16.apply(*ys, N, True, msk, pin, EPS)
        g = torch.autograd.grad((out[0] * wv).sum() + (out[1] * wa).sum(), ys, allow_unused=True)
        return out, g

    # 1. the stored z5 against the restatement's own: the same bf16 value except where the f32 accumulation order moved a sum across a
    #    rounding boundary (one bf16 ulp there)
    free, _ = oracle(None, None)
    z_ref = free[6]
    d = (z5f - z_ref).abs()
    assert float((d > 0).double().mean()) <= 2e-3, "z5 differs from the restatement's in %.2e of the elements" % float((d > 0).double().mean())
    # (one bf16 ulp; the absolute term: a sum that cancels to ~1e-6 carries the f32 accumulation's own error, many of ITS ulps)
    assert bool((d <= 2.0 ** -7 * z_ref.abs() + 2e-5).all()), float((d - 2.0 ** -7 * z_ref.abs()).max())
    rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max().clamp(min=1e-30))
    assert rel(mean5, free[2]) <= 5e-6 and rel(var5, free[3]) <= 5e-5, (rel(mean5, free[2]), rel(var5, free[3]))
    # 2. continued from the stored z5 and the forward's masks: outputs, statistics, every gradient
    pinned, g_ref = oracle(z5f, mask)
    u = pinned[7]
    rn_ref = torch.rsqrt(torch.clamp((u * u).sum(1), min=1e-12))
    fwd = dict(vlad=rel(vlad.detach(), pinned[0]), a_sum=rel(a_sum.detach(), pinned[1]), mean_c=rel(mean_c, pinned[4]),
               var_c=rel(var_c, pinned[5]), rn=rel(rn, rn_ref))
    print("head16 %dx%d forward vs the rounded restatement (max error / max magnitude): %s" % (
        B, N, ", ".join("%s %.1e" % kv for kv in fwd.items())))
    # (an operand u that sits on a bf16 rounding boundary may round the other way: 2^-9 of single products)
    assert fwd["vlad"] <= 1e-3 and fwd["a_sum"] <= 1e-3 and fwd["mean_c"] <= 1e-3 and fwd["var_c"] <= 1e-3 and fwd["rn"] <= 1e-5, fwd
    worst = (0.0, "")
    for k, g, gr in zip(names, grads, g_ref):
        if k == "b5":
            assert g is None                          # exactly zero in front of a training-mode BatchNorm: not computed
            continue
        e = float((g.double().cpu() - gr).norm() / gr.norm().clamp(min=1e-30))
        worst = max(worst, (e, k))
    print("head16 %dx%d: worst gradient relative L2 error vs the rounded restatement %.2e (%s)" % (B, N, worst[0], worst[1]))
    # (what is left: f32 accumulation order, and du / dz5 elements that round the other way -- 2^-9 of single elements)
    assert worst[0] <= 3e-3, worst


@pytest.mark.parametrize("B,N", [(3, 96), (18, 256), (4, 4096)])
def test_head32_node_matches_the_exact_graph(dev, B, N):
    """The same node in the default f32-accurate arithmetic (csrc/train_head32.hip: f32 tensors, split products) against the EXACT
    function -- _Head16 with the rounding off, i.e. the plain graph in float64 -- with the ReLU mask of the HIP forward pinned: outputs to
    1e-5, every gradient to 1e-4 relative L2 (three-product backward GEMMs: 2^-16 per product, averaged over thousands of terms)."""
    import epcnet_oracle_torch as T
    ops = H.pkg("ops")
    leaves, (wv, wa) = _inputs(B, N, 31 + N, dev)
    names = list(leaves)
    xs = [leaves[k].float().to(dev).requires_grad_(True) for k in names]
    vlad, a_sum, mean5, var5, mean_c, var_c, z5, rn = ops.Conv5VladHead.apply(xs[0], xs[1], xs[2], xs[3], xs[4], EPS, xs[5], xs[6],
                                                                              xs[7], EPS, N, "f32")
    loss = (vlad * wv.float().to(dev)).sum() + (a_sum * wa.float().to(dev)).sum()
    grads = torch.autograd.grad(loss, xs, allow_unused=True)
    assert z5.dtype == torch.float32
    mask = (ops.bn_apply_train(z5, mean5, var5, xs[3].detach(), xs[4].detach(), EPS, True) > 0).cpu()
    d_ = [x.detach() for x in xs]
    again = ops.Conv5VladHead.apply(d_[0], d_[1], d_[2], d_[3], d_[4], EPS, d_[5], d_[6], d_[7], EPS, N, "f32")
    assert torch.equal(again[0], vlad) and torch.equal(again[6], z5)
    torch.cuda.synchronize()
    ys = [leaves[k].clone().requires_grad_(True) for k in names]
    out = T._Head16.apply(*ys, N, False, mask, None, EPS)
    g_ref = torch.autograd.grad((out[0] * wv).sum() + (out[1] * wa).sum(), ys, allow_unused=True)
    rel = lambda a, b: float((a.detach().double().cpu() - b).abs().max() / b.abs().max().clamp(min=1e-30))
    u = out[7]
    rn_ref = torch.rsqrt(torch.clamp((u * u).sum(1), min=1e-12))
    fwd = dict(z5=rel(z5, out[6]), mean5=rel(mean5, out[2]), var5=rel(var5, out[3]), vlad=rel(vlad, out[0]), a_sum=rel(a_sum, out[1]),
               mean_c=rel(mean_c, out[4]), var_c=rel(var_c, out[5]), rn=rel(rn, rn_ref))
    print("head32 %dx%d forward vs the exact graph (max error / max magnitude): %s" % (B, N, ", ".join("%s %.1e" % kv for kv in fwd.items())))
    assert fwd["z5"] <= 2e-6 and fwd["mean5"] <= 2e-6 and fwd["var5"] <= 2e-5 and fwd["rn"] <= 2e-6, fwd
    assert fwd["vlad"] <= 2e-5 and fwd["a_sum"] <= 1e-5 and fwd["mean_c"] <= 1e-5 and fwd["var_c"] <= 2e-5, fwd
    worst = (0.0, "")
    for k, g, gr in zip(names, grads, g_ref):
        if k == "b5":
            assert g is None
            continue
        worst = max(worst, (float((g.double().cpu() - gr).norm() / gr.norm().clamp(min=1e-30)), k))
    print("head32 %dx%d: worst gradient relative L2 error vs the exact graph %.2e (%s)" % (B, N, worst[0], worst[1]))
    assert worst[0] <= 1e-4, worst


def test_gemm_b16_entry(dev):
    """epc_gemm_splitk_det_b16: C = A^T B with B stored as bf16 (dW5 = cat^T dz5's shape), slices added in a fixed order."""
    L, ops = H.pkg("lib"), H.pkg("ops")
    g = torch.Generator().manual_seed(3)
    R = 4608
    A = torch.randn(R, 256, generator=g).to(dev)
    Bf = torch.randn(R, 1024, generator=g)
    B16 = Bf.to(torch.bfloat16).to(dev)
    outs = []
    for _ in range(2):
        C = torch.empty((256, 1024), dtype=torch.float32, device=dev)
        splitk = 8
        ws = torch.empty(splitk * 256 * 1024, dtype=torch.float32, device=dev)
        L.check(L.lib().epc_gemm_splitk_det_b16(A.data_ptr(), B16.data_ptr(), C.data_ptr(), None, 256, 1024, R, 1, 256, 1024, 1, 1024, 1,
                                                0, 0, 0, splitk, 0, 1, ws.data_ptr(), ws.numel(), L.current_stream()))
        outs.append(C)
    # the head's own kernel for this product (epc_h16_conv5_dw), f32 and bf16 left operand: two implementations, one answer
    for a_in, is16 in ((A, 0), (A.to(torch.bfloat16), 1)):
        C = torch.empty((256, 1024), dtype=torch.float32, device=dev)
        nb = L.lib().epc_h16_conv5_dw_scratch_bytes(R)
        sc = torch.empty(nb, dtype=torch.uint8, device=dev)
        L.check(L.lib().epc_h16_conv5_dw(a_in.data_ptr(), is16, B16.data_ptr(), R, C.data_ptr(), sc.data_ptr(), nb, L.current_stream()))
        outs.append(C)
    torch.cuda.synchronize()
    ref = A.to(torch.bfloat16).double().cpu().t() @ B16.double().cpu()
    for C in outs:
        assert float((C.double().cpu() - ref).abs().max() / ref.abs().max()) <= 1e-5
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[2], outs[3])
    with pytest.raises(L.EpcNetError):
        L.check(L.lib().epc_gemm_splitk_det_b16(A.data_ptr(), B16.data_ptr(), C.data_ptr(), None, 32, 1024, R, 1, 256, 1024, 1, 1024, 1,
                                                0, 0, 0, 1, 0, 1, None, 0, L.current_stream()))


@pytest.mark.parametrize("R", [32, 40, 4608, 8200])
def test_conv5_weight_gradient_on_f32_rows(dev, R):
    """epc_h32_conv5_dw -- dW5 = cat^T dz5 on f32 rows, two bf16 pieces per operand, three products (models/epc-net.py:136's weight
    gradient in the default arithmetic): against the float64 product at the three-product form's accuracy, against the split-K tile
    product it replaces on this path (the same arithmetic, another order of sums), bit-identical across calls; row counts that are
    not a multiple of the 32-row step or of the slice count, operands with a wide range of magnitudes."""
    L, ops = H.pkg("lib"), H.pkg("ops")
    g = torch.Generator().manual_seed(R)
    A = (torch.randn(R, 256, generator=g) * torch.exp(2 * torch.randn(R, 1, generator=g))).to(dev)
    Bf = (torch.randn(R, 1024, generator=g) * torch.exp(torch.randn(1, 1024, generator=g))).to(dev)
    nb = L.lib().epc_h32_conv5_dw_scratch_bytes(R)
    outs = []
    for _ in range(2):
        C = torch.full((256, 1024), float("nan"), dtype=torch.float32, device=dev)
        sc = torch.full((nb,), 0xFF, dtype=torch.uint8, device=dev)         # (NaN patterns: every slice partial must be written)
        L.check(L.lib().epc_h32_conv5_dw(A.data_ptr(), Bf.data_ptr(), R, C.data_ptr(), sc.data_ptr(), nb, L.current_stream()))
        outs.append(C)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1])
    ref = A.double().cpu().t() @ Bf.double().cpu()
    scale = (A.double().cpu().abs().t() @ Bf.double().cpu().abs())           # the sum of |terms|: what a product's rounding is relative to
    err = float(((outs[0].double().cpu() - ref).abs() / scale).max())
    assert err <= 3e-5, err                                                   # 2^-16 per product before averaging over the rows
    prev = ops.set_gemm_precision("bf16x6")
    try:
        tile = ops.gemm(A, Bf, trans_a=True, splitk=max(1, min(16, R // 512)), fast=True, deterministic=True)
    finally:
        ops.set_gemm_precision(prev)
    assert float((outs[0] - tile).abs().max() / ref.abs().max()) <= 2e-5
    with pytest.raises(L.EpcNetError):
        L.check(L.lib().epc_h32_conv5_dw(A.data_ptr(), Bf.data_ptr(), R, C.data_ptr(), sc.data_ptr(), nb - 1, L.current_stream()))
    with pytest.raises(L.EpcNetError):
        L.check(L.lib().epc_h32_conv5_dw(A.data_ptr(), Bf.data_ptr(), 0, C.data_ptr(), sc.data_ptr(), nb, L.current_stream()))


def test_expand16_and_argument_checks(dev):
    L, ops = H.pkg("lib"), H.pkg("ops")
    z = torch.randn(64, 1024, device=dev).to(torch.bfloat16)
    assert torch.equal(ops.expand16(z), z.float())
    mean, var = torch.randn(1024, device=dev), torch.rand(1024, device=dev) + 0.5
    gamma, beta = torch.randn(1024, device=dev), torch.randn(1024, device=dev)
    rn = torch.rand(64, device=dev) + 0.5
    f = ops.expand16(z, (mean, var, gamma, beta, EPS), rn)
    s = gamma.double() / torch.sqrt(var.double() + EPS)
    ref = torch.relu(z.double() * s + (beta.double() - mean.double() * s)) * rn.double()[:, None]
    assert float((f.double() - ref).abs().max()) <= 1e-5 * float(ref.abs().max())
    # rows that are not a multiple of 32 are refused, not mis-tiled
    with pytest.raises(L.EpcNetError):
        sc = torch.empty(1 << 22, dtype=torch.uint8, device=dev)
        L.check(L.lib().epc_h16_conv5_fwd(z.data_ptr(), 0, z.data_ptr(), z.data_ptr(), 48, z.data_ptr(), z.data_ptr(), z.data_ptr(),
                                          sc.data_ptr(), sc.numel(), L.current_stream()))


def test_streamed_head_equals_the_per_layer_operators_in_a_training_step(dev):
    """Two implementations of the head inside a whole EPC-Net step (default f32-accurate arithmetic): ops.HEAD_STREAM = True (the one
    node of csrc/train_head32.hip) against False (LinearBatchNormTrain with the row norm + VladAssignAggregate through the feature map):
    the same loss and the same 62 gradients to 1e-3 relative L2 -- the bar both are held to against the float64 oracle (measured: 2.5e-4
    on VLAD/cluster_bn/beta, 64 sums of cancelling terms; three- vs six-product forward GEMMs, f32 accumulation orders)."""
    from helpers import O
    TR, ops = H.pkg("training"), H.pkg("ops")
    w0 = O.seeded_weights("epc-net", 4)
    pcs = O.synthetic_clouds(18, 256, 9)
    tup = [torch.from_numpy(a).to(dev) for a in (pcs[None, :1], pcs[None, 1:3], pcs[None, 3:17], pcs[None, 17:])]
    out = {}
    for stream in (True, False):
        st = H.make_store("epc-net", w0, dev)
        ts = TR.TrainStep(dict(H.PARAMS, ARCH="epc-net", BATCH_NUM_QUERIES=1), st, outer=H.OUTER)
        grads, orig, prev = {}, ops.adam_multi, ops.HEAD_STREAM

        def spy(ws, ms, vs, gs, *a):
            for w_, g in zip(ws, gs):
                for k, t_ in st.vars.items():
                    if t_.data_ptr() == w_.data_ptr():
                        grads[k] = g.detach().double().cpu()
            return orig(ws, ms, vs, gs, *a)

        ops.adam_multi, ops.HEAD_STREAM = spy, stream
        try:
            loss, _, _ = ts.step(*tup, epoch=0)
        finally:
            ops.adam_multi, ops.HEAD_STREAM = orig, prev
        out[stream] = (float(loss), grads)
    (l1, g1), (l0, g0) = out[True], out[False]
    assert l1 == pytest.approx(l0, rel=1e-5)
    assert set(g1) == set(g0) and len(g1) == 62
    worst = (0.0, "")
    for k in g0:
        n0 = float(g0[k].norm())
        if k.endswith("/biases") or n0 <= 1e-12:
            continue
        worst = max(worst, (float((g1[k] - g0[k]).norm()) / n0, k))
    print("streamed head vs per-layer operators, 18 x 256: worst gradient relative L2 difference %.2e (%s)" % worst)
    assert worst[0] <= 1e-3, worst



@pytest.mark.parametrize("h16", [True, False])
@pytest.mark.parametrize("rows", [96, 18 * 256, 4 * 4096])
def test_dcat_product_with_the_batchnorm_backward_formed_inside(dev, h16, rows):
    """epc_h16_conv5_dx_bn / epc_h32_conv5_dx_bn -- dz5 = gamma rstd (du - dbeta / R - zhat dgamma / R) formed from du and z5 as they stream
    through dcat's product, written once, multiplied from registers -- against the two launches they replace (epc_h16_bn_bwd_apply /
    epc_bn_apply_bwd_given, then epc_h16_conv5_dx / epc_h32_conv5_dx) and against float64: utils/tf_util.py:94-106 seen from the gradient
    side, models/epc-net.py:136.  The expression inside the product folds the mean into one coefficient: dz5 within float32 rounding of
    the apply kernel's (bf16: a value on a rounding boundary may land on the neighbouring bf16), in place over du as the step calls it."""
    L = H.pkg("lib")
    lib = L.lib()
    g = torch.Generator().manual_seed(rows)
    rnd = lambda *s: torch.randn(*s, generator=g)
    st = L.current_stream()
    W5 = (rnd(256, 1024) * 0.08).to(dev)
    mean5, var5, g5, bt5 = (0.2 * rnd(1024)).to(dev), (torch.rand(1024, generator=g) + 0.5).to(dev), (1 + 0.2 * rnd(1024)).to(dev), rnd(1024).to(dev)
    dbeta, dgamma = (rnd(1024) * rows ** 0.5).to(dev), (rnd(1024) * rows ** 0.5).to(dev)
    dt = torch.bfloat16 if h16 else torch.float32
    du, z5 = rnd(rows, 1024).to(dev).to(dt), rnd(rows, 1024).to(dev).to(dt)
    sc = torch.empty(1 << 21, dtype=torch.uint8, device=dev)
    # the two launches
    dz_ref = torch.empty_like(du)
    if h16:
        L.check(lib.epc_h16_bn_bwd_apply(du.data_ptr(), z5.data_ptr(), mean5.data_ptr(), var5.data_ptr(), g5.data_ptr(), bt5.data_ptr(), EPS,
                                         dbeta.data_ptr(), dgamma.data_ptr(), rows, dz_ref.data_ptr(), st))
    else:
        L.check(lib.epc_bn_apply_bwd_given(du.data_ptr(), z5.data_ptr(), mean5.data_ptr(), var5.data_ptr(), g5.data_ptr(), bt5.data_ptr(),
                                           dbeta.data_ptr(), dgamma.data_ptr(), EPS, rows, 1024, dz_ref.data_ptr(), st))
    dcat_ref = torch.empty((rows, 256), dtype=torch.float32, device=dev)
    L.check((lib.epc_h16_conv5_dx if h16 else lib.epc_h32_conv5_dx)(dz_ref.data_ptr(), W5.data_ptr(), rows, dcat_ref.data_ptr(), sc.data_ptr(),
                                                                   sc.numel(), st))
    # the one launch, in place over du
    buf = du.clone()
    dcat = torch.empty_like(dcat_ref)
    L.check((lib.epc_h16_conv5_dx_bn if h16 else lib.epc_h32_conv5_dx_bn)(buf.data_ptr(), z5.data_ptr(), mean5.data_ptr(), var5.data_ptr(),
                                                                         g5.data_ptr(), EPS, dbeta.data_ptr(), dgamma.data_ptr(), W5.data_ptr(),
                                                                         rows, buf.data_ptr(), dcat.data_ptr(), sc.data_ptr(), sc.numel(), st))
    torch.cuda.synchronize()
    # float64 from the same stored operands
    r = torch.rsqrt(var5.double() + EPS)
    dz64 = (g5.double() * r) * (du.double() - dbeta.double() / rows - (z5.double() - mean5.double()) * r * (dgamma.double() / rows))
    scale = float(dz64.abs().max())
    tol = (2.0 ** -8 if h16 else 2e-6) * scale       # bf16: half an ulp of the largest value, and the neighbouring value on a boundary
    assert float((buf.double() - dz64).abs().max()) <= tol
    assert float((buf.double() - dz_ref.double()).abs().max()) <= tol
    if h16:
        assert float((buf != dz_ref).float().mean()) <= 2e-3          # (the folded coefficient moves a rounding boundary now and then)
    dx64 = (buf.double() if h16 else dz64) @ (W5.double().to(dt).double() if h16 else W5.double()).T
    rel = float((dcat.double() - dx64).abs().max() / dx64.abs().max())
    assert rel <= (1e-5 if h16 else 2e-5), rel      # bf16: the stored operand times bf16(W5), f32 accumulation; f32: 2^-16 per product, K = 1024
    assert float((dcat - dcat_ref).abs().max()) <= (2e-2 if h16 else 1e-4) * float(dcat_ref.abs().max())
