"""GPU tests of the training-step operators (csrc/train_ops.hip via epc-net_amd/ops.py): forward and backward against
a float64 torch-CPU autograd restatement of the same reference op (tolerances are fp32 rounding, relative to the
largest magnitude of each tensor)."""
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


def rel(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else a
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else b
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def run_pair(fn_gpu, fn_ref, inputs, dev, tol=2e-5):
    """inputs: list of float64 CPU tensors.  Compares outputs and the gradients of sum(out * r) w.r.t. every input."""
    g_in = [t.float().to(dev).requires_grad_(True) for t in inputs]
    r_in = [t.clone().requires_grad_(True) for t in inputs]
    og, orf = fn_gpu(*g_in), fn_ref(*r_in)
    assert rel(og, orf) <= tol, "forward: %.3e" % rel(og, orf)
    up = torch.randn(orf.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(7))
    (og * up.float().to(dev)).sum().backward()
    (orf * up).sum().backward()
    for i, (a, b) in enumerate(zip(g_in, r_in)):
        assert rel(a.grad, b.grad) <= tol * 5, "grad of input %d: %.3e" % (i, rel(a.grad, b.grad))


@pytest.mark.parametrize("rows,cin,cout", [(4096, 64, 64), (1000, 3, 64), (2048, 256, 1024), (72, 16384, 256), (300, 1024, 64),
                                           (333, 64, 64), (255, 64, 64),   # 64 -> 64: the specialised layer kernel (ragged / just below its threshold)
                                           (18, 256, 256), (1, 100, 40), (32, 333, 72)])   # <= 32 rows: gemm_small_m_kernel (the gating layer; ragged N, K)
def test_linear(dev, rows, cin, cout):
    ops = H.pkg("ops")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(rows, cin, dtype=torch.float64, generator=g)
    W = torch.randn(cin, cout, dtype=torch.float64, generator=g) / np.sqrt(cin)
    b = torch.randn(cout, dtype=torch.float64, generator=g)
    run_pair(lambda x, W, b: ops.Linear.apply(x, W, b), lambda x, W, b: x @ W + b, [x, W, b], dev)


@pytest.mark.parametrize("rows,cin,cout,relu,need_dx", [(4096, 64, 64, 1, True), (1000, 64, 64, 0, True), (333, 64, 64, 1, True),
                                                         (8192, 64, 64, 1, False), (2048, 256, 1024, 1, True),
                                                         (600, 64, 128, 1, True)])
def test_linear_batch_norm_train_node(dev, rows, cin, cout, relu, need_dx):
    """ops.LinearBatchNormTrain (utils/tf_util.py:94-106 in training mode as ONE autograd node; 64 -> 64 layers take the fused
    backward epc_linear_bn_bwd64, the others the operator chain) against float64: output, batch moments, and the gradients of
    x, W, gamma, beta -- twice, bit-identical for the 64 -> 64 case (its dW partials are added in a fixed order)."""
    ops = H.pkg("ops")
    g = torch.Generator().manual_seed(rows + cin)
    x = torch.randn(rows, cin, dtype=torch.float64, generator=g)
    W = torch.randn(cin, cout, dtype=torch.float64, generator=g) / np.sqrt(cin)
    b = torch.randn(cout, dtype=torch.float64, generator=g)
    gamma = torch.rand(cout, dtype=torch.float64, generator=g) + 0.5
    beta = torch.randn(cout, dtype=torch.float64, generator=g) * 0.3
    up = torch.randn(rows, cout, dtype=torch.float64, generator=g)

    def ref(x, W, gamma, beta):
        z = x @ W + b
        mean, var = z.mean(0), z.var(0, unbiased=False)
        y = gamma * (z - mean) / torch.sqrt(var + 1e-3) + beta
        return (torch.relu(y) if relu else y), mean, var

    r_in = [t.clone().requires_grad_(True) for t in (x, W, gamma, beta)]
    yr, mr, vr = ref(*r_in)
    (yr * up).sum().backward()
    runs = []
    for _ in range(2):
        xg = x.float().to(dev).requires_grad_(need_dx)
        g_in = [xg] + [t.float().to(dev).requires_grad_(True) for t in (W, gamma, beta)]
        y, mean, var = ops.LinearBatchNormTrain.apply(g_in[0], g_in[1], b.float().to(dev), g_in[2], g_in[3], 1e-3, relu, False)
        (y * up.float().to(dev)).sum().backward()
        runs.append((y.detach(), [t.grad for t in g_in]))
        assert rel(y, yr) <= 2e-5 and rel(mean, mr) <= 2e-5 and rel(var, vr) <= 2e-5
        for name, a, r in zip(("x", "W", "gamma", "beta"), g_in, r_in):
            if name == "x" and not need_dx:
                assert a.grad is None
                continue
            assert rel(a.grad, r.grad) <= 1e-4, "grad of %s: %.3e" % (name, rel(a.grad, r.grad))
    if cin == 64 and cout == 64:
        assert torch.equal(runs[0][0], runs[1][0])
        for a, bb in zip(runs[0][1], runs[1][1]):
            assert (a is None and bb is None) or torch.equal(a, bb)


@pytest.mark.parametrize("rows,cin,cout", [(5000, 64, 64), (300, 64, 64), (4096, 256, 1024), (777, 64, 128)])
def test_fused_batch_statistics_with_mean_far_above_spread(dev, rows, cin, cout):
    """the statistics epilogues (epc_linear_stats64 for the thin layers, epc_gemm_f32_stats for the others)
    must not lose the variance of a channel whose |mean| is ~10^3 standard deviations (post-ReLU inputs, trained checkpoints):
    the per-tile sums are shifted by a pivot row of the column itself and merged Chan-style.  x = a constant large offset along
    one input direction + unit noise; the bias adds another 10^3.  Variance against float64 at 1e-4 (plain sum / sum-of-squares
    in float32 is off by tens of per cent here, or negative)."""
    ops = H.pkg("ops")
    g = torch.Generator().manual_seed(rows)
    W = torch.randn(cin, cout, dtype=torch.float64, generator=g) / np.sqrt(cin)
    x = torch.randn(rows, cin, dtype=torch.float64, generator=g) + 300.0        # every output channel: mean ~ 300 * sum_k W[k][c]
    b = torch.full((cout,), 1000.0, dtype=torch.float64)
    z = x.float().double() @ W.float().double() + b.float().double()
    mean_ref, var_ref = z.mean(0), z.var(0, unbiased=False)
    assert float((mean_ref.abs() / var_ref.sqrt()).median()) > 300.0               # the regime the test is about
    zz, mean, var = ops._gemm_with_stats(x.float().to(dev).contiguous(), W.float().to(dev).contiguous(), b.float().to(dev))
    assert rel(zz, z) <= 2e-6
    assert float(((mean.double().cpu() - mean_ref).abs() / mean_ref.abs()).max()) <= 1e-6
    assert float(((var.double().cpu() - var_ref).abs() / var_ref).max()) <= 1e-4, ((var.double().cpu() - var_ref).abs() / var_ref).max()


@pytest.mark.parametrize("rows,C,relu", [(8192, 64, 1), (5000, 1024, 1), (72, 256, 0), (18, 256, 0), (4096, 64, 0)])
def test_batch_norm_train(dev, rows, C, relu):
    ops = H.pkg("ops")
    g = torch.Generator().manual_seed(1)
    z = torch.randn(rows, C, dtype=torch.float64, generator=g) * 2 + 0.5
    gamma = torch.rand(C, dtype=torch.float64, generator=g) + 0.5
    beta = torch.randn(C, dtype=torch.float64, generator=g) * 0.1

    def ref(z, gamma, beta):
        mean = z.mean(0)
        var = ((z - mean) ** 2).mean(0)
        y = (z - mean) * torch.rsqrt(var + 1e-3) * gamma + beta
        return torch.relu(y) if relu else y

    run_pair(lambda z, g_, b_: ops.BatchNormTrain.apply(z, g_, b_, 1e-3, relu)[0], ref, [z, gamma, beta], dev, tol=5e-5)
    y, mean, var = ops.BatchNormTrain.apply(z.float().to(dev), gamma.float().to(dev), beta.float().to(dev), 1e-3, relu)
    assert rel(mean, z.mean(0)) <= 1e-5 and rel(var, z.var(0, unbiased=False)) <= 1e-5


@pytest.mark.parametrize("kind,n", [("uniform", 256), ("lattice", 512), ("zeros", 64)])
def test_neighbour_mean(dev, kind, n):
    ops = H.pkg("ops")
    pc = O.synthetic_clouds(2, n, 0, kind)
    mask = torch.tensor(O.pairwise_distance_mask(pc), dtype=torch.float64)
    graph = ops.KnnGraph(torch.from_numpy(pc).to(dev))
    x = torch.randn(2, n, 64, dtype=torch.float64, generator=torch.Generator().manual_seed(2))
    run_pair(lambda x: ops.NeighbourMean.apply(x.reshape(-1, 64), graph, 20).reshape(2, n, 64),
             lambda x: torch.matmul(mask, x) / 20.0, [x], dev)


@pytest.mark.parametrize("kind,n", [("uniform", 256), ("lattice", 512), ("zeros", 64)])
def test_neighbour_mean_and_diff(dev, kind, n):
    """ops.NeighbourMeanDiff = (mask @ x / k, mask @ x / k - x) (models/epc-net.py:70-72): both outputs carry gradient
    (the block adds xm back after conv_b); lattice / zeros clouds take the overflow (cnt > cap) path."""
    ops = H.pkg("ops")
    pc = O.synthetic_clouds(2, n, 0, kind)
    mask = torch.tensor(O.pairwise_distance_mask(pc), dtype=torch.float64)
    graph = ops.KnnGraph(torch.from_numpy(pc).to(dev))
    x = torch.randn(2, n, 64, dtype=torch.float64, generator=torch.Generator().manual_seed(2))
    w = torch.randn(2, n, 64, dtype=torch.float64, generator=torch.Generator().manual_seed(3))

    def fused(x):
        xm, d = ops.NeighbourMeanDiff.apply(x.reshape(-1, 64), graph, 20)
        return (xm.reshape(2, n, 64) * 0.7 + d.reshape(2, n, 64) * w.to(x.dtype).to(x.device))

    def ref(x):
        xm = torch.matmul(mask, x) / 20.0
        return xm * 0.7 + (xm - x) * w

    run_pair(fused, ref, [x], dev)


@pytest.mark.parametrize("kind,n", [("uniform", 512), ("lattice", 512), ("uniform", 333)])
def test_proxyconv_tail_node(dev, kind, n):
    """ops.ProxyConvTail -- x1 = mask x / k; conv_b(conv_a(x1 - x)) + x1 (models/epc-net.py:70-86) as one node -- against float64:
    the output, both layers' batch moments, and the gradients of x and of both layers' weights / gamma / beta; bit-equal across two
    runs; and the op-by-op path (NeighbourMeanDiff, two LinearBatchNormTrain, add) gives the same numbers."""
    ops = H.pkg("ops")
    B = 3
    pc = O.synthetic_clouds(B, n, 1, kind)
    mask = torch.tensor(O.pairwise_distance_mask(pc), dtype=torch.float64)
    graph = ops.KnnGraph(torch.from_numpy(pc).to(dev))
    g = torch.Generator().manual_seed(n)
    x = torch.randn(B * n, 64, dtype=torch.float64, generator=g)
    mk = lambda *sh, sc=1.0: torch.randn(*sh, dtype=torch.float64, generator=g) * sc
    Wa, Wb, ba, bb = mk(64, 64, sc=0.2), mk(64, 64, sc=0.2), mk(64, sc=0.1), mk(64, sc=0.1)
    ga, gb, bta, btb = 1 + mk(64, sc=0.1), 1 + mk(64, sc=0.1), mk(64, sc=0.2), mk(64, sc=0.2)
    up = mk(B * n, 64)

    def bn_relu(z, gamma, beta):
        m, v = z.mean(0), z.var(0, unbiased=False)
        return torch.relu((z - m) * torch.rsqrt(v + 1e-3) * gamma + beta), m, v

    def ref(x, Wa, ga, bta, Wb, gb, btb):
        xm = torch.matmul(mask, x.reshape(B, n, 64)).reshape(-1, 64) / 20.0
        t, ma, va = bn_relu((xm - x) @ Wa + ba, ga, bta)
        t, mb, vb = bn_relu(t @ Wb + bb, gb, btb)
        return t + xm, ma, va, mb, vb

    ins = [x, Wa, ga, bta, Wb, gb, btb]
    r_in = [t.clone().requires_grad_(True) for t in ins]
    out_r, ma_r, va_r, mb_r, vb_r = ref(*r_in)
    (out_r * up).sum().backward()
    f32 = lambda t: t.float().to(dev)
    runs = []
    for _ in range(2):
        g_in = [f32(t).requires_grad_(True) for t in ins]
        xg, Wag, gag, btag, Wbg, gbg, btbg = g_in
        out, za, ma, va, zb, mb, vb = ops.ProxyConvTail.apply(xg, graph, 20, Wag, f32(ba), gag, btag, Wbg, f32(bb), gbg, btbg, 1e-3)
        (out * f32(up)).sum().backward()
        assert rel(out, out_r) <= 2e-5
        for a, r in ((ma, ma_r), (va, va_r), (mb, mb_r), (vb, vb_r)):
            assert rel(a, r) <= 2e-5
        for name, a, r in zip(("x", "Wa", "gamma_a", "beta_a", "Wb", "gamma_b", "beta_b"), g_in, r_in):
            assert rel(a.grad, r.grad) <= 2e-4, "grad of %s: %.3e" % (name, rel(a.grad, r.grad))
        runs.append([out.detach()] + [t.grad for t in g_in])
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    # op by op
    g_in = [f32(t).requires_grad_(True) for t in ins]
    xg, Wag, gag, btag, Wbg, gbg, btbg = g_in
    xm, d = ops.NeighbourMeanDiff.apply(xg, graph, 20)
    t, _, _ = ops.LinearBatchNormTrain.apply(d, Wag, f32(ba), gag, btag, 1e-3, 1, False)
    t, _, _ = ops.LinearBatchNormTrain.apply(t, Wbg, f32(bb), gbg, btbg, 1e-3, 1, False)
    out2 = t + xm
    (out2 * f32(up)).sum().backward()
    assert rel(out2, runs[0][0]) <= 1e-6
    for a, b in zip([t.grad for t in g_in], runs[0][1:]):
        assert rel(a, b) <= 2e-5


@pytest.mark.parametrize("which,rows,cin,cout", [("conv5", 4096, 256, 1024), ("assign", 3000, 1024, 64)])
def test_split_fp16_stats_product(dev, which, rows, cin, cout):
    """ops._gemm_with_stats in the split-fp16 three-product form (epc_gemm_f16x3_stats: conv5 and the VLAD assignment of the
    training forward) against float64: the product to 4e-6 of its largest entry and the moments to 2e-5 over four decades of
    operand magnitude, as close as the six-product form within a factor of eight.  RANGE GUARD: a value that leaves the
    scaled fp16 range (|x| >= 256 for conv5's activations), an Inf or a NaN makes the library recompute the product in the
    six-product form inside the same stream -- the outputs are then exactly that form's, NaN rows and statistics included: nothing is
    clamped to a finite wrong value."""
    ops = H.pkg("ops")
    g = torch.Generator().manual_seed(cin)
    scales = ops.F16X3_CONV5 if which == "conv5" else ops.F16X3_ASSIGN
    for amp in (1.0, 1e-2, 1e-4, 30.0):
        if which == "conv5":
            x = torch.randn(rows, cin, dtype=torch.float64, generator=g).clamp(min=-0.5) * amp      # relu-like, heavy on one side
        else:
            x = torch.randn(rows, cin, dtype=torch.float64, generator=g).clamp(min=0)
            x = x / x.norm(dim=1, keepdim=True) * min(amp, 1.0)
        W = torch.randn(cin, cout, dtype=torch.float64, generator=g) / np.sqrt(cin)
        b = torch.randn(cout, dtype=torch.float64, generator=g) * 0.1
        ref = x @ W + b
        xg, Wg, bg = x.float().to(dev), W.float().to(dev), b.float().to(dev)
        z3, m3, v3 = ops._gemm_with_stats(xg, Wg, bg, scales)
        z6, m6, v6 = ops._gemm_with_stats(xg, Wg, bg)
        ref32 = (x.float().double() @ W.float().double()) + b.float().double()     # what the rounded operands give exactly
        e3, e6 = rel(z3, ref32), rel(z6, ref32)
        assert e3 <= 4e-6 and e3 <= max(8 * e6, 5e-7), (which, amp, e3, e6)
        assert rel(m3, ref32.mean(0)) <= 2e-5 and rel(v3, ref32.var(0, unbiased=False)) <= 2e-5
        assert rel(z3, ref) <= 2e-5
    for bad in (1e9, 300.0 if which == "conv5" else 5.0, float("inf"), float("nan")):
        big = xg.clone()
        big[7, 3] = bad
        z, m, v = ops._gemm_with_stats(big, Wg, bg, scales)
        z6g, m6g, v6g = ops._gemm_with_stats(big, Wg, bg)
        same = lambda a, b: torch.equal(torch.nan_to_num(a, nan=12345.0), torch.nan_to_num(b, nan=12345.0))
        assert same(z, z6g) and same(m, m6g) and same(v, v6g), bad
        if bad != bad:
            assert bool(torch.isnan(z[7]).all()) and bool(torch.isnan(m).all())       # a NaN reaches the loss
        elif bad == 1e9:
            assert float(z[7].abs().max()) > 1e6                                       # not saturated at 255.9
    bigw = Wg.clone()
    bigw[3, 5] = 40.0                                                                  # a weight beyond 2^4: the same guard
    z, m, v = ops._gemm_with_stats(xg, bigw, bg, scales)
    z6g, m6g, v6g = ops._gemm_with_stats(xg, bigw, bg)
    assert torch.equal(z, z6g) and torch.equal(m, m6g) and torch.equal(v, v6g)
    z3b, _, _ = ops._gemm_with_stats(xg, Wg, bg, scales)                               # and the guard leaves no state behind
    assert torch.equal(z3b, z3)
    prev = ops.set_forward_f16x3(False)
    try:
        z6b, _, _ = ops._gemm_with_stats(xg, Wg, bg, scales)
        assert torch.equal(z6b, z6)
    finally:
        ops.set_forward_f16x3(prev)


def test_rownorm_and_softmax(dev):
    ops = H.pkg("ops")
    g = torch.Generator().manual_seed(3)
    x = torch.randn(3000, 1024, dtype=torch.float64, generator=g).clamp(min=0)
    x[5] = 0.0                                                          # clamp branch of l2_normalize
    run_pair(lambda x: ops.RowL2Normalize.apply(x), lambda x: x * torch.rsqrt(torch.clamp((x * x).sum(1, keepdim=True), min=1e-12)), [x], dev)
    a = torch.randn(5000, 64, dtype=torch.float64, generator=g) * 3
    run_pair(lambda a: ops.Softmax64.apply(a), lambda a: torch.softmax(a, 1), [a], dev)


@pytest.mark.parametrize("rows", [2048, 1000, 77])
def test_conv5_tail_node(dev, rows):
    """f = l2_normalize(relu(batch_norm_train(z))) over 1024 channels (models/epc-net.py:136-148) as ops.BatchNormReluRowNorm and
    as the tail of ops.LinearBatchNormTrain(rownorm=True): output, moments and the gradients (the fused two-pass backward
    epc_bn_relu_rownorm_bwd) against float64, a row whose channels are all switched off included; bit-equal across two runs."""
    ops = H.pkg("ops")
    g = torch.Generator().manual_seed(rows)
    z = torch.randn(rows, 1024, dtype=torch.float64, generator=g) * 2 + torch.randn(1024, dtype=torch.float64, generator=g)
    z[3] = -50.0                                    # relu(bn(z)) == 0 on the whole row: the l2_normalize clamp
    gamma = torch.rand(1024, dtype=torch.float64, generator=g) + 0.5
    beta = torch.randn(1024, dtype=torch.float64, generator=g) * 0.3
    up = torch.randn(rows, 1024, dtype=torch.float64, generator=g)

    def tail(z, gamma, beta):
        mean, var = z.mean(0), z.var(0, unbiased=False)
        u = torch.relu(gamma * (z - mean) / torch.sqrt(var + 1e-3) + beta)
        return u * torch.rsqrt(torch.clamp((u * u).sum(1, keepdim=True), min=1e-12)), mean, var

    r_in = [t.clone().requires_grad_(True) for t in (z, gamma, beta)]
    fr, mr, vr = tail(*r_in)
    assert float(fr[3].abs().max()) == 0.0
    (fr * up).sum().backward()
    runs = []
    for _ in range(2):
        g_in = [t.float().to(dev).requires_grad_(True) for t in (z, gamma, beta)]
        f, mean, var = ops.BatchNormReluRowNorm.apply(g_in[0], g_in[1], g_in[2], 1e-3)
        (f * up.float().to(dev)).sum().backward()
        assert rel(f, fr) <= 2e-5 and rel(mean, mr) <= 2e-5 and rel(var, vr) <= 2e-5
        for name, a, r in zip(("z", "gamma", "beta"), g_in, r_in):
            assert rel(a.grad, r.grad) <= 1e-4, "grad of %s: %.3e" % (name, rel(a.grad, r.grad))
        runs.append([t.grad for t in g_in])
    for a, b in zip(*runs):
        assert torch.equal(a, b)
    # the same tail behind the 256 -> 1024 product (conv5): gradients of x, W, gamma, beta
    x = torch.randn(rows, 256, dtype=torch.float64, generator=g)
    W = torch.randn(256, 1024, dtype=torch.float64, generator=g) / 16
    b = torch.randn(1024, dtype=torch.float64, generator=g)
    r_in = [t.clone().requires_grad_(True) for t in (x, W, gamma, beta)]
    fr, mr, vr = tail(r_in[0] @ r_in[1] + b, r_in[2], r_in[3])
    (fr * up).sum().backward()
    g_in = [t.float().to(dev).requires_grad_(True) for t in (x, W, gamma, beta)]
    if ops.fused_linear_bn_ok(rows, 256, 1024):
        f, mean, var = ops.LinearBatchNormTrain.apply(g_in[0], g_in[1], b.float().to(dev), g_in[2], g_in[3], 1e-3, 1, True)
        (f * up.float().to(dev)).sum().backward()
        assert rel(f, fr) <= 2e-5 and rel(mean, mr) <= 2e-5 and rel(var, vr) <= 2e-5
        for name, a, r in zip(("x", "W", "gamma", "beta"), g_in, r_in):
            assert rel(a.grad, r.grad) <= 2e-4, "grad of %s: %.3e" % (name, rel(a.grad, r.grad))


def test_context_gating_product(dev):
    """loupe.py:99-100: y * sigmoid(g) and both gradients, including saturated gates."""
    ops = H.pkg("ops")
    g = torch.Generator().manual_seed(11)
    y = torch.randn(18, 256, dtype=torch.float64, generator=g)
    gt = torch.randn(18, 256, dtype=torch.float64, generator=g) * 4
    gt[0, :4] = torch.tensor([-60.0, 60.0, 0.0, -20.0], dtype=torch.float64)
    run_pair(lambda y, gt: ops.GateMul.apply(y, gt), lambda y, gt: y * torch.sigmoid(gt), [y, gt], dev)


@pytest.mark.parametrize("B,N", [(3, 512), (2, 1000), (18, 4096)])
def test_assign_aggregate_node_with_cloud_sums(dev, B, N):
    """ops.VladAssignAggregate: vlad and a_sum (loupe.py:276) against float64, the gradients with BOTH outputs in the loss
    (a_sum's gradient is the per-cloud row the softmax backward adds to every point), and a_sum bit-equal across runs."""
    ops = H.pkg("ops")
    g = torch.Generator().manual_seed(12)
    f = torch.rand(B * N, 1024, dtype=torch.float64, generator=g)
    f = f / f.norm(dim=1, keepdim=True)
    Wc = torch.randn(1024, 64, dtype=torch.float64, generator=g) / 32
    gamma = 1 + 0.1 * torch.randn(64, dtype=torch.float64, generator=g)
    beta = 0.1 * torch.randn(64, dtype=torch.float64, generator=g)
    up_v = torch.randn(B, 1024, 64, dtype=torch.float64, generator=g)
    up_s = torch.randn(B, 1, 64, dtype=torch.float64, generator=g)

    def ref(f, Wc, gamma, beta):
        z = f @ Wc
        m, v = z.mean(0), z.var(0, unbiased=False)
        a = torch.softmax((z - m) * torch.rsqrt(v + 1e-3) * gamma + beta, 1).reshape(B, N, 64)
        return torch.matmul(f.reshape(B, N, 1024).transpose(1, 2), a), a.sum(dim=1, keepdim=True)

    ins = [f, Wc, gamma, beta]
    g_in = [t.float().to(dev).requires_grad_(True) for t in ins]
    r_in = [t.clone().requires_grad_(True) for t in ins]
    vg, sg, _, _ = ops.VladAssignAggregate.apply(g_in[0], g_in[1], g_in[2], g_in[3], 1e-3, N)
    vr, sr = ref(*r_in)
    assert rel(vg, vr) <= 2e-5 and rel(sg, sr) <= 2e-5, (rel(vg, vr), rel(sg, sr))
    ((vg * up_v.float().to(dev)).sum() + (sg * up_s.float().to(dev)).sum()).backward()
    ((vr * up_v).sum() + (sr * up_s).sum()).backward()
    for i, (a, b) in enumerate(zip(g_in, r_in)):
        assert rel(a.grad, b.grad) <= 2e-3, "grad of input %d: %.3e" % (i, rel(a.grad, b.grad))   # 'fast' backward products
    with torch.no_grad():
        again = ops.VladAssignAggregate.apply(g_in[0], g_in[1], g_in[2], g_in[3], 1e-3, N)[1]
    assert torch.equal(again, sg)


@pytest.mark.parametrize("B,N", [(3, 512), (2, 4096)])
def test_tail_backward_inside_the_feature_gradient_product(dev, B, N):
    """conv5's tail f = l2_normalize(relu(bn(x W5))) followed by the VLAD assignment / aggregation (models/epc-net.py:136-148,
    loupe.py:255-291): with ops.FUSE_TAIL_BACKWARD the second node's backward continues through the first one's (epc_vlad_df_tail:
    row dot products from the assignment's tensors, du and the BatchNorm sums from the product's accumulators).  Every gradient
    must be the unfused path's up to the rounding of the re-associated dot products -- and both must be the float64 reference's."""
    ops = H.pkg("ops")
    g = torch.Generator().manual_seed(21)
    rows = B * N
    x = torch.randn(rows, 256, dtype=torch.float64, generator=g)
    W5 = torch.randn(256, 1024, dtype=torch.float64, generator=g) / 16
    g5 = 1 + 0.1 * torch.randn(1024, dtype=torch.float64, generator=g)
    b5 = 0.1 * torch.randn(1024, dtype=torch.float64, generator=g)
    Wc = torch.randn(1024, 64, dtype=torch.float64, generator=g) / 32
    gc = 1 + 0.1 * torch.randn(64, dtype=torch.float64, generator=g)
    bc = 0.1 * torch.randn(64, dtype=torch.float64, generator=g)
    up_v = torch.randn(B, 1024, 64, dtype=torch.float64, generator=g)
    up_s = torch.randn(B, 1, 64, dtype=torch.float64, generator=g)
    ins = [x, W5, g5, b5, Wc, gc, bc]

    def ref(x, W5, g5, b5, Wc, gc, bc):
        z = x @ W5
        u = torch.relu((z - z.mean(0)) * torch.rsqrt(z.var(0, unbiased=False) + 1e-3) * g5 + b5)
        f = u / u.norm(dim=1, keepdim=True).clamp_min(1e-6)
        zc = f @ Wc
        a = torch.softmax((zc - zc.mean(0)) * torch.rsqrt(zc.var(0, unbiased=False) + 1e-3) * gc + bc, 1).reshape(B, N, 64)
        return torch.matmul(f.reshape(B, N, 1024).transpose(1, 2), a), a.sum(dim=1, keepdim=True)

    def run(fuse):
        t = [v.float().to(dev).requires_grad_(True) for v in ins]
        link = ops.TailLink() if fuse else None
        bias = torch.zeros(1024, device=dev)
        f, _, _ = ops.LinearBatchNormTrain.apply(t[0], t[1], bias, t[2], t[3], 1e-3, 1, True, None, link)
        vg, sg, _, _ = ops.VladAssignAggregate.apply(f, t[4], t[5], t[6], 1e-3, N, link)
        ((vg * up_v.float().to(dev)).sum() + (sg * up_s.float().to(dev)).sum()).backward()
        return [v.grad for v in t]

    was = ops.FUSE_TAIL_BACKWARD
    try:
        ops.FUSE_TAIL_BACKWARD = True
        fused = run(True)
        plain = run(False)
    finally:
        ops.FUSE_TAIL_BACKWARD = was
    r = [v.clone().requires_grad_(True) for v in ins]
    vr, sr = ref(*r)
    ((vr * up_v).sum() + (sr * up_s).sum()).backward()
    for i, (a, b, c) in enumerate(zip(fused, plain, r)):
        assert rel(a, b) <= 2e-4, "fused vs unfused, input %d: %.3e" % (i, rel(a, b))
        assert rel(a, c.grad) <= 3e-3 and rel(b, c.grad) <= 3e-3, "input %d vs float64: %.3e %.3e" % (i, rel(a, c.grad), rel(b, c.grad))


def test_vlad_aggregate(dev):
    ops = H.pkg("ops")
    g = torch.Generator().manual_seed(4)
    f = torch.rand(3, 512, 1024, dtype=torch.float64, generator=g)
    a = torch.softmax(torch.randn(3, 512, 64, dtype=torch.float64, generator=g), -1)
    run_pair(lambda f, a: ops.VladAggregate.apply(f, a), lambda f, a: torch.matmul(f.transpose(1, 2), a), [f, a], dev)


@pytest.mark.parametrize("shape", [("nn", 18, 16384, 256, 1, 64), ("tn", 1024, 4096, 64, 3, 8), ("nn", 40, 700, 64, 1, 5)])
def test_deterministic_split_k(dev, shape):
    """epc_gemm_splitk_det: the split-K slices are added in a fixed order -- equal to float64 within the f32-accurate bar, the
    SAME BITS on every run and in every split-K workspace state, bias and accumulate included (the atomic form differs from run
    to run in the last bits, which is what the training step's forward must not do)."""
    ops = H.pkg("ops")
    form, M, K, N, nb, splitk = shape
    g = torch.Generator().manual_seed(11)
    A = torch.randn((nb, K, M) if form == "tn" else (nb, M, K), dtype=torch.float64, generator=g)
    B = torch.randn(nb, K, N, dtype=torch.float64, generator=g) / np.sqrt(K)
    bias = torch.randn(N, dtype=torch.float64, generator=g)
    ref = (A.transpose(1, 2) if form == "tn" else A) @ B + bias
    Ag, Bg, bg = A.float().to(dev), B.float().to(dev), bias.float().to(dev)
    if nb == 1:
        Ag, Bg, ref = Ag[0], Bg[0], ref[0]
    runs = []
    for _ in range(3):
        ops._SPLITK_WS.clear()                         # a fresh (uninitialised) workspace each time
        runs.append(ops.gemm(Ag, Bg, bias=bg, trans_a=form == "tn", splitk=splitk, deterministic=True))
    assert rel(runs[0], ref) <= 2e-6
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])
    acc = runs[0].clone()
    ops.gemm(Ag, Bg, out=acc, trans_a=form == "tn", splitk=splitk, accumulate=True, deterministic=True)
    assert rel(acc, 2 * ref - bias) <= 2e-6


def test_adam_matches_tensorflow_rule(dev):
    ops = H.pkg("ops")
    g = torch.Generator().manual_seed(5)
    w = torch.randn(10000, generator=g); m = torch.randn(10000, generator=g) * 0.01; v = torch.rand(10000, generator=g) * 1e-4
    gr = torch.randn(10000, generator=g) * 0.1
    wd, md, vd = w.to(dev), m.to(dev), v.to(dev)
    ops.adam_step(wd, md, vd, gr.to(dev), 5e-5, 7)
    torch.cuda.synchronize()
    w64, m64, v64, g64 = (t.double() for t in (w, m, v, gr))
    m2 = 0.9 * m64 + 0.1 * g64; v2 = 0.999 * v64 + 0.001 * g64 * g64
    lr_t = 5e-5 * np.sqrt(1 - 0.999 ** 7) / (1 - 0.9 ** 7)
    assert rel(md, m2) <= 1e-6 and rel(vd, v2) <= 2e-5      # (1 - 0.999f) is 0.00100005 in f32, as in TensorFlow's kernel
    assert np.abs((wd.cpu().double() - (w64 - lr_t * m2 / (v2.sqrt() + 1e-8))).numpy()).max() <= 5e-7   # f32 ulp of |w| ~ 2 is 2.4e-7


@pytest.mark.parametrize("cls,groups", [("G_VLAD", 4), ("G_VLAD", 1), ("NetVLAD", None)])
@pytest.mark.parametrize("is_training", [False, True])
def test_loupe_classes_match_oracle(dev, cls, groups, is_training):
    """The pooling classes as op-level API (reference loupe.py:103-333): G_VLAD / NetVLAD(...).forward(features) on
    random unit features against the oracle's restatement, inference and training mode (batch statistics)."""
    V = H.pkg("variables")
    lp = H.pkg("loupe")
    B, N, F, C, D = 2, 256, 1024, 64, 256
    rng = np.random.RandomState(3)
    feats = rng.randn(B * N, F).astype(np.float32)
    feats /= np.linalg.norm(feats, axis=1, keepdims=True)
    st = V.reset_default_store(device=dev, seed=5)
    with V.variable_scope("query_triplets"), V.variable_scope("VLAD"):
        kw = dict(feature_size=F, max_samples=N, cluster_size=C, output_dim=D, gating=True, add_batch_norm=True,
                  is_training=is_training)
        pool = lp.G_VLAD(groups=groups, **kw) if cls == "G_VLAD" else lp.NetVLAD(**kw)
        pool.declare_variables()
        st.randomize_statistics(1)
        w = {k[len("query_triplets/"):]: v.detach().cpu().numpy() for k, v in st.vars.items()}
        with torch.no_grad():
            out = pool.forward(torch.from_numpy(feats).to(dev)).cpu().numpy()
    ost = O.State(w, np.float32)
    if cls == "G_VLAD":
        ref = O.g_vlad_forward(ost, feats, N, groups, is_training)
    else:
        ref = O.netvlad_forward(ost, feats, N, is_training)
    assert out.shape == (B, D)
    assert np.abs(out - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-6) + 1e-6, np.abs(out - ref).max()


@pytest.mark.parametrize("cls", ["G_VLAD", "NetVLAD"])
def test_loupe_without_batch_norm(dev, cls):
    """add_batch_norm=False (loupe.py:264-270: cluster_biases instead of cluster_bn).  With gating the reference cannot even
    build its graph -- gating_biases is given tf.random_normal(stddev=...) (the function, called without a shape) as its
    initializer, loupe.py:88-92: TypeError -- and neither can this."""
    V, lp = H.pkg("variables"), H.pkg("loupe")
    B, N, F, C, D = 2, 128, 1024, 64, 256
    rng = np.random.RandomState(4)
    feats = rng.randn(B * N, F).astype(np.float32)
    feats /= np.linalg.norm(feats, axis=1, keepdims=True)
    st = V.reset_default_store(device=dev, seed=6)
    with V.variable_scope("query_triplets"), V.variable_scope("VLAD"):
        kw = dict(feature_size=F, max_samples=N, cluster_size=C, output_dim=D, gating=False, add_batch_norm=False, is_training=False)
        pool = lp.G_VLAD(groups=4, **kw) if cls == "G_VLAD" else lp.NetVLAD(**kw)
        pool.declare_variables()
        st.randomize_statistics(2)
        assert "query_triplets/VLAD/cluster_biases" in st.vars and "query_triplets/VLAD/cluster_bn/gamma" not in st.vars
        w = {k[len("query_triplets/"):]: v.detach().cpu().numpy() for k, v in st.vars.items()}
        with torch.no_grad():
            out = pool.forward(torch.from_numpy(feats).to(dev)).cpu().numpy()
        gated = (lp.G_VLAD(groups=4, **dict(kw, gating=True)) if cls == "G_VLAD" else lp.NetVLAD(**dict(kw, gating=True)))
        with pytest.raises(TypeError, match="shape"):
            gated.forward(torch.from_numpy(feats).to(dev))
    ost = O.State(w, np.float32)
    ref = (O.g_vlad_forward(ost, feats, N, 4, False, gating=False, add_batch_norm=False) if cls == "G_VLAD"
           else O.netvlad_forward(ost, feats, N, False, gating=False, add_batch_norm=False))
    assert np.abs(out - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-6) + 1e-6


def test_column_reductions_are_deterministic_under_workspace_reuse(dev):
    """The one-launch column reductions hand partial sums from every workgroup to the last one through a workspace that
    all layers of a step share (include/epcnet.h, workspace contract).  A partial that is read before it is visible
    would show up as a result that depends on what the previous call left in the workspace: interleave calls of
    different shapes many times and require bit-identical results, and agreement with float64."""
    ops = H.pkg("ops")
    g = torch.Generator().manual_seed(7)
    shapes = [(73728, 64), (18 * 4096, 1024), (4097, 256), (72, 256), (300, 64)]
    data = [(torch.randn(r, c, generator=g) * 3 + 1).to(dev) for r, c in shapes]
    gam = [torch.rand(c, generator=g).to(dev) + 0.5 for _, c in shapes]
    bet = [torch.randn(c, generator=g).to(dev) for _, c in shapes]
    dys = [torch.randn(r, c, generator=g).to(dev) for r, c in shapes]
    first = None
    for rep in range(25):
        outs = []
        for z, ga, be, dy in zip(data, gam, bet, dys):
            z_ = z.clone().requires_grad_(True)
            ga_, be_ = ga.clone().requires_grad_(True), be.clone().requires_grad_(True)
            y, mean, var = ops.BatchNormTrain.apply(z_, ga_, be_, 1e-3, 1)
            y.backward(dy)
            outs += [mean.clone(), var.clone(), ga_.grad.clone(), be_.grad.clone(), z_.grad[:7].clone()]
        if first is None:
            first = outs
            for (z, ga, be, dy), (mean, var) in zip(zip(data, gam, bet, dys), zip(outs[0::5], outs[1::5])):
                z64 = z.double()
                assert rel(mean, z64.mean(0)) <= 1e-5 and rel(var, z64.var(0, unbiased=False)) <= 2e-5
        else:
            for a, b in zip(first, outs):
                assert torch.equal(a, b), "a column reduction changed between identical calls (rep %d)" % rep


@pytest.mark.parametrize("degenerate", [False, True])
def test_vlad_normalize(dev, degenerate):
    """ops.VladNormalize against the op-by-op composition of loupe.py:284,292-298 in float64 (forward and the three
    gradients); `degenerate` zeroes one cluster column of one cloud so that the inner 1e-12 clamp is active there."""
    ops = H.pkg("ops")
    g = torch.Generator().manual_seed(11)
    B, F, C = 3, 1024, 64
    raw = torch.randn(B, F, C, dtype=torch.float64, generator=g)
    a_sum = torch.rand(B, 1, C, dtype=torch.float64, generator=g) * 50
    w2 = torch.randn(1, F, C, dtype=torch.float64, generator=g) / 32
    if degenerate:
        raw[1, :, 5] = 0
        a_sum[1, 0, 5] = 0

    def ref(raw, a_sum, w2):
        v = raw - a_sum * w2
        v = v * torch.rsqrt(torch.clamp((v * v).sum(1, keepdim=True), min=1e-12))
        v = v.reshape(B, -1)
        return (v * torch.rsqrt(torch.clamp((v * v).sum(1, keepdim=True), min=1e-12))).reshape(B, F, C)

    run_pair(lambda r, a, w: ops.VladNormalize.apply(r, a, w), ref, [raw, a_sum, w2], dev, tol=2e-5)


@pytest.mark.parametrize("B,P,Nn,far", [(1, 2, 14, False), (1, 2, 18, False), (3, 2, 5, False), (2, 2, 6, True)])
def test_lazy_quadruplet_loss_operator(dev, B, P, Nn, far):
    """ops.LazyQuadrupletLoss against the op-by-op composition (models/_common.py, float64 on the CPU): value and the four
    gradients; `far` = every hinge inactive (loss 0, all gradients 0)."""
    ops, M = H.pkg("ops"), H.pkg("models._common")
    g = torch.Generator().manual_seed(5)
    D = 256
    unit = lambda t: t / t.norm(dim=-1, keepdim=True)
    q, pos, neg, oth = (unit(torch.randn(B, n, D, dtype=torch.float64, generator=g)) for n in (1, P, Nn, 1))
    if far:
        pos = q.repeat(1, P, 1).clone()
        neg = -q.repeat(1, Nn, 1) + 1e-3 * neg
        oth = q.clone()
    ins64 = [t.clone().requires_grad_(True) for t in (q, pos, neg, oth)]
    ins32 = [t.float().to(dev).requires_grad_(True) for t in (q, pos, neg, oth)]
    l64 = M.lazy_triplet_loss(ins64[0], ins64[1], ins64[2], 0.5) + \
        M._neg_terms(ins64[0], ins64[1], ins64[2], ins64[3], 0.2).max(1).values.mean()
    l32 = M.lazy_quadruplet_loss(*ins32, 0.5, 0.2)
    assert l32.grad_fn is not None and type(l32.grad_fn).__name__.startswith("LazyQuadrupletLoss")
    assert float(l32) == pytest.approx(float(l64), rel=1e-5, abs=1e-7)
    (l64 * 1.7).backward()
    (l32 * 1.7).backward()
    if far:
        assert float(l64) == 0.0
    for a, b in zip(ins32, ins64):
        assert (a.grad.double().cpu() - b.grad).abs().max() <= 1e-5 * max(b.grad.abs().max().item(), 1.0)


def test_maxpool_points_op(dev):
    """EPC-Net-L's global max over a cloud's points in training mode (models/epc-net-l.py:88-92; utils/tf_util.py:349-372) as a library op:
    values equal torch's, the gradient goes to the maximum's row -- the FIRST one on ties, as tf.nn.max_pool's gradient does -- and a
    NaN entry wins."""
    ops, tf_util = H.pkg("ops"), H.pkg("utils.tf_util")
    g = torch.Generator().manual_seed(2)
    x = torch.randn(5, 1000, 192, generator=g).to(dev).requires_grad_(True)
    y = ops.MaxPoolPoints.apply(x)
    ref = x.detach().amax(dim=1)
    assert torch.equal(y, ref)
    w = torch.randn(5, 192, generator=g).to(dev)
    (gx,) = torch.autograd.grad((y * w).sum(), x)
    xr = x.detach().clone().requires_grad_(True)
    (gr,) = torch.autograd.grad((xr.max(dim=1).values * w).sum(), xr)      # (continuous data: no ties, one arg-max per column)
    assert torch.equal(gx, gr)
    # the wrapper with the reference's name and shapes: (B, N, 1, C) -> (B, 1, 1, C)
    z = tf_util.max_pool2d(x.detach().reshape(5, 1000, 1, 192), [1000, 1], "maxpool", padding="VALID")
    assert tuple(z.shape) == (5, 1, 1, 192) and torch.equal(z.reshape(5, 192), ref)
    # ties: the first row takes the gradient; NaN: propagates
    t = torch.zeros(2, 64, 64, device=dev)
    t[0, 7, :] = 3.0
    t[0, 40, :] = 3.0
    t[1, 5, 3] = float("nan")
    t.requires_grad_(True)
    yt = ops.MaxPoolPoints.apply(t)
    (gt,) = torch.autograd.grad(yt[0].sum() + yt[1, :3].sum(), t)
    assert float(gt[0, 7].sum()) == 64.0 and float(gt[0, 40].sum()) == 0.0 and float(gt[0].sum()) == 64.0
    assert bool(torch.isnan(yt[1, 3])) and float(yt[1, 2]) == 0.0
    assert float(gt[1, 0, :3].sum()) == 3.0      # (all-zero columns: row 0 is the first maximum)


def test_vlad_w2_grad_and_group_sum_ops(dev):
    """The two small sums of the VLAD head as library ops (loupe.py:284,292 and :326-328): cluster_weights2's gradient through
    ops.VladNormalize and ops.GroupSum, forward and backward, against torch float64."""
    ops = H.pkg("ops")
    g = torch.Generator().manual_seed(4)
    B, F, C = 5, 256, 64
    raw, a_sum, w2 = (torch.randn(B, F, C, generator=g).to(dev), torch.rand(B, 1, C, generator=g).to(dev) * 30,
                      (torch.randn(1, F, C, generator=g) * 0.1).to(dev))
    xs = [t.clone().requires_grad_(True) for t in (raw, a_sum, w2)]
    out = ops.VladNormalize.apply(*xs)
    wgt = torch.randn(B, F, C, generator=g).to(dev)
    gr = torch.autograd.grad((out * wgt).sum(), xs)
    ys = [t.double().cpu().clone().requires_grad_(True) for t in (raw, a_sum, w2)]
    v = ys[0] - ys[1] * ys[2]
    v = v * torch.rsqrt(torch.clamp((v * v).sum(1, keepdim=True), min=1e-12))
    flat = v.reshape(B, -1)
    ref = (flat * torch.rsqrt(torch.clamp((flat * flat).sum(1, keepdim=True), min=1e-12))).reshape(B, F, C)
    g_ref = torch.autograd.grad((ref * wgt.double().cpu()).sum(), ys)
    assert float((out.double().cpu() - ref).abs().max()) <= 1e-6
    for a, b in zip(gr, g_ref):
        assert float((a.double().cpu() - b).norm() / b.norm()) <= 2e-5
    x = torch.randn(6 * 4, 256, generator=g).to(dev).requires_grad_(True)
    y = ops.GroupSum.apply(x, 4)
    assert torch.allclose(y, x.detach().reshape(6, 4, 256).sum(1), atol=1e-6)
    w = torch.randn(6, 256, generator=g).to(dev)
    (gx,) = torch.autograd.grad((y * w).sum(), x)
    assert torch.equal(gx, w[:, None, :].expand(6, 4, 256).reshape(24, 256))


@pytest.mark.parametrize("ncl,n,kind", [(3, 4096, "uniform"), (2, 1000, "uniform"), (5, 96, "uniform"), (2, 2048, "ties"), (1, 8192 + 64, "uniform")])
def test_transposed_graph_lists_are_the_sorted_inverse_of_the_knn_lists(dev, ncl, n, kind):
    """epc_knn_transpose: for every point j, rdeg[j] entries at rlist[roff[j]..] = the ABSOLUTE rows i whose neighbour list holds j,
    ascending -- against a numpy inversion of the lists; rows whose own list overflowed (cnt > cap: "ties": 200 copies of one point) are
    not in the graph; twice the same bits."""
    ops, L = H.pkg("ops"), H.pkg("lib")
    pc = O.synthetic_clouds(ncl, n, 17)
    if kind == "ties":
        pc[0, 300:500] = pc[0, 11]
    x = ops.morton_sort(torch.from_numpy(pc).to(dev))
    g = ops.KnnGraph(x)
    rdeg, roff, rlist = (t.cpu().numpy() for t in g.transposed())
    idx, cnt = g.idx.cpu().numpy().reshape(ncl * n, -1), g.cnt.cpu().numpy().reshape(-1)
    cap = idx.shape[1]
    want = [[] for _ in range(ncl * n)]
    for i in range(ncl * n):
        if cnt[i] <= cap:
            base = (i // n) * n
            for j in idx[i, :cnt[i]]:
                want[base + int(j)].append(i)
    assert kind != "ties" or (cnt > cap).any()
    for j in range(ncl * n):
        assert rdeg[j] == len(want[j]), j
        assert list(rlist[roff[j]:roff[j] + rdeg[j]]) == want[j], j          # (ascending: i runs upward)
    for c in range(ncl):                                                      # a cloud's lists are packed back to back from its base
        assert roff[c * n] == c * n * cap
        assert np.array_equal(roff[c * n + 1:(c + 1) * n], roff[c * n] + np.cumsum(rdeg[c * n:(c + 1) * n - 1]))
    g2 = ops.KnnGraph(x)
    rdeg2, roff2, rlist2 = (t.cpu().numpy() for t in g2.transposed())
    assert np.array_equal(rdeg2, rdeg) and np.array_equal(roff2, roff)
    for c in range(ncl):
        used = int(rdeg[c * n:(c + 1) * n].sum())
        assert np.array_equal(rlist2[c * n * cap:c * n * cap + used], rlist[c * n * cap:c * n * cap + used])


def test_transposed_graph_with_a_hub_longer_than_a_wave(dev):
    """epc_knn_transpose on hand-made lists: target 5 of cloud 0 is in every third source's list (683 entries: the one-lane insertion sort
    of lists longer than 64)."""
    L = H.pkg("lib")
    lib = L.lib()
    ncl, n, cap = 2, 2048, 32
    rng = np.random.default_rng(3)
    idx = np.zeros((ncl * n, cap), dtype=np.int32)
    cnt = np.zeros(ncl * n, dtype=np.int32)
    for i in range(ncl * n):
        t = set(rng.choice(n, size=int(rng.integers(18, 25)), replace=False).tolist())
        if i % 3 == 0 and i < n:
            t.add(5)
        t = sorted(t)[:cap]
        idx[i, :len(t)], cnt[i] = t, len(t)
    d_idx, d_cnt = torch.from_numpy(idx).to(dev), torch.from_numpy(cnt).to(dev)
    M = ncl * n
    rdeg, roff, cursor = (torch.empty(M, dtype=torch.int32, device=dev) for _ in range(3))
    rlist = torch.empty(M * cap, dtype=torch.int32, device=dev)
    L.check(lib.epc_knn_transpose(d_idx.data_ptr(), d_cnt.data_ptr(), cap, ncl, n, rdeg.data_ptr(), roff.data_ptr(), cursor.data_ptr(),
                                  rlist.data_ptr(), L.current_stream()))
    torch.cuda.synchronize()
    rdeg, roff, rlist = rdeg.cpu().numpy(), roff.cpu().numpy(), rlist.cpu().numpy()
    want = [[] for _ in range(M)]
    for i in range(M):
        for j in idx[i, :cnt[i]]:
            want[(i // n) * n + int(j)].append(i)
    for j in range(M):
        assert rdeg[j] == len(want[j]) and list(rlist[roff[j]:roff[j] + rdeg[j]]) == want[j], j
    assert rdeg[5] > 600
