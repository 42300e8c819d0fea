"""Differentiable operators of the training step: ``torch.autograd.Function`` wrappers whose forward AND backward are
calls into libepcnet_hip.so (``csrc/train_ops.hip``).  autograd is used as the tape only; there is no torch arithmetic
in these functions and no CPU fallback.  Reference semantics are cited per operator."""
from __future__ import annotations

import ctypes

import torch

from . import lib as L
from .lib import EpcNetError

BN_EPS = 1e-3


def _st():
    return L.current_stream()


_WORKSPACES = {}


def _ws(rows, C, device):
    """The column-reduction workspace of (device, current stream), reused by every reduction of that stream
    (include/epcnet.h, workspace contract)."""
    n = L.lib().epc_colreduce_workspace_bytes(int(rows), int(C))
    key = (device.index, int(_st() or 0))
    buf = _WORKSPACES.get(key)
    if buf is None or buf.numel() < n:
        buf = torch.zeros(max(n, 1 << 20), dtype=torch.uint8, device=device)
        _WORKSPACES[key] = buf
    return buf, buf.numel()


# Arithmetic of the training step's GEMMs.  "bf16x6" (default): every f32 operand as three bf16 pieces, six products --
# f32-accurate, what the reference's fp32 TensorFlow graph computes; backward GEMMs use two pieces.  "bf16": operands
# rounded to one bf16 value, f32 accumulation, forward and backward (BASELINE.json configs[2] names this arithmetic for
# the training step; set by TrainStep from params["TRAIN_PRECISION"]).
_GEMM_PRECISION = "bf16x6"


def set_gemm_precision(name: str) -> str:
    """Select the GEMM arithmetic of the differentiable operators; returns the previous setting."""
    global _GEMM_PRECISION
    if name not in ("bf16x6", "bf16"):
        raise ValueError("unknown GEMM precision %r (bf16x6 | bf16)" % (name,))
    prev, _GEMM_PRECISION = _GEMM_PRECISION, name
    return prev


_SPLITK_WS = {}


def _splitk_ws(floats, device):
    """The split-K partial-product workspace of (device, current stream) for the deterministic forward GEMMs."""
    key = (device.index, int(_st() or 0))
    buf = _SPLITK_WS.get(key)
    if buf is None or buf.numel() < floats:
        buf = _SPLITK_WS[key] = torch.empty(max(int(floats), 1 << 18), dtype=torch.float32, device=device)
    return buf


def gemm(A, B, out=None, bias=None, trans_a=False, trans_b=False, splitk=1, accumulate=False, fast=False,
         deterministic=False):
    """out = op(A) @ op(B) (+ bias).  A, B: 2-D, or 3-D with a leading batch dim (same batch).  f32-accurate split-bf16
    MFMA arithmetic (three pieces per operand); ``fast`` = two pieces (backward GEMMs: linear in the gradient);
    one piece under set_gemm_precision("bf16").  ``deterministic``: a split-K product adds its slices in a fixed order
    (epc_gemm_splitk_det) instead of with f32 atomics -- the forward products use it, so that a step's activations, ReLU masks
    and loss are the same bits on every run."""
    L.require_gpu()
    batched = A.dim() == 3
    a2 = A[0] if batched else A
    b2 = B[0] if batched else B
    M, K = (a2.shape[1], a2.shape[0]) if trans_a else (a2.shape[0], a2.shape[1])
    Kb, N = (b2.shape[1], b2.shape[0]) if trans_b else (b2.shape[0], b2.shape[1])
    assert K == Kb, "inner dimensions differ: %d vs %d" % (K, Kb)
    nb = A.shape[0] if batched else 1
    if out is None:
        out = torch.empty(((nb, M, N) if batched else (M, N)), dtype=torch.float32, device=A.device)
    sa = a2.stride()
    sb = b2.stride()
    sAm, sAk = (sa[1], sa[0]) if trans_a else (sa[0], sa[1])
    sBk, sBn = (sb[1], sb[0]) if trans_b else (sb[0], sb[1])
    if deterministic and splitk > 1:
        pieces = 1 if _GEMM_PRECISION == "bf16" else (2 if fast else 3)
        ws = _splitk_ws(nb * int(splitk) * M * N, A.device)
        L.check(L.lib().epc_gemm_splitk_det(
            A.data_ptr(), B.data_ptr(), out.data_ptr(), bias.data_ptr() if bias is not None else None,
            M, N, K, sAm, sAk, sBk, sBn, out.stride(-2), nb, A.stride(0) if batched else 0,
            B.stride(0) if batched else 0, out.stride(0) if batched else 0, int(splitk), 1 if accumulate else 0, pieces,
            ws.data_ptr(), ws.numel(), _st()))
        return out
    if _GEMM_PRECISION == "bf16":
        fn = L.lib().epc_gemm_bf16
    else:
        fn = L.lib().epc_gemm_f32_fast if fast else L.lib().epc_gemm_f32
    L.check(fn(A.data_ptr(), B.data_ptr(), out.data_ptr(), bias.data_ptr() if bias is not None else None,
             M, N, K, sAm, sAk, sBk, sBn, out.stride(-2), nb, A.stride(0) if batched else 0,
             B.stride(0) if batched else 0, out.stride(0) if batched else 0, int(splitk), 1 if accumulate else 0, _st()))
    return out


def _splitk_for(M, N, K):
    """Split-K factor of a deep-K product (dW = x^T dy over all rows): about two workgroups per CU (three before the GEMM's
    branch-free tile fetch: conv5's dW is 182 us at 32 slices, 227 at 48; the assignment's dWc 91 us at 64).  The kernel's
    tile is 128 wide on a side of at least 128, 64 otherwise (gemm_impl); measured on the training shapes
    (scripts/time_gemm.py): 256x1024x73728 16 -> 48 splits 550 -> 250 us; 64x64x73728 is best at 256."""
    tile = lambda d: 128 if d >= 128 else 64
    tiles = ((M + tile(M) - 1) // tile(M)) * ((N + tile(N) - 1) // tile(N))
    return int(max(1, min(512 // max(tiles, 1), K // 128, 256)))


_ZEROS = {}


def const_zeros_like(t):
    """A shared READ-ONLY zero tensor of t's shape (gradients that are exactly zero by construction: no fill launch per
    step).  Never write to it."""
    key = (t.device.index, tuple(t.shape))
    z = _ZEROS.get(key)
    if z is None:
        z = _ZEROS[key] = torch.zeros(tuple(t.shape), dtype=torch.float32, device=t.device)
    return z


class Linear(torch.autograd.Function):
    """y = x @ W + b on (rows, Cin): tf.nn.conv1d with kernel_size 1 (utils/tf_util.py:94-99) / tf.matmul + bias_add
    (:336-339).  ``bias_before_batch_stats``: the layer feeds a training-mode BatchNorm, whose backward returns a dz
    with zero column sums -- the bias gradient sum_rows dz is exactly 0 (the reference's BiasAddGrad evaluates the same
    sum and gets rounding noise; the bias has no effect on the network's output there), so it is not computed."""

    @staticmethod
    def forward(ctx, x, W, b, bias_before_batch_stats=False):
        x = x.contiguous()
        ctx.save_for_backward(x, W)
        ctx.has_bias = b is not None
        ctx.zero_bias_grad = bool(bias_before_batch_stats)
        rows, cin = x.shape
        if cin <= 4 and W.shape[1] % 4 == 0 and W.is_contiguous():     # conv1 on the coordinates: three FMAs per output
            z = torch.empty((rows, W.shape[1]), dtype=torch.float32, device=x.device)
            L.check(L.lib().epc_linear_smallk_fwd(x.data_ptr(), W.data_ptr(), b.data_ptr() if b is not None else None, rows, cin,
                                                  W.shape[1], z.data_ptr(), _st()))
            return z
        # few output tiles but a deep K (the 16384-wide hidden projection on a handful of rows): split K over workgroups
        tiles = ((rows + 63) // 64) * ((W.shape[1] + 63) // 64)
        splitk = int(max(1, min(256 // tiles, cin // 256, 64))) if tiles <= 32 else 1
        return gemm(x, W, bias=b, splitk=splitk, deterministic=True)

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        rows, cin = x.shape
        cout = W.shape[1]
        dx = gemm(dy, W, trans_b=True, fast=True) if ctx.needs_input_grad[0] else None
        if cin <= 4 and cout == 64:
            dW = torch.empty(W.shape, dtype=torch.float32, device=W.device)   # dense [cin][64] (the kernel's layout), whatever W's strides
            pf = L.lib().epc_linear_smallk_dw_partial_floats(rows, cin)
            part = _splitk_ws(pf, x.device)
            L.check(L.lib().epc_linear_smallk_dw(x.data_ptr(), dy.data_ptr(), rows, cin, cout, dW.data_ptr(), part.data_ptr(),
                                                 part.numel(), _st()))
        else:
            dW = gemm(x, dy, trans_a=True, splitk=_splitk_for(cin, cout, rows), fast=True, deterministic=True)
        db = None      # exactly zero in front of a training-mode BatchNorm: left undefined (TrainStep reads it as zeros)
        if ctx.has_bias and not ctx.zero_bias_grad:
            db = torch.empty(cout, dtype=torch.float32, device=x.device)
            ws, n = _ws(rows, cout, x.device)
            L.check(L.lib().epc_col_sum(dy.data_ptr(), rows, cout, db.data_ptr(), ws.data_ptr(), n, _st()))
        return dx, dW, db, None


class BatchNormTrain(torch.autograd.Function):
    """Training-mode batch normalisation over the rows of z (rows, C) (+ReLU): batch mean / POPULATION variance
    (tf.nn.moments), y = act((z-mean)*rsqrt(var+eps)*gamma + beta) (utils/tf_util.py:472-490; slim.batch_norm in
    loupe.py:257-263).  Returns (y, mean, var); mean/var feed the moving averages and carry no gradient."""

    @staticmethod
    def forward(ctx, z, gamma, beta, eps, relu):
        z = z.contiguous()
        rows, C = z.shape
        mean = torch.empty(C, dtype=torch.float32, device=z.device)
        var = torch.empty(C, dtype=torch.float32, device=z.device)
        ws, n = _ws(rows, C, z.device)
        L.check(L.lib().epc_col_moments(z.data_ptr(), rows, C, mean.data_ptr(), var.data_ptr(), ws.data_ptr(), n, _st()))
        y = torch.empty_like(z)
        L.check(L.lib().epc_bn_apply_fwd(z.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                         float(eps), int(relu), rows, C, y.data_ptr(), _st()))
        ctx.save_for_backward(z, mean, var, gamma, beta)       # the ReLU mask is recomputed from z: y is not kept
        ctx.eps, ctx.relu = float(eps), int(relu)
        ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)      # no zero-filled "gradients" of mean / var (two fill launches per layer)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        if dy is None:
            return None, None, None, None, None
        z, mean, var, gamma, beta = ctx.saved_tensors
        dy = dy.contiguous()
        rows, C = z.shape
        dz = torch.empty_like(z)
        dgamma = torch.empty(C, dtype=torch.float32, device=z.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=z.device)
        ws, n = _ws(rows, C, z.device)
        L.check(L.lib().epc_bn_apply_bwd(dy.data_ptr(), z.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(),
                                         beta.data_ptr(), ctx.eps, ctx.relu, rows, C, dz.data_ptr(), dgamma.data_ptr(),
                                         dbeta.data_ptr(), ws.data_ptr(), n, _st()))
        return dz, dgamma, dbeta, None, None


# The two wide forward products of the training step whose operands are bounded by construction -- conv5 (BatchNorm'd block
# outputs against its weights) and the VLAD assignment (l2-normalised features against the cluster weights) -- take the
# split-fp16 three-product arithmetic (epc_gemm_f16x3_stats: 2^-22 per product, half the matrix work of the six-product form).
# (operand scale exponents: activations, weights)
# A scaled value beyond fp16's range (or a NaN / Inf) makes the library recompute the product in the six-product form, in the same
# stream (epc_gemm_f16x3_stats' range guard): nothing is clamped silently.
_FWD_F16X3 = True
F16X3_CONV5 = (8, 12)      # block outputs below 2^8 = 256 and |w| < 2^4 = 16 take the fast form
F16X3_ASSIGN = (14, 12)    # |f| <= 1, |w| < 2^4


def set_forward_f16x3(on: bool) -> bool:
    """Enable / disable the split-fp16 arithmetic of those two products (off: the six-product form everywhere)."""
    global _FWD_F16X3
    prev, _FWD_F16X3 = _FWD_F16X3, bool(on)
    return prev


def _gemm_with_stats(x, W, b, f16x3=None):
    """z = x @ W + b and the batch moments of z from the GEMM's own epilogue (epc_gemm_f32_stats): (z, mean, var).
    ``f16x3`` = (activation, weight) scale exponents: the split-fp16 form (epc_gemm_f16x3_stats) when enabled."""
    rows, cin = x.shape
    cout = W.shape[1]
    z = torch.empty((rows, cout), dtype=torch.float32, device=x.device)
    mean = torch.empty(cout, dtype=torch.float32, device=x.device)
    var = torch.empty(cout, dtype=torch.float32, device=x.device)
    if cin == 64 and cout == 64 and x.is_contiguous() and W.is_contiguous() and _GEMM_PRECISION == "bf16x6":
        ws, n = _ws(rows, 64, x.device)      # the thin layers: one launch, moments finished by the last workgroup
        L.check(L.lib().epc_linear_stats64(x.data_ptr(), W.data_ptr(), b.data_ptr() if b is not None else None, rows,
                                           z.data_ptr(), mean.data_ptr(), var.data_ptr(), ws.data_ptr(), n, _st()))
        return z, mean, var
    tiles = L.lib().epc_gemm_stats_tiles(rows)
    stats = torch.empty(tiles * 3 * cout + 1, dtype=torch.float32, device=x.device)   # per row tile: sum, sum of squares, pivot (+ the f16x3 range word)
    if f16x3 is not None and _FWD_F16X3 and _GEMM_PRECISION == "bf16x6":
        L.check(L.lib().epc_gemm_f16x3_stats(x.data_ptr(), W.data_ptr(), z.data_ptr(), b.data_ptr() if b is not None else None,
                                             rows, cout, cin, x.stride(0), x.stride(1), W.stride(0), W.stride(1), cout,
                                             int(f16x3[0]), int(f16x3[1]), stats.data_ptr(), stats.numel(), mean.data_ptr(),
                                             var.data_ptr(), _st()))
        return z, mean, var
    fn = L.lib().epc_gemm_bf16_stats if _GEMM_PRECISION == "bf16" else L.lib().epc_gemm_f32_stats
    L.check(fn(x.data_ptr(), W.data_ptr(), z.data_ptr(), b.data_ptr() if b is not None else None,
               rows, cout, cin, x.stride(0), x.stride(1), W.stride(0), W.stride(1), cout,
               stats.data_ptr(), stats.numel(), mean.data_ptr(), var.data_ptr(), _st()))
    return z, mean, var


def fused_linear_bn_ok(rows, cin, cout):
    """Shapes the statistics epilogue covers (the split GEMM kernel, either arithmetic of the step)."""
    return rows >= 64 and cout >= 64 and cin >= 32


# conv5's tail and the VLAD assignment / aggregation are two autograd nodes whose backward passes share work: the feature gradient df
# the second one forms is what the first one's backward reads twice (row dot products + BatchNorm sums, then dz).  A TailLink, handed
# from the first node's call site to the second's, lets the second continue its product through the tail's backward while the
# accumulators are in registers (epc_vlad_df_tail): it then returns du -- the gradient of the BatchNorm output -- in df's place and
# leaves the column sums in the link; the first node's backward finishes with one pass.  Only legal when the features have no other
# consumer with a gradient (knowledge distillation with GAMMA != 0 has one: kd_training turns FUSE_TAIL_BACKWARD off).
FUSE_TAIL_BACKWARD = True


class TailLink:
    __slots__ = ("z", "rn", "mean", "var", "gamma", "beta", "eps", "du", "sums")

    def __init__(self):
        self.z = self.rn = self.mean = self.var = self.gamma = self.beta = None
        self.eps = 0.0
        self.du = self.sums = None


def tail_link_of(f):
    """The TailLink of a feature tensor made by tf_util.conv1d_l2_normalized in training mode (None otherwise)."""
    return getattr(f, "_epc_tail_link", None)


class LinearBatchNormTrain(torch.autograd.Function):
    """Linear followed by BatchNormTrain (utils/tf_util.py:94-106 in training mode) as ONE node: the batch statistics come out
    of the GEMM's epilogue instead of a pass over z.  ``rownorm``: conv5's tail -- l2_normalize(relu(bn(z))) over the channels
    (models/epc-net.py:136-148), the BatchNorm output not materialised.  Returns (y, mean, var); the backward is the BatchNorm
    backward (ReLU mask recomputed from z) followed by the two GEMMs of the linear layer; the bias gradient in front of a
    training-mode BatchNorm is exactly zero and is not computed."""

    @staticmethod
    def forward(ctx, x, W, b, gamma, beta, eps, relu, rownorm, f16x3=None, link=None):
        x = x.contiguous()
        ctx.link = link if rownorm else None
        # f16x3: the (activation, weight) scale exponents of the split-fp16 form, passed by the call site that knows its operands
        # are BatchNorm'd block outputs (conv5 of either network: tf_util.conv1d_l2_normalized / conv1d with 1024 outputs)
        z, mean, var = _gemm_with_stats(x, W, b, f16x3)
        rows, C = z.shape
        if rownorm:
            y = torch.empty_like(z)
            rn = torch.empty(rows, dtype=torch.float32, device=z.device)
            L.check(L.lib().epc_bn_relu_rownorm_fwd(z.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(),
                                                    beta.data_ptr(), float(eps), rows, C, y.data_ptr(), rn.data_ptr(), _st()))
            ctx.save_for_backward(x, W, z, mean, var, gamma, beta, rn)
            if ctx.link is not None:
                lk = ctx.link
                lk.z, lk.rn, lk.mean, lk.var, lk.gamma, lk.beta, lk.eps = z, rn, mean, var, gamma, beta, float(eps)
        else:
            y = torch.empty_like(z)
            L.check(L.lib().epc_bn_apply_fwd(z.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                             float(eps), int(relu), rows, C, y.data_ptr(), _st()))
            ctx.save_for_backward(x, W, z, mean, var, gamma, beta)
        ctx.eps, ctx.relu, ctx.rownorm = float(eps), int(relu), bool(rownorm)
        ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)
        return y, mean, var

    @staticmethod
    def backward(ctx, dy, _dm, _dv):
        if dy is None:
            return (None,) * 10
        if ctx.rownorm:
            x, W, z, mean, var, gamma, beta, rn = ctx.saved_tensors
        else:
            x, W, z, mean, var, gamma, beta = ctx.saved_tensors
        dy = dy.contiguous()
        rows, C = z.shape
        cin = x.shape[1]
        if ctx.rownorm:
            lk = ctx.link
            if lk is not None and lk.du is not None:
                # the consumer's backward went through the tail already (epc_vlad_df_tail): dy IS du, the column sums are known
                du, sums = lk.du, lk.sums
                lk.du = lk.sums = None
                if dy.data_ptr() != du.data_ptr():
                    raise EpcNetError(-1, "conv5's features have a second consumer with a gradient: set ops.FUSE_TAIL_BACKWARD = False")
                dbeta, dgamma = sums[0], sums[1]
                dz = du          # in place (out of place measured the same: 2.92 ms)
                L.check(L.lib().epc_bn_apply_bwd_given(du.data_ptr(), z.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(),
                                                       beta.data_ptr(), dbeta.data_ptr(), dgamma.data_ptr(), ctx.eps, rows, C,
                                                       dz.data_ptr(), _st()))
            else:
                dz, dgamma, dbeta = _bn_relu_rownorm_bwd(dy, z, rn, mean, var, gamma, beta, ctx.eps)
            dx = gemm(dz, W, trans_b=True, fast=True) if ctx.needs_input_grad[0] else None
            dW = gemm(x, dz, trans_a=True, splitk=_splitk_for(cin, C, rows), fast=True, deterministic=True)
            return dx, dW, None, dgamma, dbeta, None, None, None, None, None
        dgamma = torch.empty(C, dtype=torch.float32, device=z.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=z.device)
        ws, n = _ws(rows, C, z.device)
        if C == 64 and cin == 64 and not ctx.rownorm and _GEMM_PRECISION == "bf16x6" and W.is_contiguous():
            # the thin layers: BatchNorm sums, then ONE pass for dz (never written), dx and dW (epc_linear_bn_bwd64)
            dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
            dW = torch.empty_like(W)
            pf = L.lib().epc_linear_bn_bwd64_partial_floats(rows)
            part = _splitk_ws(pf, z.device)
            L.check(L.lib().epc_linear_bn_bwd64(dy.data_ptr(), z.data_ptr(), x.data_ptr(), W.data_ptr(), mean.data_ptr(),
                                                var.data_ptr(), gamma.data_ptr(), beta.data_ptr(), ctx.eps, ctx.relu, rows,
                                                dx.data_ptr() if dx is not None else None, dW.data_ptr(), dgamma.data_ptr(),
                                                dbeta.data_ptr(), ws.data_ptr(), n, part.data_ptr(), part.numel(), _st()))
            return dx, dW, None, dgamma, dbeta, None, None, None, None, None
        dz = torch.empty_like(z)
        L.check(L.lib().epc_bn_apply_bwd(dy.data_ptr(), z.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(),
                                         beta.data_ptr(), ctx.eps, ctx.relu, rows, C, dz.data_ptr(),
                                         dgamma.data_ptr(), dbeta.data_ptr(), ws.data_ptr(), n, _st()))
        dx = gemm(dz, W, trans_b=True, fast=True) if ctx.needs_input_grad[0] else None
        dW = gemm(x, dz, trans_a=True, splitk=_splitk_for(cin, C, rows), fast=True, deterministic=True)
        return dx, dW, None, dgamma, dbeta, None, None, None, None, None


class BatchNormReluRowNorm(torch.autograd.Function):
    """l2_normalize(relu(batch_norm_train(z)), 1) for conv5's 1024 channels (models/epc-net.py:136-148) without
    materialising the BatchNorm output: (f, mean, var).  Backward: two passes over (df, z) (epc_bn_relu_rownorm_bwd)."""

    @staticmethod
    def forward(ctx, z, gamma, beta, eps):
        z = z.contiguous()
        rows, C = z.shape
        mean = torch.empty(C, dtype=torch.float32, device=z.device)
        var = torch.empty(C, dtype=torch.float32, device=z.device)
        ws, n = _ws(rows, C, z.device)
        L.check(L.lib().epc_col_moments(z.data_ptr(), rows, C, mean.data_ptr(), var.data_ptr(), ws.data_ptr(), n, _st()))
        f = torch.empty_like(z)
        rn = torch.empty(rows, dtype=torch.float32, device=z.device)
        L.check(L.lib().epc_bn_relu_rownorm_fwd(z.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(),
                                                beta.data_ptr(), float(eps), rows, C, f.data_ptr(), rn.data_ptr(), _st()))
        ctx.save_for_backward(z, mean, var, gamma, beta, rn)
        ctx.eps = float(eps)
        ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)
        return f, mean, var

    @staticmethod
    def backward(ctx, df, _dm, _dv):
        if df is None:
            return None, None, None, None
        z, mean, var, gamma, beta, rn = ctx.saved_tensors
        dz, dgamma, dbeta = _bn_relu_rownorm_bwd(df.contiguous(), z, rn, mean, var, gamma, beta, ctx.eps)
        return dz, dgamma, dbeta, None


def _bn_relu_rownorm_bwd(df, z, rn, mean, var, gamma, beta, eps):
    """Backward of f = l2_normalize(relu(bn_train(z))) (epc_bn_relu_rownorm_bwd): (dz, dgamma, dbeta)."""
    rows, C = z.shape
    dz = torch.empty_like(z)
    sums = torch.empty((2, C), dtype=torch.float32, device=z.device)
    rowdot = torch.empty(rows, dtype=torch.float32, device=z.device)
    pf = L.lib().epc_bn_relu_rownorm_bwd_partial_floats(rows)
    part = _splitk_ws(pf, z.device)
    L.check(L.lib().epc_bn_relu_rownorm_bwd(df.data_ptr(), z.data_ptr(), rn.data_ptr(), mean.data_ptr(), var.data_ptr(),
                                            gamma.data_ptr(), beta.data_ptr(), float(eps), rows, C, dz.data_ptr(),
                                            sums.data_ptr(), rowdot.data_ptr(), part.data_ptr(), part.numel(), _st()))
    return dz, sums[1], sums[0]


def bn_apply_train(z, mean, var, gamma, beta, eps, relu):
    """act(batch_norm(z)) with GIVEN batch moments (epc_bn_apply_fwd), no autograd: what a fused node leaves unmaterialised."""
    z = z.detach().contiguous()
    y = torch.empty_like(z)
    L.check(L.lib().epc_bn_apply_fwd(z.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                     float(eps), int(bool(relu)), z.shape[0], z.shape[1], y.data_ptr(), _st()))
    return y


def bn_inference(z, mean, var, gamma, beta, eps=BN_EPS, relu=False):
    """Inference-mode BN with stored statistics (no gradient path is needed by the reference in this mode)."""
    z = z.contiguous()
    rows, C = z.shape
    y = torch.empty_like(z)
    L.check(L.lib().epc_bn_apply_fwd(z.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                     float(eps), int(relu), rows, C, y.data_ptr(), _st()))
    return y


class KnnGraph:
    """Static kNN graph of a batch of clouds in index form (utils/tf_util.py:647-666): non-differentiable."""

    def __init__(self, xyz):
        from .utils import tf_util
        self.xyz = xyz.contiguous().float()   # callers pass Z-ordered clouds (morton_sort) for speed; any order is exact
        self.num_clouds, self.n = int(xyz.shape[0]), int(xyz.shape[1])
        self.kth, self.idx, self.cnt = tf_util.knn_index(self.xyz)
        self._transposed = None
        self._overflow = None

    def overflow(self):
        """(ovf_cnt (clouds,), ovf_list (clouds, n)): per cloud the points whose own list overflowed (cnt > cap: exact ties of
        duplicated / zero-padded clouds) -- the transposed graph does not list them, the chain's gather backward visits them with
        the exact test (epc_knn_overflow_lists).  Built on first use."""
        if self._overflow is None:
            dev = self.xyz.device
            oc = torch.empty(self.num_clouds, dtype=torch.int32, device=dev)
            ol = torch.empty((self.num_clouds, self.n), dtype=torch.int32, device=dev)
            L.check(L.lib().epc_knn_overflow_lists(self.cnt.data_ptr(), L.EPC_KNN_CAP, self.num_clouds, self.n, oc.data_ptr(),
                                                   ol.data_ptr(), _st()))
            self._overflow = (oc, ol)
        return self._overflow

    def transposed(self):
        """(rdeg, roff, rlist): for every point the points that list it (epc_knn_transpose), built on first use -- the
        backward of every block of a step gathers over it."""
        if self._transposed is None:
            rdeg, roff, cursor, rlist, oc, ol = self._build_transposed()
            L.check(L.lib().epc_knn_transpose(self.idx.data_ptr(), self.cnt.data_ptr(), L.EPC_KNN_CAP, self.num_clouds, self.n,
                                              rdeg.data_ptr(), roff.data_ptr(), cursor.data_ptr(), rlist.data_ptr(), _st()))
            L.check(L.lib().epc_knn_overflow_lists(self.cnt.data_ptr(), L.EPC_KNN_CAP, self.num_clouds, self.n, oc.data_ptr(),
                                                   ol.data_ptr(), _st()))
            self._transposed, self._overflow = (rdeg, roff, rlist), (oc, ol)
        return self._transposed

    def _build_transposed(self):
        M = self.num_clouds * self.n
        dev = self.xyz.device
        rdeg = torch.empty(M, dtype=torch.int32, device=dev)
        roff = torch.empty(M, dtype=torch.int32, device=dev)
        cursor = torch.empty(M, dtype=torch.int32, device=dev)
        rlist = torch.empty(M * L.EPC_KNN_CAP, dtype=torch.int32, device=dev)
        oc = torch.empty(self.num_clouds, dtype=torch.int32, device=dev)
        ol = torch.empty((self.num_clouds, self.n), dtype=torch.int32, device=dev)
        self._scratch = cursor
        return rdeg, roff, cursor, rlist, oc, ol



class NeighbourMean(torch.autograd.Function):
    """xm = matmul(mask, x) / float(k) (models/epc-net.py:70-71) in index form; backward = mask^T @ dxm / k."""

    @staticmethod
    def forward(ctx, x, graph, k):
        x = x.contiguous()
        ctx.graph, ctx.k = graph, int(k)
        xm = torch.empty_like(x)
        g = graph
        L.check(L.lib().epc_neighbour_mean_fwd(x.data_ptr(), g.xyz.data_ptr(), g.idx.data_ptr(), g.cnt.data_ptr(),
                                               g.kth.data_ptr(), L.EPC_KNN_CAP, g.num_clouds, g.n, int(k), xm.data_ptr(),
                                               _st()))
        return xm

    @staticmethod
    def backward(ctx, dxm):
        g = ctx.graph
        dxm = dxm.contiguous()
        dx = torch.empty_like(dxm)
        rdeg, roff, rlist = g.transposed()
        L.check(L.lib().epc_neighbour_mean_bwd_gather(dxm.data_ptr(), g.xyz.data_ptr(), g.cnt.data_ptr(), g.kth.data_ptr(),
                                                      L.EPC_KNN_CAP, rdeg.data_ptr(), roff.data_ptr(), rlist.data_ptr(),
                                                      g.num_clouds, g.n, ctx.k, dx.data_ptr(), _st()))
        return dx, None, None


class NeighbourMeanDiff(torch.autograd.Function):
    """(xm, xm - x) with xm = matmul(mask, x) / float(k) (models/epc-net.py:70-72): the mean and the difference the block
    feeds to conv_a, one launch forward, one gather backward (dx = mask^T (dxm + ddiff) / k - ddiff)."""

    @staticmethod
    def forward(ctx, x, graph, k):
        x = x.contiguous()
        ctx.graph, ctx.k = graph, int(k)
        xm, diff = torch.empty_like(x), torch.empty_like(x)
        g = graph
        L.check(L.lib().epc_neighbour_mean_diff_fwd(x.data_ptr(), g.xyz.data_ptr(), g.idx.data_ptr(), g.cnt.data_ptr(),
                                                    g.kth.data_ptr(), L.EPC_KNN_CAP, g.num_clouds, g.n, int(k),
                                                    xm.data_ptr(), diff.data_ptr(), _st()))
        return xm, diff

    @staticmethod
    def backward(ctx, dxm, ddiff):
        g = ctx.graph
        dxm, ddiff = dxm.contiguous(), ddiff.contiguous()
        dx = torch.empty_like(dxm)
        rdeg, roff, rlist = g.transposed()
        L.check(L.lib().epc_neighbour_mean_diff_bwd_gather(dxm.data_ptr(), ddiff.data_ptr(), g.xyz.data_ptr(),
                                                           g.cnt.data_ptr(), g.kth.data_ptr(), L.EPC_KNN_CAP, rdeg.data_ptr(),
                                                           roff.data_ptr(), rlist.data_ptr(), g.num_clouds, g.n, ctx.k,
                                                           dx.data_ptr(), _st()))
        return dx, None, None


class ProxyConvTail(torch.autograd.Function):
    """The rest of a ProxyConv block behind its leading conv (models/epc-net.py:70-86) as ONE node:
        x1 = matmul(mask, x) / k;  t = x1 - x;  t = conv_a(t);  t = conv_b(t);  out = t + x1        (conv = 64 -> 64 + BN + ReLU)
    -> (out, z_a, mean_a, var_a, z_b, mean_b, var_b) (the pre-activations and batch moments: moving averages, mask taps).
    What the single node buys: the activation between conv_a and conv_b is never written (conv_b forms relu(bn(z_a)) as it loads
    z_a, forward and backward), the residual is added inside conv_b's BatchNorm pass, and in the backward conv_a's input gradient
    leaves with the residual path's gradient already added, so the neighbour backward gathers ONE tensor instead of two."""

    @staticmethod
    def forward(ctx, x, graph, k, Wa, ba, ga, bta, Wb, bb, gb, btb, eps):
        x = x.contiguous()
        rows = x.shape[0]
        assert x.shape[1] == 64 and Wa.shape == (64, 64) and Wb.shape == (64, 64) and _GEMM_PRECISION == "bf16x6"
        Wa, Wb = Wa.contiguous(), Wb.contiguous()
        g = graph
        lib = L.lib()
        xm, diff = torch.empty_like(x), torch.empty_like(x)
        L.check(lib.epc_neighbour_mean_diff_fwd(x.data_ptr(), g.xyz.data_ptr(), g.idx.data_ptr(), g.cnt.data_ptr(),
                                                g.kth.data_ptr(), L.EPC_KNN_CAP, g.num_clouds, g.n, int(k), xm.data_ptr(),
                                                diff.data_ptr(), _st()))
        za, ma, va = _gemm_with_stats(diff, Wa, ba)
        zb = torch.empty_like(za)
        mb = torch.empty(64, dtype=torch.float32, device=x.device)
        vb = torch.empty(64, dtype=torch.float32, device=x.device)
        ws, n = _ws(rows, 64, x.device)
        L.check(lib.epc_linear_stats64_bn(za.data_ptr(), ma.data_ptr(), va.data_ptr(), ga.data_ptr(), bta.data_ptr(), float(eps),
                                          Wb.data_ptr(), bb.data_ptr() if bb is not None else None, rows, zb.data_ptr(),
                                          mb.data_ptr(), vb.data_ptr(), ws.data_ptr(), n, _st()))
        out = torch.empty_like(x)
        L.check(lib.epc_bn_apply_add_fwd(zb.data_ptr(), mb.data_ptr(), vb.data_ptr(), gb.data_ptr(), btb.data_ptr(), float(eps), 1,
                                         rows, 64, xm.data_ptr(), out.data_ptr(), _st()))
        ctx.save_for_backward(diff, Wa, za, ma, va, ga, bta, Wb, zb, mb, vb, gb, btb)
        ctx.graph, ctx.k, ctx.eps = graph, int(k), float(eps)
        ctx.mark_non_differentiable(za, ma, va, zb, mb, vb)
        ctx.set_materialize_grads(False)
        return out, za, ma, va, zb, mb, vb

    @staticmethod
    def backward(ctx, dout, *_unused):
        if dout is None:
            return (None,) * 12
        diff, Wa, za, ma, va, ga, bta, Wb, zb, mb, vb, gb, btb = ctx.saved_tensors
        dout = dout.contiguous()
        rows = za.shape[0]
        dev = za.device
        lib, g, eps = L.lib(), ctx.graph, ctx.eps
        new = lambda: torch.empty(64, dtype=torch.float32, device=dev)
        ws, n = _ws(rows, 64, dev)
        pf = lib.epc_linear_bn_bwd64_partial_floats(rows)
        part = _splitk_ws(pf, dev)
        # conv_b: its input is relu(bn(z_a)), re-formed from z_a
        dta, dWb, dgb, dbtb = torch.empty_like(za), torch.empty_like(Wb), new(), new()
        L.check(lib.epc_linear_bn_bwd64_ex(dout.data_ptr(), zb.data_ptr(), za.data_ptr(), ma.data_ptr(), va.data_ptr(),
                                           ga.data_ptr(), bta.data_ptr(), Wb.data_ptr(), mb.data_ptr(), vb.data_ptr(),
                                           gb.data_ptr(), btb.data_ptr(), eps, 1, rows, dta.data_ptr(), None, dWb.data_ptr(),
                                           dgb.data_ptr(), dbtb.data_ptr(), ws.data_ptr(), n, part.data_ptr(), part.numel(), _st()))
        # conv_a: the gradient of its input (xm - x) leaves as s = ddiff + dout, dout being what reaches xm by the residual
        s, dWa, dga, dbta = torch.empty_like(za), torch.empty_like(Wa), new(), new()
        L.check(lib.epc_linear_bn_bwd64_ex(dta.data_ptr(), za.data_ptr(), diff.data_ptr(), None, None, None, None, Wa.data_ptr(),
                                           ma.data_ptr(), va.data_ptr(), ga.data_ptr(), bta.data_ptr(), eps, 1, rows, s.data_ptr(),
                                           dout.data_ptr(), dWa.data_ptr(), dga.data_ptr(), dbta.data_ptr(), ws.data_ptr(), n,
                                           part.data_ptr(), part.numel(), _st()))
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty_like(za)
            rdeg, roff, rlist = g.transposed()
            L.check(lib.epc_neighbour_mean_diff_bwd_gather_sum(s.data_ptr(), dout.data_ptr(), g.xyz.data_ptr(), g.cnt.data_ptr(),
                                                               g.kth.data_ptr(), L.EPC_KNN_CAP, rdeg.data_ptr(), roff.data_ptr(),
                                                               rlist.data_ptr(), g.num_clouds, g.n, ctx.k, dx.data_ptr(), _st()))
        # (bias gradients in front of a training-mode BatchNorm are exactly zero: LinearBatchNormTrain)
        return dx, None, None, dWa, None, dga, dbta, dWb, None, dgb, dbtb, None


# The backbone chain's FORWARD as one persistent launch (csrc/train_chain_persist.hip: grid-wide barriers instead of kernel boundaries)
# whenever the library covers the row count (epc_chain_persist_ok); the backward stays the launch chain (its persistent form was
# built, was parity-green and measured slower: DESIGN.md 4).  Nothing runs beside the forward in the step (the data-parallel step's
# collectives overlap the BACKWARD), so no kernel that waits for it shares the device with it.
CHAIN_PERSIST_FWD = True
CHAIN_SPIN_TICKS = 0          # spin budget of a grid barrier in 10-ns ticks; 0: the library's default (a quarter second)
_CHAIN_WS = {}


def chain_workspace(device):
    """The persistent chain's workspace of `device` (sequence number, error word, the barriers' tagged partials): zeroed once, then
    left to the library.  One launch at a time: the chain's launches of a device are stream-ordered."""
    key = (device.type, device.index)
    t = _CHAIN_WS.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("the persistent chain's workspace must exist before a graph capture (its one-time zeroing would be "
                               "replayed): run one eager step first, as TrainStep's warm-up does")
        t = torch.empty(L.lib().epc_chain_persist_workspace_bytes(), dtype=torch.uint8, device=device)
        L.check(L.lib().epc_chain_persist_init(t.data_ptr(), _st()))
        _CHAIN_WS[key] = t
    return t


def chain_persist_check(device=None):
    """Synchronises and raises EpcNetError when a persistent chain launch was abandoned (a barrier ran out of its spin budget);
    the workspace is then reset so that the next step can run."""
    for key, t in list(_CHAIN_WS.items()):
        if device is not None and key != (device.type, device.index):
            continue
        rc = L.lib().epc_chain_persist_status(t.data_ptr(), _st())
        if rc != L.EPC_OK:
            L.lib().epc_chain_persist_reset(t.data_ptr(), _st())
            L.check(rc)


def chain_ok(rows):
    """The fused backbone chain (ProxyConvChain) covers both GEMM arithmetics of the step and any row count."""
    return rows >= 1


class ProxyConvChain(torch.autograd.Function):
    """The whole 64-channel backbone behind conv1's product as ONE autograd node on the fused chain launches of
    csrc/train_chain.hip (models/epc-net.py:66-134 in training mode): for block b = 1 .. nblocks
        x = relu(bn0_b(z0_b));  xm = matmul(mask, x) / k;  d = xm - x;  za = d Wa + ba;  zb = relu(bna(za)) Wb + bb;
        out_b = relu(bnb(zb)) + xm -> columns [64 (b - 1), 64 b) of the concat (:134);  z0_{b+1} = out_b W0_{b+1} + b0_{b+1}
    with z0_1 = conv1's pre-activation (the input; its K = 3 product keeps its own small kernels).  Every BatchNorm is in
    training mode: batch moments from the producer's partials, pooled in the consumer's prologue.  Forward: 3 launches per block
    (+ 1 for z0_1's moments); backward: 4 per block + 2 -- against 9 + 1 and ~14 of the per-layer operators.

    apply(z01, graph, k, eps, nblocks, pieces_fwd, pieces_bwd, want_bf16, *params) with params = for block 1: gamma0, beta0, then
    Wa, ba, gamma_a, beta_a, Wb, bb, gamma_b, beta_b; for every later block: W0, b0, gamma0, beta0 and the same eight.
    Returns (cat (rows, 64 nblocks), then per block: [z0 -- blocks after the first --], mean0, var0, za, mean_a, var_a, zb, mean_b,
    var_b, then the concat's bf16 copy or None) -- the pre-activations
    and batch moments feed the moving averages and the test hook's mask taps; only cat is differentiable.  Bias gradients in front of
    a training-mode BatchNorm are exactly zero and are not computed (LinearBatchNormTrain)."""

    @staticmethod
    def _split(nblocks, params):
        blocks, at = [], 0
        for b in range(nblocks):
            n = 10 if b == 0 else 12
            p = list(params[at:at + n])
            at += n
            if b == 0:
                p = [None, None] + p
            blocks.append(p)      # [W0, b0, g0, bt0, Wa, ba, ga, bta, Wb, bb, gb, btb]
        assert at == len(params)
        return blocks

    @staticmethod
    def forward(ctx, z01, graph, k, eps, nblocks, pieces_fwd, pieces_bwd, want_bf16, *params):
        lib = L.lib()
        z01 = z01.contiguous()
        rows = int(z01.shape[0])
        assert z01.shape[1] == 64 and rows == graph.num_clouds * graph.n
        dev = z01.device
        blocks = ProxyConvChain._split(nblocks, [p.contiguous() if p is not None else None for p in params])
        P = lib.epc_chain_parts(rows)
        width = 64 * nblocks
        new = lambda: torch.empty((rows, 64), dtype=torch.float32, device=dev)
        vec = lambda: torch.empty(64, dtype=torch.float32, device=dev)
        stats = lambda: torch.empty(P * 192, dtype=torch.float32, device=dev)
        ptr = lambda t: t.data_ptr() if t is not None else None
        cat = torch.empty((rows, width), dtype=torch.float32, device=dev)
        # the bf16 head (Conv5VladHead, mode "bf16") reads the concat as bf16: written beside the f32 tensor by the launches that form it
        # (``want_bf16``: the caller knows that the bf16 streamed head is what consumes the concat; the copy is an OUTPUT of the node,
        # handed on explicitly -- tf_util.proxyconv_backbone -> conv1d_l2_normalized -> LazyConv5Features -> Conv5VladHead)
        cat16 = torch.empty((rows, width), dtype=torch.bfloat16, device=dev) if want_bf16 else None
        g = graph
        if CHAIN_PERSIST_FWD and nblocks <= L.EPC_CHAIN_MAX_BLOCKS and lib.epc_chain_persist_ok(rows):
            saved, outs = ProxyConvChain._forward_persistent(lib, z01, g, k, eps, nblocks, pieces_fwd, blocks, cat, cat16)
            ctx.save_for_backward(cat, *saved, *[p for p in params])
            ctx.graph, ctx.k, ctx.eps, ctx.nblocks = graph, int(k), float(eps), int(nblocks)
            ctx.pieces_bwd, ctx.n_params = int(pieces_bwd), len(params)
            ctx.mark_non_differentiable(*outs, *([cat16] if cat16 is not None else []))
            ctx.set_materialize_grads(False)
            return (cat,) + tuple(outs) + (cat16,)
        st0 = stats()
        L.check(lib.epc_chain_stats(z01.data_ptr(), rows, st0.data_ptr(), _st()))
        z0, in_stats, in_bias = z01, st0, None
        saved, outs = [], []
        for b, (W0, b0, g0, bt0, Wa, ba, ga, bta, Wb, bb, gb, btb) in enumerate(blocks):
            m0, v0, ma, va, mb, vb = vec(), vec(), vec(), vec(), vec(), vec()
            xm, d, za, zb = new(), new(), new(), new()
            st_a, st_b = stats(), stats()
            L.check(lib.epc_chain_fwd_gather(z0.data_ptr(), in_stats.data_ptr(), ptr(in_bias), m0.data_ptr(), v0.data_ptr(),
                                             g0.data_ptr(), bt0.data_ptr(), float(eps), g.xyz.data_ptr(), g.idx.data_ptr(),
                                             g.cnt.data_ptr(), g.kth.data_ptr(), L.EPC_KNN_CAP, g.num_clouds, g.n, int(k),
                                             Wa.data_ptr(), ptr(ba), xm.data_ptr(), d.data_ptr(), za.data_ptr(), st_a.data_ptr(),
                                             int(pieces_fwd), _st()))
            L.check(lib.epc_chain_fwd_linear(za.data_ptr(), st_a.data_ptr(), ptr(ba), ma.data_ptr(), va.data_ptr(), ga.data_ptr(),
                                             bta.data_ptr(), float(eps), None, None, 0, None, Wb.data_ptr(), ptr(bb), zb.data_ptr(),
                                             st_b.data_ptr(), rows, int(pieces_fwd), _st()))
            slice_ptr = cat.data_ptr() + 4 * 64 * b
            slice16 = cat16.data_ptr() + 2 * 64 * b if cat16 is not None else None
            if b + 1 < nblocks:
                W0n, b0n = blocks[b + 1][0], blocks[b + 1][1]
                z0n, st0n = new(), stats()
                L.check(lib.epc_chain_fwd_linear(zb.data_ptr(), st_b.data_ptr(), ptr(bb), mb.data_ptr(), vb.data_ptr(), gb.data_ptr(),
                                                 btb.data_ptr(), float(eps), xm.data_ptr(), slice_ptr, width, slice16, W0n.data_ptr(),
                                                 ptr(b0n), z0n.data_ptr(), st0n.data_ptr(), rows, int(pieces_fwd), _st()))
            else:
                z0n, st0n, b0n = None, None, None
                L.check(lib.epc_chain_fwd_linear(zb.data_ptr(), st_b.data_ptr(), ptr(bb), mb.data_ptr(), vb.data_ptr(), gb.data_ptr(),
                                                 btb.data_ptr(), float(eps), xm.data_ptr(), slice_ptr, width, slice16, None, None, None,
                                                 None, rows, int(pieces_fwd), _st()))
            saved += [z0, d, za, zb, m0, v0, ma, va, mb, vb]
            outs += ([z0] if b > 0 else []) + [m0, v0, za, ma, va, zb, mb, vb]      # (block 1's z0 is the input itself)
            z0, in_stats, in_bias = z0n, st0n, b0n
        ctx.save_for_backward(cat, *saved, *[p for p in params])
        ctx.graph, ctx.k, ctx.eps, ctx.nblocks = graph, int(k), float(eps), int(nblocks)
        ctx.pieces_bwd, ctx.n_params = int(pieces_bwd), len(params)
        ctx.mark_non_differentiable(*outs, *([cat16] if cat16 is not None else []))
        ctx.set_materialize_grads(False)
        return (cat,) + tuple(outs) + (cat16,)

    @staticmethod
    def _forward_persistent(lib, z01, g, k, eps, nblocks, pieces_fwd, blocks, cat, cat16):
        """The whole forward as one launch (epc_chain_fwd_persist); returns (saved, outs) in forward()'s order."""
        rows, dev = int(z01.shape[0]), z01.device
        new = lambda: torch.empty((rows, 64), dtype=torch.float32, device=dev)
        vec = lambda: torch.empty(64, dtype=torch.float32, device=dev)
        ptr = lambda t: t.data_ptr() if t is not None else None
        a = L.ChainFwdArgs()
        a.nblocks = nblocks
        a.xyz, a.idx, a.cnt, a.kth = g.xyz.data_ptr(), g.idx.data_ptr(), g.cnt.data_ptr(), g.kth.data_ptr()
        a.cap, a.num_clouds, a.n, a.knn = L.EPC_KNN_CAP, g.num_clouds, g.n, int(k)
        a.cat, a.cat_bf16 = cat.data_ptr(), ptr(cat16)
        a.eps = float(eps)
        a.workspace, a.spin_ticks = chain_workspace(dev).data_ptr(), int(CHAIN_SPIN_TICKS)
        saved, outs = [], []
        z0 = z01
        for b, (W0, b0, g0, bt0, Wa, ba, ga, bta, Wb, bb, gb, btb) in enumerate(blocks):
            m0, v0, ma, va, mb, vb = vec(), vec(), vec(), vec(), vec(), vec()
            d, za, zb = new(), new(), new()
            nxt = blocks[b + 1] if b + 1 < nblocks else None
            z0n = new() if nxt is not None else None
            B = a.blk[b]
            B.gamma0, B.beta0, B.in_bias = g0.data_ptr(), bt0.data_ptr(), (ptr(b0) if b > 0 else None)
            B.Wa, B.ba, B.gamma_a, B.beta_a = Wa.data_ptr(), ptr(ba), ga.data_ptr(), bta.data_ptr()
            B.Wb, B.bb, B.gamma_b, B.beta_b = Wb.data_ptr(), ptr(bb), gb.data_ptr(), btb.data_ptr()
            B.W0_next, B.b0_next = (nxt[0].data_ptr(), ptr(nxt[1])) if nxt is not None else (None, None)
            B.z0, B.z0_next = z0.data_ptr(), ptr(z0n)
            B.mean0, B.var0, B.mean_a, B.var_a, B.mean_b, B.var_b = (t.data_ptr() for t in (m0, v0, ma, va, mb, vb))
            B.d, B.za, B.zb = d.data_ptr(), za.data_ptr(), zb.data_ptr()
            saved += [z0, d, za, zb, m0, v0, ma, va, mb, vb]
            outs += ([z0] if b > 0 else []) + [m0, v0, za, ma, va, zb, mb, vb]
            z0 = z0n
        L.check(lib.epc_chain_fwd_persist(ctypes.byref(a), int(pieces_fwd), _st()))
        return saved, outs

    @staticmethod
    def backward(ctx, dcat, *_unused):
        nb = ctx.nblocks
        n_in = 8 + ctx.n_params
        if dcat is None:
            return (None,) * n_in
        lib = L.lib()
        sv = ctx.saved_tensors
        cat = sv[0]
        per = [sv[1 + 10 * b: 11 + 10 * b] for b in range(nb)]          # z0, d, za, zb, m0, v0, ma, va, mb, vb
        params = list(sv[1 + 10 * nb:])
        blocks = ProxyConvChain._split(nb, params)
        dcat = dcat.contiguous()
        rows, width = int(cat.shape[0]), int(cat.shape[1])
        dev = cat.device
        g, eps, k, pc = ctx.graph, ctx.eps, ctx.k, ctx.pieces_bwd
        P = lib.epc_chain_parts(rows)
        new = lambda: torch.empty((rows, 64), dtype=torch.float32, device=dev)
        vec = lambda: torch.empty(64, dtype=torch.float32, device=dev)
        sums = lambda: torch.empty(P * 128, dtype=torch.float32, device=dev)
        n_layers = 3 * nb - 1
        parts = _splitk_ws(n_layers * P * 4096, dev)[: n_layers * P * 4096].view(n_layers, P * 4096)
        layer_dw, layer_at = [], 0
        rdeg, roff, rlist = g.transposed()
        ovc, ovl = g.overflow()
        grads = [None] * ctx.n_params
        at_of = lambda b: 0 if b == 0 else 10 + 12 * (b - 1)          # index of block b's first parameter in `params`
        z_b, m_b, v_b = per[nb - 1][3], per[nb - 1][8], per[nb - 1][9]
        sums_b = sums()
        last = dcat.data_ptr() + 4 * 64 * (nb - 1)
        L.check(lib.epc_chain_sums(last, width, z_b.data_ptr(), m_b.data_ptr(), v_b.data_ptr(), blocks[nb - 1][10].data_ptr(),
                                   blocks[nb - 1][11].data_ptr(), eps, rows, sums_b.data_ptr(), _st()))
        gout = None
        dz01 = None
        for b in range(nb - 1, -1, -1):
            z0, d, za, zb, m0, v0, ma, va, mb, vb = per[b]
            W0, b0, g0, bt0, Wa, ba, ga, bta, Wb, bb, gb, btb = blocks[b]
            base = at_of(b) + (2 if b == 0 else 4)                   # Wa's index
            dy_ptr, dy_stride = (gout.data_ptr(), 64) if gout is not None else (dcat.data_ptr() + 4 * 64 * b, width)
            # conv_b: its input is relu(bn_a(za)), re-formed from za; leaves conv_a's BatchNorm sums
            dya, dgb, dbtb, sums_a = new(), vec(), vec(), sums()
            L.check(lib.epc_chain_bwd_linear(dy_ptr, dy_stride, zb.data_ptr(), mb.data_ptr(), vb.data_ptr(), gb.data_ptr(),
                                             btb.data_ptr(), eps, sums_b.data_ptr(), dgb.data_ptr(), dbtb.data_ptr(), Wb.data_ptr(),
                                             za.data_ptr(), 64, ma.data_ptr(), va.data_ptr(), ga.data_ptr(), bta.data_ptr(),
                                             dya.data_ptr(), None, 0, parts[layer_at].data_ptr(), za.data_ptr(), ma.data_ptr(),
                                             va.data_ptr(), ga.data_ptr(), bta.data_ptr(), sums_a.data_ptr(), rows, pc, _st()))
            dWb = torch.empty_like(Wb)
            layer_dw.append(dWb)
            layer_at += 1
            grads[base + 4], grads[base + 6], grads[base + 7] = dWb, dgb, dbtb
            # conv_a: the gradient of its input d = xm - x leaves as s = dd + dout (dout reaches xm by the residual path)
            s_, dga, dbta = new(), vec(), vec()
            L.check(lib.epc_chain_bwd_linear(dya.data_ptr(), 64, za.data_ptr(), ma.data_ptr(), va.data_ptr(), ga.data_ptr(),
                                             bta.data_ptr(), eps, sums_a.data_ptr(), dga.data_ptr(), dbta.data_ptr(), Wa.data_ptr(),
                                             d.data_ptr(), 64, None, None, None, None, s_.data_ptr(), dy_ptr, dy_stride,
                                             parts[layer_at].data_ptr(), None, None, None, None, None, None, rows, pc, _st()))
            dWa = torch.empty_like(Wa)
            layer_dw.append(dWa)
            layer_at += 1
            grads[base + 0], grads[base + 2], grads[base + 3] = dWa, dga, dbta
            # the gather's transpose: dx = mask^T s / k - (s - dout); leaves the leading BatchNorm's sums
            dxx, sums_0 = new(), sums()
            L.check(lib.epc_chain_bwd_gather(s_.data_ptr(), dy_ptr, dy_stride, rdeg.data_ptr(), roff.data_ptr(), rlist.data_ptr(),
                                             ovc.data_ptr(), ovl.data_ptr(), g.xyz.data_ptr(), g.kth.data_ptr(), g.num_clouds, g.n,
                                             k, z0.data_ptr(), m0.data_ptr(), v0.data_ptr(), g0.data_ptr(), bt0.data_ptr(), eps,
                                             sums_0.data_ptr(), dxx.data_ptr(), _st()))
            dg0, dbt0 = vec(), vec()
            if b > 0:
                # the leading conv: its input is the previous block's output (a slice of cat); its dx, plus the concat's gradient of
                # that slice, is d(out_{b-1}); leaves the previous block's conv_b BatchNorm sums
                pz = per[b - 1]
                pgb, pbtb = blocks[b - 1][10], blocks[b - 1][11]
                gnew, sums_prev = new(), sums()
                prev_slice = 4 * 64 * (b - 1)
                L.check(lib.epc_chain_bwd_linear(dxx.data_ptr(), 64, z0.data_ptr(), m0.data_ptr(), v0.data_ptr(), g0.data_ptr(),
                                                 bt0.data_ptr(), eps, sums_0.data_ptr(), dg0.data_ptr(), dbt0.data_ptr(),
                                                 W0.data_ptr(), cat.data_ptr() + prev_slice, width, None, None, None, None,
                                                 gnew.data_ptr(), dcat.data_ptr() + prev_slice, width, parts[layer_at].data_ptr(),
                                                 pz[3].data_ptr(), pz[8].data_ptr(), pz[9].data_ptr(), pgb.data_ptr(), pbtb.data_ptr(),
                                                 sums_prev.data_ptr(), rows, pc, _st()))
                dW0 = torch.empty_like(W0)
                layer_dw.append(dW0)
                layer_at += 1
                a0 = at_of(b)
                grads[a0 + 0], grads[a0 + 2], grads[a0 + 3] = dW0, dg0, dbt0
                gout, sums_b = gnew, sums_prev
            else:
                dz01 = new()
                L.check(lib.epc_chain_bn_bwd(dxx.data_ptr(), z0.data_ptr(), m0.data_ptr(), v0.data_ptr(), g0.data_ptr(), bt0.data_ptr(),
                                             eps, sums_0.data_ptr(), dg0.data_ptr(), dbt0.data_ptr(), rows, dz01.data_ptr(), _st()))
                grads[0], grads[1] = dg0, dbt0
        pa = (ctypes.c_void_p * n_layers)(*[parts[l].data_ptr() for l in range(n_layers)])
        pw = (ctypes.c_void_p * n_layers)(*[w.data_ptr() for w in layer_dw])
        L.check(lib.epc_chain_dw_sum(n_layers, pa, pw, rows, _st()))
        return (dz01 if ctx.needs_input_grad[0] else None, None, None, None, None, None, None, None) + tuple(grads)


class RowL2Normalize(torch.autograd.Function):
    """tf.nn.l2_normalize(x, 1) on (rows, C) (models/epc-net.py:148)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        rows, C = x.shape
        y = torch.empty_like(x)
        rn = torch.empty(rows, dtype=torch.float32, device=x.device)
        L.check(L.lib().epc_rownorm_fwd(x.data_ptr(), rows, C, y.data_ptr(), rn.data_ptr(), _st()))
        ctx.save_for_backward(y, rn)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, rn = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        L.check(L.lib().epc_rownorm_bwd(dy.data_ptr(), y.data_ptr(), rn.data_ptr(), y.shape[0], y.shape[1], dx.data_ptr(), _st()))
        return dx


class VladNormalize(torch.autograd.Function):
    """(raw - a_sum * cluster_weights2) -> intra-normalisation over F -> L2 normalisation of the flattened vector
    (loupe.py:284,292-298) in one launch; backward in one launch plus the cross-cloud cluster_weights2 sum.
    raw (B,F,64), a_sum (B,1,64), w2 (1,F,64) -> (B,F,64)."""

    @staticmethod
    def forward(ctx, raw, a_sum, w2):
        raw, a_sum, w2 = raw.contiguous(), a_sum.contiguous(), w2.contiguous()
        B, F, C = raw.shape
        out = torch.empty_like(raw)
        r1 = torch.empty((B, C), dtype=torch.float32, device=raw.device)
        r2 = torch.empty((B,), dtype=torch.float32, device=raw.device)
        L.check(L.lib().epc_vlad_normalize_fwd(raw.data_ptr(), a_sum.data_ptr(), w2.data_ptr(), B, F, C, out.data_ptr(),
                                               r1.data_ptr(), r2.data_ptr(), _st()))
        ctx.save_for_backward(out, r1, r2, w2, a_sum)
        return out

    @staticmethod
    def backward(ctx, dout):
        out, r1, r2, w2, a_sum = ctx.saved_tensors
        dout = dout.contiguous()
        B, F, C = out.shape
        draw = torch.empty_like(out)
        da = torch.empty_like(a_sum)
        L.check(L.lib().epc_vlad_normalize_bwd(dout.data_ptr(), out.data_ptr(), r1.data_ptr(), r2.data_ptr(), w2.data_ptr(),
                                               B, F, C, draw.data_ptr(), da.data_ptr(), _st()))
        dw2 = None
        if ctx.needs_input_grad[2]:
            dw2 = torch.empty(w2.shape, dtype=torch.float32, device=out.device)
            L.check(L.lib().epc_vlad_w2_grad(draw.data_ptr(), a_sum.data_ptr(), B, F, C, dw2.data_ptr(), _st()))
        return draw, da, dw2


class LazyQuadrupletLoss(torch.autograd.Function):
    """models/epc-net.py:269-284 as one launch forward and one backward (q (B,1,D), pos (B,P,D), neg (B,Nn,D),
    other (B,1,D) -> scalar)."""

    @staticmethod
    def forward(ctx, q, pos, neg, other, m1, m2):
        q, pos, neg, other = q.contiguous(), pos.contiguous(), neg.contiguous(), other.contiguous()
        B, P, D = pos.shape
        Nn = neg.shape[1]
        loss = torch.empty((), dtype=torch.float32, device=q.device)
        sel = torch.empty((B, 3), dtype=torch.int32, device=q.device)
        L.check(L.lib().epc_lazy_quadruplet_loss_fwd(q.data_ptr(), pos.data_ptr(), neg.data_ptr(), other.data_ptr(), B, P, Nn,
                                                     D, float(m1), float(m2), loss.data_ptr(), sel.data_ptr(), _st()))
        ctx.save_for_backward(q, pos, neg, other, sel)
        return loss

    @staticmethod
    def backward(ctx, dloss):
        q, pos, neg, other, sel = ctx.saved_tensors
        dloss = dloss.contiguous().float()
        B, P, D = pos.shape
        Nn = neg.shape[1]
        dq, dpos, dneg, dother = (torch.empty_like(t) for t in (q, pos, neg, other))
        L.check(L.lib().epc_lazy_quadruplet_loss_bwd(q.data_ptr(), pos.data_ptr(), neg.data_ptr(), other.data_ptr(),
                                                     sel.data_ptr(), dloss.data_ptr(), B, P, Nn, D, dq.data_ptr(),
                                                     dpos.data_ptr(), dneg.data_ptr(), dother.data_ptr(), _st()))
        return dq, dpos, dneg, dother, None, None


class Softmax64(torch.autograd.Function):
    """tf.nn.softmax over the 64 clusters (loupe.py:272)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        assert x.shape[1] == 64
        y = torch.empty_like(x)
        L.check(L.lib().epc_softmax64_fwd(x.data_ptr(), x.shape[0], y.data_ptr(), _st()))
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(y)
        L.check(L.lib().epc_softmax64_bwd(dy.data_ptr(), y.data_ptr(), y.shape[0], dx.data_ptr(), _st()))
        return dx


class VladAggregate(torch.autograd.Function):
    """vlad[b] = f[b]^T @ a[b]: (B,N,F),(B,N,C) -> (B,F,C) (loupe.py:286-291: transpose, batched matmul, transpose)."""

    @staticmethod
    def forward(ctx, f, a):
        f, a = f.contiguous(), a.contiguous()
        ctx.save_for_backward(f, a)
        return gemm(f, a, trans_a=True, splitk=max(1, min(8, f.shape[1] // 256)), deterministic=True)

    @staticmethod
    def backward(ctx, dv):
        f, a = ctx.saved_tensors
        dv = dv.contiguous()
        df = gemm(a, dv, trans_b=True, fast=True)   # (B,N,C) @ (B,F,C)^T -> (B,N,F)
        da = gemm(f, dv, fast=True)                 # (B,N,F) @ (B,F,C)   -> (B,N,C)
        return df, da


class VladAssignAggregate(torch.autograd.Function):
    """The soft assignment and the aggregation of loupe.py:255-291 in training mode as ONE node:
        z = f @ cluster_weights; a = softmax(batch_norm(z)) (slim.batch_norm, batch statistics); vlad[b] = f[b]^T @ a[b]
    -> (vlad (B, F, 64), a_sum (B, 1, 64) = sum of a over the cloud's points (:276), mean, var).  The assignment a itself stays
    inside the node: its only other consumer is a_sum, whose gradient (one row per cloud) is added to every point's row inside the
    softmax backward (epc_softmax64_bwd_bcast) instead of as an expanded (B, N, 64) tensor.  What the single node buys is the backward: the two gradients of the shared
    input f -- a dvlad^T from the aggregation, dz Wc^T from the assignment -- are ONE product, [a | dz] (rows, 128) times the
    per-cloud [dvlad^T ; Wc^T] (128, F): the (rows, 1024) gradient is written once instead of twice and never re-read for an
    addition (three passes over a 302-MB tensor less per step).  Forward products in the f32-accurate arithmetic (the
    aggregation's split-K slices added in a fixed order), backward products in two pieces, like the separate operators."""

    @staticmethod
    def forward(ctx, f, Wc, gamma, beta, eps, n_points, link=None):
        f = f.contiguous()
        rows, F = f.shape
        ctx.link = link
        assert Wc.shape == (F, 64) and rows % n_points == 0
        if fused_linear_bn_ok(rows, F, 64):
            z, mean, var = _gemm_with_stats(f, Wc, None, F16X3_ASSIGN)
        else:
            z = gemm(f, Wc)
            mean = torch.empty(64, dtype=torch.float32, device=f.device)
            var = torch.empty(64, dtype=torch.float32, device=f.device)
            ws, n = _ws(rows, 64, f.device)
            L.check(L.lib().epc_col_moments(z.data_ptr(), rows, 64, mean.data_ptr(), var.data_ptr(), ws.data_ptr(), n, _st()))
        B = rows // n_points
        a = torch.empty_like(z)
        a_sum = torch.empty((B, 1, 64), dtype=torch.float32, device=f.device)
        nparts = L.lib().epc_cloud_colsum64_partial_floats(B)
        parts = _splitk_ws(nparts, f.device)
        # a = softmax(batch_norm(z)) and a_sum in one pass over z (epc_assign_softmax_fwd)
        L.check(L.lib().epc_assign_softmax_fwd(z.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                               float(eps), B, n_points, a.data_ptr(), a_sum.data_ptr(), parts.data_ptr(),
                                               parts.numel(), _st()))
        f3, a3 = f.view(B, n_points, F), a.view(B, n_points, 64)
        vlad = gemm(f3, a3, trans_a=True, splitk=max(1, min(8, n_points // 256)), deterministic=True)
        ctx.save_for_backward(f, Wc, z, mean, var, gamma, beta, a)
        ctx.eps, ctx.n_points = float(eps), int(n_points)
        ctx.mark_non_differentiable(mean, var)
        ctx.set_materialize_grads(False)
        return vlad, a_sum, mean, var

    @staticmethod
    def backward(ctx, dvlad, dasum, _dm, _dv):
        f, Wc, z, mean, var, gamma, beta, a = ctx.saved_tensors
        rows, F = f.shape
        N = ctx.n_points
        B = rows // N
        f3 = f.view(B, N, F)
        if dvlad is None:
            dvlad = torch.zeros((B, F, 64), dtype=torch.float32, device=f.device)
        dvlad = dvlad.contiguous()
        # da = f dvlad + (for every point of the cloud) the gradient of a_sum, the latter added inside the softmax backward
        da = gemm(f3, dvlad, fast=True)
        dz = torch.empty_like(z)
        dgamma = torch.empty(64, dtype=torch.float32, device=z.device)
        dbeta = torch.empty(64, dtype=torch.float32, device=z.device)
        ws, n = _ws(rows, 64, z.device)
        if dasum is not None:
            dasum = dasum.contiguous()
        lk = ctx.link
        through_tail = (lk is not None and lk.z is not None and FUSE_TAIL_BACKWARD and ctx.needs_input_grad[0] and F == 1024
                        and N % 32 == 0 and Wc.is_contiguous() and tuple(lk.z.shape) == (rows, F))
        trow = torch.empty(rows, dtype=torch.float32, device=z.device) if through_tail else None
        # softmax backward (+ the a_sum gradient of every point's cloud) with BatchNorm's sums, then dz in place
        L.check(L.lib().epc_assign_softmax_bwd(da.data_ptr(), dasum.data_ptr() if dasum is not None else None, a.data_ptr(),
                                               z.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr(),
                                               ctx.eps, B, N, dz.data_ptr(), dgamma.data_ptr(), dbeta.data_ptr(),
                                               trow.data_ptr() if through_tail else None, ws.data_ptr(), n, _st()))
        dWc = gemm(f, dz, trans_a=True, splitk=_splitk_for(F, 64, rows), fast=True, deterministic=True)
        df = None
        if through_tail:
            # df continued through conv5's l2-norm / ReLU / BatchNorm-sums backward in the product's epilogue (epc_vlad_df_tail)
            du = torch.empty((rows, F), dtype=torch.float32, device=f.device)
            sums = torch.empty((2, F), dtype=torch.float32, device=f.device)
            nbytes = L.lib().epc_vlad_df_packed_bytes(B, F)
            pfl = L.lib().epc_vlad_df_tail_partial_floats(B, N)
            scratch = _splitk_ws((nbytes + 3) // 4 + pfl, f.device)
            part_ptr = scratch.data_ptr() + 4 * ((nbytes + 3) // 4)
            L.check(L.lib().epc_vlad_df_tail(a.data_ptr(), dz.data_ptr(), dvlad.data_ptr(), Wc.data_ptr(), B, N,
                                             1 if _GEMM_PRECISION == "bf16" else 2, scratch.data_ptr(), nbytes, lk.z.data_ptr(),
                                             lk.rn.data_ptr(), trow.data_ptr(), lk.mean.data_ptr(), lk.var.data_ptr(),
                                             lk.gamma.data_ptr(), lk.beta.data_ptr(), lk.eps, du.data_ptr(), sums.data_ptr(),
                                             part_ptr, pfl, _st()))
            lk.du, lk.sums = du, sums
            df = du
        elif ctx.needs_input_grad[0]:
            if F % 64 == 0 and Wc.is_contiguous():
                # df = a dvlad^T + dz Wc^T in one pass, no concatenated operands (epc_vlad_df)
                df = torch.empty((rows, F), dtype=torch.float32, device=f.device)
                nbytes = L.lib().epc_vlad_df_packed_bytes(B, F)
                packed = _splitk_ws((nbytes + 3) // 4, f.device)
                L.check(L.lib().epc_vlad_df(a.data_ptr(), dz.data_ptr(), dvlad.data_ptr(), Wc.data_ptr(), B, N, F,
                                            1 if _GEMM_PRECISION == "bf16" else 2, packed.data_ptr(), packed.numel() * 4,
                                            df.data_ptr(), _st()))
            else:
                lhs = torch.cat((a, dz), dim=1).view(B, N, 128)                                    # [a | dz]
                rhs = torch.cat((dvlad.transpose(1, 2), Wc.t().unsqueeze(0).expand(B, 64, F)), dim=1)   # [dvlad^T ; Wc^T]  (B, 128, F)
                df = gemm(lhs, rhs, fast=True).view(rows, F)
        return df, dWc, dgamma, dbeta, None, None, None


# The head of the training step -- conv5, the per-point l2 norm, the soft assignment and the aggregation -- as ONE autograd node whose
# kernels are single streaming passes over the (rows, 1024) tensors and never write the feature map f (csrc/train_head16.hip for
# set_gemm_precision("bf16"): bf16-stored tensors, one bf16 value per operand; csrc/train_head32.hip for the default f32-accurate
# arithmetic: f32 tensors, split products).  Taken when the call sites hand conv5's operands over un-evaluated (LazyConv5Features:
# tf_util.conv1d_l2_normalized(..., lazy=True) -> loupe.G_VLAD.forward).  False: the per-layer operators (LinearBatchNormTrain with
# the row norm + VladAssignAggregate) -- the second implementation the tests hold this one to.
HEAD_STREAM = True


def head_stream_mode(rows, cin, cout, n_points=None):
    """"bf16" / "f32": the arithmetic of the streamed head for these shapes under the current GEMM precision; None: not applicable."""
    if not (HEAD_STREAM and cin == 256 and cout == 1024 and rows % 32 == 0 and rows * 1024 < (1 << 32)
            and (n_points is None or (n_points % 32 == 0 and rows % n_points == 0))):
        return None
    return "bf16" if _GEMM_PRECISION == "bf16" else "f32"


class LazyConv5Features:
    """conv5's operands and BatchNorm variables, handed from tf_util.conv1d_l2_normalized to loupe.G_VLAD.forward so that
    l2_normalize(relu(batch_norm(x W5 + b5))) (models/epc-net.py:136-148) and the VLAD assignment / aggregation (loupe.py:255-291)
    run as one node (Conv5VladHead).  ``on_stats(mean, var, z5, rn)`` is conv5's side of the bookkeeping (moving averages, the
    mask- and value-tap test hooks), called by whoever evaluates the node.  ``shape`` = the feature map's."""

    def __init__(self, x, W, b, gamma, beta, eps, on_stats, x_bf16=None, materialize=None):
        self.x, self.W, self.b, self.gamma, self.beta, self.eps, self.on_stats = x, W, b, gamma, beta, float(eps), on_stats
        self.x_bf16 = x_bf16                      # the chain's bf16 copy of x (the bf16 head's operand), when it made one
        self._materialize = materialize           # evaluates the layer through the per-layer operators: the consumer's way out
        self.shape = (int(x.shape[0]), int(W.shape[1]))

    def materialize(self):
        """The feature map itself, (rows, 1024), through the per-layer operators (for a consumer that cannot take the streamed head)."""
        if self._materialize is None:
            raise EpcNetError(-1, "this lazy conv5 feature map cannot be evaluated here")
        return self._materialize()

    def reshape(self, *shape):
        shape = tuple(shape[0]) if len(shape) == 1 and isinstance(shape[0], (tuple, list)) else tuple(shape)
        if shape in ((-1, self.shape[1]), self.shape):
            return self
        raise EpcNetError(-1, "conv5's lazy feature map can only be consumed as (rows, 1024) (loupe.G_VLAD.forward)")


def _scratch_bytes(nbytes, device):
    buf = _splitk_ws((int(nbytes) + 3) // 4, device)
    return buf, buf.numel() * 4


class Conv5VladHead(torch.autograd.Function):
    """(vlad (B, 1024, 64), a_sum (B, 1, 64), mean5, var5, mean_c, var_c, z5, rn) from the backbone's output cat (rows, 256):
        z5 = cat W5 + b5;  u = relu(batch_norm_train(z5));  f = u rn (tf.nn.l2_normalize over the channels);
        za = f Wc;  a = softmax(batch_norm_train(za));  vlad[b] = f[b]^T a[b];  a_sum = sum of a over the cloud's points
    (models/epc-net.py:136-148, loupe.py:255-291).  mode "bf16" (train_head16.hip): z5, du and dz5 are bf16 tensors, every product
    rounds its operands to one bf16 value, statistics and accumulators are f32.  mode "f32" (train_head32.hip): f32 tensors, conv5's
    forward in the scaled split-fp16 three-product arithmetic, every other product in three bf16 products
    -- with the feature gradient through conv5's tail (epc_vlad_df_tail), epc_bn_apply_bwd_given and the split-K dW5 of the per-layer
    operators.  The bias gradient in front of a training-mode BatchNorm is exactly zero and is not computed (LinearBatchNormTrain)."""

    @staticmethod
    def forward(ctx, cat, W5, b5, g5, bt5, eps5, Wc, gc, btc, epsc, n_points, mode, cat16=None):
        lib = L.lib()
        cat, W5, Wc = cat.contiguous(), W5.contiguous(), Wc.contiguous()
        rows = int(cat.shape[0])
        N = int(n_points)
        B = rows // N
        dev = cat.device
        h16 = mode == "bf16"
        f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        z5 = torch.empty((rows, 1024), dtype=torch.bfloat16 if h16 else torch.float32, device=dev)
        mean5, var5 = f32(1024), f32(1024)
        bn5 = lambda: (mean5.data_ptr(), var5.data_ptr(), g5.data_ptr(), bt5.data_ptr(), float(eps5))
        lhs = cat
        if h16:
            if cat16 is not None and tuple(cat16.shape) == (rows, 256) and cat16.dtype == torch.bfloat16:
                lhs = cat16                                    # the chain's bf16 copy of this very tensor
            sc, n = _scratch_bytes(lib.epc_h16_conv5_fwd_scratch_bytes(rows), dev)
            L.check(lib.epc_h16_conv5_fwd(lhs.data_ptr(), int(lhs is not cat), W5.data_ptr(), b5.data_ptr(), rows, z5.data_ptr(),
                                          mean5.data_ptr(), var5.data_ptr(), sc.data_ptr(), n, _st()))
        else:
            sc, n = _scratch_bytes(lib.epc_h32_conv5_fwd_scratch_bytes(rows), dev)
            L.check(lib.epc_h32_conv5_fwd(cat.data_ptr(), W5.data_ptr(), b5.data_ptr(), rows, z5.data_ptr(), mean5.data_ptr(),
                                          var5.data_ptr(), sc.data_ptr(), n, _st()))
        za, rn, mean_c, var_c = f32(rows, 64), f32(rows), f32(64), f32(64)
        assign = lib.epc_h16_assign if h16 else lib.epc_h32_assign
        sc, n = _scratch_bytes((lib.epc_h16_assign_scratch_bytes if h16 else lib.epc_h32_assign_scratch_bytes)(B, N, 0), dev)
        L.check(assign(z5.data_ptr(), *bn5(), Wc.data_ptr(), 0, B, N, za.data_ptr(), rn.data_ptr(), mean_c.data_ptr(), var_c.data_ptr(),
                       sc.data_ptr(), n, _st()))
        a, a_sum = f32(rows, 64), f32(B, 1, 64)
        parts = _splitk_ws(lib.epc_cloud_colsum64_partial_floats(B), dev)
        L.check(lib.epc_assign_softmax_fwd(za.data_ptr(), mean_c.data_ptr(), var_c.data_ptr(), gc.data_ptr(), btc.data_ptr(),
                                           float(epsc), B, N, a.data_ptr(), a_sum.data_ptr(), parts.data_ptr(), parts.numel(), _st()))
        vlad = f32(B, 1024, 64)
        sc, n = _scratch_bytes((lib.epc_h16_colgemm_scratch_bytes if h16 else lib.epc_h32_colgemm_scratch_bytes)(B, N), dev)
        L.check((lib.epc_h16_colgemm if h16 else lib.epc_h32_colgemm)(z5.data_ptr(), *bn5(), a.data_ptr(), rn.data_ptr(), B, N, 1,
                                                                     vlad.data_ptr(), sc.data_ptr(), n, _st()))
        ctx.save_for_backward(lhs, W5, z5, mean5, var5, g5, bt5, rn, Wc, za, mean_c, var_c, gc, btc, a)
        ctx.eps5, ctx.epsc, ctx.n_points, ctx.h16 = float(eps5), float(epsc), N, h16
        ctx.mark_non_differentiable(mean5, var5, mean_c, var_c, z5, rn)
        ctx.set_materialize_grads(False)
        return vlad, a_sum, mean5, var5, mean_c, var_c, z5, rn      # (z5, rn: for the call site's test hooks)

    @staticmethod
    def backward(ctx, dvlad, dasum, *_unused):
        lib = L.lib()
        cat, W5, z5, mean5, var5, g5, bt5, rn, Wc, za, mean_c, var_c, gc, btc, a = ctx.saved_tensors
        rows = int(cat.shape[0])
        N, h16 = ctx.n_points, ctx.h16
        B = rows // N
        dev = cat.device
        f32 = lambda *shape: torch.empty(shape, dtype=torch.float32, device=dev)
        if dvlad is None:
            dvlad = torch.zeros((B, 1024, 64), dtype=torch.float32, device=dev)
        dvlad = dvlad.contiguous()
        bn5 = (mean5.data_ptr(), var5.data_ptr(), g5.data_ptr(), bt5.data_ptr(), ctx.eps5)
        # da = f dvlad[cloud] (the a_sum gradient of every point's cloud is added inside the softmax backward)
        da = f32(rows, 64)
        sc, n = _scratch_bytes((lib.epc_h16_assign_scratch_bytes if h16 else lib.epc_h32_assign_scratch_bytes)(B, N, 1), dev)
        L.check((lib.epc_h16_assign if h16 else lib.epc_h32_assign)(z5.data_ptr(), *bn5, dvlad.data_ptr(), 1, B, N, da.data_ptr(), None,
                                                                   None, None, sc.data_ptr(), n, _st()))
        dz, dgc, dbtc, trow = f32(rows, 64), f32(64), f32(64), f32(rows)
        ws, wn = _ws(rows, 64, dev)
        L.check(lib.epc_assign_softmax_bwd(da.data_ptr(), dasum.contiguous().data_ptr() if dasum is not None else None, a.data_ptr(),
                                           za.data_ptr(), mean_c.data_ptr(), var_c.data_ptr(), gc.data_ptr(), btc.data_ptr(), ctx.epsc,
                                           B, N, dz.data_ptr(), dgc.data_ptr(), dbtc.data_ptr(), trow.data_ptr(), ws.data_ptr(), wn, _st()))
        dWc = f32(1024, 64)
        sc, n = _scratch_bytes((lib.epc_h16_colgemm_scratch_bytes if h16 else lib.epc_h32_colgemm_scratch_bytes)(B, N), dev)
        L.check((lib.epc_h16_colgemm if h16 else lib.epc_h32_colgemm)(z5.data_ptr(), *bn5, dz.data_ptr(), rn.data_ptr(), B, N, 0,
                                                                     dWc.data_ptr(), sc.data_ptr(), n, _st()))
        # du = [f > 0] rn ([a | dz] [dvlad^T ; Wc^T] - f trow), the BatchNorm's column sums, then dz5 in place
        need_dx = bool(ctx.needs_input_grad[0])
        du = torch.empty_like(z5)
        sums = f32(2, 1024)
        if h16:
            sc, n = _scratch_bytes(lib.epc_h16_df_tail_scratch_bytes(B, N), dev)
            L.check(lib.epc_h16_df_tail(a.data_ptr(), dz.data_ptr(), dvlad.data_ptr(), Wc.data_ptr(), B, N, z5.data_ptr(), rn.data_ptr(),
                                        trow.data_ptr(), *bn5, du.data_ptr(), sums.data_ptr(), sc.data_ptr(), n, _st()))
            if not need_dx:
                L.check(lib.epc_h16_bn_bwd_apply(du.data_ptr(), z5.data_ptr(), *bn5, sums[0].data_ptr(), sums[1].data_ptr(), rows,
                                                 du.data_ptr(), _st()))
        else:
            nbytes = lib.epc_vlad_df_packed_bytes(B, 1024)
            pfl = lib.epc_vlad_df_tail_partial_floats(B, N)
            scratch = _splitk_ws((nbytes + 3) // 4 + pfl, dev)
            L.check(lib.epc_vlad_df_tail(a.data_ptr(), dz.data_ptr(), dvlad.data_ptr(), Wc.data_ptr(), B, N, 2, scratch.data_ptr(), nbytes,
                                         z5.data_ptr(), rn.data_ptr(), trow.data_ptr(), *bn5, du.data_ptr(), sums.data_ptr(),
                                         scratch.data_ptr() + 4 * ((nbytes + 3) // 4), pfl, _st()))
            if not need_dx:
                L.check(lib.epc_bn_apply_bwd_given(du.data_ptr(), z5.data_ptr(), mean5.data_ptr(), var5.data_ptr(), g5.data_ptr(),
                                                   bt5.data_ptr(), sums[0].data_ptr(), sums[1].data_ptr(), ctx.eps5, rows, 1024, du.data_ptr(),
                                                   _st()))
        dcat = None
        if need_dx:
            # dz5 = gamma rstd (du - dbeta / R - zhat dgamma / R) is formed INSIDE dcat's product as du and z5 stream, and written over du for
            # dW5's: one pass over the (rows, 1024) tensors instead of the apply pass + the product's own read
            dcat = f32(rows, 256)
            sc, n = _scratch_bytes((lib.epc_h16_dx_scratch_bytes if h16 else lib.epc_h32_dx_scratch_bytes)(), dev)
            L.check((lib.epc_h16_conv5_dx_bn if h16 else lib.epc_h32_conv5_dx_bn)(
                du.data_ptr(), z5.data_ptr(), mean5.data_ptr(), var5.data_ptr(), g5.data_ptr(), ctx.eps5, sums[0].data_ptr(), sums[1].data_ptr(),
                W5.data_ptr(), rows, du.data_ptr(), dcat.data_ptr(), sc.data_ptr(), n, _st()))
        # dW5 = cat^T dz5, row slices added in a fixed order
        if h16:
            dW5 = f32(256, 1024)
            sc, n = _scratch_bytes(lib.epc_h16_conv5_dw_scratch_bytes(rows), dev)
            L.check(lib.epc_h16_conv5_dw(cat.data_ptr(), int(cat.dtype == torch.bfloat16), du.data_ptr(), rows, dW5.data_ptr(),
                                         sc.data_ptr(), n, _st()))
        else:
            dW5 = f32(256, 1024)
            sc, n = _scratch_bytes(lib.epc_h32_conv5_dw_scratch_bytes(rows), dev)
            L.check(lib.epc_h32_conv5_dw(cat.data_ptr(), du.data_ptr(), rows, dW5.data_ptr(), sc.data_ptr(), n, _st()))
        return dcat, dW5, None, sums[1], sums[0], None, dWc, dgc, dbtc, None, None, None, None


def expand16(z16, bn=None, rn=None):
    """(rows, 1024) f32 from a bf16 tensor of train_head16.hip: its values, or -- bn = (mean, var, gamma, beta, eps) -- the feature
    map relu(batch_norm(z5)) rn the fused kernels never write (epc_h16_expand).  Test taps / materialised features."""
    rows = int(z16.shape[0])
    y = torch.empty((rows, 1024), dtype=torch.float32, device=z16.device)
    if bn is None:
        L.check(L.lib().epc_h16_expand(z16.data_ptr(), None, None, None, None, 0.0, None, rows, y.data_ptr(), _st()))
    else:
        mean, var, gamma, beta, eps = bn
        L.check(L.lib().epc_h16_expand(z16.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr(), float(eps),
                                       rn.data_ptr() if rn is not None else None, rows, y.data_ptr(), _st()))
    return y


class GroupSum(torch.autograd.Function):
    """tf.reduce_sum over the G group rows behind G_VLAD's shared hidden projection (loupe.py:326-328): (B G, O) -> (B, O)."""

    @staticmethod
    def forward(ctx, x, G):
        x = x.contiguous()
        rows, O = (int(v) for v in x.shape)
        ctx.G, ctx.shape = int(G), (rows, O)
        y = torch.empty((rows // int(G), O), dtype=torch.float32, device=x.device)
        L.check(L.lib().epc_group_sum_fwd(x.data_ptr(), rows // int(G), int(G), O, y.data_ptr(), _st()))
        return y

    @staticmethod
    def backward(ctx, dy):
        rows, O = ctx.shape
        dx = torch.empty((rows, O), dtype=torch.float32, device=dy.device)
        L.check(L.lib().epc_group_sum_bwd(dy.contiguous().data_ptr(), rows // ctx.G, ctx.G, O, dx.data_ptr(), _st()))
        return dx, None


# The grouped hidden projection (loupe.py:302-322) and its two gradients on the skinny kernels of csrc/train_hidden.hip -- one pass over
# the 16-MB weight matrix per product -- where the shape is covered (64 .. 128 rows: 16 .. 32 clouds x 4 groups); False, or any other
# shape: ops.Linear's tile GEMMs (the second implementation tests/test_gpu_hidden_tail.py holds it to).
HIDDEN_PROJ = True


def hidden_proj_ok(rows, cin, cout):
    return bool(HIDDEN_PROJ and L.lib().epc_hidden_proj_ok(int(rows), int(cin), int(cout)))


class HiddenProjection(torch.autograd.Function):
    """y = x @ W for the (B G, C F / G) x (C F / G, O) product of loupe.py:322 (no bias: a BatchNorm follows).  Arithmetic as
    ops.Linear's: under "bf16x6" three bf16 pieces per operand forward (six products, f32-accurate), two backward; under "bf16" one
    (every side of the three products is at least 64: oracle/epcnet_oracle_torch.py, bf16_product_rule)."""

    @staticmethod
    def forward(ctx, x, W):
        x, W = x.contiguous(), W.contiguous()
        M, K = int(x.shape[0]), int(x.shape[1])
        lib = L.lib()
        bf16 = _GEMM_PRECISION == "bf16"
        y = torch.empty((M, int(W.shape[1])), dtype=torch.float32, device=x.device)
        sc, n = _scratch_bytes(lib.epc_hidden_proj_scratch_bytes(M, K), x.device)
        L.check(lib.epc_hidden_proj_fwd(x.data_ptr(), W.data_ptr(), M, K, 1 if bf16 else 3, y.data_ptr(), sc.data_ptr(), n, _st()))
        ctx.save_for_backward(x, W)
        ctx.pieces = 1 if bf16 else 2
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W = ctx.saved_tensors
        dy = dy.contiguous()
        M, K = int(x.shape[0]), int(x.shape[1])
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dW = torch.empty_like(W) if ctx.needs_input_grad[1] else None
        L.check(L.lib().epc_hidden_proj_bwd(x.data_ptr(), W.data_ptr(), dy.data_ptr(), M, K, ctx.pieces,
                                            dx.data_ptr() if dx is not None else None, dW.data_ptr() if dW is not None else None, _st()))
        return dx, dW


# The VLAD tail behind the hidden projection -- its BatchNorm, the group sum, context gating (product, BatchNorm, sigmoid gate) -- as ONE
# launch each way (csrc/train_head.hip: epc_hidden_tail_fwd / _bwd) instead of nine and ten of 2-8 us; False: the per-op path (the
# second implementation tests/test_gpu_hidden_tail.py holds it to).
HIDDEN_TAIL = True


def hidden_tail_ok(rows, G, O):
    return bool(HIDDEN_TAIL and rows % int(G) == 0 and L.lib().epc_hidden_tail_ok(int(rows) // int(G), int(G), int(O)))


class HiddenTail(torch.autograd.Function):
    """loupe.py:323-331 + :61-101 on h (B G, O), the hidden projection's output, in training mode:
        y = slim.batch_norm(h);  v = reduce_sum over the G group rows;  out = v * sigmoid(slim.batch_norm(v @ gating_weights))
    Returns (out (B, O), mean1, var1u, mean2, var2u): the batch means and the Bessel-corrected batch variances -- what the fused slim
    op feeds its moving averages (the population variances normalise).  The gating product is f32-accurate in both arithmetics of the
    step (at most 32 rows: oracle/epcnet_oracle_torch.py, bf16_product_rule)."""

    @staticmethod
    def forward(ctx, h, gamma1, beta1, G, Wg, gamma2, beta2, eps):
        lib = L.lib()
        h, Wg = h.contiguous(), Wg.contiguous()
        R, O = (int(v) for v in h.shape)
        G = int(G)
        B = R // G
        dev = h.device
        vec = lambda: torch.empty(O, dtype=torch.float32, device=dev)
        mat = lambda: torch.empty((B, O), dtype=torch.float32, device=dev)
        mean1, var1, var1u, mean2, var2, var2u, v, gl, out = vec(), vec(), vec(), vec(), vec(), vec(), mat(), mat(), mat()
        L.check(lib.epc_hidden_tail_fwd(h.data_ptr(), B, G, O, gamma1.data_ptr(), beta1.data_ptr(), Wg.data_ptr(), gamma2.data_ptr(),
                                        beta2.data_ptr(), float(eps), R / max(R - 1, 1), B / max(B - 1, 1), mean1.data_ptr(), var1.data_ptr(),
                                        var1u.data_ptr(), v.data_ptr(), gl.data_ptr(), mean2.data_ptr(), var2.data_ptr(), var2u.data_ptr(),
                                        out.data_ptr(), _st()))
        ctx.save_for_backward(h, gamma1, mean1, var1, v, gl, Wg, gamma2, beta2, mean2, var2)
        ctx.G, ctx.eps = G, float(eps)
        ctx.mark_non_differentiable(mean1, var1u, mean2, var2u)
        ctx.set_materialize_grads(False)
        return out, mean1, var1u, mean2, var2u

    @staticmethod
    def backward(ctx, dout, *_unused):
        if dout is None:
            return (None,) * 8
        lib = L.lib()
        h, gamma1, mean1, var1, v, gl, Wg, gamma2, beta2, mean2, var2 = ctx.saved_tensors
        R, O = (int(x) for x in h.shape)
        B = R // ctx.G
        dev = h.device
        dout = dout.contiguous()
        vec = lambda: torch.empty(O, dtype=torch.float32, device=dev)
        dh, dWg = torch.empty_like(h), torch.empty_like(Wg)
        dg1, db1, dg2, db2 = vec(), vec(), vec(), vec()
        L.check(lib.epc_hidden_tail_bwd(dout.data_ptr(), h.data_ptr(), B, ctx.G, O, gamma1.data_ptr(), mean1.data_ptr(), var1.data_ptr(),
                                        v.data_ptr(), gl.data_ptr(), Wg.data_ptr(), gamma2.data_ptr(), beta2.data_ptr(), mean2.data_ptr(),
                                        var2.data_ptr(), ctx.eps, dh.data_ptr(), dg1.data_ptr(), db1.data_ptr(), dWg.data_ptr(),
                                        dg2.data_ptr(), db2.data_ptr(), _st()))
        return dh, dg1, db1, None, dWg, dg2, db2, None


class MaxPoolPoints(torch.autograd.Function):
    """EPC-Net-L's global max over a cloud's points (models/epc-net-l.py:88-92: tf_util.max_pool2d with the kernel covering all N
    points): (B, N, C) -> (B, C); the gradient goes to the row that held the maximum (the first on ties, tf.nn.max_pool's)."""

    @staticmethod
    def forward(ctx, x):
        x = x.contiguous()
        B, N, C = (int(v) for v in x.shape)
        out = torch.empty((B, C), dtype=torch.float32, device=x.device)
        arg = torch.empty((B, C), dtype=torch.int32, device=x.device)
        L.check(L.lib().epc_maxpool_points_fwd(x.data_ptr(), B, N, C, out.data_ptr(), arg.data_ptr(), _st()))
        ctx.save_for_backward(arg)
        ctx.shape = (B, N, C)
        return out

    @staticmethod
    def backward(ctx, dy):
        (arg,) = ctx.saved_tensors
        B, N, C = ctx.shape
        dx = torch.empty((B, N, C), dtype=torch.float32, device=dy.device)
        L.check(L.lib().epc_maxpool_points_bwd(dy.contiguous().data_ptr(), arg.data_ptr(), B, N, C, dx.data_ptr(), _st()))
        return dx


class GateMul(torch.autograd.Function):
    """Context gating's product, y * sigmoid(g) (loupe.py:99-100), forward and backward one kernel each."""

    @staticmethod
    def forward(ctx, y, g):
        y, g = y.contiguous(), g.contiguous()
        assert y.shape == g.shape
        out = torch.empty_like(y)
        L.check(L.lib().epc_gate_fwd(y.data_ptr(), g.data_ptr(), y.numel(), out.data_ptr(), _st()))
        ctx.save_for_backward(y, g)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, g = ctx.saved_tensors
        dout = dout.contiguous()
        dy, dg = torch.empty_like(y), torch.empty_like(g)
        L.check(L.lib().epc_gate_bwd(dout.data_ptr(), y.data_ptr(), g.data_ptr(), y.numel(), dy.data_ptr(), dg.data_ptr(),
                                     _st()))
        return dy, dg


class SquaredError(torch.autograd.Function):
    """square_error_sum / square_error_mean (kd_train.py:330-340) of the student's tensor ``a`` against the teacher's ``b``:
    sum (or mean) of (a - b)^2 in one read of both (epc_sq_err_fwd); only ``a`` gets a gradient (the reference feeds the
    teacher's outputs through placeholders, kd_train.py:786-790)."""

    @staticmethod
    def forward(ctx, a, b, mean):
        a, b = a.contiguous(), b.detach().contiguous()
        assert a.shape == b.shape and a.dtype == torch.float32 and b.dtype == torch.float32
        n = a.numel()
        loss = torch.empty((), dtype=torch.float32, device=a.device)
        pf = L.lib().epc_sq_err_partial_floats(n)
        part = _splitk_ws(pf, a.device)
        L.check(L.lib().epc_sq_err_fwd(a.data_ptr(), b.data_ptr(), n, int(bool(mean)), loss.data_ptr(), part.data_ptr(),
                                       part.numel(), _st()))
        ctx.save_for_backward(a, b)
        ctx.mean = int(bool(mean))
        return loss

    @staticmethod
    def backward(ctx, dloss):
        a, b = ctx.saved_tensors
        da = torch.empty_like(a)
        dloss = dloss.contiguous().float()
        L.check(L.lib().epc_sq_err_bwd(a.data_ptr(), b.data_ptr(), a.numel(), ctx.mean, dloss.data_ptr(), da.data_ptr(), _st()))
        return da, None, None


def morton_sort(xyz):
    """Z-order every cloud of (B, N, 3) (epc_morton_sort): a pure re-ordering, legal because the whole network is
    permutation-equivariant and the pooling invariant; it makes the kNN culling and the gathers cache-local."""
    L.require_gpu()
    xyz = xyz.contiguous().float()
    if xyz.shape[1] > 16384:
        return xyz
    out = torch.empty_like(xyz)
    L.check(L.lib().epc_morton_sort(xyz.data_ptr(), int(xyz.shape[0]), int(xyz.shape[1]), out.data_ptr(), None, _st()))
    return out


def adam_step(w, m, v, g, lr, t, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer update of one tensor, in place (train.py:273).  ``lr`` may be a one-element device tensor
    holding the bias-corrected rate lr_t (``t`` is then ignored): the form a captured HIP graph of the step uses."""
    if torch.is_tensor(lr):
        L.check(L.lib().epc_adam_step_dev(w.data_ptr(), m.data_ptr(), v.data_ptr(), g.contiguous().data_ptr(), w.numel(),
                                          lr.data_ptr(), float(beta1), float(beta2), float(eps), _st()))
        return
    L.check(L.lib().epc_adam_step(w.data_ptr(), m.data_ptr(), v.data_ptr(), g.contiguous().data_ptr(), w.numel(), float(lr),
                                  float(beta1), float(beta2), float(eps), int(t), _st()))


def _ptr_array(tensors):
    import ctypes
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


def adam_multi(ws, ms, vs, gs, lr, t, beta1=0.9, beta2=0.999, eps=1e-8):
    """adam_step over lists of tensors in one launch per 64 tensors (epc_adam_multi)."""
    import ctypes
    gs = [g.contiguous() for g in gs]
    n = (ctypes.c_long * len(ws))(*[w.numel() for w in ws])
    dev_lr = lr.data_ptr() if torch.is_tensor(lr) else None
    L.check(L.lib().epc_adam_multi(len(ws), _ptr_array(ws), _ptr_array(ms), _ptr_array(vs), _ptr_array(gs), n,
                                   0.0 if torch.is_tensor(lr) else float(lr), float(beta1), float(beta2), float(eps),
                                   int(t), dev_lr, _st()))


def ema_multi(shadows, values, scheduled, fixed_decay, sched_decay):
    """All moving-average updates of a step in one launch (epc_ema_multi).  ``sched_decay``: float or 0-d device tensor."""
    import ctypes
    values = [v.detach().contiguous() for v in values]
    n = (ctypes.c_long * len(shadows))(*[s_.numel() for s_ in shadows])
    flags = (ctypes.c_int * len(shadows))(*[1 if f else 0 for f in scheduled])
    dev = sched_decay.data_ptr() if torch.is_tensor(sched_decay) else None
    L.check(L.lib().epc_ema_multi(len(shadows), _ptr_array(shadows), _ptr_array(values), n, flags, float(fixed_decay),
                                  0.0 if torch.is_tensor(sched_decay) else float(sched_decay), dev, _st()))
