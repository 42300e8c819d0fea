"""Gradient ORACLE for the EPC-Net training step -- TEST INFRASTRUCTURE, NOT PRODUCT CODE (see epcnet_oracle.py).

A torch-CPU (float64 by default) restatement of the same graph as ``epcnet_oracle.forward`` so that autograd yields
the gradients the reference's ``optimizer.minimize(loss)`` applies (train.py:264-277), plus TensorFlow's Adam update
rule and the BN moving-average updates that run with the step (UPDATE_OPS, train.py:275-277).  Its forward is pinned
against the numpy oracle in tests/test_oracle_cpu.py (training and inference mode); like the numpy oracle it is
PARITY UNPINNED against a running TensorFlow.

The kNN mask is non-differentiable (Cast(GreaterEqual), utils/tf_util.py:664-665) and is taken from the numpy oracle.
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional, Tuple

import numpy as np
import torch

import epcnet_oracle as O


def _bn_train(x, gamma, beta, axes, eps=O.BN_EPS):
    mean = x.mean(dim=axes)
    var = ((x - mean) ** 2).mean(dim=axes)          # tf.nn.moments: population variance
    inv = torch.rsqrt(var + eps) * gamma
    return x * inv + (beta - mean * inv), mean, var


def _bn_infer(x, gamma, beta, mean, var, eps=O.BN_EPS):
    inv = torch.rsqrt(var + eps) * gamma
    return x * inv + (beta - mean * inv)


def _l2n(x, dim):
    ss = (x * x).sum(dim=dim, keepdim=True)
    return x * torch.rsqrt(torch.clamp(ss, min=O.L2_EPS))


def _round_bf16(t):
    """Round to the nearest bf16 value (ties to even), keep the dtype."""
    return t.to(torch.float32).to(torch.bfloat16).to(t.dtype)


def bf16_product_rule(M: int, N: int, K: int) -> bool:
    """Which products of the "bf16" training arithmetic (BASELINE.json configs[2]; params["TRAIN_PRECISION"] = "bf16") round
    their operands to bf16: the ones the matrix-pipe GEMM kernel takes -- every side at least 64 (K at least 32).  Products with
    at most 32 rows (the 18-row gating / fc layers) and K = 3 (conv1) are computed in float32 by their own small kernels."""
    return M >= 64 and N >= 64 and K >= 32


class _RoundedMatmul(torch.autograd.Function):
    """C = A @ B (2-D, or 3-D with a leading batch dimension) with the operands of every product rounded to bf16 where
    ``bf16_product_rule`` says so, forward AND backward (dA = dC B^T, dB = A^T dC each round their own two operands): the
    arithmetic of a training step whose GEMMs take one bf16 value per operand and accumulate in float32 -- restated in the
    oracle's dtype, so that what is left between it and the HIP step is accumulation order."""

    @staticmethod
    def forward(ctx, A, B):
        ctx.save_for_backward(A, B)
        M, K, N = A.shape[-2], A.shape[-1], B.shape[-1]
        if bf16_product_rule(M, N, K):
            return torch.matmul(_round_bf16(A), _round_bf16(B))
        return torch.matmul(A, B)

    @staticmethod
    def backward(ctx, dC):
        A, B = ctx.saved_tensors
        M, K, N = A.shape[-2], A.shape[-1], B.shape[-1]
        r = lambda t, on: _round_bf16(t) if on else t
        on_a = bf16_product_rule(M, K, N)          # dA (M, K) = dC (M, N) @ B^T (N, K)
        dA = torch.matmul(r(dC, on_a), r(B, on_a).transpose(-1, -2))
        on_b = bf16_product_rule(K, N, M)          # dB (K, N) = A^T (K, M) @ dC (M, N)
        dB = torch.matmul(r(A, on_b).transpose(-1, -2), r(dC, on_b))
        if B.dim() == 2 and dB.dim() == 3:
            dB = dB.sum(0)
        return dA, dB


class _Head16(torch.autograd.Function):
    """conv5 + per-point l2 norm + VLAD soft assignment + aggregation (models/epc-net.py:136-148, loupe.py:255-291, training mode) with
    the ROUNDING POINTS of the bf16-stored head of the HIP step (epc-net_amd/csrc/train_head16.hip), forward and backward written out:

      forward   zacc = r(cat) r(W5) [f32 accumulators: exact here];  moments of conv5 from zacc (+ b5);  z5 = r(zacc + b5) [stored];
                u = relu(z5 s5 + t5);  rn = rsqrt(max(sum_c u^2, 1e-12));  za = rn (r(u) r(Wc));  a = softmax(bn(za));
                vlad[b] = r(u)[b]^T r(rn a)[b];  a_sum[b] = sum_n a[b]
      backward  da = rn (r(u) r(dvlad[b]));  softmax / BatchNorm backward in exact arithmetic -> dz;  t = sum_k a da + sum_k dz za;
                dWc = r(u)^T r(rn dz);  df = r([a | dz]) r([dvlad[b]^T ; Wc^T]);  f = u rn;  du = [f > 0] (rn df - (rn t) f);
                dbeta5 = sum du, dgamma5 = sum du zhat5 [from the unrounded du];  dz5 = r(gamma5 rstd5 (r(du) - dbeta5 / R - zhat5 dgamma5 / R));
                dcat = dz5 r(W5)^T;  dW5 = r(cat)^T dz5;  db5 = 0 (a bias in front of a training-mode BatchNorm)

    r = round to bf16 (``rounding``) or the identity -- with the identity this is the exact function, and tests/test_oracle_cpu.py holds
    the hand-written backward to autograd's of the plain composition.  ``mask``: conv5's ReLU mask pinned (TorchOracle.relu_masks);
    ``z5_pin``: the stored z5 of the implementation under test, continued from instead of this function's own (TorchOracle.value_pins).
    Returns (vlad (B, 1024, 64), a_sum (B, 1, 64), mean5, var5, mean_c, var_c, z5, u); only the first two are differentiable."""

    @staticmethod
    def forward(ctx, cat, W5, b5, g5, bt5, Wc, gc, btc, n_points, rounding, mask, z5_pin, eps):
        r = _round_bf16 if rounding else (lambda t: t)
        R = cat.shape[0]
        B = R // n_points
        cat_r, W5_r, Wc_r = r(cat), r(W5), r(Wc)
        zacc = cat_r @ W5_r
        mean5 = zacc.mean(0) + b5
        var5 = ((zacc - zacc.mean(0)) ** 2).mean(0)
        z5 = r(zacc + b5)
        if z5_pin is not None:
            z5 = z5_pin
        rs5 = torch.rsqrt(var5 + eps)
        s5 = g5 * rs5
        pre = z5 * s5 + (bt5 - mean5 * s5)
        m5 = (pre > 0) if mask is None else mask
        u = pre * m5.to(pre.dtype)
        rn = torch.rsqrt(torch.clamp((u * u).sum(1), min=O.L2_EPS))
        u_r = r(u)
        za = rn[:, None] * (u_r @ Wc_r)
        mean_c = za.mean(0)
        var_c = ((za - mean_c) ** 2).mean(0)
        rs_c = torch.rsqrt(var_c + eps)
        a = torch.softmax((za - mean_c) * (rs_c * gc) + btc, dim=1)
        a3 = a.reshape(B, n_points, 64)
        vlad = torch.matmul(u_r.reshape(B, n_points, -1).transpose(1, 2), r(rn[:, None] * a).reshape(B, n_points, 64))
        a_sum = a3.sum(1, keepdim=True)
        ctx.save_for_backward(cat_r, W5_r, Wc_r, z5, mean5, rs5, g5, m5, u, u_r, rn, za, mean_c, rs_c, gc, a)
        ctx.n_points, ctx.r = n_points, r
        ctx.mark_non_differentiable(mean5, var5, mean_c, var_c, z5, u)
        return vlad, a_sum, mean5, var5, mean_c, var_c, z5, u

    @staticmethod
    def backward(ctx, dvlad, dasum, *_unused):
        cat_r, W5_r, Wc_r, z5, mean5, rs5, g5, m5, u, u_r, rn, za, mean_c, rs_c, gc, a = ctx.saved_tensors
        r, N = ctx.r, ctx.n_points
        R = cat_r.shape[0]
        B = R // N
        if dvlad is None:
            dvlad = torch.zeros((B, 1024, 64), dtype=cat_r.dtype)
        dv_r = r(dvlad)
        da = rn[:, None] * torch.matmul(u_r.reshape(B, N, -1), dv_r).reshape(R, 64)
        dy = da if dasum is None else da + dasum.reshape(B, 1, 64).expand(B, N, 64).reshape(R, 64)
        dpre = a * (dy - (dy * a).sum(1, keepdim=True))
        zhat_c = (za - mean_c) * rs_c
        dbtc = dpre.sum(0)
        dgc = (dpre * zhat_c).sum(0)
        dz = gc * rs_c * (dpre - dbtc / R - zhat_c * dgc / R)
        trow = (a * da).sum(1) + (dz * za).sum(1)
        dWc = u_r.t() @ r(rn[:, None] * dz)
        lhs = r(torch.cat((a, dz), dim=1)).reshape(B, N, 128)
        rhs = torch.cat((dv_r.transpose(1, 2), Wc_r.t().unsqueeze(0).expand(B, 64, Wc_r.shape[0])), dim=1)
        df = torch.matmul(lhs, rhs).reshape(R, -1)
        f = u * rn[:, None]
        rt = torch.where(rn >= 0.99e6, torch.zeros_like(rn), rn * trow)          # (the clamped zero row: no projection term)
        du = (rn[:, None] * df - rt[:, None] * f) * m5.to(df.dtype)
        zhat5 = (z5 - mean5) * rs5
        dbt5 = du.sum(0)
        dg5 = (du * zhat5).sum(0)
        dz5 = r(g5 * rs5 * (r(du) - dbt5 / R - zhat5 * dg5 / R))
        dcat = dz5 @ W5_r.t()
        dW5 = cat_r.t() @ dz5
        return dcat, dW5, torch.zeros_like(mean5), dg5, dbt5, dWc, dgc, dbtc, None, None, None, None, None


class TorchOracle:
    def __init__(self, weights: Dict[str, np.ndarray], arch="epc-net", params=None, dtype=torch.float64,
                 outer="query_triplets"):
        self.arch, self.dtype, self.outer = arch, dtype, outer
        self.p = dict(O.DEFAULT_PARAMS)
        self.p.update(params or {})
        table = O.variable_table(arch, self.p, outer)
        self.w: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        for k, (shape, kind) in table.items():
            t = torch.tensor(np.asarray(weights[k], dtype=np.float64), dtype=dtype)
            t.requires_grad_(kind in ("weight", "bias", "gamma", "beta"))
            self.w[k] = t
        self.trainable = [k for k, (_, kind) in table.items() if kind in ("weight", "bias", "gamma", "beta")]
        self.new_stats: Dict[str, torch.Tensor] = {}
        # Mask pinning (tests/test_gpu_train_step.py): {scope: bool array of the layer's output shape}.  A ReLU'd layer named
        # here computes y * mask instead of relu(y) -- the same piecewise-linear function as the implementation whose forward
        # produced the masks, so that the two gradients differ by arithmetic only, not by which side of zero a pre-activation
        # within float32 rounding of zero fell on.  ``relu_mask_disagreement`` counts the elements where relu(y) would have
        # chosen otherwise.
        self.relu_masks: Optional[Dict[str, np.ndarray]] = None
        self.relu_mask_disagreement: Dict[str, int] = {}
        # "bf16": every dense product rounds its operands to bf16 where the HIP step's bf16 arithmetic does (_RoundedMatmul), and
        # EPC-Net's head -- conv5 to the VLAD aggregation -- has the rounding points of the bf16-STORED head (_Head16)
        self.gemm_rounding: Optional[str] = None
        # Value pinning (tests/test_gpu_train_step.py, the bf16 step): {scope: array of the layer's pre-activation as the HIP step
        # stored it}.  A layer named here continues from that value (its own result's gradient path kept: z + (pin - z).detach()) -- the
        # two implementations then round the SAME numbers at every later rounding point, and what is left between their gradients is
        # the arithmetic of each layer, not the compounded drift of twelve normalised layers.  ``value_pin_gap`` records, per layer,
        # max |pin - z| / max |z|: a forward bug shows up there.
        self.value_pins: Optional[Dict[str, np.ndarray]] = None
        self.value_pin_gap: Dict[str, float] = {}

    def _mm(self, A, B):
        """A (rows, K) @ B (K, N) (or batched 3-D @ 3-D) in the step's GEMM arithmetic."""
        if self.gemm_rounding == "bf16":
            return _RoundedMatmul.apply(A, B)
        return torch.matmul(A, B)

    def _relu(self, y, scope):
        if self.relu_masks is None or scope not in self.relu_masks:
            return torch.relu(y)
        m = torch.as_tensor(np.asarray(self.relu_masks[scope]).reshape(tuple(y.shape)), dtype=torch.bool)
        self.relu_mask_disagreement[scope] = int(((y.detach() > 0) != m).sum())
        return y * m.to(y.dtype)

    # ---- layers ----------------------------------------------------------------------------------------------
    def _tfutil_bn(self, z, scope, axes, training, bn_decay):
        g, b = self.w[scope + "/bn/gamma"], self.w[scope + "/bn/beta"]
        mname, vname = O.ema_names(scope, self.outer)
        if training:
            y, mean, var = _bn_train(z, g, b, axes)
            decay = 0.9 if bn_decay is None else bn_decay
            with torch.no_grad():
                self.new_stats[mname] = self.w[mname] - (1 - decay) * (self.w[mname] - mean)
                self.new_stats[vname] = self.w[vname] - (1 - decay) * (self.w[vname] - var)
            return y
        return _bn_infer(z, g, b, self.w[mname], self.w[vname])

    def conv1d(self, x, scope, training, bn_decay):
        W = self.w[scope + "/weights"]
        z = self._mm(x.reshape(-1, x.shape[-1]), W.reshape(W.shape[-2], W.shape[-1])).reshape(
            tuple(x.shape[:-1]) + (W.shape[-1],)) + self.w[scope + "/biases"]
        z = self._pin(z, scope)
        return self._relu(self._tfutil_bn(z, scope, (0, 1), training, bn_decay), scope)

    def _pin(self, z, scope):
        if self.value_pins is None or scope not in self.value_pins:
            return z
        pin = torch.as_tensor(np.asarray(self.value_pins[scope], dtype=np.float64).reshape(tuple(z.shape)), dtype=z.dtype)
        self.value_pin_gap[scope] = float((pin - z.detach()).abs().max() / z.detach().abs().max().clamp(min=1e-30))
        return z + (pin - z).detach()

    def _slim_bn(self, x, scope, training, fused):
        g, b = self.w[scope + "/gamma"], self.w[scope + "/beta"]
        mm, mv = self.w[scope + "/moving_mean"], self.w[scope + "/moving_variance"]
        if training:
            y, mean, var = _bn_train(x, g, b, (0,))
            rows = x.shape[0]
            var_upd = var * (rows / max(rows - 1, 1)) if fused else var
            with torch.no_grad():
                self.new_stats[scope + "/moving_mean"] = mm - (mm - mean) * (1 - O.SLIM_DECAY)
                self.new_stats[scope + "/moving_variance"] = mv - (mv - var_upd) * (1 - O.SLIM_DECAY)
            return y
        return _bn_infer(x, g, b, mm, mv)

    def _head16(self, cat, n_points, bn_decay):
        """conv5 .. the VLAD aggregation in the bf16-stored arithmetic (_Head16), with the moving-average updates of the two
        BatchNorms (tf_util's scheduled decay for conv5, slim's 0.999 with the population variance for cluster_bn)."""
        sc = "fastdgcnn/conv5"
        W5 = self.w[sc + "/weights"]
        mask = None
        if self.relu_masks is not None and sc in self.relu_masks:
            mask = torch.as_tensor(np.asarray(self.relu_masks[sc]).reshape(-1, 1024), dtype=torch.bool)
        pin = None
        if self.value_pins is not None and sc in self.value_pins:
            pin = torch.as_tensor(np.asarray(self.value_pins[sc], dtype=np.float64).reshape(-1, 1024), dtype=cat.dtype)
            with torch.no_grad():     # (this layer's own stored value, from the pinned input: what the pin replaces)
                own = _round_bf16(_round_bf16(cat) @ _round_bf16(W5.reshape(W5.shape[-2], W5.shape[-1])) + self.w[sc + "/biases"])
                self.value_pin_gap[sc] = float((pin - own).abs().max() / own.abs().max().clamp(min=1e-30))
        vlad, a_sum, mean5, var5, mean_c, var_c, z5, u = _Head16.apply(
            cat, W5.reshape(W5.shape[-2], W5.shape[-1]), self.w[sc + "/biases"], self.w[sc + "/bn/gamma"], self.w[sc + "/bn/beta"],
            self.w["VLAD/cluster_weights"], self.w["VLAD/cluster_bn/gamma"], self.w["VLAD/cluster_bn/beta"], n_points, True, mask, pin,
            O.BN_EPS)
        if mask is not None:
            rs5 = torch.rsqrt(var5 + O.BN_EPS) * self.w[sc + "/bn/gamma"].detach()
            pre = z5 * rs5 + (self.w[sc + "/bn/beta"].detach() - mean5 * rs5)
            self.relu_mask_disagreement[sc] = int(((pre > 0) != mask).sum())
        decay = 0.9 if bn_decay is None else bn_decay
        mname, vname = O.ema_names(sc, self.outer)
        with torch.no_grad():
            self.new_stats[mname] = self.w[mname] - (1 - decay) * (self.w[mname] - mean5)
            self.new_stats[vname] = self.w[vname] - (1 - decay) * (self.w[vname] - var5)
            cm, cv = self.w["VLAD/cluster_bn/moving_mean"], self.w["VLAD/cluster_bn/moving_variance"]
            self.new_stats["VLAD/cluster_bn/moving_mean"] = cm - (cm - mean_c) * (1 - O.SLIM_DECAY)
            self.new_stats["VLAD/cluster_bn/moving_variance"] = cv - (cv - var_c) * (1 - O.SLIM_DECAY)
        return vlad, a_sum

    # ---- forward -----------------------------------------------------------------------------------------------
    def forward(self, point_cloud: np.ndarray, is_training: bool, bn_decay: Optional[float] = None,
                mask: Optional[np.ndarray] = None, return_features: bool = False):
        """``return_features``: the KD variants' second output (models/kd_epc-net.py:158, models/kd_epc-net-l.py:102):
        (l2_normalize(conv5 output reshaped to (-1, 1024), 1), descriptors), rows in the INPUT point order."""
        B, P, N, D = point_cloud.shape
        pc32 = np.ascontiguousarray(point_cloud, dtype=np.float32).reshape(B * P, N, D)
        if mask is None:
            mask = O.pairwise_distance_mask(pc32)
        m = torch.tensor(mask, dtype=self.dtype)
        k = float(self.p["KNN"])
        inp = torch.tensor(pc32, dtype=self.dtype)
        nblocks = 4 if self.arch == "epc-net" else 2
        outs = []
        tr, bd = is_training, bn_decay
        for b in range(1, nblocks + 1):
            x = self.conv1d(inp, "fastdgcnn/conv%d" % b, tr, bd)
            xm = torch.matmul(m, x) / k
            t = xm - x
            t = self.conv1d(t, "fastdgcnn/conv%d_a" % b, tr, bd)
            t = self.conv1d(t, "fastdgcnn/conv%d_b" % b, tr, bd)
            inp = t + xm
            outs.append(inp)
        head16 = self.arch == "epc-net" and self.gemm_rounding == "bf16" and tr and not return_features
        if head16:
            vlad, a_sum = self._head16(torch.cat(outs, dim=-1).reshape(-1, 256), N, bd)
            features = None
        else:
            x = self.conv1d(torch.cat(outs, dim=-1), "fastdgcnn/conv5", tr, bd)
            features = _l2n(x.reshape(-1, 1024), 1) if return_features else None
        if self.arch == "epc-net":
            G = self.p["GROUPS"]
            if not head16:
                f = _l2n(x.reshape(-1, 1024), 1)
                act = self._mm(f, self.w["VLAD/cluster_weights"])
                act = torch.softmax(self._slim_bn(act, "VLAD/cluster_bn", tr, fused=False), dim=1).reshape(-1, N, 64)
                a_sum = act.sum(dim=-2, keepdim=True)
                # (the HIP step forms vlad[b] = f[b]^T @ act[b] directly: the product whose shape the rounding rule sees)
                vlad = self._mm(f.reshape(-1, N, 1024).transpose(1, 2), act)
            a = a_sum * self.w["VLAD/cluster_weights2"]
            vlad = vlad - a
            vlad = _l2n(vlad, 1).reshape(-1, 64 * 1024)
            vlad = _l2n(vlad, 1)
            y = self._mm(vlad.reshape(-1, 65536 // G), self.w["VLAD/hidden1_weights"])
            y = self._slim_bn(y, "VLAD/bn", tr, fused=True).reshape(-1, G, 256).sum(dim=-2)
            gates = torch.sigmoid(self._slim_bn(self._mm(y, self.w["VLAD/gating_weights"]), "VLAD/gating_bn", tr, fused=True))
            out = y * gates
        else:
            net = x.max(dim=1).values
            z = self._mm(net, self.w["VLAD/fc1/weights"]) + self.w["VLAD/fc1/biases"]
            out = self._relu(self._tfutil_bn(z, "VLAD/fc1", (0,), tr, bd), "VLAD/fc1")
        out = _l2n(out, 1).reshape(B, P, self.p["FEATURE_OUTPUT_DIM"])
        return (features, out) if return_features else out


def lazy_quadruplet_loss(q, pos, neg, other, m1, m2, select_on=None):
    """models/epc-net.py:269-284.  ``select_on`` = (q, pos, neg, other) of ANOTHER evaluation of the same descriptors (the
    implementation under test's): the loss's three selections -- the closest positive, the hardest negative of each term -- are then made
    on those numbers and applied to these (same function wherever the two evaluations select alike; descriptors of synthetic clouds lie
    a rounding apart, and two different selections would compare two different gradients)."""
    d_pos, d_neg, d_oth = ((pos - q) ** 2).sum(2), ((neg - q) ** 2).sum(2), ((neg - other) ** 2).sum(2)
    if select_on is None:
        best = d_pos.min(1).values.reshape(-1, 1)
        l1 = torch.clamp(m1 + best - d_neg, min=0).max(1).values.mean()
        l2 = torch.clamp(m2 + best - d_oth, min=0).max(1).values.mean()
        return l1 + l2
    qs, ps, ns, os_ = select_on
    s_pos, s_neg, s_oth = ((ps - qs) ** 2).sum(2), ((ns - qs) ** 2).sum(2), ((ns - os_) ** 2).sum(2)
    ib = s_pos.argmin(1, keepdim=True)
    s_best = s_pos.gather(1, ib)
    i1 = torch.clamp(m1 + s_best - s_neg, min=0).argmax(1, keepdim=True)
    i2 = torch.clamp(m2 + s_best - s_oth, min=0).argmax(1, keepdim=True)
    best = d_pos.gather(1, ib)
    l1 = torch.clamp(m1 + best - d_neg, min=0).gather(1, i1).mean()
    l2 = torch.clamp(m2 + best - d_oth, min=0).gather(1, i2).mean()
    return l1 + l2


def train_step(weights: Dict[str, np.ndarray], query, positives, negatives, other_neg, step: int, epoch: int,
               adam_m: Optional[Dict[str, np.ndarray]] = None, adam_v: Optional[Dict[str, np.ndarray]] = None,
               arch="epc-net", params=None, m1=0.5, m2=0.2, base_lr=5e-5, batch_num_queries=1, dtype=torch.float64,
               relu_masks: Optional[Dict[str, np.ndarray]] = None, gemm_rounding: Optional[str] = None,
               value_pins: Optional[Dict[str, np.ndarray]] = None):
    """One reference training step (train.py:251-277, 484-495): returns dict(loss, grads, new_weights, adam_m, adam_v).

    ``step`` = value of the global-step variable BEFORE the step (``batch``, train.py:246): bn_decay is evaluated with
    it; Adam's bias correction uses t = step + 1 (TensorFlow's beta*_power are multiplied after each apply)."""
    orc = TorchOracle(weights, arch, params, dtype)
    orc.relu_masks = relu_masks        # None = the reference's relu; a dict pins the masks (see TorchOracle.__init__)
    orc.gemm_rounding = gemm_rounding  # None = exact products; "bf16" = the configs[2] arithmetic (TorchOracle.__init__)
    orc.value_pins = value_pins        # None, or the stored pre-activations of the implementation under test (TorchOracle.__init__)
    vecs = np.concatenate([query, positives, negatives, other_neg], axis=1)          # train.py:252
    bn_decay = O.get_bn_decay(step, batch_num_queries)
    out = orc.forward(vecs, True, bn_decay)
    npos, nneg = positives.shape[1], negatives.shape[1]
    q, pos, neg, oth = torch.split(out, [1, npos, nneg, 1], dim=1)                    # train.py:255
    select_on = None
    if value_pins is not None and "descriptors" in value_pins:
        # the descriptors of the implementation under test: the loss SELECTS on them (lazy_quadruplet_loss), the values stay this
        # oracle's own
        pin = torch.as_tensor(np.asarray(value_pins["descriptors"], dtype=np.float64).reshape(tuple(out.shape)), dtype=out.dtype)
        orc.value_pin_gap["descriptors"] = float((pin - out.detach()).abs().max() / out.detach().abs().max())
        select_on = torch.split(pin, [1, npos, nneg, 1], dim=1)
    loss = lazy_quadruplet_loss(q, pos, neg, oth, m1, m2, select_on)
    grads = torch.autograd.grad(loss, [orc.w[k] for k in orc.trainable], allow_unused=True)
    lr = O.get_learning_rate(epoch, base_lr)
    b1, b2, eps = 0.9, 0.999, 1e-8                                                    # tf.train.AdamOptimizer defaults
    t = step + 1
    lr_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    new_w = {k: v.detach().numpy().copy() for k, v in orc.w.items()}
    am, av, g_out = {}, {}, {}
    for k, g in zip(orc.trainable, grads):
        g = np.zeros_like(new_w[k]) if g is None else g.detach().numpy()
        m_prev = np.zeros_like(g) if adam_m is None else adam_m[k]
        v_prev = np.zeros_like(g) if adam_v is None else adam_v[k]
        am[k] = b1 * m_prev + (1 - b1) * g
        av[k] = b2 * v_prev + (1 - b2) * g * g
        new_w[k] = new_w[k] - lr_t * am[k] / (np.sqrt(av[k]) + eps)
        g_out[k] = g
    for k, v in orc.new_stats.items():
        new_w[k] = v.numpy().copy()
    return {"loss": float(loss.detach()), "grads": g_out, "new_weights": new_w, "adam_m": am, "adam_v": av,
            "lr": lr, "bn_decay": bn_decay, "descriptors": out.detach().numpy(),
            "relu_mask_disagreement": dict(orc.relu_mask_disagreement), "value_pin_gap": dict(orc.value_pin_gap)}


def distill_step(teacher_weights: Dict[str, np.ndarray], student_weights: Dict[str, np.ndarray], query, positives, negatives,
                 other_neg, alpha: float = 0.1, beta: float = 1.0, gamma: float = 0.0, loss_type: str = "square_error_sum",
                 step: int = 0, teacher_arch="epc-net", student_arch="epc-net-l", params=None, m1=0.5, m2=0.2,
                 batch_num_queries=1, dtype=torch.float64, relu_masks: Optional[Dict[str, np.ndarray]] = None,
                 mask: Optional[np.ndarray] = None):
    """One distillation step of the reference up to the gradients (kd_train.py:255-425, configs/epc-net-l-d.yaml):

      teacher (kd_train.py:262-279): ``out_fea, out_vecs = MODEL_teacher.forward(vecs, is_training=False)`` -- fed False at
        :762 "always False" -- soft_label = reshape(out_vecs, [-1, 256]); its outputs reach the student graph through
        placeholders (:786-790), so nothing flows back into the teacher;
      student (:355-387): ``out_fea_student, out_vecs = MODEL_student.forward(vecs, is_training=True, bn_decay)``;
        loss_q = lazy_quadruplet_loss(split(out_vecs), MARGIN_1, MARGIN_2) (:371);
        square_error_sum (:330-331, :376-380): loss_soft = sum((soft_s - soft_t)^2), loss_fea = sum((fea_s - fea_t)^2);
        square_error_mean (:336-337, :381-383): the same with means;  any other LOSS_TYPE leaves loss_fea undefined (:387);
        loss = loss_q * beta + loss_soft * alpha + loss_fea * gamma (:387);  minimize over the student's variables (:401-403).

    Returns dict(loss, loss_q, loss_soft, loss_fea, grads {student name: array}, descriptors, relu_mask_disagreement).
    Weight names are the oracle's (relative to ``query_triplets``) for both models.  ``relu_masks`` pins the STUDENT's ReLU
    masks (TorchOracle.__init__); ``mask``: the kNN mask of the tuple (both models share the clouds), computed when None."""
    if loss_type not in ("square_error_sum", "square_error_mean"):
        raise NameError("name 'loss_fea' is not defined")
    vecs = np.concatenate([query, positives, negatives, other_neg], axis=1)
    B, P, N, D = vecs.shape
    if mask is None:
        mask = O.pairwise_distance_mask(np.ascontiguousarray(vecs, dtype=np.float32).reshape(B * P, N, D))
    teacher = TorchOracle(teacher_weights, teacher_arch, params, dtype)
    with torch.no_grad():
        fea_t, out_t = teacher.forward(vecs, False, None, mask=mask, return_features=True)
    soft_t = out_t.reshape(-1, out_t.shape[-1])
    student = TorchOracle(student_weights, student_arch, params, dtype)
    student.relu_masks = relu_masks
    bn_decay = O.get_bn_decay(step, batch_num_queries)
    fea_s, out_s = student.forward(vecs, True, bn_decay, mask=mask, return_features=True)
    npos, nneg = positives.shape[1], negatives.shape[1]
    q, pos, neg, oth = torch.split(out_s, [1, npos, nneg, 1], dim=1)
    loss_q = lazy_quadruplet_loss(q, pos, neg, oth, m1, m2)
    red = torch.sum if loss_type == "square_error_sum" else torch.mean
    loss_soft = red((out_s.reshape(-1, out_s.shape[-1]) - soft_t) ** 2)
    loss_fea = red((fea_s - fea_t) ** 2)
    loss = loss_q * beta + loss_soft * alpha + loss_fea * gamma
    grads = torch.autograd.grad(loss, [student.w[k] for k in student.trainable], allow_unused=True)
    g_out = {k: (np.zeros(tuple(student.w[k].shape)) if g is None else g.detach().numpy())
             for k, g in zip(student.trainable, grads)}
    return {"loss": float(loss.detach()), "loss_q": float(loss_q.detach()), "loss_soft": float(loss_soft.detach()),
            "loss_fea": float(loss_fea.detach()), "grads": g_out, "descriptors": out_s.detach().numpy(),
            "teacher_descriptors": out_t.detach().numpy(), "bn_decay": bn_decay, "new_stats": {k: v.numpy().copy() for k, v in student.new_stats.items()},
            "relu_mask_disagreement": dict(student.relu_mask_disagreement)}
