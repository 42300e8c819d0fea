"""CPU tests of the oracle: pinned against the reference's own artefacts where any exist (variable tables, parameter
counts, graph constants -- SURVEY.md 8c) and against its own internal consistency elsewhere (PARITY UNPINNED vs a
running TensorFlow: not installable here)."""
import json
import os

import numpy as np
import pytest

from helpers import O, ROOT

TABLES = json.load(open(os.path.join(ROOT, "tests", "golden", "ckpt_tables.json")))


def _ref_vars(tag):
    return {e["name"]: tuple(e["shape"]) for e in TABLES[tag]["entries"]
            if not e["name"].endswith(("/Adam", "/Adam_1")) and e["name"] not in ("Variable", "beta1_power", "beta2_power")}


@pytest.mark.parametrize("arch,count", [("epc-net", 4704832), ("epc-net-l", 418880)])
def test_variable_table_matches_reference_checkpoint(arch, count):
    t = O.variable_table(arch)
    assert {"query_triplets/" + k: v[0] for k, v in t.items()} == _ref_vars(arch)
    assert O.count_trainable(t) == count        # README/ckpt-derived parameter counts


def test_every_trainable_has_adam_slots_in_reference_table():
    names = {e["name"] for e in TABLES["epc-net"]["entries"]}
    t = O.variable_table("epc-net")
    for k, (_, kind) in t.items():
        if kind in ("weight", "bias", "gamma", "beta"):
            assert "query_triplets/%s/Adam" % k in names and "query_triplets/%s/Adam_1" % k in names


def test_mask_semantics_ties_and_padding():
    # self is always selected; row sum >= 20; all-zero cloud selects everything
    pc = O.synthetic_clouds(1, 128, 0)
    m = O.pairwise_distance_mask(pc)
    assert m.shape == (1, 128, 128) and m.dtype == np.float32
    assert np.all(np.diagonal(m[0]) == 1.0)
    assert np.all(m.sum(-1) >= 20)
    z = O.pairwise_distance_mask(O.synthetic_clouds(1, 64, 0, "zeros"))
    assert np.all(z == 1.0)
    lat = O.pairwise_distance_mask(O.synthetic_clouds(1, 512, 0, "lattice"))
    assert lat.sum(-1).max() > 20              # exact ties are all kept (greater_equal)


def test_mask_k_argument_is_ignored_like_the_reference():
    pc = O.synthetic_clouds(1, 64, 1)
    assert np.array_equal(O.pairwise_distance_mask(pc, k=5), O.pairwise_distance_mask(pc, k=20))


@pytest.mark.parametrize("arch", ["epc-net", "epc-net-l"])
def test_dense_and_index_formulations_agree(arch):
    w = O.seeded_weights(arch, 0)
    pc = O.synthetic_clouds(2, 128, 3, "lidar")[:, None]
    a, _ = O.forward(pc, w, arch=arch, formulation="dense")
    b, _ = O.forward(pc, w, arch=arch, formulation="lists")
    assert np.abs(a - b).max() <= 2e-6
    assert np.allclose(np.linalg.norm(a, axis=-1), 1.0, atol=1e-6)


@pytest.mark.parametrize("arch", ["epc-net", "epc-net-l"])
def test_fp32_tolerance_budget(arch):
    """fp32 vs the fp64 shadow (same neighbour sets): the error budget the 1e-4 GPU bar sits far above."""
    w = O.seeded_weights(arch, 1)
    pc = O.synthetic_clouds(2, 256, 4)
    mask = O.pairwise_distance_mask(pc)
    a, _ = O.forward(pc[:, None], w, arch=arch)
    b, _ = O.forward(pc[:, None], w, arch=arch, dtype=np.float64, mask=mask)
    assert np.linalg.norm(a - b, axis=-1).max() <= 5e-6


def test_grouped_projection_fold_identity():
    """with affine (inference) BN and one shared weight, sum_g BN(v_g W) == s*((sum_g v_g) W) + G*t."""
    w = O.seeded_weights("epc-net", 2)
    st = O.State(w, np.float64)
    rng = np.random.RandomState(0)
    v = rng.randn(3, 65536)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    H = st.w["VLAD/hidden1_weights"]
    y = O.slim_batch_norm(st, (v.reshape(-1, 16384) @ H), "VLAD/bn", False, True).reshape(3, 4, 256).sum(1)
    s = st.w["VLAD/bn/gamma"] / np.sqrt(st.w["VLAD/bn/moving_variance"] + 1e-3)
    t = st.w["VLAD/bn/beta"] - st.w["VLAD/bn/moving_mean"] * s
    y2 = (v.reshape(3, 4, 16384).sum(1) @ H) * s + 4 * t
    assert np.abs(y - y2).max() <= 1e-10


def test_training_mode_statistics_and_ema():
    w = O.seeded_weights("epc-net", 0, mode="init")
    pc = O.synthetic_clouds(2, 64, 0)[None]                  # (1, 2, 64, 3)
    out, st = O.forward(pc, w, is_training=True, bn_decay=0.5)
    assert np.allclose(np.linalg.norm(out, axis=-1), 1.0, atol=1e-5)
    m, v = O.ema_names("fastdgcnn/conv1")
    bm, bv = st.batch_stats["fastdgcnn/conv1/bn"]
    assert np.allclose(st.new_stats[m], 0.5 * bm, atol=1e-7)  # shadow starts at 0: 0 - 0.5*(0 - mean)
    assert np.allclose(st.new_stats[v], 0.5 * bv, atol=1e-7)
    # fused slim BN feeds the Bessel-corrected variance to the moving average (rows = B*P*G = 8)
    _, var = st.batch_stats["VLAD/bn"]
    assert np.allclose(st.new_stats["VLAD/bn/moving_variance"], 1 - (1 - var * 8 / 7) * 0.001, atol=1e-6)


def test_losses_known_answers():
    q = np.zeros((1, 1, 4)); q[0, 0, 0] = 1
    pos = np.stack([q[0, 0], np.array([0, 1, 0, 0.])])[None]           # best positive distance 0
    neg = np.array([[[-1, 0, 0, 0.], [0, 0, 1, 0.]]])                   # d^2 = 4 and 2
    other = np.array([[[0, 0, 0, 1.]]])
    assert O.lazy_triplet_loss(q, pos, neg, 0.5) == 0.0                  # negatives are far
    assert O.triplet_loss(q, pos, neg, 2.5) == pytest.approx(0.5)        # only the d^2 = 2 negative violates
    assert O.lazy_quadruplet_loss(q, pos, neg, other, 0.5, 0.2) == 0.0
    assert O.lazy_quadruplet_loss(q, pos, neg, other, 2.5, 2.5) == pytest.approx(0.5 + 0.5)


def test_schedules():
    assert O.get_bn_decay(0) == 0.5 and O.get_bn_decay(200000) == 0.75 and O.get_bn_decay(10 ** 8) == 0.99
    assert O.get_learning_rate(0) == 5e-5 and O.get_learning_rate(5) == pytest.approx(4.5e-5)
    assert O.get_learning_rate(1000) == 1e-5


def test_get_recall_semantics_against_sklearn():
    from sklearn.neighbors import KDTree
    rng = np.random.RandomState(0)
    db = rng.randn(250, 256).astype(np.float32); db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = db[rng.randint(0, 250, 40)] + 0.2 * rng.randn(40, 256).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    truth = [list(rng.choice(250, size=rng.randint(0, 4), replace=False)) for _ in range(40)]
    _, ind = KDTree(db).query(q, k=25)
    rec, sim, one = O.get_recall(db, q, truth, indices=ind)
    rec2, sim2, one2 = O.get_recall(db, q, truth)                     # brute-force neighbours
    assert np.array_equal(rec, rec2) and one == one2 and np.allclose(sim, sim2)
    assert max(int(round(250 / 100.0)), 1) == 2 and max(int(round(50 / 100.0)), 1) == 1  # banker's rounding, :470
    assert rec.shape == (25,) and np.all(np.diff(rec) >= 0)


def test_distill_oracle_composition_and_finite_difference():
    """oracle/epcnet_oracle_torch.distill_step (kd_train.py:255-425): loss = beta * quadruplet + alpha * soft + gamma * feature
    term with the teacher in inference mode; its autograd gradient of a student weight against a central finite difference of
    the same float64 function; the teacher gets no gradient by construction (its forward runs under no_grad)."""
    import epcnet_oracle_torch as T
    wt, ws = O.seeded_weights("epc-net", 2), O.seeded_weights("epc-net-l", 3)
    pcs = O.synthetic_clouds(18, 64, 5)
    tup = (pcs[None, :1], pcs[None, 1:3], pcs[None, 3:17], pcs[None, 17:])
    for lt in ("square_error_sum", "square_error_mean"):
        r = T.distill_step(wt, ws, *tup, alpha=0.1, beta=1.0, gamma=0.5, loss_type=lt)
        assert abs(r["loss"] - (r["loss_q"] + 0.1 * r["loss_soft"] + 0.5 * r["loss_fea"])) < 1e-10
        soft_s, soft_t = r["descriptors"].reshape(-1, 256), r["teacher_descriptors"].reshape(-1, 256)
        red = np.sum if lt == "square_error_sum" else np.mean
        assert abs(r["loss_soft"] - red((soft_s - soft_t) ** 2)) < 1e-9
    with pytest.raises(NameError):
        T.distill_step(wt, ws, *tup, loss_type="mse")
    r = T.distill_step(wt, ws, *tup, gamma=0.5)
    assert len(r["grads"]) == 32
    k, idx, eps = "fastdgcnn/conv2_a/weights", (0, 3, 5), 1e-6
    w2 = {a: b.astype(np.float64).copy() for a, b in ws.items()}
    w2[k][idx] += eps
    up = T.distill_step(wt, w2, *tup, gamma=0.5)["loss"]
    w2[k][idx] -= 2 * eps
    dn = T.distill_step(wt, w2, *tup, gamma=0.5)["loss"]
    assert abs((up - dn) / (2 * eps) - r["grads"][k][idx]) < 1e-5 * max(1.0, abs(r["grads"][k][idx]))


def test_head16_restatement_without_rounding_is_the_plain_graph():
    """oracle/epcnet_oracle_torch.py: _Head16 writes out the forward AND the backward of conv5 .. the VLAD aggregation so that it can
    round where the bf16-stored head of the HIP step rounds.  With the rounding switched off it must BE the plain graph: its outputs equal
    the composition's and its hand-written backward equals autograd's of that composition (float64, 1e-9) -- a derivation error in the
    restatement cannot hide behind the rounding noise of the GPU comparison."""
    import torch
    import epcnet_oracle_torch as T
    g = torch.Generator().manual_seed(5)
    B, N = 3, 64
    R = B * N
    rnd = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    leaves = [rnd(R, 256), rnd(256, 1024) * 0.1, rnd(1024) * 0.1, 1 + 0.2 * rnd(1024), 0.3 * rnd(1024), rnd(1024, 64) * 0.2,
              1 + 0.2 * rnd(64), 0.3 * rnd(64)]
    wv, wa = rnd(B, 1024, 64), rnd(B, 1, 64)                  # random cotangents

    def plain(cat, W5, b5, g5, bt5, Wc, gc, btc):
        z = cat @ W5 + b5
        y, _, _ = T._bn_train(z, g5, bt5, (0,))
        f = T._l2n(torch.relu(y), 1)
        za = f @ Wc
        ya, _, _ = T._bn_train(za, gc, btc, (0,))
        a = torch.softmax(ya, dim=1).reshape(B, N, 64)
        return torch.matmul(f.reshape(B, N, 1024).transpose(1, 2), a), a.sum(1, keepdim=True)

    outs = []
    for fn in (plain, lambda *xs: T._Head16.apply(*xs, N, False, None, None, O.BN_EPS)[:2]):
        xs = [x.clone().requires_grad_(True) for x in leaves]
        vlad, a_sum = fn(*xs)
        grads = torch.autograd.grad((vlad * wv).sum() + (a_sum * wa).sum(), xs, allow_unused=True)
        outs.append((vlad.detach(), a_sum.detach(), grads))
    (v0, s0, g0), (v1, s1, g1) = outs
    assert (v0 - v1).abs().max() <= 1e-12 and (s0 - s1).abs().max() <= 1e-12
    for k, (a, b) in enumerate(zip(g0, g1)):
        if k == 2:       # b5: exactly zero in exact arithmetic (a bias in front of a training-mode BatchNorm); autograd holds rounding noise
            assert a.abs().max() <= 1e-10 and b.abs().max() == 0
            continue
        assert (a - b).abs().max() <= 1e-9 * max(1.0, float(a.abs().max())), (k, float((a - b).abs().max()))


def test_selection_pinned_loss_is_the_loss_where_the_selections_agree():
    """lazy_quadruplet_loss(select_on=...) (the bf16 step test's last value pin): selecting on the same descriptors is the plain loss,
    value and gradient; selecting on other numbers applies THEIR closest positive / hardest negatives to these distances."""
    import torch
    import epcnet_oracle_torch as T
    g = torch.Generator().manual_seed(0)
    mk = lambda *s: torch.randn(*s, generator=g, dtype=torch.float64)
    q, pos, neg, oth = mk(3, 1, 8).requires_grad_(True), mk(3, 2, 8), mk(3, 5, 8), mk(3, 1, 8)
    a = T.lazy_quadruplet_loss(q, pos, neg, oth, 5.0, 2.0)
    b = T.lazy_quadruplet_loss(q, pos, neg, oth, 5.0, 2.0, (q.detach(), pos, neg, oth))
    assert float(a) == float(b)
    assert torch.equal(torch.autograd.grad(a, q)[0], torch.autograd.grad(b, q)[0])
    # selecting on OTHER numbers: their argmin / argmax, applied to these distances -- by hand
    qs, ps, ns, os_ = mk(3, 1, 8), mk(3, 2, 8), mk(3, 5, 8), mk(3, 1, 8)
    c = T.lazy_quadruplet_loss(q, pos, neg, oth, 5.0, 2.0, (qs, ps, ns, os_))
    d_pos, d_neg, d_oth = ((pos - q) ** 2).sum(2), ((neg - q) ** 2).sum(2), ((neg - oth) ** 2).sum(2)
    s_pos, s_neg, s_oth = ((ps - qs) ** 2).sum(2), ((ns - qs) ** 2).sum(2), ((ns - os_) ** 2).sum(2)
    want = 0.0
    for b_ in range(3):
        ib = int(s_pos[b_].argmin())
        i1 = int(torch.clamp(5.0 + s_pos[b_, ib] - s_neg[b_], min=0).argmax())
        i2 = int(torch.clamp(2.0 + s_pos[b_, ib] - s_oth[b_], min=0).argmax())
        want += float(torch.clamp(5.0 + d_pos[b_, ib] - d_neg[b_, i1], min=0) + torch.clamp(2.0 + d_pos[b_, ib] - d_oth[b_, i2], min=0)) / 3
    assert float(c) == pytest.approx(want, rel=1e-12)
