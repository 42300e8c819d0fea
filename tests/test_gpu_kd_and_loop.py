"""GPU tests of the rows SURVEY 8f ranks 3-4: the KD models / distillation step (kd_train.py:255-425) and the training
driver with hard-negative mining and checkpoint save / resume (train.py:330-617)."""
import importlib
import logging

import numpy as np
import pytest
import torch

import helpers as H
from helpers import O

pytestmark = pytest.mark.gpu
N = 128


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _kd_store(dev, wt, ws, N=N):
    """One store holding the teacher under teacher/query_triplets (fastdgcnn) and the student under
    student/query_triplets (BACKBONE), filled with oracle-named weights."""
    V = H.pkg("variables")
    st = V.reset_default_store(device=dev, seed=0)
    T = H.pkg("models.kd_epc-net")
    S = H.pkg("models.kd_epc-net-l")
    with V.variable_scope("teacher/query_triplets"):
        T.declare_variables(H.PARAMS, N)
    with V.variable_scope("student/query_triplets"):
        S.declare_variables(H.PARAMS, N)

    def rename(k, root, backbone):
        # EMA shadow names embed the full scope a second time (utils/tf_util.py:474-487)
        k = k.replace("bn/query_triplets/", "bn/" + root + "/")
        return (root + "/" + k).replace("/fastdgcnn/", "/" + backbone + "/")
    both = {rename(k, "teacher/query_triplets", "fastdgcnn"): v for k, v in wt.items()}
    both.update({rename(k, "student/query_triplets", "BACKBONE"): v for k, v in ws.items()})
    st.load_state_dict(both, strict=True)
    return st, T, S


def test_kd_models_match_the_plain_models(dev):
    """kd_epc-net / kd_epc-net-l are epc-net / epc-net-l with renamed scopes and a second output: same descriptors as
    the oracle-checked models, features = L2-normalised conv5 rows, fused shortcut == op-by-op graph."""
    wt, ws = O.seeded_weights("epc-net", 2), O.seeded_weights("epc-net-l", 3)
    st, T, S = _kd_store(dev, wt, ws)
    V = H.pkg("variables")
    assert all(k in st.vars for k in ["student/query_triplets/BACKBONE/conv1/weights",
                                      "student/query_triplets/VLAD/fc1/weights",
                                      "teacher/query_triplets/fastdgcnn/conv5/weights"])
    pc = O.synthetic_clouds(3, N, 1)
    x = torch.from_numpy(pc[None]).to(dev)                               # (1, 3, N, 3)
    for mod, scope, arch, w in ((T, "teacher/query_triplets", "epc-net", wt), (S, "student/query_triplets", "epc-net-l", ws)):
        ref, stt = O.forward(pc[:, None], w, arch=arch)
        with V.variable_scope(scope), torch.no_grad():
            fea, out = mod.forward(x, False, params=H.PARAMS)
            fused = mod.descriptors(x, H.PARAMS)
        assert tuple(fea.shape) == (3 * N, 1024) and tuple(out.shape) == (1, 3, 256)
        assert np.linalg.norm(out[0].cpu().numpy() - ref.reshape(3, 256), axis=1).max() < 1e-4
        assert np.linalg.norm((fused - out)[0].cpu().numpy(), axis=1).max() < 1e-4
        assert float((fea.norm(dim=1) - 1).abs().max()) < 1e-4
        # rows follow each cloud's Morton order: compare as sets of rows through the per-row feature norm profile
        conv5 = stt.taps["fastdgcnn/conv5"].reshape(3, N, 1024)
        want = conv5 / np.sqrt(np.maximum((conv5 ** 2).sum(-1, keepdims=True), 1e-12))
        got = fea.reshape(3, N, 1024).cpu().numpy()
        for c in range(3):
            d = np.abs(np.sort(got[c].sum(1)) - np.sort(want[c].sum(1))).max()
            assert d < 1e-3


def test_distill_step(dev):
    wt, ws = O.seeded_weights("epc-net", 2), O.seeded_weights("epc-net-l", 3)
    st, T, S = _kd_store(dev, wt, ws)
    KD = H.pkg("kd_training")
    V = H.pkg("variables")
    params = dict(H.PARAMS, ARCH_TEACHER="kd_epc-net", ARCH_STUDENT="kd_epc-net-l", LOSS_TYPE="square_error_sum",
                  ALPHA=0.1, BETA=1.0, GAMMA=0.0, BATCH_NUM_QUERIES=1, BASE_LEARNING_RATE=1e-3, DECAY_STEP=200000,
                  MARGIN_1=0.5, MARGIN_2=0.2)
    ds = KD.DistillStep(params, st)
    pcs = O.synthetic_clouds(18, N, 5)
    to = lambda a: torch.from_numpy(a).to(dev)
    q, pos, neg, oth = to(pcs[None, :1]), to(pcs[None, 1:3]), to(pcs[None, 3:17]), to(pcs[None, 17:])
    teacher_before = {k: v.detach().clone() for k, v in st.vars.items() if k.startswith("teacher/")}
    student_before = {k: v.detach().clone() for k, v in st.vars.items() if k.startswith("student/")}
    loss, lr, bn_decay = ds.step(q, pos, neg, oth, epoch=1)
    assert lr == pytest.approx(1e-3) and bn_decay == pytest.approx(0.5)
    aux = ds.last_aux
    # the composition of kd_train.py:387 from its separately tested parts
    vecs = torch.cat([q, pos, neg, oth], 1)
    with V.variable_scope("teacher/query_triplets"), torch.no_grad():
        soft_t = T.descriptors(vecs, params).reshape(-1, 256)
    soft_s = torch.cat([aux["q_vec"], aux["pos_vecs"], aux["neg_vecs"], aux["other_neg_vec"]], 1).reshape(-1, 256)
    want_soft = float(((soft_s - soft_t) ** 2).sum())
    assert float(aux["loss_soft"]) == pytest.approx(want_soft, rel=1e-4)
    want_q = float(S.lazy_quadruplet_loss(aux["q_vec"], aux["pos_vecs"], aux["neg_vecs"], aux["other_neg_vec"], 0.5, 0.2))
    assert float(aux["loss_q"]) == pytest.approx(want_q, rel=1e-5)
    assert float(loss) == pytest.approx(want_q + 0.1 * want_soft, rel=1e-5)
    # only the student moves; its optimizer slots are the 32 trainable student tensors x 2 (64 /Adam* slots)
    for k, v in teacher_before.items():
        assert torch.equal(st.vars[k], v), k
    moved = [k for k, v in student_before.items() if k in st.trainable and not torch.equal(st.vars[k], v)]
    # the 8 biases sit in front of a training-mode BatchNorm: their gradient is exactly zero (ops.Linear) and they stay put
    still = [k for k in student_before if k in st.trainable and k not in moved]
    assert len(moved) >= 24 and all(k.endswith("/biases") for k in still), still
    slots = [k for k in ds.optimizer_state() if k.endswith("/Adam") or k.endswith("/Adam_1")]
    assert len(slots) == 64 and all(k.startswith("student/") for k in slots)
    # GAMMA != 0 builds the teacher's feature map and adds the feature term
    ds2 = KD.DistillStep(dict(params, GAMMA=0.5, LOSS_TYPE="square_error_mean"), st)
    ds2._ensure_built(N)
    with torch.no_grad():
        l2 = ds2.compute_loss(q, pos, neg, oth, False, None)
    a2 = ds2.last_aux
    assert float(a2["loss_fea"]) > 0
    assert float(l2) == pytest.approx(float(a2["loss_q"]) + 0.1 * float(a2["loss_soft"]) + 0.5 * float(a2["loss_fea"]), rel=1e-5)
    with pytest.raises(NameError):                                      # kd_train.py:373-387 for LOSS_TYPE "mse"
        KD.DistillStep(dict(params, LOSS_TYPE="mse"), st).compute_loss(q, pos, neg, oth, False, None)


@pytest.mark.parametrize("n,gamma,loss_type", [(256, 0.0, "square_error_sum"), (256, 0.5, "square_error_sum"),
                                               (256, 0.5, "square_error_mean"), (4096, 0.0, "square_error_sum"),
                                               (4096, 0.5, "square_error_sum")])
def test_distill_step_matches_the_float64_oracle(dev, n, gamma, loss_type):
    """the distillation step of kd_train.py:255-425 against its float64 restatement
    (oracle/epcnet_oracle_torch.distill_step: teacher in inference mode, student in training mode, loss = beta * quadruplet +
    alpha * soft term + gamma * feature term, gradients of the student's 30 trainable tensors).  Loss terms to 1e-5 relative; with
    the ReLU masks of the HIP forward pinned in the oracle every student gradient to 1e-3 relative L2 -- at 18 x 256 and at the
    full 18 x 4096, without (GAMMA 0, the shipped config) and with the feature term."""
    import epcnet_oracle_torch as T
    wt, ws = O.seeded_weights("epc-net", 2), O.seeded_weights("epc-net-l", 3)
    st, _, _ = _kd_store(dev, wt, ws, n)
    KD, TR, TFU = H.pkg("kd_training"), H.pkg("training"), H.pkg("utils.tf_util")
    alpha, beta = 0.1, 1.0
    params = dict(H.PARAMS, ARCH_TEACHER="kd_epc-net", ARCH_STUDENT="kd_epc-net-l", LOSS_TYPE=loss_type, ALPHA=alpha, BETA=beta,
                  GAMMA=gamma, BATCH_NUM_QUERIES=1, BASE_LEARNING_RATE=1e-3, DECAY_STEP=200000, MARGIN_1=0.5, MARGIN_2=0.2)
    ds = KD.DistillStep(params, st, feature_loss_when_unused=True)
    pcs = O.synthetic_clouds(18, n, 5)
    to = lambda a: torch.from_numpy(a).to(dev)
    grads = {}
    orig = H.pkg("ops").adam_multi

    def spy(ws_, ms, vs, gs, lr, t, *a):
        for w, g in zip(ws_, gs):
            for k, t_ in st.vars.items():
                if t_.data_ptr() == w.data_ptr():
                    grads[k] = g.detach().cpu().numpy().copy()
        return orig(ws_, ms, vs, gs, lr, t, *a)

    H.pkg("ops").adam_multi = spy
    TR.ops.adam_multi = spy
    TFU.RELU_MASK_TAPS = {}
    try:
        loss, _, _ = ds.step(to(pcs[None, :1]), to(pcs[None, 1:3]), to(pcs[None, 3:17]), to(pcs[None, 17:]), epoch=1)
    finally:
        H.pkg("ops").adam_multi = orig
        TR.ops.adam_multi = orig
        masks, TFU.RELU_MASK_TAPS = TFU.RELU_MASK_TAPS, None
    torch.cuda.synchronize()
    pre = "student/query_triplets/"
    masks = {k[len(pre):].replace("BACKBONE/", "fastdgcnn/"): v.cpu().numpy() for k, v in masks.items() if k.startswith(pre)}
    assert len(masks) == 8, sorted(masks)
    # the training forward Morton-sorts every cloud (teacher and student alike): the oracle gets the sorted clouds, so that the
    # pinned masks' rows and the feature rows correspond; the loss and the gradients do not depend on the order
    srt = H.pkg("ops").morton_sort(to(pcs)).cpu().numpy()[None]
    ref = T.distill_step(wt, ws, srt[:, :1], srt[:, 1:3], srt[:, 3:17], srt[:, 17:], alpha=alpha, beta=beta, gamma=gamma,
                         loss_type=loss_type, step=0, relu_masks=masks)
    aux = ds.last_aux
    flips = sum(ref["relu_mask_disagreement"].values())
    total = sum(int(np.prod(m.shape)) for m in masks.values())
    assert flips <= 2e-5 * total, (flips, total)
    assert float(aux["loss_q"]) == pytest.approx(ref["loss_q"], rel=2e-5, abs=1e-6)
    assert float(aux["loss_soft"]) == pytest.approx(ref["loss_soft"], rel=1e-5)
    assert float(aux["loss_fea"]) == pytest.approx(ref["loss_fea"], rel=1e-5)
    assert float(loss) == pytest.approx(ref["loss"], rel=1e-5)
    worst = (0.0, "")
    assert len(ref["grads"]) == 32
    scale = max(np.linalg.norm(v) for v in ref["grads"].values())
    for k, g_ref in ref["grads"].items():
        name = pre + k.replace("fastdgcnn/", "BACKBONE/")
        g = grads[name].reshape(g_ref.shape)
        if k.endswith("/biases") or np.linalg.norm(g_ref) <= 1e-9:
            # exactly-zero true gradients: every bias sits in front of a training-mode BatchNorm, and EPC-Net-L's conv5 beta shifts
            # every cloud's pooled feature alike, which fc1's training-mode BatchNorm over the tuple removes (both sides hold
            # rounding noise there)
            assert np.abs(g).max() <= 5e-5, (k, np.abs(g).max())
            continue
        rel_l2 = np.linalg.norm(g - g_ref) / max(np.linalg.norm(g_ref), 1e-30)
        # (a tensor whose true gradient is a tiny remainder -- EPC-Net-L's conv5 beta under the feature term, which breaks the exact
        # cancellation above by 1e-7 of the network's gradient scale -- is held by that scale: float32 noise, not a relative error)
        if np.linalg.norm(g - g_ref) <= 1e-6 * scale:
            continue
        worst = max(worst, (rel_l2, k))
        assert rel_l2 <= 1e-3, "mask-pinned student gradient of %s: relative L2 error %.3e" % (k, rel_l2)
    print("distill step 18x%d gamma %.1f %s: loss terms q %.6f soft %.6f fea %.6f; worst student gradient rel L2 %.2e (%s), "
          "%d of %d mask elements differ" % (n, gamma, loss_type, ref["loss_q"], ref["loss_soft"], ref["loss_fea"], worst[0],
                                             worst[1], flips, total))


def _dataset(T, n, seed=0):
    rng = np.random.default_rng(seed)
    data = rng.uniform(-1, 1, (T, n, 3)).astype(np.float32)
    queries = {}
    for i in range(T):
        queries[i] = {"query": "%d.bin" % i, "positives": [j for j in range(T) if j != i and abs(j - i) <= 2],
                      "negatives": [j for j in range(T) if abs(j - i) > 4]}
    return queries, data


@pytest.mark.parametrize("arch", ["epc-net-l"])
def test_training_loop_with_mining_and_resume(dev, tmp_path, arch):
    V = H.pkg("variables")
    TR = H.pkg("training")
    TL = H.pkg("train_loop")
    tb = H.pkg("tf_bundle")
    params = dict(H.PARAMS, ARCH=arch, BATCH_NUM_QUERIES=1, POSITIVES_PER_QUERY=2, NEGATIVES_PER_QUERY=6,
                  NUM_POINTS=N, BASE_LEARNING_RATE=1e-3, MAX_EPOCH=8)
    queries, data = _dataset(40, N)
    st = V.reset_default_store(device=dev, seed=0)
    ts = TR.TrainStep(params, st)
    ts._ensure_built(N)
    st.randomize_statistics(0)
    tr = TL.Trainer(ts, queries, data, queries, data, save_path=str(tmp_path), logger=logging.getLogger("t"))
    np.random.seed(0)
    import random
    random.seed(0)
    # cached descriptors -> the mining branch (train.py:373-377): hard negatives are the nearest cached descriptors
    tr.TRAINING_LATENT_VECTORS = tr.get_latent_vectors()
    assert tr.TRAINING_LATENT_VECTORS.shape == (40, 256)
    hard = tr._hard_negatives(7)
    lat = tr.TRAINING_LATENT_VECTORS
    negs = np.asarray(queries[7]["negatives"])
    d = np.linalg.norm(lat[negs] - tr.get_feature_representation(7), axis=1)
    assert sorted(hard) == sorted(negs[np.argsort(d, kind="stable")[:10]].tolist())
    losses = tr.train_one_epoch(6, max_iters=4)
    assert len(losses) == 4 and all(np.isfinite(losses)) and ts.global_step == 4
    tr.graph = True                                  # the same loop replaying the captured HIP graph of the step
    losses_g = tr.train_one_epoch(6, max_iters=3)
    assert len(losses_g) == 3 and all(np.isfinite(losses_g)) and ts.global_step == 7
    tr.graph = False
    ts.global_step = 4
    assert np.isfinite(tr.evaluate_loss(6))
    # save (train.py:611-617) -> a fresh process state -> restore (train.py:308-315) -> identical continuation
    prefix = tr.save(6, 101)
    ent = tb.read_index(prefix + ".index")
    assert "Variable" in ent and "beta1_power" in ent
    assert "query_triplets/fastdgcnn/conv1/weights/Adam_1" in ent and "query_triplets/VLAD/fc1/weights" in ent
    q = [torch.from_numpy(data[i][None, None]).to(dev) for i in range(4)]
    batch = (q[0], torch.cat([q[1], q[2]], 1), torch.cat([q[3]] * 6, 1), q[1])
    l_a, _, _ = ts.step(*batch, epoch=6)
    w_a = st.vars["query_triplets/VLAD/fc1/weights"].detach().clone()
    st2 = V.reset_default_store(device=dev, seed=123)
    ts2 = TR.TrainStep(params, st2)
    tr2 = TL.Trainer(ts2, queries, data, save_path=str(tmp_path))
    tr2.restore(prefix)
    assert ts2.global_step == 4
    l_b, _, _ = ts2.step(*batch, epoch=6)
    assert float(l_a) == pytest.approx(float(l_b), rel=1e-6)
    assert torch.allclose(st2.vars["query_triplets/VLAD/fc1/weights"], w_a, rtol=1e-6, atol=1e-9)
