"""The HIP path against the committed golden vectors G1-G6 (tests/golden/vectors.npz).

Same inputs as scripts/make_golden_vectors.py (regenerated from their seeds), expected outputs read from the fixture
instead of being recomputed: these tests pass or fail on numbers that were fixed when the fixture was committed.
Bars: bit-exact for the kNN graph and the retrieval indices, descriptor L2 error <= 1e-4, stage tolerances as in
test_gpu_parity.py / test_gpu_train_step.py."""
import importlib.util
import os

import numpy as np
import pytest
import torch

import helpers as H
from helpers import O

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("make_golden_vectors",
                                               os.path.join(ROOT, "scripts", "make_golden_vectors.py"))
G = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(G)
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "vectors.npz"))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X; there is no CPU fallback"
    return torch.device("cuda:0")


@pytest.mark.parametrize("kind,n,seed", G.KNN_CASES)
def test_g1_knn_graph(dev, kind, n, seed):
    tf_util = H.pkg("utils.tf_util")
    key = "g1/%s_%d_%d/" % (kind, n, seed)
    pc = O.synthetic_clouds(2, n, seed, kind)
    kth, idx, cnt = (t.cpu().numpy() for t in tf_util.knn_index(torch.from_numpy(pc).to(dev)))
    assert np.array_equal(kth, GOLD[key + "kth"])
    assert np.array_equal(cnt, GOLD[key + "cnt"])
    cap = idx.shape[-1]
    gidx = GOLD[key + "idx"].astype(np.int32)
    width = min(cap, gidx.shape[-1])
    listed = np.minimum(cnt, width)[..., None] > np.arange(width)
    assert np.array_equal(np.where(listed, idx[..., :width], -1), np.where(listed, gidx[..., :width], -1))
    whole = cnt <= cap                                   # rows the index list holds completely: checksum all of it
    sel = np.arange(cap) < cnt[..., None]
    assert np.array_equal(np.where(sel, idx, 0).astype(np.int64).sum(-1)[whole], GOLD[key + "isum"][whole])
    assert np.array_equal(np.bitwise_xor.reduce(np.where(sel, idx, 0), axis=-1)[whole], GOLD[key + "ixor"][whole])


@pytest.mark.parametrize("prec", ["f32", "fast"])
def test_g2_proxyconv_block(dev, prec):
    w = O.seeded_weights("epc-net", 5)
    pc = O.synthetic_clouds(2, 256, 5, "lidar")
    ref = GOLD["g2/block1"]
    eng, _ = H.make_engine("epc-net", w, dev, precision=prec)
    got = H.run_stages(eng, torch.from_numpy(pc).to(dev))
    blk = got["cat"][..., :64].float().cpu().numpy()
    # the default arithmetic keeps f32 rows (2e-5, the stage bar); the fast one fp16 rows (include/epcnet.h)
    assert np.abs(blk - ref).max() <= (2e-5 if prec == "f32" else 1e-3) * np.abs(ref).max()
    # the f32 operator chain (training / unfused path): neighbour mean and block output at f32 accuracy
    V, tf_util, ops = H.pkg("variables"), H.pkg("utils.tf_util"), H.pkg("ops")
    x = torch.from_numpy(pc).to(dev)
    with V.variable_scope(H.OUTER), V.variable_scope("fastdgcnn"), torch.no_grad():
        conv = lambda t, s: tf_util.conv1d(t, 64, 1, padding='VALID', stride=1, bn=True, is_training=False, scope=s)
        graph = ops.KnnGraph(x)
        c1 = conv(x, 'conv1')
        mean = ops.NeighbourMean.apply(c1.reshape(-1, 64), graph, 20).reshape(c1.shape)
        out = conv(conv(mean - c1, 'conv1_a'), 'conv1_b') + mean
    assert np.abs(mean.cpu().numpy() - GOLD["g2/mean1"]).max() <= 2e-5 * np.abs(GOLD["g2/mean1"]).max()
    assert np.abs(out.cpu().numpy() - ref).max() <= 2e-5 * np.abs(ref).max()


def test_g3_gvlad(dev):
    V, lp = H.pkg("variables"), H.pkg("loupe")
    w = O.seeded_weights("epc-net", 6)
    H.make_store("epc-net", w, dev)
    feats = torch.from_numpy(G.g3_features()).to(dev)
    with V.variable_scope(H.OUTER), V.variable_scope("VLAD"), torch.no_grad():
        pool = lp.G_VLAD(feature_size=1024, max_samples=256, cluster_size=64, output_dim=256, groups=4, gating=True,
                         add_batch_norm=True, is_training=False)
        out = pool.forward(feats).cpu().numpy()
    ref = GOLD["g3/gated"]
    assert np.abs(out - ref).max() <= 2e-4 * np.abs(ref).max() + 1e-6


@pytest.mark.parametrize("arch", ["epc-net", "epc-net-l"])
def test_g4_full_network_descriptors(dev, arch):
    V = H.pkg("variables")
    w = O.seeded_weights(arch, 7)
    pc = O.synthetic_clouds(2, 4096, 7, "uniform")
    H.make_store(arch, w, dev)
    M = H.pkg("models." + arch)
    with V.variable_scope(H.OUTER), torch.no_grad():
        out = M.forward(torch.from_numpy(pc).to(dev)[:, None], False, params=H.PARAMS).cpu().numpy().reshape(2, 256)
    for ref in (GOLD["g4/%s/desc_f32" % arch], GOLD["g4/%s/desc_f64" % arch]):
        err = np.linalg.norm(out - ref, axis=1).max()
        assert err <= 1e-4, "descriptor L2 error %.3e" % err


def test_g5_training_step(dev):
    TR, ops = H.pkg("training"), H.pkg("ops")
    w = O.seeded_weights("epc-net", 9, mode="init")
    st = H.make_store("epc-net", w, dev)
    params = dict(H.PARAMS, ARCH="epc-net", BATCH_NUM_QUERIES=1, DECAY_STEP=200000, BASE_LEARNING_RATE=5e-5,
                  MARGIN_1=0.5, MARGIN_2=0.2)
    ts = TR.TrainStep(params, st, outer=H.OUTER)
    grads = {}
    orig = ops.adam_multi

    def spy(ws, ms, vs, gs, *a):
        for w_, g in zip(ws, gs):
            for k, t_ in st.vars.items():
                if t_.data_ptr() == w_.data_ptr():
                    grads[k[len(H.OUTER) + 1:]] = g.detach().cpu().numpy().copy()
        return orig(ws, ms, vs, gs, *a)

    ops.adam_multi = spy
    try:
        loss, lr, bn_decay = ts.step(*(torch.from_numpy(a).to(dev) for a in G.g5_inputs()), epoch=0)
    finally:
        ops.adam_multi = orig
    torch.cuda.synchronize()
    assert bn_decay == float(GOLD["g5/bn_decay"]) and lr == pytest.approx(5e-5)
    assert float(loss) == pytest.approx(float(GOLD["g5/loss"]), rel=2e-5)
    desc = np.concatenate([ts.last_aux[k].cpu().numpy() for k in ("q_vec", "pos_vecs", "neg_vecs", "other_neg_vec")], 1)
    assert np.linalg.norm(desc - GOLD["g5/descriptors"], axis=-1).max() <= 1e-4
    for name in G.G5_GRADS:                                 # bars of test_gpu_train_step.py
        ref = GOLD["g5/grad/" + name]
        g = grads[name].reshape(ref.shape)
        assert np.abs(g - ref).max() <= 3e-2 * np.abs(ref).max() + 2e-6, name
        assert np.linalg.norm(g - ref) <= 8e-3 * np.linalg.norm(ref) + 2e-6 * np.sqrt(g.size), name
    for key in (k for k in GOLD.files if k.startswith("g5/ema/")):
        ref = GOLD[key]
        v = st.vars[H.OUTER + "/" + key[len("g5/ema/"):]].detach().cpu().numpy().reshape(ref.shape)
        assert np.abs(v - ref).max() <= 2e-6 + 2e-5 * np.abs(ref).max(), key


def test_g6_retrieval(dev):
    R = H.pkg("retrieval")
    db, q, truth = G.g6_inputs()
    _, idx = R.knn_search(torch.from_numpy(db).to(dev), torch.from_numpy(q).to(dev), 25)
    assert np.array_equal(idx.cpu().numpy(), GOLD["g6/indices"])
    rec, sim, one = R.get_recall(db, q, truth, device=dev)
    assert np.allclose(rec, GOLD["g6/recall"]) and one == pytest.approx(float(GOLD["g6/one_percent_recall"]))
    assert np.allclose(sim, GOLD["g6/top1_similarity"], atol=1e-6)
