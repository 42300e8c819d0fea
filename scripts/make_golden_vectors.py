#!/usr/bin/env python3
"""Generate tests/golden/vectors.npz: the golden vectors G1-G6 of SURVEY.md 8c.

The reference cannot run here (TensorFlow 1.12 is absent), so these vectors come from the CPU restatement in oracle/
("PARITY UNPINNED", see oracle/README.md).  What they pin is the restatement itself: every later edit of the oracle, of
the seeded generators or of the HIP path is checked against the same committed numbers (tests/test_golden_cpu.py on the
CPU, tests/test_gpu_golden.py on the GPU).  Inputs are not stored: they are regenerated from the seeds through
numpy's RandomState (a frozen stream) by the functions named beside each group.

    G1  kNN graph: idx / kth / count for seeded clouds incl. tie, lattice and all-zero clouds   (tf_util.py:647-666)
    G2  one ProxyConv block at N=256: neighbour mean and block output                            (epc-net.py:66-81)
    G3  G_VLAD on random unit features (B=2, N=256): V, grouped projection, gated output         (loupe.py:233-333)
    G4  full EPC-Net and EPC-Net-L at N=4096, B=2: descriptors, fp32 and the fp64 shadow         (epc-net.py:29-157)
    G5  training-mode forward on an 18-cloud tuple at N=256: loss, batch statistics, new moving averages, three
        gradients (fp64 autograd, cross-checked by central differences here)                     (train.py:251-277)
    G6  retrieval bookkeeping on seeded descriptors: recall@1..25, top-1 similarity, one-percent recall
                                                                                                 (evaluate.py:455-537)

Run from the repo root:  python scripts/make_golden_vectors.py   (about two minutes on 8 cores).
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import epcnet_oracle as O  # noqa: E402

KNN_CASES = [("uniform", 64, 0), ("uniform", 256, 1), ("uniform", 4096, 2), ("uniform", 4096, 3), ("lidar", 4096, 0),
             ("dup", 256, 0), ("lattice", 512, 0), ("zeros", 128, 0)]
KNN_CAP = 32            # lists longer than this (ties) are stored through their length and checksums only
G5_GRADS = ["fastdgcnn/conv1/weights", "fastdgcnn/conv3_a/bn/gamma", "VLAD/cluster_weights"]


def knn_summary(pc):
    """Per row: count, kth, the first KNN_CAP indices (ascending, -1 padded), sum and xor of all indices."""
    kth, lists = O.knn_lists(pc)
    B, n = kth.shape
    cnt = np.zeros((B, n), np.int32)
    idx = np.full((B, n, KNN_CAP), -1, np.int32)
    isum = np.zeros((B, n), np.int64)
    ixor = np.zeros((B, n), np.int32)
    for b in range(B):
        for i in range(n):
            l = lists[b][i]
            cnt[b, i] = len(l)
            m = min(len(l), KNN_CAP)
            idx[b, i, :m] = l[:m]
            isum[b, i] = int(l.astype(np.int64).sum())
            ixor[b, i] = int(np.bitwise_xor.reduce(l))
    return {"kth": kth, "cnt": cnt, "idx": idx, "isum": isum, "ixor": ixor}


def g1(out):
    for kind, n, seed in KNN_CASES:
        pc = O.synthetic_clouds(2, n, seed, kind)
        for k, v in knn_summary(pc).items():
            if n == 4096 and k == "idx":
                v = v[:, :, :20].astype(np.int16)      # first 20 of each row (a tie beyond that shows in cnt / isum / ixor); keeps the file small
            out["g1/%s_%d_%d/%s" % (kind, n, seed, k)] = v


def g2(out):
    w = O.seeded_weights("epc-net", 5)
    pc = O.synthetic_clouds(2, 256, 5, "lidar")
    _, st = O.forward(pc[:, None], w, arch="epc-net")
    out["g2/mean1"] = st.taps["mean1"]
    out["g2/block1"] = st.taps["block1"]


def g3_features():
    rng = np.random.RandomState(33)
    f = rng.randn(2 * 256, 1024).astype(np.float32)
    f = np.maximum(f, 0)                              # conv5 ends in a ReLU: features are non-negative
    return O.l2_normalize(f, 1)


def g3(out):
    w = O.seeded_weights("epc-net", 6)
    st = O.State(w, np.float32)
    y = O.g_vlad_forward(st, g3_features(), 256, 4, False, gating=True)
    out["g3/vlad_raw"] = st.taps["vlad_raw"]
    out["g3/vlad_hidden"] = st.taps["vlad_hidden"]
    out["g3/gated"] = y
    # "fold the groups first" identity: sum_g BN(v_g W) == s * ((sum_g v_g) W) + G t
    s = w["VLAD/bn/gamma"] / np.sqrt(w["VLAD/bn/moving_variance"] + np.float32(1e-3))
    t = w["VLAD/bn/beta"] - w["VLAD/bn/moving_mean"] * s
    folded = (st.taps["vlad_flat"].astype(np.float64).reshape(2, 4, 16384).sum(1)
              @ w["VLAD/hidden1_weights"].astype(np.float64)) * s + 4 * t
    assert np.abs(folded - st.taps["vlad_hidden"]).max() < 1e-5


def g4(out):
    for arch in ("epc-net", "epc-net-l"):
        w = O.seeded_weights(arch, 7)
        pc = O.synthetic_clouds(2, 4096, 7, "uniform")
        mask = O.pairwise_distance_mask(pc)
        a, _ = O.forward(pc[:, None], w, arch=arch, mask=mask)
        b, _ = O.forward(pc[:, None], w, arch=arch, dtype=np.float64, mask=mask)
        out["g4/%s/desc_f32" % arch] = a.reshape(2, 256)
        out["g4/%s/desc_f64" % arch] = b.reshape(2, 256)


def g5_inputs():
    pc = O.synthetic_clouds(18, 256, 11, "uniform")
    return pc[None, 0:1], pc[None, 1:3], pc[None, 3:17], pc[None, 17:18]


def g5(out):
    import torch
    import epcnet_oracle_torch as OT
    torch.set_num_threads(8)
    w = O.seeded_weights("epc-net", 9, mode="init")
    q, p, n, o = g5_inputs()
    r = OT.train_step(w, q, p, n, o, step=0, epoch=0)
    out["g5/loss"] = np.float64(r["loss"])
    out["g5/bn_decay"] = np.float64(r["bn_decay"])
    out["g5/descriptors"] = r["descriptors"]
    for name in G5_GRADS:
        out["g5/grad/" + name] = r["grads"][name]
    m, v = O.ema_names("fastdgcnn/conv5")
    for name in (m, v, "VLAD/bn/moving_variance", "VLAD/cluster_bn/moving_mean"):
        out["g5/ema/" + name] = r["new_weights"][name]
    # fp64 numpy forward agrees with the torch graph; batch statistics come from it
    vecs = np.concatenate([q, p, n, o], 1)
    desc, st = O.forward(vecs, w, is_training=True, bn_decay=r["bn_decay"], dtype=np.float64)
    assert np.abs(desc - r["descriptors"]).max() < 1e-9
    out["g5/batch_mean/conv5"], out["g5/batch_var/conv5"] = st.batch_stats["fastdgcnn/conv5/bn"]
    # central differences on a few coordinates of each stored gradient (fp64)
    rng = np.random.RandomState(0)

    def loss_of(weights):
        d, _ = O.forward(vecs, weights, is_training=True, bn_decay=r["bn_decay"], dtype=np.float64)
        return O.lazy_quadruplet_loss(d[:, 0:1], d[:, 1:3], d[:, 3:17], d[:, 17:18], 0.5, 0.2)

    assert abs(loss_of({k: v.astype(np.float64) for k, v in w.items()}) - r["loss"]) < 1e-9
    for name in G5_GRADS:
        g = r["grads"][name]
        for _ in range(2):
            pos = tuple(rng.randint(0, s) for s in g.shape)
            h = 1e-7                                 # the loss has a ReLU kink every few 1e-6 along a conv1 weight
            wp = {k: v.astype(np.float64) for k, v in w.items()}
            wm = {k: v.astype(np.float64) for k, v in w.items()}
            wp[name][pos] += h
            wm[name][pos] -= h
            fd = (loss_of(wp) - loss_of(wm)) / (2 * h)
            assert abs(fd - g[pos]) <= 2e-3 * abs(fd) + 1e-7, (name, pos, fd, g[pos])


def g6_inputs():
    rng = np.random.RandomState(21)
    db = rng.randn(500, 256).astype(np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    src = rng.randint(0, 500, 50)
    q = db[src] + 0.25 * rng.randn(50, 256).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    truth = []
    for i in range(50):
        k = rng.randint(0, 4)
        t = list(rng.choice(500, size=k, replace=False))
        if k and rng.rand() < 0.7:
            t[0] = int(src[i])
        truth.append([int(x) for x in t])
    return db, q.astype(np.float32), truth


def g6(out):
    db, q, truth = g6_inputs()
    rec, sim, one = O.get_recall(db, q, truth)
    _, ind = O.knn_bruteforce(db, q, 25)
    out["g6/recall"] = np.asarray(rec, np.float64)
    out["g6/top1_similarity"] = np.asarray(sim, np.float64)
    out["g6/one_percent_recall"] = np.float64(one)
    out["g6/indices"] = ind.astype(np.int32)


def main():
    out = {}
    for f in (g1, g2, g3, g4, g5, g6):
        f(out)
        print(f.__name__, "done", flush=True)
    path = os.path.join(ROOT, "tests", "golden", "vectors.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "arrays")


if __name__ == "__main__":
    main()
