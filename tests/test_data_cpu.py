"""CPU tests of the host data path (utils/loading_pointclouds.py mirror)."""
import os
import pickle
import random

import numpy as np
import torch

import helpers as H
from helpers import O


def test_bin_loader_and_pickles(tmp_path):
    LP = H.pkg("utils.loading_pointclouds")
    rng = np.random.RandomState(0)
    pcs = rng.uniform(-1, 1, size=(3, 4096, 3))
    for i in range(3):
        pcs[i].astype(np.float64).tofile(tmp_path / ("c%d.bin" % i))
    (tmp_path / "bad.bin").write_bytes(np.zeros(100, dtype=np.float64).tobytes())
    got = LP.load_pc_file("c1.bin", str(tmp_path))
    assert got.shape == (4096, 3) and got.dtype == np.float64 and np.array_equal(got, pcs[1])
    assert np.array_equal(LP.load_pc_file("bad.bin", str(tmp_path)), np.zeros((4096, 3)))     # reference: zeros + message
    q = {i: {"query": "c%d.bin" % i, "positives": [(i + 1) % 3], "negatives": [(i + 2) % 3]} for i in range(3)}
    with open(tmp_path / "q.pickle", "wb") as f:
        pickle.dump(q, f)
    assert LP.get_queries_dict(str(tmp_path / "q.pickle")) == q
    data = LP.load_pc_data(q, str(tmp_path))
    assert data.shape == (3, 4096, 3) and data.dtype == np.float32 and np.allclose(data, pcs, atol=1e-7)


def test_query_tuple_semantics():
    LP = H.pkg("utils.loading_pointclouds")
    random.seed(0)
    T = 30
    data = np.arange(T, dtype=np.float32)[:, None, None] * np.ones((1, 8, 3), dtype=np.float32)
    Q = {i: {"query": "x", "positives": [(i + d) % T for d in (1, 2, 3)], "negatives": [(i + d) % T for d in range(8, 25)]}
         for i in range(T)}
    q, pos, neg, oth = LP.get_query_tuple(5, Q[5], 2, 14, Q, hard_neg=[], other_neg=True, data=data)
    assert q[0, 0] == 5 and pos.shape[0] == 2 and neg.shape[0] == 14
    assert set(pos[:, 0, 0].astype(int)) <= {6, 7, 8} and set(neg[:, 0, 0].astype(int)) <= {(5 + d) % T for d in range(8, 25)}
    forbidden = set(Q[5]["positives"])
    for n_ in neg[:, 0, 0].astype(int):
        forbidden |= set(Q[n_]["positives"])
    assert int(oth[0, 0]) not in forbidden
    hard = [13, 14]
    _, _, neg2 = LP.get_query_tuple(5, Q[5], 2, 14, Q, hard_neg=hard, other_neg=False, data=data)
    ids = neg2[:, 0, 0].astype(int).tolist()
    assert ids[:2] == hard and len(set(ids)) == 14                       # hard negatives first, no duplicates


def test_hard_negative_selection_matches_kdtree():
    from sklearn.neighbors import KDTree
    LP = H.pkg("utils.loading_pointclouds")
    rng = np.random.RandomState(1)
    lat = rng.randn(500, 256).astype(np.float32)
    lat /= np.linalg.norm(lat, axis=1, keepdims=True)
    negs = rng.choice(500, size=200, replace=False).tolist()
    qv = lat[negs[17]] + 0.01 * rng.randn(256).astype(np.float32)

    def cpu_search(db, q, k):
        d, i = O.knn_bruteforce(db.numpy(), q.numpy(), k)
        return torch.from_numpy(d), torch.from_numpy(i)

    got = LP.get_random_hard_negatives(qv, negs, 10, lat, search=cpu_search)
    _, ind = KDTree(lat[negs]).query(np.array([qv]), k=10)                  # train.py:865-867
    assert got == np.squeeze(np.array(negs)[ind[0]]).tolist()
