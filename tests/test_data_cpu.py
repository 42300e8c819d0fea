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


def test_evaluate_runs_books_every_ordered_pair_like_the_reference():
    """retrieval.evaluate_runs (evaluate.py:305-332) searches ONCE per database run -- the queries of all other runs in one call --
    and must still book every ordered pair (m, n), m != n, in the reference's order with the reference's numbers: checked on the
    CPU with an injected brute-force search against the oracle's pair-by-pair protocol (ragged run sizes, queries without a true
    neighbour, a run pair whose queries all lack one is not generated here: the reference divides by zero there)."""
    R = H.pkg("retrieval")
    rng = np.random.RandomState(5)
    sizes_db, sizes_q = [31, 47, 40, 25], [12, 9, 15, 11]
    unit = lambda n: (lambda v: (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32))(rng.randn(n, 256))
    dbs, qs = [unit(n) for n in sizes_db], [unit(n) for n in sizes_q]
    truth = {(m, n): [sorted(rng.choice(sizes_db[m], size=rng.randint(0, 4), replace=False).tolist()) if i else [0]
                      for i in range(sizes_q[n])] for m in range(4) for n in range(4)}
    calls = []

    def search(db, q, k):
        calls.append((tuple(db.shape), tuple(q.shape)))
        dist, idx = O.knn_bruteforce(db.numpy(), q.numpy(), min(k, db.shape[0]))
        return torch.from_numpy(dist.astype(np.float32)), torch.from_numpy(idx.astype(np.int32))

    res = R.evaluate_runs(dbs, qs, lambda m, n: truth[(m, n)], device=torch.device("cpu"), search=search)
    assert len(calls) == 4 and all(c[1][0] == sum(sizes_q) - sizes_q[m] for m, c in enumerate(calls))
    rec = np.zeros(25); opr = []; sim = []
    for m in range(4):
        for n in range(4):
            if m != n:
                r_, s_, o_ = O.get_recall(dbs[m], qs[n], truth[(m, n)])
                rec += r_; opr.append(o_); sim.extend(s_)
    assert np.allclose(res["ave_recall"], rec / 12) and np.isclose(res["ave_one_percent_recall"], np.mean(opr))
    assert np.isclose(res["average_similarity"], np.mean(sim), atol=1e-6)
