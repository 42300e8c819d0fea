"""world_size-2 gloo tests (CPU) of the sharding / all-gather / index-gather arithmetic of the multi-GPU path.
The neighbour search itself is injected (oracle brute force) -- on a GPU box it is epc_pairwise_topk."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H
from helpers import O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _oracle_search(db, q, k):
    d, i = O.knn_bruteforce(db.numpy(), q.numpy(), k)
    return torch.from_numpy(d.astype(np.float32)), torch.from_numpy(i.astype(np.int32))


def _worker(rank, ws, port, n_db, n_q, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    D = H.pkg("distributed")
    rng = np.random.RandomState(0)
    db = rng.randn(n_db, 256).astype(np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = rng.randn(n_q, 256).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    a, b = D.shard_bounds(n_db, rank, ws)
    qa, qb = D.shard_bounds(n_q, rank, ws)
    full = D.all_gather_rows(torch.from_numpy(db[a:b]), n_db)
    assert torch.equal(full, torch.from_numpy(db)), "all-gathered database differs from the unsharded one"
    idx = D.sharded_knn(torch.from_numpy(db[a:b]), n_db, torch.from_numpy(q[qa:qb]), n_q, 25, _oracle_search)
    if rank == 0:
        _, ref = O.knn_bruteforce(db, q, 25)
        assert np.array_equal(idx, ref)
        ret.put("ok")
    else:
        assert idx is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_db,n_q", [(101, 37), (64, 64), (3, 5)])
def test_sharded_retrieval_world2_gloo(n_db, n_q):
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    n_db_eff = max(n_db, 25)  # k = 25 neighbours need >= 25 database rows
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_db_eff, n_q, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == "ok"


def test_shard_bounds_cover_exactly():
    D = H.pkg("distributed")
    for n in (0, 1, 7, 64, 1000):
        for ws in (1, 2, 3, 8):
            spans = [D.shard_bounds(n, r, ws) for r in range(ws)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(ws - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == D.shard_sizes(n, ws)


def test_recall_bookkeeping_matches_oracle():
    R = H.pkg("retrieval")
    rng = np.random.RandomState(1)
    db = rng.randn(250, 256).astype(np.float32); db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = db[rng.randint(0, 250, 40)] + 0.2 * rng.randn(40, 256).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    truth = [list(rng.choice(250, size=rng.randint(0, 4), replace=False)) for _ in range(40)]
    _, ind = O.knn_bruteforce(db, q, 25)
    a = R.recall_from_indices(ind, db, q, truth)
    b = O.get_recall(db, q, truth)
    assert np.array_equal(a[0], b[0]) and a[2] == b[2] and np.allclose(a[1], b[1])
    # the array form against the reference-shaped loop: identical, also with invalid (-1) slots, short lists, a 1 % threshold
    # above 1 and truth entries outside the database
    for trial in range(6):
        nd, nq, k = [250, 1000, 30, 400, 26, 333][trial], 40 + trial, [25, 25, 25, 7, 25, 25][trial]
        db2 = rng.randn(nd, 256).astype(np.float32)
        q2 = rng.randn(nq, 256).astype(np.float32)
        _, ind2 = O.knn_bruteforce(db2, q2, min(k, nd))
        ind2 = ind2.astype(np.int32)
        ind2[rng.rand(*ind2.shape) < 0.05] = -1
        truth2 = [list(rng.choice(nd + 5, size=rng.randint(0, 5), replace=False)) for _ in range(nq)]
        truth2[0] = [int(ind2[0, 0])] if ind2[0, 0] >= 0 else [int(ind2[0][ind2[0] >= 0][0])]
        va = R.recall_from_indices(ind2, db2, q2, truth2)
        vb = R.recall_from_indices_loop(ind2, db2, q2, truth2)
        assert np.array_equal(va[0], vb[0]) and va[1] == vb[1] and va[2] == vb[2], trial


def _grad_worker(rank, ws, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    D = H.pkg("distributed")
    shapes = [(1, 3, 64), (64,), (1, 256, 1024), (1024, 64), (16384, 8), ()]
    g = torch.Generator().manual_seed(7)
    base = [torch.randn(s, generator=g) for s in shapes]            # identical on both ranks
    mine = [b * (rank + 1) for b in base]                             # rank r holds (r+1) * base
    D.all_reduce_gradients(mine, bucket_bytes=300000)                 # several buckets, one tensor larger than a bucket
    for m, b in zip(mine, base):
        assert torch.allclose(m, b * (sum(range(1, ws + 1)) / ws), rtol=1e-6, atol=1e-7)
    summed = [b * (rank + 1) for b in base]
    D.all_reduce_gradients(summed, average=False)
    for m, b in zip(summed, base):
        assert torch.allclose(m, b * sum(range(1, ws + 1)), rtol=1e-6, atol=1e-7)
    if rank == 0:
        ret.put("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_averaging_world2_gloo():
    """Data-parallel training: bucketed flat all-reduce leaves the per-rank mean in every tensor."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == "ok"


def _sync_worker(rank, ws, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    D, TR, V = H.pkg("distributed"), H.pkg("training"), H.pkg("variables")
    # un-seeded stores, as the reference's training is (MANUAL_SEED is a dead key): the ranks start from DIFFERENT inits
    st = V.reset_default_store(device="cpu", seed=1000 + rank)
    ts = TR.TrainStep(dict(H.PARAMS, ARCH="epc-net-l"), st, outer=H.OUTER, arch="epc-net-l")
    ts.global_step = 5 * rank
    ts._ensure_built(256)                                  # declares the variables, creates the Adam slots, broadcasts
    flat = torch.cat([v.detach().reshape(-1) for v in st.vars.values()] +
                     [ts.m[n].reshape(-1) for n in ts.trainable_names()])
    both = [torch.zeros_like(flat) for _ in range(ws)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1]), "variables / Adam moments differ between the ranks after the initial sync"
    assert ts.global_step == 0                             # rank 0's step count
    # gradient + moving-statistics averaging as TrainStep applies it (the Adam kernel then sees identical inputs everywhere)
    g = torch.Generator().manual_seed(3 + rank)
    grads = [torch.randn(st.vars[n].shape, generator=g) for n in ts.trainable_names()]
    with torch.no_grad():
        for k, v in st.vars.items():
            if k not in st.trainable:
                v.add_(float(rank + 1))                    # pretend this step updated the moving statistics differently
    grads = ts._average_over_ranks(grads)                  # (returns the averaged gradients: views of the exchange buffer)
    flat = torch.cat([x.reshape(-1) for x in grads] + [v.detach().reshape(-1) for v in st.vars.values()])
    both = [torch.zeros_like(flat) for _ in range(ws)]
    dist.all_gather(both, flat)
    assert torch.equal(both[0], both[1])
    # the skip vote: one rank with a faulty tuple makes every rank skip
    assert D.all_true(True) is True
    assert D.all_true(rank != 1) is False
    if rank == 0:
        ret.put("ok")
    dist.barrier()
    dist.destroy_process_group()


def test_train_state_sync_and_skip_vote_world2_gloo():
    """data-parallel TrainStep ranks must start from rank 0's weights / Adam moments / step count, average
    gradients AND moving statistics, and skip an iteration together."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sync_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == "ok"


# ---- configs[4] as one composed, rank-aware flow: evaluate_sharded == single-process evaluate_runs --------------------------
def _synthetic_runs(seed=5, n_runs=3, n_points=32):
    """Three runs of (database clouds, query clouds) + truth sets shaped like QUERY_SETS[n][i][m] (evaluate.py:476)."""
    rng = np.random.RandomState(seed)
    n_db = [41, 29, 37][:n_runs]
    n_q = [13, 17, 9][:n_runs]
    dbs = [rng.uniform(-1, 1, (n, n_points, 3)).astype(np.float32) for n in n_db]
    # queries = noisy copies of database clouds of OTHER runs, so that true neighbours are actually retrieved
    qs, truth = [], {}
    for n in range(n_runs):
        q = rng.uniform(-1, 1, (n_q[n], n_points, 3)).astype(np.float32)
        for m in range(n_runs):
            if m == n:
                continue
            t = []
            for i in range(n_q[n]):
                k = rng.randint(0, 4)                     # some queries have no true neighbour in run m (skipped, :477)
                pick = list(rng.choice(n_db[m], size=k, replace=False))
                if k and rng.rand() < 0.7:
                    q[i] = dbs[m][pick[0]] + 0.01 * rng.randn(n_points, 3).astype(np.float32)
                t.append(pick)
            truth[(m, n)] = t
        qs.append(q)
    return dbs, qs, truth


def _cpu_extract(chunk):
    """Stand-in for engine.forward in the CPU tests: a fixed per-cloud map to a unit 256-vector (chunking-independent)."""
    x = torch.as_tensor(np.asarray(chunk), dtype=torch.float32).reshape(len(chunk), -1)
    g = torch.Generator().manual_seed(99)
    w = torch.randn((x.shape[1], 256), generator=g)
    v = torch.stack([torch.mv(w.t(), torch.tanh(r)) for r in x]) if len(x) else torch.empty((0, 256))   # row by row: bit-equal in any chunking
    return v / v.norm(dim=1, keepdim=True)


def _single_process_reference(dbs, qs, truth):
    R = H.pkg("retrieval")
    dbv = [_cpu_extract(d).numpy() for d in dbs]
    qv = [_cpu_extract(q).numpy() for q in qs]
    return R.evaluate_runs(dbv, qv, lambda m, n: truth[(m, n)], device=torch.device("cpu"), search=_oracle_search)


def _eval_worker(rank, ws, port, ret, force):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    D, R = H.pkg("distributed"), H.pkg("retrieval")
    D.force_collective(force)
    dbs, qs, truth = _synthetic_runs()
    tm = {}
    # every rank books ITS share of the queries (device-side, integer counters) and the ranks all-reduce the counters: every
    # rank needs the truth sets, every rank gets the result.  Rank 1 passes the packed form (what repeated evaluations use).
    tr = (lambda m, n: truth[(m, n)]) if rank == 0 else R.pack_truth(lambda m, n: truth[(m, n)], list(map(len, dbs)), list(map(len, qs)))
    res = R.evaluate_sharded(_cpu_extract, dbs, qs, tr, device=torch.device("cpu"), search=_oracle_search, batch_size=8,
                             timings=tm)
    assert tm["clouds_total"] == sum(map(len, dbs)) + sum(map(len, qs))
    assert tm["reduce_bytes"] == (len(dbs) * len(qs) * 28 + 1) * 8
    ref = _single_process_reference(dbs, qs, truth)
    assert np.array_equal(res["ave_recall"], ref["ave_recall"]), (res["ave_recall"], ref["ave_recall"])
    assert res["ave_one_percent_recall"] == ref["ave_one_percent_recall"]
    assert res["average_similarity"] == ref["average_similarity"]
    assert res["ave_recall"][-1] > 20.0            # the synthetic truth really is retrieved: not a vacuous comparison
    if rank == 0:
        for m, d in enumerate(dbs):
            assert np.array_equal(res["database_vectors"][m], _cpu_extract(d).numpy())
        ret.put("ok")
    else:
        assert "database_vectors" not in res
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("ws,force", [(2, False), (1, True)])
def test_evaluate_sharded_equals_single_rank(ws, force):
    """extraction sharded over the ranks -> one all-gather -> every rank ranks its query share -> one
    all-reduce of the per-pair integer counters each rank booked for ITS queries (no host loop, no serial
    tail on rank 0); the numbers must equal the single-process evaluate_runs bit for bit ON EVERY RANK.
    (1, True) = a world of one with the collectives forced on: the code path the 1-GPU nccl test takes."""
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, ws, port, ret, force)) for r in range(ws)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == "ok"


def test_evaluate_sharded_without_process_group():
    R = H.pkg("retrieval")
    dbs, qs, truth = _synthetic_runs(seed=8)
    res = R.evaluate_sharded(_cpu_extract, dbs, qs, lambda m, n: truth[(m, n)], device=torch.device("cpu"),
                             search=_oracle_search, batch_size=16)
    ref = _single_process_reference(dbs, qs, truth)
    assert np.array_equal(res["ave_recall"], ref["ave_recall"])
    assert res["ave_one_percent_recall"] == ref["ave_one_percent_recall"]
    assert res["average_similarity"] == ref["average_similarity"]


def test_sharded_counters_equal_single_rank_counters_and_the_loop_form():
    """The device-side bookkeeping (retrieval._book_rows) for any split of the queries over ranks: the summed integer counters are
    those of one rank, and they are the reference-shaped loop's numbers (recall_from_indices_loop) pair by pair -- also with
    invalid (-1) slots, a database run shorter than 25 rows, truth entries outside the database and a 1 % threshold above 1."""
    R = H.pkg("retrieval")
    rng = np.random.RandomState(11)
    n_db, n_q = [260, 19, 57], [14, 9, 21]
    unit = lambda n: (lambda v: (v / np.linalg.norm(v, axis=1, keepdims=True)).astype(np.float32))(rng.randn(n, 256))
    dbv, qv = [torch.from_numpy(unit(n)) for n in n_db], [torch.from_numpy(unit(n)) for n in n_q]
    truth = {(m, n): [list(rng.choice(n_db[m] + 4, size=rng.randint(0, 5), replace=False)) if i else [1]
                      for i in range(n_q[n])] for m in range(3) for n in range(3)}

    def search(db, q, k):
        d, i = O.knn_bruteforce(db.numpy(), q.numpy(), min(k, db.shape[0]))
        i = i.astype(np.int32)
        i[i % 17 == 3] = -1     # "no such neighbour" slots (a function of the entry, not of the row's position in the call)
        return torch.from_numpy(d.astype(np.float32)), torch.from_numpy(i)

    packed = R.pack_truth(lambda m, n: truth[(m, n)], n_db, n_q)
    one, sim1 = R._rank_and_book(dbv, qv, packed, search, 0, 1, torch.device("cpu"))
    for ws in (2, 3, 5):
        parts = [R._rank_and_book(dbv, qv, packed, search, r, ws, torch.device("cpu")) for r in range(ws)]
        assert torch.equal(sum(p[0] for p in parts), one) and int(sum(p[1] for p in parts)) == int(sim1)
    K = R.NUM_NEIGHBORS
    sims = []
    for m in range(3):
        for n in range(3):
            if m == n:
                continue
            _, ind = search(dbv[m], qv[n], K)
            rec, sim, opr = R.recall_from_indices_loop(ind.numpy(), dbv[m].numpy(), qv[n].numpy(), truth[(m, n)])
            c = one[m * 3 + n].numpy()
            ne = c[K + 1]
            assert ne == sum(1 for t in truth[(m, n)] if len(t))
            assert np.array_equal(np.cumsum(c[:K]) / float(ne) * 100, rec) and c[K] / float(ne) * 100 == opr
            assert c[K + 2] == len(sim)
            sims.extend(sim)
    assert abs(int(sim1) / float(1 << R.SIM_FIXED_BITS) - float(np.sum(sims))) < 1e-5


def test_all_gather_var_rows_world2_gloo():
    ctx = mp.get_context("spawn")
    ret = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_var_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert ret.get(timeout=5) == "ok"


def _var_worker(rank, ws, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    D = H.pkg("distributed")
    for sizes in ([3, 5], [4, 4], [0, 6], [0, 0]):
        full = torch.arange(sum(sizes) * 2, dtype=torch.int32).reshape(-1, 2)
        a = sum(sizes[:rank])
        got = D.all_gather_var_rows(full[a:a + sizes[rank]], sizes)
        assert torch.equal(got, full), (sizes, got)
    if rank == 0:
        ret.put("ok")
    dist.barrier()
    dist.destroy_process_group()
