"""The RCCL path on ONE GPU.  An 8-GPU node is not ours to launch, so the code the multi-GPU runs
depend on -- backend "nccl" (= RCCL), collectives on DEVICE tensors -- is executed here in a process group of one rank with
the single-rank short-circuits of epc-net_amd/distributed.py switched off (``force_collective``): same calls, same buffers,
a degenerate ring.  Covered: all_gather_rows / all_gather_var_rows, all_reduce_gradients, broadcast_tensors, all_true,
evaluate_sharded end to end (extraction -> all-gather -> rank -> index gather -> recall) against the single-process
evaluate_runs, and data-parallel TrainStep.step -- eager and as the two HIP graphs around the all-reduce -- against the
plain single-process step (bit-identical state: the mean over one rank is the identity)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

import helpers as H
from helpers import O

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.fixture(scope="module")
def rccl():
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    D = H.pkg("distributed")
    prev = D.force_collective(True)
    assert D.collectives_active() and D.backend_name() == "nccl"
    yield D
    D.force_collective(prev)
    dist.barrier()
    dist.destroy_process_group()


def test_collective_helpers_on_device_tensors(rccl):
    D = rccl
    dev = torch.device("cuda:0")
    x = torch.randn((37, 256), device=dev)
    assert torch.equal(D.all_gather_rows(x, 37), x)
    assert torch.equal(D.all_gather_var_rows(x.to(torch.int32), [37]), x.to(torch.int32))
    g = [torch.randn(s, device=dev) for s in [(1, 3, 64), (64,), (1, 256, 1024), (16384, 8), ()]]
    ref = [t.clone() for t in g]
    D.all_reduce_gradients(g, bucket_bytes=300000)          # several buckets; mean over one rank = identity
    assert all(torch.equal(a, b) for a, b in zip(g, ref))
    D.all_reduce_gradients(g, average=False)
    assert all(torch.equal(a, b) for a, b in zip(g, ref))
    D.broadcast_tensors(g, src=0)
    assert all(torch.equal(a, b) for a, b in zip(g, ref))
    assert D.all_true(True) is True and D.all_true(False) is False
    idx = D.sharded_knn(x, 37, x[:5].contiguous(), 5, 25, H.pkg("retrieval").knn_search)
    assert idx.shape == (5, 25) and np.array_equal(idx[:, 0], np.arange(5))


def test_evaluate_sharded_over_rccl_equals_evaluate_runs(rccl):
    """BASELINE.json configs[4] composed: extraction (the HIP pipeline) -> RCCL all-gather -> epc_pairwise_topk -> RCCL index
    gather -> recall, against get_latent_vectors + evaluate_runs."""
    R = H.pkg("retrieval")
    dev = torch.device("cuda:0")
    w = O.seeded_weights("epc-net-l", 0)
    eng, _ = H.make_engine("epc-net-l", w, dev)
    rng = np.random.RandomState(4)
    dbs = [O.synthetic_clouds(40 + 7 * r, 256, 60 + r) for r in range(3)]
    qs = [O.synthetic_clouds(11 + 3 * r, 256, 80 + r) for r in range(3)]
    truth = {(m, n): [list(rng.choice(len(dbs[m]), size=rng.randint(0, 4), replace=False)) for _ in range(len(qs[n]))]
             for m in range(3) for n in range(3)}
    tm = {}
    res = R.evaluate_sharded(lambda c: eng.forward(torch.as_tensor(c, dtype=torch.float32).to(dev)), dbs, qs,
                             lambda m, n: truth[(m, n)], device=dev, batch_size=16, timings=tm)
    dbv = [R.get_latent_vectors(eng, c, batch_size=16, device=dev) for c in dbs]
    qv = [R.get_latent_vectors(eng, c, batch_size=16, device=dev) for c in qs]
    ref = R.evaluate_runs(dbv, qv, lambda m, n: truth[(m, n)], device=dev)
    for m in range(3):
        assert np.array_equal(res["database_vectors"][m], dbv[m]) and np.array_equal(res["query_vectors"][m], qv[m])
    assert np.array_equal(res["ave_recall"], ref["ave_recall"])
    assert res["ave_one_percent_recall"] == ref["ave_one_percent_recall"]
    assert res["average_similarity"] == ref["average_similarity"]
    assert tm["all_gather_bytes"] == (sum(map(len, dbs)) + sum(map(len, qs))) * 1024


def _tuple(seed, dev, n=256):
    g = torch.Generator().manual_seed(seed)
    mk = lambda p: (torch.rand((1, p, n, 3), generator=g) * 2 - 1).to(dev)
    return mk(1), mk(2), mk(14), mk(1)


def _state(ts, st):
    names = ts.trainable_names()
    return torch.cat([v.detach().reshape(-1) for v in st.vars.values()] + [ts.m[n].reshape(-1) for n in names] +
                     [ts.v[n].reshape(-1) for n in names])


@pytest.mark.parametrize("graph,overlap", [(False, True), (True, True), (False, False), (True, False)])
def test_data_parallel_train_step_through_rccl(rccl, graph, overlap):
    """TrainStep.step with the data-parallel exchange live must leave exactly the state of the plain single-process step.
    overlap (the default): the backward is cut at the backbone's output, the head's gradients (17.9 of 18.8 MB) start their RCCL
    all-reduce asynchronously and travel under the backbone's backward, the rest follows -- eager, and as THREE HIP graphs around the
    two collectives; without it: one flat all-reduce after the whole backward (two graphs)."""
    D = rccl
    dev = torch.device("cuda:0")
    TR, V = H.pkg("training"), H.pkg("variables")
    params = dict(H.PARAMS, ARCH="epc-net", BATCH_NUM_QUERIES=1, DP_OVERLAP=overlap)
    states, losses = [], []
    for dp in (False, True):
        D.force_collective(dp)
        st = V.reset_default_store(device=dev, seed=321)
        ts = TR.TrainStep(params, st, outer=H.OUTER)
        ls = []
        for k in range(3):
            loss, _, _ = ts.step(*_tuple(70 + k, dev), epoch=0, graph=graph)
            ls.append(float(loss))
        if dp:
            assert ts._exchange is not None and ts._exchange["flat"].numel() >= 4704832      # the flat RCCL message
            assert ts._exchange["head_end"] >= 4600000 and len(ts._exchange["below"]) == 48     # conv1 .. conv4_b: 12 layers x (W, b, gamma, beta)
            if graph:
                assert ts._graph["dp"] and "graph2" in ts._graph and ts._graph["cut"] == overlap
                assert (ts._graph["graph_mid"] is not None) == overlap
        states.append(_state(ts, st).cpu())
        losses.append(ls)
    D.force_collective(True)
    assert torch.isfinite(states[0]).all()
    # the step is bit-reproducible (ordered split-K, sorted transposed lists) and the exchange adds nothing on top of it (a sum
    # over one rank, a multiplication by 1.0, copies): identical losses and identical state
    assert losses[0] == losses[1], losses
    assert torch.equal(states[0], states[1])
