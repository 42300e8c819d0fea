"""Two ranks on the device (both on cuda:0 of the one-GPU box, backend gloo: RCCL refuses two ranks on one GPU): the data-parallel
training step of training.TrainStep with a REAL second rank -- different tuples per rank, gradients and moving statistics averaged
-- eager and as HIP graphs around the collectives, with and without the overlapped exchange (train.py:251-277 under SURVEY 8e's
data-parallel reading).  What a world of one cannot show: that the ranks issue their collectives in the same order and do not wait
for each other inside a captured graph, that both end with the same weights, and that graph replay equals the eager step bit for
bit when the mean really mixes two gradients.  Each case runs in two spawned processes; results come back through files."""
import os
import socket
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, graph, overlap, out_dir):
    sys.path.insert(0, HERE)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch.distributed as dist
    import helpers as H
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    TR, V = H.pkg("training"), H.pkg("variables")
    # (two processes on ONE GPU: the persistent forward chain wants its workgroups co-resident, and two such launches from two processes
    # could hold each other's CUs until a spin budget ran out -- real ranks have a GPU each; here the launch chain)
    H.pkg("ops").CHAIN_PERSIST_FWD = False
    params = dict(H.PARAMS, ARCH="epc-net", BATCH_NUM_QUERIES=1, DP_OVERLAP=overlap)
    st = V.reset_default_store(device=dev, seed=321)
    ts = TR.TrainStep(params, st, outer=H.OUTER)

    def tup(seed, n=256):
        g = torch.Generator().manual_seed(seed)
        mk = lambda p: (torch.rand((1, p, n, 3), generator=g) * 2 - 1).to(dev)
        return mk(1), mk(2), mk(14), mk(1)
    losses = []
    for k in range(3):
        loss, _, _ = ts.step(*tup(500 + 10 * k + rank), epoch=0, graph=graph)
        losses.append(float(loss))
    names = ts.trainable_names()
    state = torch.cat([v.detach().reshape(-1) for v in st.vars.values()] + [ts.m[n].reshape(-1) for n in names] +
                      [ts.v[n].reshape(-1) for n in names]).cpu()
    torch.save({"state": state, "losses": losses, "dp": bool(getattr(ts, "_exchange", None) is not None)},
               os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def _run(graph, overlap, tmp_path):
    import torch.multiprocessing as mp
    out = tmp_path / ("g%d_o%d" % (graph, overlap))
    out.mkdir()
    mp.spawn(_worker, args=(2, _free_port(), graph, overlap, str(out)), nprocs=2, join=True)
    return [torch.load(str(out / ("r%d.pt" % r))) for r in range(2)]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("overlap", [True, False])
def test_two_ranks_agree_and_graph_equals_eager(tmp_path, overlap):
    assert torch.cuda.is_available()
    eager = _run(False, overlap, tmp_path)
    graph = _run(True, overlap, tmp_path)
    for res in (eager, graph):
        assert res[0]["dp"] and res[1]["dp"]
        assert torch.isfinite(res[0]["state"]).all()
        assert torch.equal(res[0]["state"], res[1]["state"]), "the ranks' weights / Adam moments diverged"
        assert res[0]["losses"] != res[1]["losses"]          # (each rank saw its own tuples)
    # replaying the captured step == launching it eagerly, bit for bit, with two gradients in the mean
    assert torch.equal(eager[0]["state"], graph[0]["state"])
    assert eager[0]["losses"] == graph[0]["losses"] and eager[1]["losses"] == graph[1]["losses"]
