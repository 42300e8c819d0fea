"""Descriptor extraction over a dataset and the retrieval protocol of the reference's ``evaluate.py``.

  get_latent_vectors   evaluate.py:351-452  (the reference feeds ONE cloud per sess.run, :86-90; here any batch size --
                       in inference every cloud's descriptor is independent, so batching changes nothing but speed)
  knn_search           evaluate.py:463,481  sklearn KDTree(database).query(q, k=25) -> epc_pairwise_topk on the GPU
  get_recall           evaluate.py:455-537  recall@1..25, top-1 % recall, top-1 similarity for one (m, n) run pair
  evaluate_runs        evaluate.py:293-332  average over all ordered pairs m != n
  write_results        evaluate.py:336-348  results.txt
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import lib as L

NUM_NEIGHBORS = 25  # evaluate.py:465


def get_latent_vectors(engine, clouds, batch_size: int = 64, device: Optional[torch.device] = None) -> np.ndarray:
    """clouds (n, N, 3) numpy or tensor (host or device) -> (n, 256) float32 numpy, in order.
    Host-resident input (what the reference feeds, evaluate.py:378-383) is converted and uploaded batch by batch: the
    launches are asynchronous, so the 48 KB per cloud of batch i+1 cross PCIe while batch i is being extracted
    (measured from pageable host memory: 55 k clouds/s vs 61 k device-resident; a pinned double-buffer pipeline with a
    copy stream was tried and is not faster)."""
    dev = device or (clouds.device if torch.is_tensor(clouds) and clouds.is_cuda
                     else torch.device("cuda", torch.cuda.current_device()))
    n = int(clouds.shape[0])
    out = torch.empty((n, 256), dtype=torch.float32, device=dev)
    for i in range(0, n, batch_size):
        chunk = torch.as_tensor(clouds[i:i + batch_size], dtype=torch.float32).to(dev, non_blocking=True)
        engine.forward(chunk, out=out[i:i + chunk.shape[0]])
    if n:
        torch.cuda.synchronize(dev)
    return out.cpu().numpy()


def knn_search(database: torch.Tensor, queries: torch.Tensor, k: int = NUM_NEIGHBORS) -> Tuple[torch.Tensor, torch.Tensor]:
    """Exact k nearest database rows per query (Euclidean), ties -> lower index.  Returns (dist (Q,k), idx (Q,k))."""
    L.require_gpu()
    database = database.contiguous().float()
    queries = queries.contiguous().float()
    q, d = int(queries.shape[0]), int(database.shape[0])
    k = min(k, d)
    idx = torch.empty((q, k), dtype=torch.int32, device=queries.device)
    dist = torch.empty((q, k), dtype=torch.float32, device=queries.device)
    dim = int(database.shape[1])
    if q == 0:
        return dist, idx
    if dim % 8 == 0 and k <= 56:
        # pairwise matrix on the matrix pipe + exact re-rank of the candidates (epc_pairwise_topk_ws): any database size
        need = L.lib().epc_pairwise_topk_workspace_bytes(d, q)
        ws = torch.empty(need, dtype=torch.uint8, device=queries.device)
        L.check(L.lib().epc_pairwise_topk_ws(L.ptr(database), d, L.ptr(queries), q, dim, k, L.ptr(idx), L.ptr(dist),
                                             ws.data_ptr(), need, L.current_stream()))
        ws.record_stream(torch.cuda.current_stream(queries.device))
    else:
        L.check(L.lib().epc_pairwise_topk(L.ptr(database), d, L.ptr(queries), q, dim, k, L.ptr(idx), L.ptr(dist),
                                          L.current_stream()))
    return dist, idx


def recall_from_indices(indices: np.ndarray, database_output: np.ndarray, queries_output: np.ndarray,
                        true_neighbors: Sequence[Sequence[int]], num_neighbors: int = NUM_NEIGHBORS):
    """The bookkeeping of evaluate.py:467-537 given each query's sorted neighbour indices."""
    recall = [0] * num_neighbors
    top1_similarity_score: List[float] = []
    one_percent_retrieved = 0
    threshold = max(int(round(len(database_output) / 100.0)), 1)          # :470 (Python banker's rounding)
    num_evaluated = 0
    for i in range(len(queries_output)):
        truth = true_neighbors[i]
        if len(truth) == 0:                                               # :477-478
            continue
        num_evaluated += 1
        ind = indices[i]
        truth_set = set(int(t) for t in truth)
        for j in range(len(ind)):                                         # :512-521
            if int(ind[j]) in truth_set:
                if j == 0:
                    top1_similarity_score.append(float(np.dot(queries_output[i], database_output[ind[j]])))
                recall[j] += 1
                break
        if len(set(int(v) for v in ind[0:threshold]).intersection(truth_set)) > 0:   # :526-527
            one_percent_retrieved += 1
    if num_evaluated == 0:
        raise ZeroDivisionError("no query of this pair has a true neighbour (evaluate.py:529 divides by zero too)")
    one_percent_recall = (one_percent_retrieved / float(num_evaluated)) * 100
    recall_pct = (np.cumsum(recall) / float(num_evaluated)) * 100
    return recall_pct, top1_similarity_score, one_percent_recall


def get_recall(database_output: np.ndarray, queries_output: np.ndarray, true_neighbors: Sequence[Sequence[int]],
               device: Optional[torch.device] = None, search: Optional[Callable] = None):
    """evaluate.py:455-537 for one (database run m, query run n) pair; ``true_neighbors[i]`` = QUERY_SETS[n][i][m]."""
    dev = device or torch.device("cuda", torch.cuda.current_device())
    search = search or knn_search
    _, idx = search(torch.as_tensor(database_output, dtype=torch.float32, device=dev),
                    torch.as_tensor(queries_output, dtype=torch.float32, device=dev), NUM_NEIGHBORS)
    return recall_from_indices(idx.cpu().numpy(), database_output, queries_output, true_neighbors)


def evaluate_runs(database_vectors: Sequence[np.ndarray], query_vectors: Sequence[np.ndarray],
                  truth: Callable[[int, int], Sequence[Sequence[int]]], device=None, search=None) -> Dict[str, object]:
    """evaluate.py:305-332: loop over ordered pairs m != n; ``truth(m, n)[i]`` = QUERY_SETS[n][i][m]."""
    recall = np.zeros(NUM_NEIGHBORS)
    count = 0
    similarity: List[float] = []
    one_percent: List[float] = []
    dev = device or torch.device("cuda", torch.cuda.current_device())
    search = search or knn_search
    # One search per DATABASE run: the queries of all the other runs against it in a single call (a query's neighbours do
    # not depend on which other queries ride along) -- 23 searches instead of 506 at Oxford scale, every array uploaded once.
    # The pairs are then booked in the reference's order (m outer, n inner).
    q_dev = [torch.as_tensor(np.ascontiguousarray(q), dtype=torch.float32, device=dev) for q in query_vectors]
    for m in range(len(database_vectors)):
        others = [n for n in range(len(query_vectors)) if n != m]
        if not others:
            continue
        db = torch.as_tensor(np.ascontiguousarray(database_vectors[m]), dtype=torch.float32, device=dev)
        _, idx = search(db, torch.cat([q_dev[n] for n in others], dim=0), NUM_NEIGHBORS)
        idx = idx.cpu().numpy()
        at = 0
        for n in others:
            nq = len(query_vectors[n])
            pair_recall, pair_sim, pair_opr = recall_from_indices(idx[at:at + nq], database_vectors[m], query_vectors[n],
                                                                  truth(m, n))
            at += nq
            recall += np.array(pair_recall)
            count += 1
            one_percent.append(pair_opr)
            similarity.extend(pair_sim)
    return {"ave_recall": recall / count, "average_similarity": float(np.mean(similarity)) if similarity else float("nan"),
            "ave_one_percent_recall": float(np.mean(one_percent))}


def write_results(path: str, res: Dict[str, object], arch: str = "epc-net") -> None:
    """evaluate.py:336-348 (appends, like the reference)."""
    with open(path, "a") as output:
        output.write(arch)
        output.write("\n\n")
        output.write("Average Recall @N:\n")
        output.write(str(res["ave_recall"]))
        output.write("\n\n")
        output.write("Average Similarity:\n")
        output.write(str(res["average_similarity"]))
        output.write("\n\n")
        output.write("Average Top 1% Recall:\n")
        output.write(str(res["ave_one_percent_recall"]))
        output.write("\n\n")
