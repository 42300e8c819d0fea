"""Descriptor extraction over a dataset and the retrieval protocol of the reference's ``evaluate.py``.

  get_latent_vectors   evaluate.py:351-452  (the reference feeds ONE cloud per sess.run, :86-90; here any batch size --
                       in inference every cloud's descriptor is independent, so batching changes nothing but speed)
  knn_search           evaluate.py:463,481  sklearn KDTree(database).query(q, k=25) -> epc_pairwise_topk on the GPU
  get_recall           evaluate.py:455-537  recall@1..25, top-1 % recall, top-1 similarity for one (m, n) run pair
  evaluate_runs        evaluate.py:293-332  average over all ordered pairs m != n
  evaluate_sharded     evaluate.py:293-332 end to end over the ranks of a process group (BASELINE.json configs[4]):
                       extraction of every run sharded over the ranks -> ONE all-gather of the descriptors -> every rank
                       ranks its share of the queries -> ONE index gather -> rank 0 books the recall
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
    out = latent_vectors_device(engine, clouds, batch_size, device)
    if out.shape[0]:
        torch.cuda.synchronize(out.device)
    return out.cpu().numpy()


def latent_vectors_device(engine, clouds, batch_size: int = 64, device: Optional[torch.device] = None) -> torch.Tensor:
    """``get_latent_vectors`` without the trip to the host: (n, 256) float32 on the device, asynchronous."""
    dev = device or (clouds.device if torch.is_tensor(clouds) and clouds.is_cuda
                     else torch.device("cuda", torch.cuda.current_device()))
    n = int(clouds.shape[0])
    out = torch.empty((n, 256), dtype=torch.float32, device=dev)
    for i in range(0, n, batch_size):
        chunk = torch.as_tensor(clouds[i:i + batch_size], dtype=torch.float32).to(dev, non_blocking=True)
        engine.forward(chunk, out=out[i:i + chunk.shape[0]])
    return out


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
    """The bookkeeping of evaluate.py:467-537 given each query's sorted neighbour indices -- the reference's per-query Python
    loop (set membership per neighbour, :512-527) as array operations with the same results: at Oxford scale the loop form
    costs as much host time as extracting the descriptors costs the GPU.  ``recall_from_indices_loop`` keeps the
    reference-shaped loop; the tests hold the two to each other and to the oracle.  An index < 0 = "no such neighbour"
    (fewer than k finite database rows, e.g. a NaN query descriptor): never a hit."""
    import itertools
    nq, nd = len(queries_output), len(database_output)
    threshold = max(int(round(nd / 100.0)), 1)                            # :470 (Python banker's rounding)
    lens = np.fromiter((len(t) for t in true_neighbors[:nq]), dtype=np.int64, count=nq)
    evaluated = lens > 0                                                  # :477-478
    num_evaluated = int(evaluated.sum())
    if num_evaluated == 0:
        raise ZeroDivisionError("no query of this pair has a true neighbour (evaluate.py:529 divides by zero too)")
    ind = np.asarray(indices)[:nq, :num_neighbors].astype(np.int64)
    is_true = np.zeros((nq, nd + 1), dtype=bool)                          # column nd: where the invalid indices point
    cols = np.fromiter(itertools.chain.from_iterable(true_neighbors[:nq]), dtype=np.int64, count=int(lens.sum()))
    rows = np.repeat(np.arange(nq), lens)
    ok = (cols >= 0) & (cols < nd)
    is_true[rows[ok], cols[ok]] = True
    hits = is_true[np.arange(nq)[:, None], np.where((ind < 0) | (ind >= nd), nd, ind)]      # (nq, k)
    hits &= evaluated[:, None]
    any_hit = hits.any(axis=1)
    first = hits.argmax(axis=1)                                           # :512-521: the FIRST true neighbour in the list
    recall = np.bincount(first[any_hit], minlength=num_neighbors)[:num_neighbors]
    top1_similarity_score = [float(np.dot(queries_output[i], database_output[ind[i, 0]]))  # :515-517, in query order
                             for i in np.nonzero(any_hit & (first == 0))[0]]
    one_percent_retrieved = int(hits[:, :threshold].any(axis=1).sum())    # :526-527
    one_percent_recall = (one_percent_retrieved / float(num_evaluated)) * 100
    recall_pct = (np.cumsum(recall) / float(num_evaluated)) * 100
    return recall_pct, top1_similarity_score, one_percent_recall


def recall_from_indices_loop(indices: np.ndarray, database_output: np.ndarray, queries_output: np.ndarray,
                             true_neighbors: Sequence[Sequence[int]], num_neighbors: int = NUM_NEIGHBORS):
    """evaluate.py:467-537 in the reference's own loop shape (what ``recall_from_indices`` must reproduce)."""
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
            if int(ind[j]) < 0:       # no such neighbour: fewer than k finite database rows (e.g. a NaN query descriptor)
                continue
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


class _Sliced:
    """Rows [a, b) of the concatenation of several (n_i, N, 3) arrays, without building the concatenation."""

    def __init__(self, parts: Sequence, a: int, b: int):
        self.parts, self.a, self.b = parts, a, b
        self.starts = np.concatenate([[0], np.cumsum([len(p) for p in parts])])

    def chunks(self, batch: int):
        at = self.a
        while at < self.b:
            r = int(np.searchsorted(self.starts, at, side="right") - 1)
            stop = min(self.b, int(self.starts[r + 1]), at + batch)
            yield self.parts[r][at - int(self.starts[r]): stop - int(self.starts[r])]
            at = stop


def evaluate_sharded(extract: Callable, database_sets: Sequence, query_sets: Sequence,
                     truth: Optional[Callable[[int, int], Sequence[Sequence[int]]]], device=None, search=None,
                     batch_size: int = 64, timings: Optional[dict] = None) -> Optional[Dict[str, object]]:
    """The reference's whole ``evaluate()`` (evaluate.py:293-332: extract every run, rank every ordered pair of runs,
    average) as ONE rank-aware flow (BASELINE.json configs[4]).  Call it on every rank of the process group
    (or in a single process) with the same arguments:

      1. the clouds of ALL runs -- ``database_sets[m]`` (n_m, N, 3) then ``query_sets[n]`` -- form one list that is sharded
         contiguously over the ranks; ``extract(chunk (b, N, 3)) -> (b, 256) device tensor`` runs on this rank's shard in
         chunks of ``batch_size`` (e.g. ``lambda x: engine.forward(torch.as_tensor(x).to(dev))``): no collective,
         inference-mode descriptors are independent;
      2. ONE fused all-gather of the descriptor shards (not one per run): every rank holds all database and query descriptors
         (Oxford scale: 23 x (400 + 120) x 1 KB = 12 MB);
      3. for every database run m the queries of all the other runs are ranked against it; each rank searches ITS contiguous
         share of those queries (``search(db, q, 25)``, default ``knn_search`` = epc_pairwise_topk);
      4. ONE gather of the (q, 25) int32 neighbour lists of all runs;
      5. rank 0 books recall@N / top-1 % / similarity per ordered pair exactly as ``evaluate_runs`` does and returns the
         result dictionary (plus the descriptors under "database_vectors" / "query_vectors"); the other ranks return None.

    ``truth(m, n)[i]`` = QUERY_SETS[n][i][m]; only rank 0 calls it.  The result equals the single-process
    ``evaluate_runs`` on the same descriptors bit for bit: a query's neighbour list does not depend on which rank ranks it.
    ``timings`` (optional dict) receives wall-clock seconds of the phases on this rank, each closed by a device
    synchronisation (extract, all_gather, rank, index_gather, book)."""
    import time
    from . import distributed as D
    rank, ws = D.world()
    search = search or knn_search
    n_dbs = [len(x) for x in database_sets]
    n_qs = [len(x) for x in query_sets]
    parts = list(database_sets) + list(query_sets)
    total = sum(n_dbs) + sum(n_qs)
    a, b = D.shard_bounds(total, rank, ws)
    on_gpu = torch.cuda.is_available()

    def tick():
        if on_gpu:
            torch.cuda.synchronize()
        return time.perf_counter()

    t0 = tick()
    local = [extract(chunk) for chunk in _Sliced(parts, a, b).chunks(batch_size)]
    if local:
        local = torch.cat(local, dim=0) if len(local) > 1 else local[0]
    else:
        dev = device or (torch.device("cuda", torch.cuda.current_device()) if on_gpu else torch.device("cpu"))
        local = torch.empty((0, 256), dtype=torch.float32, device=dev)
    local = local.float().contiguous()
    dev = local.device
    t1 = tick()
    vectors = D.all_gather_rows(local, total)                                   # (total, 256) on every rank
    t2 = tick()
    starts = np.concatenate([[0], np.cumsum(n_dbs + n_qs)]).astype(np.int64)
    db_vec = [vectors[starts[m]: starts[m + 1]] for m in range(len(n_dbs))]
    q_vec = [vectors[starts[len(n_dbs) + n]: starts[len(n_dbs) + n + 1]] for n in range(len(n_qs))]

    # every rank's share of the queries ranked against database run m: rows [qa, qb) of cat(q_vec[n] for n != m)
    mine, counts = [], []                       # counts[m] = rows of run m's query list (same on every rank)
    for m in range(len(n_dbs)):
        others = [n for n in range(len(n_qs)) if n != m]
        nq = sum(n_qs[n] for n in others)
        counts.append(nq)
        if nq == 0 or n_dbs[m] == 0:
            continue
        qa, qb = D.shard_bounds(nq, rank, ws)
        if qb > qa:
            q = torch.cat([q_vec[n] for n in others], dim=0)[qa:qb]
            _, idx = search(db_vec[m], q, NUM_NEIGHBORS)
            idx = idx.to(torch.int32)
            if idx.shape[1] < NUM_NEIGHBORS:                                     # a database run with fewer than 25 rows
                idx = torch.cat([idx, torch.full((idx.shape[0], NUM_NEIGHBORS - idx.shape[1]), -1, dtype=torch.int32,
                                                 device=idx.device)], dim=1)
            mine.append(idx)
    width = NUM_NEIGHBORS
    mine = torch.cat(mine, dim=0) if mine else torch.empty((0, width), dtype=torch.int32, device=dev)
    t3 = tick()
    live = [m for m in range(len(n_dbs)) if counts[m] > 0 and n_dbs[m] > 0]
    per_rank = [sum(D.shard_bounds(counts[m], r, ws)[1] - D.shard_bounds(counts[m], r, ws)[0] for m in live)
                for r in range(ws)]
    gathered = D.all_gather_var_rows(mine, per_rank)                             # rank-major, run-minor
    t4 = tick()
    if timings is not None:
        timings.update(extract=t1 - t0, all_gather=t2 - t1, rank=t3 - t2, index_gather=t4 - t3,
                       all_gather_bytes=int(vectors.numel() * 4), clouds_local=b - a, clouds_total=total)
    if rank != 0:
        return None
    gathered = gathered.cpu().numpy()
    vec_np = vectors.cpu().numpy()
    db_np = [vec_np[starts[m]: starts[m + 1]] for m in range(len(n_dbs))]
    q_np = [vec_np[starts[len(n_dbs) + n]: starts[len(n_dbs) + n + 1]] for n in range(len(n_qs))]
    # un-shard: lists[m] (counts[m], 25) = concatenation over the ranks of their slices for run m
    offs = np.concatenate([[0], np.cumsum(per_rank)]).astype(np.int64)
    lists = {}
    cursor = [int(o) for o in offs[:-1]]
    for m in live:
        rows = []
        for r in range(ws):
            qa, qb = D.shard_bounds(counts[m], r, ws)
            rows.append(gathered[cursor[r]: cursor[r] + (qb - qa)])
            cursor[r] += qb - qa
        lists[m] = np.concatenate(rows, axis=0)
    recall = np.zeros(NUM_NEIGHBORS)
    count = 0
    similarity: List[float] = []
    one_percent: List[float] = []
    for m in range(len(n_dbs)):                                                  # evaluate.py:305-319 order
        at = 0
        for n in range(len(n_qs)):
            if n == m:
                continue
            nq = n_qs[n]
            if m not in lists:
                raise ValueError("database run %d is empty" % m)
            k_eff = min(NUM_NEIGHBORS, n_dbs[m])
            pair_recall, pair_sim, pair_opr = recall_from_indices(lists[m][at:at + nq, :k_eff], db_np[m], q_np[n], truth(m, n))
            at += nq
            recall += np.array(pair_recall)
            count += 1
            one_percent.append(pair_opr)
            similarity.extend(pair_sim)
    if timings is not None:
        timings["book"] = time.perf_counter() - t4
    return {"ave_recall": recall / count, "average_similarity": float(np.mean(similarity)) if similarity else float("nan"),
            "ave_one_percent_recall": float(np.mean(one_percent)), "database_vectors": db_np, "query_vectors": q_np}


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
