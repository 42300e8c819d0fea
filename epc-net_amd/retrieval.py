"""Descriptor extraction over a dataset and the retrieval protocol of the reference's ``evaluate.py``.

  get_latent_vectors   evaluate.py:351-452  (the reference feeds ONE cloud per sess.run, :86-90; here any batch size --
                       in inference every cloud's descriptor is independent, so batching changes nothing but speed)
  knn_search           evaluate.py:463,481  sklearn KDTree(database).query(q, k=25) -> epc_pairwise_topk on the GPU
  get_recall           evaluate.py:455-537  recall@1..25, top-1 % recall, top-1 similarity for one (m, n) run pair
  evaluate_runs        evaluate.py:293-332  average over all ordered pairs m != n
  evaluate_sharded     evaluate.py:293-332 end to end over the ranks of a process group (BASELINE.json configs[4]):
                       extraction of every run sharded over the ranks -> ONE all-gather of the descriptors -> every rank
                       ranks AND books its share of the queries on its device -> ONE all-reduce of integer counters
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
    # (nq, k) neighbour lists against the (nq, Lmax) padded truth sets: memory O(nq * k * Lmax), not O(nq * nd)
    width = max(int(lens.max()), 1)
    pad = np.full((nq, width), -2, dtype=np.int64)
    cols = np.fromiter(itertools.chain.from_iterable(true_neighbors[:nq]), dtype=np.int64, count=int(lens.sum()))
    rows = np.repeat(np.arange(nq), lens)
    slot = np.arange(int(lens.sum())) - np.repeat(np.cumsum(lens) - lens, lens)
    ok = (cols >= 0) & (cols < nd)
    pad[rows[ok], slot[ok]] = cols[ok]
    valid = np.where((ind < 0) | (ind >= nd), -1, ind)
    hits = np.zeros(valid.shape, dtype=bool)
    step = max(1, (1 << 24) // max(1, valid.shape[1] * width))
    for a in range(0, nq, step):
        hits[a:a + step] = (valid[a:a + step, :, None] == pad[a:a + step, None, :]).any(-1)
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


class PackedTruth:
    """The truth sets of every ordered run pair (QUERY_SETS[n][i][m], evaluate.py:476) in the array form the device-side
    bookkeeping reads, built ONCE per dataset (the sets are as static as the pickles they come from): for each database run m
    the rows are the queries of all the other runs in run order (the order ``evaluate_runs`` ranks them in), each row its true
    neighbours padded with -2 (never a neighbour index).  Entries outside [0, len(database m)) can never be retrieved and are
    dropped.  ``to(device)`` uploads the padded tables; after that an evaluation does no host work per query."""

    def __init__(self, truth: Callable[[int, int], Sequence[Sequence[int]]], n_dbs: Sequence[int], n_qs: Sequence[int]):
        self.n_dbs, self.n_qs = [int(v) for v in n_dbs], [int(v) for v in n_qs]
        self.padded: List[Optional[torch.Tensor]] = []      # per m: (rows_m, Lmax_m) int32
        self.lens: List[Optional[torch.Tensor]] = []        # per m: (rows_m,) int32 -- the ORIGINAL list lengths (:477 tests them)
        for m in range(len(self.n_dbs)):
            rows: List[Sequence[int]] = []
            for n in range(len(self.n_qs)):
                if n == m:
                    continue
                t = truth(m, n)
                if len(t) < self.n_qs[n]:
                    raise ValueError("truth(%d, %d) has %d rows for %d queries" % (m, n, len(t), self.n_qs[n]))
                rows.extend(t[: self.n_qs[n]])
            lens = np.fromiter((len(r) for r in rows), dtype=np.int64, count=len(rows))
            width = max(int(lens.max()) if len(rows) else 0, 1)
            pad = np.full((len(rows), width), -2, dtype=np.int32)
            if len(rows) and lens.sum():
                import itertools
                flat = np.fromiter(itertools.chain.from_iterable(rows), dtype=np.int64, count=int(lens.sum()))
                r_ix = np.repeat(np.arange(len(rows)), lens)
                c_ix = np.arange(int(lens.sum())) - np.repeat(np.cumsum(lens) - lens, lens)
                ok = (flat >= 0) & (flat < self.n_dbs[m])
                pad[r_ix[ok], c_ix[ok]] = flat[ok].astype(np.int32)
            self.padded.append(torch.from_numpy(pad))
            self.lens.append(torch.from_numpy(lens.astype(np.int32)))
        self.device = torch.device("cpu")

    def to(self, device) -> "PackedTruth":
        device = torch.device(device)
        if device != self.device:
            self.padded = [t.to(device) for t in self.padded]
            self.lens = [t.to(device) for t in self.lens]
            self.device = device
        return self


def pack_truth(truth, n_dbs: Sequence[int], n_qs: Sequence[int]) -> PackedTruth:
    """``truth`` (callable ``truth(m, n)[i]`` = QUERY_SETS[n][i][m], or an already packed table) -> PackedTruth."""
    if isinstance(truth, PackedTruth):
        if truth.n_dbs != [int(v) for v in n_dbs] or truth.n_qs != [int(v) for v in n_qs]:
            raise ValueError("packed truth was built for other run sizes")
        return truth
    return PackedTruth(truth, n_dbs, n_qs)


SIM_FIXED_BITS = 40      # top-1 similarities are summed as round(score * 2^40) in int64: exact, order- and shard-independent
_BOOK_CHUNK = 1 << 25    # elements of the (rows, k, Lmax) comparison per chunk


def _book_rows(counters: torch.Tensor, sim: torch.Tensor, idx: torch.Tensor, q_rows: torch.Tensor, db: torch.Tensor,
               truth_pad: torch.Tensor, truth_len: torch.Tensor, pair_of_row: torch.Tensor, threshold: int) -> None:
    """evaluate.py:467-537 for a block of query rows ranked against ONE database run, on the device the neighbour lists are on:
    ``counters[pair, 0:K]`` += first-hit rank histogram (:512-521), ``[pair, K]`` += hit within the first ``threshold`` (:526-527),
    ``[pair, K + 1]`` += evaluated queries (:477-478), ``[pair, K + 2]`` += top-1 hits; ``sim[0]`` += their similarities
    (q . db[top-1], :515-517) in fixed point.  Integer scatter-adds only: the result does not depend on how the rows are split
    over calls or ranks.  An index < 0 = "no such neighbour" (never a hit)."""
    K = NUM_NEIGHBORS
    r = int(idx.shape[0])
    if r == 0:
        return
    step = max(1, _BOOK_CHUNK // max(1, int(idx.shape[1]) * int(truth_pad.shape[1])))
    for a in range(0, r, step):
        b = min(r, a + step)
        ind = idx[a:b].to(torch.int64)
        hits = (ind[:, :, None] == truth_pad[a:b].to(torch.int64)[:, None, :]).any(-1)         # (rows, k)
        evaluated = truth_len[a:b] > 0
        hits &= evaluated[:, None]
        any_hit = hits.any(1)
        first = hits.to(torch.uint8).argmax(1)                                                 # the FIRST true neighbour
        pair = pair_of_row[a:b].to(torch.int64)
        flat = counters.view(-1)
        width = counters.shape[1]
        one = torch.ones_like(pair)
        flat.scatter_add_(0, pair * width + first, any_hit.to(torch.int64))
        flat.scatter_add_(0, pair * width + K, hits[:, :threshold].any(1).to(torch.int64))
        flat.scatter_add_(0, pair * width + K + 1, evaluated.to(torch.int64))
        top1 = any_hit & (first == 0)
        flat.scatter_add_(0, pair * width + K + 2, top1.to(torch.int64))
        score = (q_rows[a:b] * db[ind[:, 0].clamp(min=0)]).sum(1)                              # float32, row by row
        fixed = torch.round(score.double() * float(1 << SIM_FIXED_BITS)).to(torch.int64)
        sim += torch.where(top1, fixed, torch.zeros_like(fixed)).sum().reshape(1)
        del one


def _finish_counters(counters: np.ndarray, sim_fixed: int, n_runs_db: int, n_runs_q: int) -> Dict[str, object]:
    """evaluate.py:305-332 from the integer counters of every ordered pair, in the reference's order (m outer, n inner)."""
    K = NUM_NEIGHBORS
    order = [m * n_runs_q + n for m in range(n_runs_db) for n in range(n_runs_q) if n != m]
    if not order:
        raise ZeroDivisionError("no ordered pair of runs to evaluate")
    c = counters[order]
    ne = c[:, K + 1].astype(np.float64)
    if (c[:, K + 1] == 0).any():
        raise ZeroDivisionError("no query of a run pair has a true neighbour (evaluate.py:529 divides by zero too)")
    pair_recall = (np.cumsum(c[:, :K], axis=1) / ne[:, None]) * 100          # :529, per pair
    one_percent = (c[:, K] / ne) * 100                                        # :528
    ave_recall = np.cumsum(pair_recall, axis=0)[-1] / len(order)              # `recall += pair_recall` in pair order, :315
    hits1 = int(c[:, K + 2].sum())
    similarity = (sim_fixed / float(1 << SIM_FIXED_BITS)) / hits1 if hits1 else float("nan")
    return {"ave_recall": ave_recall, "average_similarity": float(similarity),
            "ave_one_percent_recall": float(np.mean(one_percent))}


def _rank_and_book(db_vec: Sequence[torch.Tensor], q_vec: Sequence[torch.Tensor], packed: PackedTruth, search, rank: int,
                   ws: int, dev) -> Tuple[torch.Tensor, torch.Tensor]:
    """For every database run m: this rank's contiguous share of the other runs' queries is ranked against it (``search``) and
    booked (``_book_rows``).  Returns this rank's (pairs, K + 3) int64 counters and the fixed-point similarity sum."""
    n_dbs, n_qs = [int(x.shape[0]) for x in db_vec], [int(x.shape[0]) for x in q_vec]
    K = NUM_NEIGHBORS
    counters = torch.zeros((len(n_dbs) * len(n_qs), K + 3), dtype=torch.int64, device=dev)
    sim = torch.zeros(1, dtype=torch.int64, device=dev)
    for m in range(len(n_dbs)):
        others = [n for n in range(len(n_qs)) if n != m]
        nq = sum(n_qs[n] for n in others)
        if nq == 0:
            continue
        if n_dbs[m] == 0:
            raise ValueError("database run %d is empty" % m)
        qa, qb = _shard(nq, rank, ws)
        if qb <= qa:
            continue
        q = torch.cat([q_vec[n] for n in others], dim=0)[qa:qb]
        _, idx = search(db_vec[m], q, NUM_NEIGHBORS)
        idx = idx.to(device=dev, dtype=torch.int32)
        if idx.shape[1] < NUM_NEIGHBORS:                                     # a database run with fewer than 25 rows
            idx = torch.cat([idx, torch.full((idx.shape[0], NUM_NEIGHBORS - idx.shape[1]), -1, dtype=torch.int32,
                                             device=dev)], dim=1)
        pair_of_row = torch.cat([torch.full((n_qs[n],), m * len(n_qs) + n, dtype=torch.int32, device=dev)
                                 for n in others])[qa:qb]
        threshold = max(int(round(n_dbs[m] / 100.0)), 1)                     # :470 (Python banker's rounding)
        _book_rows(counters, sim, idx, q, db_vec[m], packed.padded[m][qa:qb], packed.lens[m][qa:qb], pair_of_row, threshold)
    return counters, sim


def _shard(n: int, rank: int, ws: int) -> Tuple[int, int]:
    from . import distributed as D
    return D.shard_bounds(n, rank, ws)


def evaluate_runs(database_vectors: Sequence[np.ndarray], query_vectors: Sequence[np.ndarray], truth, device=None,
                  search=None) -> Dict[str, object]:
    """evaluate.py:305-332: every ordered pair m != n; ``truth(m, n)[i]`` = QUERY_SETS[n][i][m] (or a ``PackedTruth``).
    One search per DATABASE run -- the queries of all the other runs against it in a single call (a query's neighbours do not
    depend on which other queries ride along): 23 searches instead of 506 at Oxford scale -- and the bookkeeping of
    evaluate.py:467-537 on the device the lists are on (``_book_rows``); the pairs are then averaged in the reference's order."""
    dev = device or torch.device("cuda", torch.cuda.current_device())
    search = search or knn_search
    db = [torch.as_tensor(np.ascontiguousarray(v), dtype=torch.float32, device=dev) for v in database_vectors]
    qv = [torch.as_tensor(np.ascontiguousarray(v), dtype=torch.float32, device=dev) for v in query_vectors]
    packed = pack_truth(truth, [len(v) for v in db], [len(v) for v in qv]).to(dev)
    counters, sim = _rank_and_book(db, qv, packed, search, 0, 1, dev)
    return _finish_counters(counters.cpu().numpy(), int(sim.item()), len(db), len(qv))


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


def evaluate_sharded(extract: Callable, database_sets: Sequence, query_sets: Sequence, truth, device=None, search=None,
                     batch_size: int = 64, timings: Optional[dict] = None,
                     return_vectors: bool = True) -> Dict[str, object]:
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
         share of those queries (``search(db, q, 25)``, default ``knn_search`` = epc_pairwise_topk) and BOOKS them on its own
         device (``_book_rows``: hit matrix of the (q, 25) lists against the padded truth sets, first-hit rank, top-1 % test,
         top-1 similarity) into integer counters per ordered pair;
      4. ONE all-reduce of those counters (pairs x 28 int64 + the fixed-point similarity sum: 114 KB at Oxford scale);
      5. every rank finishes the averages of evaluate.py:305-332 from the counters (a few small numpy operations) and returns
         the result dictionary; rank 0's also holds the descriptors under "database_vectors" / "query_vectors" (numpy) unless
         ``return_vectors`` is False.

    ``truth``: ``truth(m, n)[i]`` = QUERY_SETS[n][i][m], or -- what repeated evaluations of one dataset should pass -- the
    ``PackedTruth`` built once from it (``pack_truth``); needed on EVERY rank (each books its own queries).  The result equals
    the single-process ``evaluate_runs`` on the same descriptors bit for bit: a query's neighbour list does not depend on
    which rank ranks it, and the counters are integers.  There is no per-query or per-pair host loop and no serial tail on
    rank 0.  ``timings`` (optional dict) receives wall-clock seconds of the phases on this rank, each closed by a device
    synchronisation (extract, all_gather, rank_book, reduce, finish, to_host; pack_truth when the truth was not packed)."""
    import time
    import torch.distributed as dist
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
    was_packed = isinstance(truth, PackedTruth)
    packed = pack_truth(truth, n_dbs, n_qs).to(dev)
    t2b = tick()
    starts = np.concatenate([[0], np.cumsum(n_dbs + n_qs)]).astype(np.int64)
    db_vec = [vectors[starts[m]: starts[m + 1]] for m in range(len(n_dbs))]
    q_vec = [vectors[starts[len(n_dbs) + n]: starts[len(n_dbs) + n + 1]] for n in range(len(n_qs))]
    counters, sim = _rank_and_book(db_vec, q_vec, packed, search, rank, ws, dev)
    t3 = tick()
    if D.collectives_active():
        flat = torch.cat([counters.view(-1), sim])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        counters, sim = flat[:-1].view(counters.shape), flat[-1:]
    t4 = tick()
    res = _finish_counters(counters.cpu().numpy(), int(sim.item()), len(n_dbs), len(n_qs))
    t5 = time.perf_counter()
    if return_vectors and rank == 0:
        vec_np = vectors.cpu().numpy()
        res["database_vectors"] = [vec_np[starts[m]: starts[m + 1]] for m in range(len(n_dbs))]
        res["query_vectors"] = [vec_np[starts[len(n_dbs) + n]: starts[len(n_dbs) + n + 1]] for n in range(len(n_qs))]
    t6 = time.perf_counter()
    if timings is not None:
        timings.update(extract=t1 - t0, all_gather=t2 - t1, rank_book=t3 - t2b, reduce=t4 - t3, finish=t5 - t4,
                       to_host=t6 - t5, pack_truth=(0.0 if was_packed else t2b - t2),
                       all_gather_bytes=int(vectors.numel() * 4), reduce_bytes=int((counters.numel() + 1) * 8),
                       clouds_local=b - a, clouds_total=total)
    return res


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
