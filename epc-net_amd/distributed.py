"""One process per GPU: sharding of descriptor extraction and the one real exchange step of the path.

The reference is single-process / single-GPU; this is new, MI355X-first work:

  * inference BN uses stored statistics, so every cloud's descriptor is independent -> clouds are sharded over the
    ranks with NO data-path collective during extraction (``shard_bounds``; retrieval.evaluate_sharded);
  * retrieval needs every query to see the whole database -> ONE all-gather of the (n_r, 256) f32 descriptor shards
    (``all_gather_rows``; RCCL over xGMI when the backend is "nccl", gloo in the CPU tests).  An Oxford-scale database
    is ~10 MB, i.e. latency-bound: a single fused all-gather of equal-sized (padded) shards, not a ring of small sends;
  * each rank then ranks ITS query shard against the full database locally (``epc_pairwise_topk``) and only the
    (Q_r, 25) int32 neighbour lists travel back to rank 0 for the recall bookkeeping of evaluate.py:476-530.

  * training: data-parallel over TUPLES -- every rank trains one 18-cloud tuple per step with its own
    BatchNorm batch statistics (the reference's statistics are per tuple too, BATCH_NUM_QUERIES = 1), the 4.70 M f32
    gradients (18.8 MB) are averaged with ONE flat all-reduce per bucket before the Adam update
    (``all_reduce_gradients``; xGMI rings are per-link bound, so a few large messages, not 62 small ones), and the
    BatchNorm moving averages are averaged the same way so that the ranks' variables stay identical.

``torch.distributed`` is plumbing here (process group, collectives); all arithmetic is in libepcnet_hip.so.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist


def world() -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


# Single-rank runs skip every collective (there is nothing to exchange).  ``force_collective(True)`` makes the helpers below
# issue their collectives even in a world of ONE rank, provided a process group exists: this is how the RCCL code the
# 8-GPU runs depend on (all_gather_into_tensor / all_reduce / broadcast on device tensors, backend "nccl") is executed and
# tested on a 1-GPU box -- same calls, same buffers, a degenerate ring.
_FORCE_COLLECTIVE = False


def force_collective(on: bool = True) -> bool:
    """Returns the previous setting."""
    global _FORCE_COLLECTIVE
    prev, _FORCE_COLLECTIVE = _FORCE_COLLECTIVE, bool(on)
    return prev


def collectives_active() -> bool:
    """True when the helpers of this module talk to the process group: world > 1, or a world of one with force_collective."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or _FORCE_COLLECTIVE


def backend_name() -> Optional[str]:
    return dist.get_backend() if dist.is_available() and dist.is_initialized() else None


def shard_bounds(n: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous, balanced shards: the first (n % world) ranks hold one extra row.  [start, stop)."""
    q, r = divmod(n, world_size)
    start = rank * q + min(rank, r)
    return start, start + q + (1 if rank < r else 0)


def shard_sizes(n: int, world_size: int) -> List[int]:
    return [shard_bounds(n, r, world_size)[1] - shard_bounds(n, r, world_size)[0] for r in range(world_size)]


def all_gather_rows(local: torch.Tensor, n_total: int) -> torch.Tensor:
    """All-gather row shards laid out by ``shard_bounds`` into the full (n_total, ...) tensor on every rank.
    Shards are padded to the largest shard so that ONE equal-size all-gather moves everything."""
    rank, ws = world()
    if not collectives_active():
        assert local.shape[0] == n_total
        return local
    return all_gather_var_rows(local, shard_sizes(n_total, ws))


def all_gather_var_rows(local: torch.Tensor, sizes: Sequence[int]) -> torch.Tensor:
    """All-gather of row blocks of KNOWN, possibly different sizes (``sizes[r]`` rows on rank r; every rank can compute the
    list, so no size exchange is needed) into their concatenation in rank order.  ONE equal-size ``all_gather_into_tensor``:
    blocks are padded to the largest one; when the blocks are equal nothing is padded or re-packed."""
    rank, ws = world()
    sizes = [int(v) for v in sizes]
    assert len(sizes) == ws and local.shape[0] == sizes[rank], \
        "local block has %d rows, expected %d" % (local.shape[0], sizes[rank])
    if not collectives_active():
        return local
    m = max(sizes)
    tail = tuple(local.shape[1:])
    if m == 0:
        return local
    if sizes[rank] == m:
        send = local.contiguous()
    else:
        send = torch.zeros((m,) + tail, dtype=local.dtype, device=local.device)
        send[: sizes[rank]] = local
    out = torch.empty((ws * m,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, send)
    if min(sizes) == m:
        return out
    return torch.cat([out[r * m: r * m + sizes[r]] for r in range(ws)], dim=0)


def sharded_knn(database_local: torch.Tensor, n_db: int, queries_local: torch.Tensor, n_q: int, k: int,
                search: Callable[[torch.Tensor, torch.Tensor, int], Tuple[torch.Tensor, torch.Tensor]]
                ) -> Optional[np.ndarray]:
    """Database and queries are sharded by rows.  All-gather the database, search this rank's queries locally,
    gather the (Q_r, k) neighbour indices to rank 0.  Returns the full (n_q, k) int32 array on rank 0, None elsewhere."""
    rank, ws = world()
    database = all_gather_rows(database_local, n_db)
    _, idx = search(database, queries_local, k)
    idx = idx.to(torch.int32)
    if not collectives_active():
        return idx.cpu().numpy()
    full = all_gather_rows(idx, n_q)       # (n_q, k) everywhere; tiny (25 ints per query)
    return full.cpu().numpy() if rank == 0 else None


def all_reduce_gradients(tensors: Sequence[torch.Tensor], bucket_bytes: int = 32 << 20, average: bool = True) -> None:
    """In-place mean (or sum) over the ranks of a list of same-dtype tensors, moved as flat buckets of about
    ``bucket_bytes`` (default 32 MB: the whole 18.8-MB EPC-Net gradient is ONE message).  Order and shapes must be the
    same on every rank.  No-op in a single process."""
    rank, ws = world()
    if not collectives_active() or not tensors:
        return
    bucket: List[torch.Tensor] = []
    size = 0

    def flush():
        nonlocal bucket, size
        if not bucket:
            return
        flat = torch.cat([t.reshape(-1) for t in bucket])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
        if average:
            flat /= ws
        o = 0
        for t in bucket:
            n = t.numel()
            t.copy_(flat[o:o + n].reshape(t.shape))
            o += n
        bucket, size = [], 0

    for t in tensors:
        nbytes = t.numel() * t.element_size()
        if bucket and size + nbytes > bucket_bytes:
            flush()
        bucket.append(t)
        size += nbytes
    flush()


def broadcast_tensors(tensors: Sequence[torch.Tensor], src: int = 0) -> None:
    """In-place broadcast of a list of same-dtype tensors from rank ``src`` as ONE flat message.  No-op in a single process."""
    rank, ws = world()
    if not collectives_active() or not tensors:
        return
    flat = torch.cat([t.detach().reshape(-1) for t in tensors])
    dist.broadcast(flat, src=src)
    o = 0
    for t in tensors:
        n = t.numel()
        t.detach().copy_(flat[o:o + n].reshape(t.shape))
        o += n


def all_true(flag: bool, device=None) -> bool:
    """Logical AND of a per-rank condition (one tiny all-reduce): used to make every rank take the same branch -- e.g. to skip
    a training iteration together when ANY rank drew a faulty tuple, so that no rank waits in a collective the others never
    enter.  Single process: the flag itself."""
    rank, ws = world()
    if not collectives_active():
        return bool(flag)
    if device is None and backend_name() == "nccl":
        device = torch.device("cuda", torch.cuda.current_device())       # RCCL moves device memory only
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(int(t.item()))
