"""The reference's training step (train.py:238-277, 484-495) on the HIP operators.

One ``TrainStep.step(query, positives, negatives, other_neg, epoch)`` = one ``sess.run([train_op, loss, ...])``:
  concat the four inputs on axis 1 -> MODEL.forward(is_training=True, bn_decay) -> split [1, P, N, 1] ->
  lazy_quadruplet_loss(m1, m2) -> gradients -> tf.train.AdamOptimizer(learning_rate) update of the 62 trainable tensors,
  with the BatchNorm moving averages updated by the same run (UPDATE_OPS).  Schedules: get_bn_decay (train.py:138-146),
  get_learning_rate (train.py:154-157).  State follows the checkpoint contract: ``Variable`` (global step), ``beta1_power``,
  ``beta2_power``, ``<var>/Adam`` (m) and ``<var>/Adam_1`` (v)  (tests/golden/ckpt_tables.json).
"""
from __future__ import annotations

import importlib
import math
from typing import Dict, Optional

import torch

from . import ops
from .variables import VariableStore, default_store, variable_scope

BN_INIT_DECAY, BN_DECAY_DECAY_RATE, BN_DECAY_CLIP = 0.5, 0.5, 0.99   # train.py:127-130 (hard-coded there)


REPLAYS_PER_SYNC = 32         # replayed steps between two stream synchronisations (TrainStep._graphed_step: a runtime limit)


def get_bn_decay(batch: int, batch_num_queries: int, decay_step: float) -> float:
    """train.py:138-146: min(0.99, 1 - 0.5 * 0.5 ** floor(batch*BATCH_NUM_QUERIES / DECAY_STEP))."""
    bn_momentum = BN_INIT_DECAY * BN_DECAY_DECAY_RATE ** math.floor(batch * batch_num_queries / float(decay_step))
    return min(BN_DECAY_CLIP, 1 - bn_momentum)


def get_learning_rate(epoch: int, base_learning_rate: float) -> float:
    """train.py:154-157."""
    return max(base_learning_rate * (0.9 ** (epoch // 5)), 0.00001)


def _joined_along_dim1(parts):
    """The tensor of which ``parts`` are the consecutive dim-1 slices -- views of ONE storage with the same strides, each starting
    where the previous one ends -- as a view of that storage, or None.  (train.py:252's concat is then the buffer itself.)"""
    p0 = parts[0]
    try:
        store = p0.untyped_storage().data_ptr()
    except Exception:
        return None
    at = p0.storage_offset()
    for p in parts:
        if p.dim() != p0.dim() or p.dim() < 2 or p.dtype != p0.dtype or p.untyped_storage().data_ptr() != store:
            return None
        if p.stride() != p0.stride() or p.shape[0] != p0.shape[0] or p.shape[2:] != p0.shape[2:] or p.storage_offset() != at:
            return None
        at += int(p.shape[1]) * int(p.stride(1))
    total = sum(int(p.shape[1]) for p in parts)
    joined = p0.as_strided((p0.shape[0], total) + tuple(p0.shape[2:]), p0.stride(), p0.storage_offset())
    if p0.shape[0] > 1 and total * int(p0.stride(1)) > int(p0.stride(0)):
        return None            # (rows of the leading dim would overlap: not slices of one (B, total, ...) tensor)
    return joined


class TrainStep:
    def __init__(self, params: dict, store: Optional[VariableStore] = None, outer: str = "query_triplets",
                 arch: Optional[str] = None):
        self.params = dict(params)
        self.arch = arch or params.get("ARCH", "epc-net")
        self.model = importlib.import_module("epc-net_amd.models." + self.arch)
        self.store = store or default_store()
        self.outer = outer                        # every variable of the trained model lives under this scope
        self.last_aux: Dict[str, torch.Tensor] = {}
        self.global_step = 0                      # tf.Variable(0) `batch`, train.py:246
        self.beta1, self.beta2, self.eps = 0.9, 0.999, 1e-8
        self.m: Dict[str, torch.Tensor] = {}
        self.v: Dict[str, torch.Tensor] = {}
        self._graph = None                        # captured HIP graph(s) of one step (step(..., graph=True))
        self._replays_since_sync = 0              # replayed steps since the last stream synchronisation (REPLAYS_PER_SYNC)
        self._exchange = None                     # flat gradient / statistics buffer of data-parallel steps
        # GEMM arithmetic of the step (ops.set_gemm_precision): "bf16x6" = f32-accurate like the reference's fp32 graph
        # (default); "bf16" = one bf16 value per operand, the arithmetic BASELINE.json configs[2] names
        self.precision = params.get("TRAIN_PRECISION", "bf16x6")
        # data-parallel steps: cut the backward at the backbone's output and exchange the head's gradients under the backbone's backward
        self.overlap_exchange = bool(params.get("DP_OVERLAP", True))

    # -- checkpoint-shaped optimizer state ---------------------------------------------------------------------------
    def optimizer_state(self) -> Dict[str, torch.Tensor]:
        dev = self.store.device
        out = {"Variable": torch.tensor(self.global_step, dtype=torch.int32),
               "beta1_power": torch.tensor(self.beta1 ** (self.global_step + 1), dtype=torch.float32),
               "beta2_power": torch.tensor(self.beta2 ** (self.global_step + 1), dtype=torch.float32)}
        for name in self.trainable_names():
            out[name + "/Adam"] = self.m.get(name, torch.zeros_like(self.store.vars[name]))
            out[name + "/Adam_1"] = self.v.get(name, torch.zeros_like(self.store.vars[name]))
        return out

    def load_optimizer_state(self, state: Dict[str, object]) -> None:
        """Inverse of optimizer_state (resume, train.py:308-315): global step and the Adam moments."""
        self.global_step = int(torch.as_tensor(state["Variable"]).item())
        for name in self.trainable_names():
            for slot, dst in (("/Adam", self.m), ("/Adam_1", self.v)):
                if name + slot in state:
                    val = torch.as_tensor(state[name + slot], dtype=torch.float32).to(self.store.device).reshape(
                        self.store.vars[name].shape)
                    if name in dst:
                        with torch.no_grad():
                            dst[name].copy_(val)      # in place: a captured HIP graph of the step updates THESE buffers
                    else:
                        dst[name] = val.clone()

    def trainable_names(self):
        """The trainable variables under this step's scope (a second model in the same store -- the KD teacher -- is
        left alone, as in kd_train.py where the loss does not depend on it)."""
        pre = self.outer + "/" if self.outer else ""
        return [n for n in self.store.trainable if n.startswith(pre)]

    def _ensure_built(self, num_points: int):
        with variable_scope(self.outer):
            self.model.declare_variables(self.params, num_points)
        fresh = False
        for name in self.trainable_names():
            self.store.vars[name].requires_grad_(True)
            if name not in self.m:
                self.m[name] = torch.zeros_like(self.store.vars[name])
                self.v[name] = torch.zeros_like(self.store.vars[name])
                fresh = True
        if fresh:
            self.sync_initial_state()

    def sync_initial_state(self) -> None:
        """Data-parallel runs: every rank starts from RANK 0's variables, Adam moments and step count (the stores are
        un-seeded by default -- the reference never seeds, MANUAL_SEED is a dead key -- so without this the ranks would
        average gradients of different models).  Called once when the step is first built and after ``restore``; one flat
        broadcast per dtype.  Single process: nothing to do."""
        from . import distributed as D
        if not D.collectives_active():
            return
        pre = self.outer + "/" if self.outer else ""
        tensors = [v for k, v in self.store.vars.items() if k.startswith(pre)]
        tensors += [self.m[n] for n in self.trainable_names()] + [self.v[n] for n in self.trainable_names()]
        with torch.no_grad():
            D.broadcast_tensors(tensors, src=0)
            step = torch.tensor([self.global_step], dtype=torch.int64, device=self.store.device)
            D.broadcast_tensors([step], src=0)
        self.global_step = int(step.item())
        self.store.bump(self.outer or None)

    def step(self, query, positives, negatives, other_neg, epoch: int = 0, graph: bool = False):
        """One training step; returns (loss, learning_rate, bn_decay).  Inputs: (B,1,N,3), (B,P,N,3), (B,Nn,N,3), (B,1,N,3).
        ``graph=True`` records the step into HIP graphs on first use and replays them afterwards -- the step is launch-bound
        from Python otherwise.  Single rank: ONE graph (forward, backward, moving averages, Adam).  Data-parallel ranks
        (SURVEY.md 8e): THREE graphs around two all-reduces -- [forward, head backward, pack the head's gradients] -> RCCL all-reduce
        of those (asynchronous) || [backbone backward, moving averages, pack the rest] -> RCCL all-reduce of the rest -> [mean, unpack
        the statistics, Adam].  The schedule values (learning rate with Adam's bias correction, BN decay) live in device memory
        and are refreshed before every replay, so replays follow train.py:138-157 exactly like eager steps."""
        p = self.params
        B = int(query.shape[0])
        self._ensure_built(int(query.shape[2]))
        bn_decay = get_bn_decay(self.global_step, p.get("BATCH_NUM_QUERIES", B), p.get("DECAY_STEP", 200000))
        lr = get_learning_rate(epoch, p.get("BASE_LEARNING_RATE", 5e-5))
        t = self.global_step + 1
        if graph:
            loss = self._graphed_step(query, positives, negatives, other_neg, lr, bn_decay, t)
        else:
            loss = self._eager_step(query, positives, negatives, other_neg, lr, bn_decay, t)
        self.store.bump(self.outer or None)
        self.global_step += 1
        # (the graphed path returns the replayed graph's static output buffer: clone it, or the next replay overwrites it)
        return (loss.detach().clone() if graph else loss.detach()), lr, bn_decay

    # -- the three phases of a step ------------------------------------------------------------------------------------
    def _forward_backward(self, query, positives, negatives, other_neg, bn_decay, between=None):
        """Forward in training mode, loss, backward, and the moving-average updates of this step (UPDATE_OPS,
        train.py:275-277).  Returns (loss, gradients in trainable_names() order).
        ``between`` (data-parallel steps): the backward is cut at the backbone's output (tf_util.BACKBONE_TAP) -- first the head
        (loss -> VLAD -> conv5: every gradient above the cut, 95 % of the bytes), then ``between(head_names, head_grads)`` is called
        (it packs them and starts their exchange), then the backbone's backward runs under that exchange.  Same kernels in the same
        order as the uncut backward: same bits."""
        from .utils import tf_util
        from .loupe import SLIM_DECAY
        names = self.trainable_names()
        for name in names:
            self.store.vars[name].grad = None
        tf_util.defer_ema_updates()                 # the 34 moving-average updates are applied in one launch below
        tf_util.BACKBONE_TAP = None
        prev = ops.set_gemm_precision(self.precision)
        try:
            loss = self.compute_loss(query, positives, negatives, other_neg, True, bn_decay)
            tap = tf_util.BACKBONE_TAP
            if between is not None and tap is not None and tap.requires_grad:
                # Which variables lie below the cut is a property of the GRAPH: those the backbone's output depends on
                # (_split_at_tap walks the tape once per model structure).  A variable on neither side of the loss gets zeros;
                # one on BOTH sides (shared above and below the cut) cannot be split and raises.
                head, below = self._split_at_tap(loss, tap, names)
                got = torch.autograd.grad(loss, [tap] + [self.store.vars[n] for n in head], allow_unused=True)
                with torch.no_grad():
                    head_grads = [g if g is not None else ops.const_zeros_like(self.store.vars[n]) for n, g in zip(head, got[1:])]
                    between(head, head_grads)
                if got[0] is not None:
                    torch.autograd.backward([tap], [got[0]])
                    # every variable classified "below" must have been reached by the second phase (a mis-split would otherwise
                    # freeze a layer silently: its missing gradient would be exchanged and applied as zeros)
                    missing = [n for n in below if self.store.vars[n].grad is None and not self._bias_before_batchnorm(n)]
                    if missing:
                        raise RuntimeError("backward cut at the backbone's output: no gradient reached %s" % missing)
                for n, g in zip(head, head_grads):
                    self.store.vars[n].grad = g
            else:
                loss.backward()
        except BaseException:
            tf_util._deferred_ema = None            # a failed forward OR backward applies nothing and leaves no deferral armed
            raise
        finally:
            ops.set_gemm_precision(prev)
            tf_util.BACKBONE_TAP = None
        with torch.no_grad():
            grads = []
            for name in names:
                w = self.store.vars[name]
                grads.append(w.grad if w.grad is not None else ops.const_zeros_like(w))
            # data-parallel runs average the moving statistics too: apply the updates before the exchange
            tf_util.flush_ema_updates(SLIM_DECAY, bn_decay if bn_decay is not None else 0.9, scope=self.outer or None)
        return loss, grads

    def _split_at_tap(self, loss, tap, names):
        """(head, below): the trainable names whose gradient is formed above / below the backbone's output ``tap`` in THIS graph.
        below = the variables ``tap`` depends on (reachable from tap.grad_fn); head = the rest.  A variable reachable from the
        loss WITHOUT passing through ``tap`` and also from ``tap`` itself would need both phases' contributions added: not a
        structure any shipped model has, refused instead of silently mis-split.  Cached per set of names (the walk costs a few
        hundred nodes); ``_below_tap`` answers from the cache for the exchange layout."""
        def leaves_under(root_fn, stop=None):
            seen, out, stack = set(), set(), [root_fn]      # (the node OBJECTS are kept: the id of a collected wrapper is reused)
            while stack:
                fn = stack.pop()
                if fn is None or fn in seen or fn is stop:
                    continue
                seen.add(fn)
                v = getattr(fn, "variable", None)          # AccumulateGrad node of a leaf
                if v is not None:
                    out.add(id(v))
                stack.extend(f for f, _ in fn.next_functions)
            return out
        under_tap = leaves_under(tap.grad_fn)
        above = leaves_under(loss.grad_fn, stop=tap.grad_fn)
        both = [n for n in names if id(self.store.vars[n]) in under_tap and id(self.store.vars[n]) in above]
        if both:
            raise RuntimeError("variables used on both sides of the backbone's output cannot be split by the backward cut: %s" % both)
        below = [n for n in names if id(self.store.vars[n]) in under_tap]
        head = [n for n in names if id(self.store.vars[n]) not in under_tap]
        self._below_names = (tuple(names), frozenset(below))
        return head, below

    def _bias_before_batchnorm(self, name: str) -> bool:
        """A bias DECLARED in front of a training-mode BatchNorm (tf_util.declare_conv1d / declare_fully_connected with bn=True): its
        gradient is exactly zero and the operators leave it undefined.  A bias with no BatchNorm behind it is not exempt."""
        from .utils import tf_util
        return name in tf_util.BIASES_BEFORE_BATCHNORM

    def _below_tap(self, name: str) -> bool:
        """Is this variable's gradient formed BELOW the backbone's output?  Answered from the last graph walk (_split_at_tap);
        before any cut step has run, from the layer names of the shipped models (the 64-channel layers conv1 .. conv4_b) -- the
        exchange layout is rebuilt when the walk disagrees."""
        cached = getattr(self, "_below_names", None)
        if cached is not None and name in cached[0]:
            return name in cached[1]
        rel = name[len(self.outer) + 1:] if self.outer and name.startswith(self.outer + "/") else name
        parts = rel.split("/")
        return len(parts) >= 2 and parts[0] != "VLAD" and parts[1].startswith("conv") and not parts[1].startswith("conv5")

    def _lr_t(self, lr: float, t: int) -> float:
        """Adam's bias-corrected rate exactly as epc_adam_multi forms it from its float arguments (csrc/train_ops.hip:
        (double)lr * sqrt(1 - pow((double)beta2, t)) / (1 - pow((double)beta1, t)), lr / beta1 / beta2 passed as C floats), so that
        a replayed graph (device-resident rate) and an eager step (rate formed by the library) apply the same bits."""
        import numpy as np
        f = lambda v: float(np.float32(v))
        return f(lr) * math.sqrt(1.0 - f(self.beta2) ** t) / (1.0 - f(self.beta1) ** t)

    def _apply(self, grads, lr, t) -> None:
        """tf.train.AdamOptimizer(learning_rate).minimize (train.py:273-277): all 62 updates in one launch."""
        names = self.trainable_names()
        with torch.no_grad():
            ops.adam_multi([self.store.vars[k] for k in names], [self.m[k] for k in names], [self.v[k] for k in names],
                           grads, lr, t, self.beta1, self.beta2, self.eps)

    def _statistics(self):
        pre = self.outer + "/" if self.outer else ""
        trainable = set(self.store.trainable)
        return [v for k, v in self.store.vars.items() if k.startswith(pre) and k not in trainable]

    def _exchange_layout(self):
        """The flat f32 exchange buffer of data-parallel steps: [gradients formed ABOVE the backbone's output | gradients formed
        below it | moving statistics], every tensor at a 64-float (256-byte) boundary.  Two RCCL messages per step: the head's
        17.9 MB (EPC-Net: hidden1_weights, conv5, the VLAD tensors) start as soon as the head's backward has formed them and travel
        while the backbone's backward runs; the backbone's 0.2 MB + 10 KB of statistics follow it (xGMI rings are per-link bound:
        few large messages, not 62 + 34 small ones).  ``views``: the gradients' slots in trainable_names() order."""
        names = self.trainable_names()
        stats = self._statistics()
        tensors = [self.store.vars[n] for n in names]
        key = (tuple((n, tuple(x.shape), self._below_tap(n)) for n, x in zip(names, tensors)) + tuple(tuple(x.shape) for x in stats))
        ex = self._exchange
        if ex is None or ex["key"] != key or ex["flat"].device != tensors[0].device:
            head = [i for i, n in enumerate(names) if not self._below_tap(n)]
            below = [i for i, n in enumerate(names) if self._below_tap(n)]
            offs, o = {}, 0
            for i in head:
                offs[i] = o
                o += (tensors[i].numel() + 63) // 64 * 64
            head_end = o
            for i in below:
                offs[i] = o
                o += (tensors[i].numel() + 63) // 64 * 64
            stat_offs = []
            for x in stats:
                stat_offs.append(o)
                o += (x.numel() + 63) // 64 * 64
            flat = torch.zeros(max(o, 64), dtype=torch.float32, device=tensors[0].device)
            views = [flat[offs[i]:offs[i] + tensors[i].numel()].view(tensors[i].shape) for i in range(len(names))]
            stat_views = [flat[a:a + x.numel()].view(x.shape) for a, x in zip(stat_offs, stats)]
            ex = self._exchange = {"key": key, "flat": flat, "views": views, "stat_views": stat_views, "head_end": head_end,
                                   "head": head, "below": below, "index": {n: i for i, n in enumerate(names)}}
        return ex

    def _pack_head(self, ex, head_names, head_grads):
        with torch.no_grad():
            torch._foreach_copy_([ex["views"][ex["index"][n]] for n in head_names], list(head_grads))

    def _pack_rest(self, ex, grads, packed_head: bool):
        """The gradients not yet in the buffer (all of them when the backward was not cut) and the moving statistics."""
        with torch.no_grad():
            idx = ex["below"] if packed_head else list(range(len(grads)))
            torch._foreach_copy_([ex["views"][i] for i in idx] + ex["stat_views"], [grads[i] for i in idx] + self._statistics())

    def _unpack_mean(self, ex, world_size: int):
        """After the all-reduce(sum): mean over the ranks in place, statistics back into the variables; returns the averaged
        gradients (views of the exchange buffer, what Adam then reads)."""
        with torch.no_grad():
            if world_size > 1:
                ex["flat"].mul_(1.0 / world_size)
            torch._foreach_copy_(self._statistics(), ex["stat_views"])
        return ex["views"]

    def _exchange_rest(self, ex, packed_head: bool):
        """The blocking part of the exchange: the message(s) not yet under way."""
        import torch.distributed as dist
        if packed_head:
            dist.all_reduce(ex["flat"][ex["head_end"]:], op=dist.ReduceOp.SUM)
        else:
            dist.all_reduce(ex["flat"], op=dist.ReduceOp.SUM)

    def _eager_step(self, query, positives, negatives, other_neg, lr, bn_decay, t):
        from . import distributed as D
        if not D.collectives_active():
            loss, grads = self._forward_backward(query, positives, negatives, other_neg, bn_decay)
            self._apply(grads, lr, t)
            return loss
        # Data parallelism over tuples: every rank its own tuple and batch statistics; gradients and moving statistics
        # are averaged so that every rank applies the same update.  The head's gradients travel under the backbone's backward.
        import torch.distributed as dist
        pending = []

        def between(head_names, head_grads):
            ex = self._exchange_layout()      # (after the graph walk of this step: the layout follows the split it found)
            self._pack_head(ex, head_names, head_grads)
            pending.append(dist.all_reduce(ex["flat"][:ex["head_end"]], op=dist.ReduceOp.SUM, async_op=True))

        loss, grads = self._forward_backward(query, positives, negatives, other_neg, bn_decay,
                                             between=between if self.overlap_exchange else None)
        ex = self._exchange_layout()
        self._pack_rest(ex, grads, bool(pending))
        self._exchange_rest(ex, bool(pending))
        for w in pending:
            w.wait()
        self._apply(self._unpack_mean(ex, D.world()[1]), lr, t)
        return loss

    def _average_over_ranks(self, grads):
        """One flat all-reduce of a list of gradients (trainable_names() order) and of the moving statistics: the uncut form of the
        exchange, kept for callers that hold finished gradients.  Returns the averaged gradients (views of the flat exchange buffer;
        the input list is left as it is).  Single process without force_collective: the input list."""
        from . import distributed as D
        if not D.collectives_active():
            return grads
        ex = self._exchange_layout()
        self._pack_rest(ex, grads, False)
        self._exchange_rest(ex, False)
        return self._unpack_mean(ex, D.world()[1])     # (no copy back: `grads` may hold shared read-only zero tensors)

    # -- HIP-graph replay ------------------------------------------------------------------------------------------------
    def _state_tensors(self):
        out = [self.store.vars[n] for n in self.store.vars if n.startswith(self.outer + "/") or not self.outer]
        out += [self.m[n] for n in self.trainable_names()] + [self.v[n] for n in self.trainable_names()]
        return out

    def _graphed_step(self, query, positives, negatives, other_neg, lr, bn_decay, t):
        from . import distributed as D
        inputs = (query, positives, negatives, other_neg)
        shapes = tuple(tuple(x.shape) for x in inputs)
        dp = D.collectives_active()
        g = self._graph
        if g is None or g["shapes"] != shapes or g["dp"] != dp:
            dev = query.device
            # the four inputs are slices of ONE buffer in train.py:252's order, so that its concat is the buffer itself
            joined = torch.empty((query.shape[0], sum(s[1] for s in shapes)) + tuple(query.shape[2:]), dtype=query.dtype,
                                 device=dev)
            g = {"shapes": shapes, "dp": dp, "joined": joined, "in": list(torch.split(joined, [s[1] for s in shapes], 1)),
                 "lr_t": torch.zeros(1, dtype=torch.float32, device=dev),
                 "bn_decay": torch.zeros((), dtype=torch.float32, device=dev)}
            for dst, src in zip(g["in"], inputs):
                dst.copy_(src)
            g["lr_t"].fill_(self._lr_t(lr, t))
            g["bn_decay"].fill_(bn_decay)
            # warm-up on a side stream (lazy allocations, kernel attributes), on a snapshot of the state that is restored.
            # No collective here: every rank restores its own snapshot, so the ranks stay in step.
            snap = [x.detach().clone() for x in self._state_tensors()]
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                cut = []
                loss, grads = self._forward_backward(*g["in"], g["bn_decay"],
                                                     between=(lambda hn, hg: (self._pack_head(self._exchange_layout(), hn, hg), cut.append(1)))
                                                     if dp and self.overlap_exchange else None)
                if dp:
                    self._pack_rest(self._exchange_layout(), grads, bool(cut))
                    grads = self._unpack_mean(self._exchange_layout(), 1)
                self._apply(grads, g["lr_t"], t)
            torch.cuda.current_stream(dev).wait_stream(side)
            ex = self._exchange_layout() if dp else None      # (after the warm-up's graph walk: the split the captures will see)
            with torch.no_grad():
                for x, s0 in zip(self._state_tensors(), snap):
                    x.copy_(s0)
            for name in self.trainable_names():
                self.store.vars[name].grad = None
            g["graph"] = torch.cuda.CUDAGraph()
            if not dp:
                with torch.cuda.graph(g["graph"]):
                    g["loss"], grads = self._forward_backward(*g["in"], g["bn_decay"])
                    self._apply(grads, g["lr_t"], t)
            else:
                # Data-parallel: [forward, head backward, pack head] | RCCL (head, asynchronous) | [backbone backward, moving averages,
                # pack rest] | RCCL (rest) | [mean, unpack, Adam] -- the captures end and begin INSIDE the step, at the cut of its
                # backward (the `between` callback runs on this thread, between two autograd calls).
                ws = D.world()[1]
                g["graph_mid"] = torch.cuda.CUDAGraph() if self.overlap_exchange else None
                g["graph2"] = torch.cuda.CUDAGraph()
                g["cut"] = False
                torch.cuda.synchronize(dev)
                cap = torch.cuda.Stream(device=dev)
                cap.wait_stream(torch.cuda.current_stream(dev))

                open_graph = [None]        # the graph whose capture is open RIGHT NOW: the only one a failure may end

                def begin(graph, **kw):
                    graph.capture_begin(**kw)
                    open_graph[0] = graph

                def end():
                    graph, open_graph[0] = open_graph[0], None
                    if graph is not None:
                        graph.capture_end()

                def between(hn, hg):
                    self._pack_head(ex, hn, hg)
                    end()
                    begin(g["graph_mid"], pool=g["graph"].pool())
                    g["cut"] = True

                with torch.cuda.stream(cap):
                    try:
                        begin(g["graph"])
                        g["loss"], grads = self._forward_backward(*g["in"], g["bn_decay"],
                                                                  between=between if self.overlap_exchange else None)
                        self._pack_rest(ex, grads, g["cut"])
                        end()
                        begin(g["graph2"], pool=g["graph"].pool())
                        self._apply(self._unpack_mean(ex, ws), g["lr_t"], t)
                    finally:
                        end()
                torch.cuda.current_stream(dev).wait_stream(cap)
                g["ex"] = ex
            self._graph = g
        src = _joined_along_dim1(inputs)
        if src is not None:                       # the caller's four slices of one tuple tensor: one copy, not four
            g["joined"].copy_(src)
        else:
            for dst, s_ in zip(g["in"], inputs):
                dst.copy_(s_)
        g["lr_t"].fill_(self._lr_t(lr, t))
        g["bn_decay"].fill_(bn_decay)
        g["graph"].replay()
        if dp:
            import torch.distributed as dist
            ex = g["ex"]
            pending = None
            if g["cut"]:
                pending = dist.all_reduce(ex["flat"][:ex["head_end"]], op=dist.ReduceOp.SUM, async_op=True)
                g["graph_mid"].replay()                                     # the backbone's backward, under the head's exchange
            self._exchange_rest(ex, g["cut"])                               # same stream: ordered between the replays
            if pending is not None:
                pending.wait()
            g["graph2"].replay()
        # A caller that never waits for the stream would queue replays without limit, and this runtime (ROCm 7.0 / PyTorch 2.10) faults
        # once the packets of replayed graphs have filled the stream's hardware queue a first time without a hipStreamSynchronize in
        # between ("Memory access fault ... Write access to a read-only page" at the ~101st un-synchronised step of ~80 launches: 8192
        # packets; measured in round 6, also on round 5's tree; event waits and `.item()` reads do not prevent it, eager launches do not
        # show it).  A stream synchronisation every REPLAYS_PER_SYNC replayed steps costs one pipeline drain (~0.1 ms) per 32 steps.
        self._replays_since_sync += 1
        if self._replays_since_sync >= REPLAYS_PER_SYNC:
            torch.cuda.current_stream(g["joined"].device).synchronize()
            self._replays_since_sync = 0
        return g["loss"]

    def compute_loss(self, query, positives, negatives, other_neg, is_training: bool, bn_decay=None):
        """train.py:251-264: concat -> forward -> split -> lazy quadruplet loss.  ``is_training=False`` is the
        evaluation loss of train.py:568-576 (stored statistics, no moving-average update)."""
        p = self.params
        with variable_scope(self.outer):
            vecs = _joined_along_dim1((query, positives, negatives, other_neg))                         # train.py:252
            if vecs is None:
                vecs = torch.cat([query, positives, negatives, other_neg], 1)
            out_vecs = self.model.forward(vecs, is_training, bn_decay=bn_decay, params=p)                # :254
            q_vec, pos_vecs, neg_vecs, other_neg_vec = torch.split(
                out_vecs, [1, int(positives.shape[1]), int(negatives.shape[1]), 1], 1)                  # :255
            loss = self.model.lazy_quadruplet_loss(q_vec, pos_vecs, neg_vecs, other_neg_vec,
                                                   p.get("MARGIN_1", 0.5), p.get("MARGIN_2", 0.2))       # :264
        self.last_aux = {"q_vec": q_vec.detach(), "pos_vecs": pos_vecs.detach(), "neg_vecs": neg_vecs.detach(),
                         "other_neg_vec": other_neg_vec.detach()}
        return loss
