"""The reference's training driver (``train.py:330-617``, and the identical loop of ``kd_train.py``) around the HIP
training step: tuple sampling, hard-negative mining against cached descriptors, periodic evaluation loss, cache refresh
and checkpointing -- the control flow and its constants restated, the compute on the GPU:

  * one step                    train.py:484-495   -> ``TrainStep.step`` / ``DistillStep.step``
  * descriptor of one cloud     train.py:820-855   -> the fused inference pipeline with the CURRENT weights
  * cached descriptors          train.py:871-965   -> ``retrieval.get_latent_vectors`` (row i = training cloud i)
  * hard negatives              train.py:857-869   -> exact GPU k-NN over the 4000 sampled negatives' cached descriptors
                                                      (sklearn KDTree in the reference)
  * checkpoints                 train.py:611-617   -> TensorFlow bundle files (``tf_bundle.write_checkpoint``), same
                                                      variable names, readable by ``tf.train.Saver`` and by this package

The host side stays numpy / random exactly like the reference (``random.shuffle`` of the positive / negative lists,
``np.random.shuffle`` of the epoch order), so a seeded run visits the same tuples."""
from __future__ import annotations

import logging
import os
from typing import Callable, Dict, List, Optional

import numpy as np
import torch

from . import tf_bundle
from .retrieval import get_latent_vectors
from .utils.loading_pointclouds import NUM_POINTS, get_query_tuple, get_random_hard_negatives
from .variables import variable_scope

SAMPLED_NEG = 4000     # train.py:340
NUM_TO_TAKE = 10       # train.py:343
EVAL_BATCHES = 5       # train.py:527


class Trainer:
    def __init__(self, step, train_queries: Dict[int, dict], train_data: np.ndarray,
                 test_queries: Optional[Dict[int, dict]] = None, test_data: Optional[np.ndarray] = None,
                 save_path: Optional[str] = None, logger: Optional[logging.Logger] = None, graph: bool = False):
        """``step``: a TrainStep / DistillStep; ``*_queries``: the pickles of generate_training_tuples (key -> {'query',
        'positives', 'negatives'}); ``*_data``: (T, 4096, INPUT_DIM) float32 arrays in key order (train.py:159-190)."""
        self.step = step
        self.params = step.params
        self.TRAINING_QUERIES, self.train_data = train_queries, train_data
        self.TEST_QUERIES, self.test_data = test_queries, test_data
        self.save_path = save_path
        self.graph = graph     # replay the step as one HIP graph (TrainStep.step(graph=True)): every tuple has the same shape
        self.log = logger or logging.getLogger("epcnet.train")
        self.HARD_NEGATIVES: Dict[int, List[int]] = {}      # train.py:97 (never filled by the reference either)
        self.TRAINING_LATENT_VECTORS = []                   # train.py:98
        p = self.params
        self.B = int(p.get("BATCH_NUM_QUERIES", 1))
        self.P = int(p.get("POSITIVES_PER_QUERY", p.get("TRAIN_POSITIVES_PER_QUERY", 2)))
        self.N = int(p.get("NEGATIVES_PER_QUERY", p.get("TRAIN_NEGATIVES_PER_QUERY", 14)))
        self.max_epoch = int(p.get("MAX_EPOCH", 20))
        self.num_points = int(p.get("NUM_POINTS", NUM_POINTS))
        self.device = step.store.device
        self.history: List[dict] = []

    # ---- inference with the current weights ------------------------------------------------------------------------
    def _engine(self):
        with variable_scope(self.step.outer):
            scope = getattr(self.step.model, "BACKBONE_SCOPE", "fastdgcnn")
            return self.step.model.engine_for(self.step.model.ARCH, self.params, backbone_scope=scope)

    def get_feature_representation(self, idx: int) -> np.ndarray:
        """train.py:820-855: the descriptor of training cloud ``idx`` (is_training=False)."""
        self.step._ensure_built(int(self.train_data.shape[1]))
        return get_latent_vectors(self._engine(), self.train_data[[idx]], batch_size=1, device=self.device)[0]

    def get_latent_vectors(self, data: Optional[np.ndarray] = None) -> np.ndarray:
        """train.py:871-965: descriptors of every training cloud, row i = cloud i."""
        self.step._ensure_built(int(self.train_data.shape[1]))
        return get_latent_vectors(self._engine(), self.train_data if data is None else data, batch_size=64,
                                  device=self.device)

    # ---- tuples ---------------------------------------------------------------------------------------------------------
    def _tuples(self, keys, queries, data, hard_negs_of: Optional[Callable[[int], List[int]]]):
        """The per-batch tuple assembly of train.py:356-425 / :535-561.  Returns (arrays or None, reason)."""
        tuples = []
        for key in keys:
            if len(queries[key]["positives"]) < self.P:
                return None, "FAULTY TUPLE"
            hard = hard_negs_of(key) if hard_negs_of is not None else []
            tuples.append(get_query_tuple(key, queries[key], self.P, self.N, queries, hard_neg=hard, other_neg=True,
                                          data=data))
            if tuples[-1][3].shape[0] != self.num_points:                                       # train.py:401
                return None, "NO OTHER NEG"
        q = np.expand_dims(np.array([t[0] for t in tuples]), axis=1)
        o = np.expand_dims(np.array([t[3] for t in tuples]), axis=1)
        pos = np.array([t[1] for t in tuples])
        neg = np.array([t[2] for t in tuples])
        if q.ndim != 4:
            return None, "FAULTY TUPLE"
        dev = lambda a: torch.as_tensor(a, dtype=torch.float32).to(self.device)
        return (dev(q), dev(pos), dev(neg), dev(o)), ""

    def _hard_negatives(self, key: int) -> List[int]:
        """train.py:373-377 / :390-395 (the three cache states)."""
        if len(self.TRAINING_LATENT_VECTORS) == 0:
            return []
        query = self.get_feature_representation(key)
        np.random.shuffle(self.TRAINING_QUERIES[key]["negatives"])
        negatives = self.TRAINING_QUERIES[key]["negatives"][0:SAMPLED_NEG]
        hard = get_random_hard_negatives(query, negatives, NUM_TO_TAKE, self.TRAINING_LATENT_VECTORS)
        if len(self.HARD_NEGATIVES.keys()) != 0:
            hard = list(set().union(self.HARD_NEGATIVES[key], hard))
        return hard

    # ---- one epoch -------------------------------------------------------------------------------------------------------
    def train_one_epoch(self, epoch: int, max_iters: Optional[int] = None) -> List[float]:
        """train.py:330-617.  ``max_iters`` truncates the epoch (tests / smoke runs); None = the whole epoch."""
        from . import distributed as D
        rank, world = D.world()
        idxs = np.arange(0, len(self.TRAINING_QUERIES.keys()))
        np.random.shuffle(idxs)
        if world > 1:
            # data-parallel over tuples: every rank must walk the SAME permutation (rank 0's) and take its own
            # slice of every global batch of world * B queries; the ranks step -- or skip -- together
            perm = torch.as_tensor(idxs, dtype=torch.int64, device=self.device)
            D.broadcast_tensors([perm], src=0)
            idxs = perm.cpu().numpy()
        iter_num = len(idxs) // (self.B * world)
        losses = []
        for i in range(iter_num if max_iters is None else min(iter_num, max_iters)):
            base = (i * world + rank) * self.B
            keys = idxs[base:base + self.B]
            batch, why = self._tuples(keys, self.TRAINING_QUERIES, self.train_data, self._hard_negatives)
            if not D.all_true(batch is not None, self.device):
                # a rank that skipped alone would leave the others waiting in the gradient all-reduce
                self.log.info("Epoch: [%d/%d][%d/%d] %s!!!", epoch, self.max_epoch, i + 1, iter_num,
                              why or "another rank drew a faulty tuple")
                continue
            loss, lr, _ = self.step.step(*batch, epoch=epoch, graph=self.graph) if self.graph else self.step.step(*batch, epoch=epoch)
            losses.append(float(loss))
            if not np.isfinite(losses[-1]):
                # a persistent chain launch that was abandoned (a grid barrier ran out of its spin budget) leaves NaN: say so, and reset
                from . import ops
                ops.chain_persist_check()
            self.history.append({"epoch": epoch, "iter": i, "loss": losses[-1], "lr": lr})
            self.log.info("Epoch: [%d/%d][%d/%d] Loss %.4f lr %.8f", epoch, self.max_epoch, i + 1, iter_num, losses[-1], lr)
            if i % 200 == 7 and self.TEST_QUERIES is not None:                                  # train.py:523-594
                self.log.info("\t\t\teval_loss: %f", self.evaluate_loss(epoch))
            if epoch > 5 and i % (1400 // self.B) == 29:                                        # train.py:597-602
                self.TRAINING_LATENT_VECTORS = self.get_latent_vectors()
                self.log.info("Updated cached feature vectors")
            if i % (6000 // self.B) == 101 and self.save_path and rank == 0:                    # train.py:605-617
                self.log.info("Model saved in file: %s", self.save(epoch, i))     # (the ranks hold identical variables)
        return losses

    def evaluate_loss(self, epoch: int) -> float:
        """train.py:523-594: the loss (is_training=False) averaged over up to 5 random test tuples."""
        idxs = np.arange(0, len(self.TEST_QUERIES.keys()))
        np.random.shuffle(idxs)
        total, counted = 0.0, 0
        for e in range(EVAL_BATCHES):
            keys = idxs[e * self.B:(e + 1) * self.B]
            if len(keys) < self.B:
                break
            batch, _ = self._tuples(keys, self.TEST_QUERIES, self.test_data, None)
            if batch is None:
                continue
            self.step._ensure_built(int(batch[0].shape[2]))
            with torch.no_grad():
                total += float(self.step.compute_loss(*batch, False, None))
            counted += 1
        return total / counted if counted else float("nan")        # (the reference divides by zero here)

    # ---- checkpoints -----------------------------------------------------------------------------------------------------
    def checkpoint_tensors(self) -> Dict[str, np.ndarray]:
        """Everything tf.train.Saver() writes at train.py:611: model variables + step + Adam slots.  The optimizer
        scalars live at the graph root (``Variable``, ``beta1_power``, ``beta2_power``), or under ``student/`` for KD."""
        root = self.step.outer.split("/")[0] + "/" if "/" in self.step.outer else ""
        # plain training: tf.train.Saver() covers the whole graph.  KD: student_saver covers scope `student` only
        # (kd_train.py:525-526) -- no teacher/* variables, and the two beta powers live at the graph root, outside it
        out = {k: v.detach().cpu().numpy() for k, v in self.step.store.state_dict().items() if k.startswith(root)}
        for k, v in self.step.optimizer_state().items():
            if k in ("Variable", "beta1_power", "beta2_power"):
                if root and k != "Variable":
                    continue
                k = root + k
            out[k] = v.detach().cpu().numpy()
        return out

    def save(self, epoch: int, i: int) -> str:
        prefix = os.path.join(self.save_path, "saved_model", "model_epoch%d_iter%d.ckpt" % (epoch, i))   # train.py:612
        os.makedirs(os.path.dirname(prefix), exist_ok=True)
        tf_bundle.write_checkpoint(prefix, self.checkpoint_tensors())
        return prefix

    def restore(self, prefix: str) -> None:
        """train.py:308-315 (RESTORE): model variables, global step and Adam moments."""
        state = tf_bundle.load_checkpoint(prefix)
        self.step._ensure_built(int(self.train_data.shape[1]))
        root_scope = self.step.outer.split("/")[0] + "/" if "/" in self.step.outer else ""
        names = set(k for k in self.step.store.vars.keys() if k.startswith(root_scope))
        missing = sorted(names - set(state.keys()))
        if missing:       # model variables are mandatory (only optimizer slots may be absent: a weights-only checkpoint)
            raise KeyError("checkpoint %s lacks %d model variables, e.g. %s" % (prefix, len(missing), missing[:3]))
        self.step.store.load_state_dict({k: v for k, v in state.items() if k in names}, strict=False)
        root = self.step.outer.split("/")[0] + "/" if "/" in self.step.outer else ""
        opt = {k[len(root):] if k.startswith(root) and k[len(root):] in ("Variable", "beta1_power", "beta2_power")
               else k: v for k, v in state.items()}
        self.step.load_optimizer_state(opt)
        self.step.sync_initial_state()

    def train(self, start_epoch: int = 1, max_iters: Optional[int] = None) -> None:
        """train.py:330-338."""
        for epoch in range(start_epoch, self.max_epoch + 1):
            self.log.info("**** EPOCH %03d ****", epoch)
            self.train_one_epoch(epoch, max_iters)
