"""Variable store with the reference's naming contract.

The reference creates its parameters through ``tf.get_variable`` / ``tf.Variable`` under nested
``tf.variable_scope``s (``train.py:251`` "query_triplets" -> ``models/epc-net.py:62`` "fastdgcnn" / ``:141`` "VLAD"
-> ``utils/tf_util.py:85-98`` "<conv>/weights|biases" -> ``:465-487`` "bn/beta|gamma|<EMA shadows>";
``loupe.py:75-79,249-316`` for the VLAD matrices and slim batch-norm variables).  Checkpoint compatibility is the
state-dict contract: every tensor here is stored under exactly the name and shape the
reference's ``tf.train.Saver`` writes (``tests/golden/ckpt_tables.json``).

Tensors are ``torch.Tensor`` (device memory = plumbing); all arithmetic on them happens in the HIP library.
"""
from __future__ import annotations

import contextlib
import math
from collections import OrderedDict
from typing import Callable, Dict, Iterable, List, Optional, Tuple

import numpy as np
import torch

EMA_MEAN_SUFFIX = "/moments/Squeeze/ExponentialMovingAverage"
EMA_VAR_SUFFIX = "/moments/Squeeze_1/ExponentialMovingAverage"


class VariableStore:
    """Flat ``{full name: tensor}`` map + the set of trainable names, in creation order."""

    def __init__(self, device: Optional[torch.device] = None, seed: Optional[int] = None):
        self.device = torch.device(device) if device is not None else torch.device(
            "cuda" if torch.cuda.is_available() else "cpu")
        self.vars: "OrderedDict[str, torch.Tensor]" = OrderedDict()
        self.trainable: List[str] = []
        self.version = 0  # bumped whenever a value changes: engines re-pack their folded weights
        self._all_version = 0                     # last change that may have touched any scope
        self._scope_version: Dict[str, int] = {}  # top-level scope -> version of its last change (version_of)
        self._gen = torch.Generator(device="cpu")
        if seed is not None:
            self._gen.manual_seed(int(seed))  # the reference itself never seeds (MANUAL_SEED is a dead key)

    def bump(self, scope: Optional[str] = None) -> None:
        """Record that values changed: under the top-level scope of ``scope`` (a variable or scope name), or anywhere."""
        self.version += 1
        if scope:
            self._scope_version[scope.split("/", 1)[0]] = self.version
        else:
            self._all_version = self.version

    def version_of(self, scope: Optional[str]) -> int:
        """Version of the last change that can have touched variables under ``scope``: a frozen model that shares the
        store with a model being trained (the KD teacher, kd_train.py:467-497) keeps its version -- and its packed
        inference weights -- while the other one steps."""
        if not scope:
            return self.version
        return max(self._all_version, self._scope_version.get(scope.split("/", 1)[0], 0))

    # -- creation --------------------------------------------------------------------------------------------
    def get_variable(self, name: str, shape: Iterable[int], initializer: Callable[[Tuple[int, ...], torch.Generator], torch.Tensor],
                     trainable: bool = True) -> torch.Tensor:
        shape = tuple(int(s) for s in shape)
        if name in self.vars:
            t = self.vars[name]
            if tuple(t.shape) != shape:
                raise ValueError("variable %s exists with shape %s, requested %s" % (name, tuple(t.shape), shape))
            return t
        t = initializer(shape, self._gen).to(dtype=torch.float32).reshape(shape).to(self.device).contiguous()
        self.vars[name] = t
        if trainable:
            self.trainable.append(name)
        self.bump(name)
        return t

    # -- state dict -------------------------------------------------------------------------------------------
    def state_dict(self) -> "OrderedDict[str, torch.Tensor]":
        return OrderedDict(self.vars)

    def assign(self, name: str, value) -> None:
        if name not in self.vars:
            raise KeyError("unknown variable %s" % name)
        cur = self.vars[name]
        v = torch.as_tensor(np.asarray(value) if not torch.is_tensor(value) else value, dtype=torch.float32)
        if tuple(v.shape) != tuple(cur.shape):
            raise ValueError("%s: shape %s != %s" % (name, tuple(v.shape), tuple(cur.shape)))
        with torch.no_grad():
            cur.copy_(v.to(cur.device))
        self.bump(name)

    def load_state_dict(self, values: Dict[str, object], strict: bool = True) -> List[str]:
        """Assign every matching name; returns the names of ``values`` that were not used (Adam slots etc.)."""
        unused = []
        for k, v in values.items():
            if k in self.vars:
                self.assign(k, v)
            else:
                unused.append(k)
        if strict:
            missing = [k for k in self.vars if k not in values]
            if missing:
                raise KeyError("checkpoint lacks %d variables, e.g. %s" % (len(missing), missing[:3]))
        return unused

    def load_checkpoint(self, prefix: str, verify_crc: bool = True) -> List[str]:
        """``saver.restore`` (train.py:309-315, evaluate.py:268) from a TF bundle, without TensorFlow.  Payload checksums are
        verified like TensorFlow's BundleReader does (``verify_crc``)."""
        from . import tf_bundle
        return self.load_state_dict(tf_bundle.load_checkpoint(prefix, verify_crc=verify_crc), strict=True)

    def randomize_statistics(self, seed: int = 0) -> None:
        """Synthetic 'trained-like' values for everything the reference initialises to a constant (biases 0,
        gamma 1, beta 0, EMA shadows 0, moving_variance 1): with the raw initial values inference-mode BN divides
        by sqrt(0 + 1e-3) in every tf_util layer and the activations are meaningless.  Used by bench.py / smoke()
        (no checkpoint payloads or datasets exist in this environment -- BASELINE.md 1)."""
        g = torch.Generator(device="cpu")
        g.manual_seed(int(seed))
        for name, t in self.vars.items():
            leaf = name.rsplit("/", 1)[-1]
            n = t.numel()
            if leaf == "biases":
                v = torch.randn(n, generator=g) * 0.05
            elif leaf == "beta":
                v = torch.randn(n, generator=g) * 0.1
            elif leaf == "gamma":
                v = torch.rand(n, generator=g) + 0.5
            elif leaf in ("moving_mean",) or name.endswith(EMA_MEAN_SUFFIX):
                v = torch.randn(n, generator=g) * 0.1
            elif leaf in ("moving_variance",) or name.endswith(EMA_VAR_SUFFIX):
                v = torch.rand(n, generator=g) + 0.5
            else:
                continue
            with torch.no_grad():
                t.copy_(v.reshape(t.shape).to(t.device))
        self.bump()

    def num_trainable_params(self) -> int:
        """``count_params()`` of train.py:202-206."""
        return int(sum(self.vars[n].numel() for n in self.trainable))


# ---- scopes (tf.variable_scope) -------------------------------------------------------------------------------
_scope_stack: List[str] = []
_default_store: Optional[VariableStore] = None


def default_store() -> VariableStore:
    global _default_store
    if _default_store is None:
        _default_store = VariableStore()
    return _default_store


def set_default_store(store: Optional[VariableStore]) -> None:
    global _default_store
    _default_store = store


def reset_default_store(device=None, seed: Optional[int] = None) -> VariableStore:
    """``tf.reset_default_graph()`` analogue."""
    global _default_store
    _default_store = VariableStore(device=device, seed=seed)
    del _scope_stack[:]
    return _default_store


@contextlib.contextmanager
def variable_scope(name: str):
    _scope_stack.append(name)
    try:
        yield "/".join(_scope_stack)
    finally:
        _scope_stack.pop()


def current_scope() -> str:
    return "/".join(_scope_stack)


@contextlib.contextmanager
def absolute_scope(path: str):
    """Re-enter the scope ``path`` (a value of current_scope()) from anywhere: the stack is replaced for the block and restored."""
    saved = list(_scope_stack)
    _scope_stack[:] = path.split("/") if path else []
    try:
        yield path
    finally:
        _scope_stack[:] = saved


def scoped(name: str) -> str:
    s = current_scope()
    return s + "/" + name if s else name


def outer_scope() -> str:
    """The caller's scope above the model scopes (``query_triplets`` in train.py:251)."""
    return current_scope()


# ---- initialisers ---------------------------------------------------------------------------------------------
def xavier_uniform(shape, gen):
    """tf.contrib.layers.xavier_initializer() (utils/tf_util.py:42): U(-l, l), l = sqrt(6/(fan_in+fan_out)),
    fans = receptive field size x channels."""
    rf = 1
    for d in shape[:-2]:
        rf *= d
    fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=gen) * 2.0 - 1.0) * lim


def truncated_normal(stddev):
    def init(shape, gen):
        t = torch.randn(shape, generator=gen)
        bad = t.abs() > 2.0
        while bool(bad.any()):
            t[bad] = torch.randn(int(bad.sum()), generator=gen)
            bad = t.abs() > 2.0
        return t * stddev
    return init


def random_normal(stddev):
    return lambda shape, gen: torch.randn(shape, generator=gen) * stddev


def constant(value):
    return lambda shape, gen: torch.full(shape, float(value))


def ema_shadow_names(bn_scope_full: str) -> Tuple[str, str]:
    """Names of the two EMA shadows ``batch_norm_template`` creates inside ``bn_scope_full`` (utils/tf_util.py:
    474-487): TF prefixes the shadow with the variable scope and then appends the FULL op name of the averaged
    tensor, which repeats the scope -- e.g. ``query_triplets/fastdgcnn/conv1/bn/query_triplets/fastdgcnn/conv1/bn/
    moments/Squeeze/ExponentialMovingAverage``."""
    return (bn_scope_full + "/" + bn_scope_full + EMA_MEAN_SUFFIX,
            bn_scope_full + "/" + bn_scope_full + EMA_VAR_SUFFIX)
