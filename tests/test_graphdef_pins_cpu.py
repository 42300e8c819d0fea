"""The oracle against the reference's SERIALIZED graphs (tests/golden/graphdef_pins.npz).

The pins are numpy evaluations of the GraphDefs the reference ships in ``exp/epc-net/saved_model/*.ckpt.meta`` and
``exp/epc-net-l/saved_model/*.ckpt.meta`` (oracle/tf_graphdef.py, scripts/make_graphdef_pins.py) on a seeded 18-cloud tuple at
N = 4096 -- the wiring (axes, reshape constants, permutations, epsilons, association of the distance expression, TopKV2 /
FusedBatchNorm attributes, the tf.cond plumbing of the BatchNorm branches, the loss and both schedules) comes from the
reference's own artefact, the op arithmetic from numpy.  The oracle, written from the reference's Python source, must
reproduce them: identical kNN masks / thresholds, every tapped activation, `last_output`, the loss, the schedules and all 32
(16) moving-average updates.  Parity with a RUNNING TensorFlow stays unpinned (oracle/README.md).
"""
import os

import numpy as np
import pytest

import helpers as H  # noqa: F401  (sys.path)
from helpers import O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PINS = np.load(os.path.join(ROOT, "tests", "golden", "graphdef_pins.npz"))
N = int(PINS["meta/N"])
_shared = {}


def tuple_clouds(training):
    """scripts/make_graphdef_pins.py: the inference tuple ends with an all-zero padding cloud, the training tuple does not."""
    pc = O.synthetic_clouds(18, N, int(PINS["meta/seed_pc"]))
    if not training:
        pc[17] = 0.0
    return pc


def mask_of_tuple(training):
    if training not in _shared:
        pc = tuple_clouds(training)
        _shared[training] = (O.pairwise_distance_mask(pc), O.kth_largest(O.neg_sq_dist(pc)))
    return _shared[training]


def sample(a):
    a = np.asarray(a)
    return a[::5, ::509, :] if a.ndim == 3 else a[::509]


def rel_err(a, b):
    return float(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64)).max() / max(np.abs(b).max(), 1e-30))


def test_knn_mask_equals_the_graphs():
    """TopKV2(k=20, sorted) -> Min -> GreaterEqual -> Cast on Neg((Sum + mul) + transpose_1): thresholds bit-equal, row
    sums (20 plus ties; 4096 on the all-zero cloud) equal -- for both graphs (they share the sub-graph)."""
    for training in (False, True):
        mask, kth = mask_of_tuple(training)
        for arch in ("epc-net", "epc-net-l"):
            tag = arch + ("/train/" if training else "/eval/")
            assert np.array_equal(PINS[tag + "kth"], kth)
            assert np.array_equal(PINS[tag + "mask_rowsum"], mask.sum(-1).astype(np.int32))
        assert int(mask.sum(-1).min()) == 20 and int(mask[17].sum()) == (20 * N if training else N * N)


@pytest.mark.parametrize("arch", ["epc-net", "epc-net-l"])
@pytest.mark.parametrize("mode", ["eval", "train"])
def test_oracle_reproduces_the_serialized_graph(arch, mode):
    tag = "%s/%s/" % (arch, mode)
    training = mode == "train"
    step, epoch = int(PINS["meta/global_step"]), float(PINS["meta/epoch"])
    bn_decay = O.get_bn_decay(step)
    assert np.isclose(bn_decay, float(PINS[tag + "bn_decay"]), rtol=1e-7)                       # train.py:138-146
    assert np.isclose(O.get_learning_rate(int(epoch)), float(PINS[tag + "learning_rate"]), rtol=1e-6)   # train.py:154-157
    w = O.seeded_weights(arch, int(PINS["meta/seed_w"]))
    pc = tuple_clouds(training)
    mask, _ = mask_of_tuple(training)
    out, st = O.forward(pc[None], w, is_training=training, bn_decay=bn_decay, arch=arch, mask=mask)
    for key in [k for k in PINS.files if k.startswith(tag + "tap/")]:
        name = key[len(tag) + 4:]
        got = st.taps[name]
        got = got[:, ::997] if name == "vlad_flat" else sample(got)
        # (float32 noise: numpy picks different GEMM blockings for the evaluator's 4-D Conv2D operands and the oracle's 3-D ones)
        assert rel_err(got, PINS[key]) <= 2e-5, (key, rel_err(got, PINS[key]))
    err = float(np.linalg.norm(out.reshape(18, -1) - PINS[tag + "last_output"].reshape(18, -1), axis=1).max())
    # training mode: the batch-statistics normalisations (18 x 4096 rows summed in float32) amplify the GEMM-blocking noise
    assert err <= (5e-5 if training else 2e-6), err
    q, pos, neg, other = np.split(out, [1, 3, 17], axis=1)
    loss = O.lazy_quadruplet_loss(q, pos, neg, other, 0.5, 0.2)
    assert abs(float(loss) - float(PINS[tag + "loss"])) <= (1e-4 if training else 1e-6)
    ema = [k for k in PINS.files if k.startswith(tag + "ema/")]
    assert len(ema) == (0 if not training else (32 if arch == "epc-net" else 16))
    for key in ema:
        name = key[len(tag) + 4:]
        assert rel_err(st.new_stats[name], PINS[key]) <= 1e-5, (key, rel_err(st.new_stats[name], PINS[key]))


@pytest.mark.skipif(not (os.path.isdir("/root/reference") and os.environ.get("EPC_RECHECK_PINS")),
                    reason="opt-in (EPC_RECHECK_PINS=1, ~1 min) and only where the reference tree exists")
def test_pins_are_what_the_reference_graph_gives_now():
    """Where the reference tree is present: re-evaluate one graph and compare with the committed pins (the fixture is not stale
    and was not edited by hand).  EPC-Net-L, inference mode: the cheapest of the four."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import make_graphdef_pins as M
    out, _ = M.evaluate("epc-net-l", False)
    for k, v in out.items():
        assert np.array_equal(np.asarray(v), PINS["epc-net-l/eval/" + k]), k
