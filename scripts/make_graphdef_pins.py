#!/usr/bin/env python3
"""Evaluate the reference's serialized graphs (``exp/*/saved_model/*.ckpt.meta``) with numpy and freeze the results.

Runs ONLY where /root/reference exists (this container); writes ``tests/golden/graphdef_pins.npz``, which
``tests/test_graphdef_pins_cpu.py`` holds the oracle to.  Nothing of the reference travels: the fixture is inputs'
seeds + output arrays.

For each model (EPC-Net ``exp/epc-net``, EPC-Net-L ``exp/epc-net-l``) the graph is evaluated twice on the same seeded
tuple of 18 clouds x 4096 points (the graph's static shapes: placeholders [1,1|2|14|1,4096,3]) and seeded
``mode='trained'`` weights:
  * ``is_training = False``: the moving statistics feed the normalisations (evaluate.py path);
  * ``is_training = True`` : batch statistics, the moving-average updates, the lazy quadruplet loss (train.py:251-277).
Frozen per run: ``last_output`` (1,18,256), the loss, the bn_decay / learning-rate schedule outputs at a fed global step,
every moving-average update, row sums of the kNN mask, and strided samples of the named activations the oracle taps.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import epcnet_oracle as O          # noqa: E402
import tf_graphdef as G            # noqa: E402

REF = "/root/reference"
MODELS = {
    "epc-net": ("exp/epc-net/saved_model/model_epoch22_iter18101.ckpt.meta", "VLAD/last_output"),
    "epc-net-l": ("exp/epc-net-l/saved_model/model_epoch13_iter18101.ckpt.meta", "VLAD/last_output"),
}
N = 4096
SEED_W, SEED_PC = 5, 77
GLOBAL_STEP = 250000      # a step at which the staircase bn_decay schedule has moved (train.py:138-146)
EPOCH_FED = 12.0          # Placeholder_5 = the epoch fed to the learning-rate schedule (train.py:154-157, 488-491)


def tuple_clouds(is_training):
    """The 18 clouds of the tuple.  Inference runs carry one all-zero cloud: the reference pads its inference batches with them
    (evaluate.py:425-430, train.py:834-844: every point a neighbour of every point, neighbour mean = 204.8 x).  Training tuples
    never hold one (train.py:373-455), and with one the batch statistics are dominated by its rows and float32 itself
    drifts by 1e-2 between summation orders -- nothing a pin could hold."""
    pc = O.synthetic_clouds(18, N, SEED_PC)
    if not is_training:
        pc[17] = 0.0
    return pc


def taps_of(arch):
    """{pin name: graph node (relative to query_triplets/)} of activations the oracle also exposes."""
    nb = 4 if arch == "epc-net" else 2
    t = {}
    for b in range(1, nb + 1):
        t["fastdgcnn/conv%d" % b] = "fastdgcnn/conv%d/Relu" % b
        t["fastdgcnn/conv%d_a" % b] = "fastdgcnn/conv%d_a/Relu" % b
        t["fastdgcnn/conv%d_b" % b] = "fastdgcnn/conv%d_b/Relu" % b
    t["fastdgcnn/conv5"] = "fastdgcnn/conv5/Relu"
    return t


def sample(a):
    """A small deterministic sample of a (clouds, points, channels) / (rows, channels) activation."""
    a = np.asarray(a)
    if a.ndim == 3:
        return a[::5, ::509, :].copy()
    return a[::509].copy()


def evaluate(arch, is_training):
    rel_meta, out_node = MODELS[arch]
    nodes = G.load_graph(os.path.join(REF, rel_meta))
    w = O.seeded_weights(arch, SEED_W)
    pc = tuple_clouds(is_training)
    feeds = G.variable_feeds(nodes, w)
    feeds.update({"Placeholder": pc[None, 0:1], "Placeholder_1": pc[None, 1:3], "Placeholder_2": pc[None, 3:17],
                  "Placeholder_3": pc[None, 17:18], "Placeholder_4": np.bool_(is_training),
                  "Placeholder_5": np.float32(EPOCH_FED), "Variable": np.int32(GLOBAL_STEP)})
    ev = G.GraphEvaluator(nodes, feeds)
    out = {}
    out["last_output"] = ev.get("query_triplets/" + out_node)
    out["loss"] = ev.get("add_2")                      # lazy_quadruplet_loss: Mean + Mean_1 (models/epc-net.py:269-284)
    out["bn_decay"] = ev.get("Minimum")                # train.py:138-146
    out["learning_rate"] = ev.get("Maximum_2")         # train.py:154-157
    mask = ev.get("query_triplets/fastdgcnn/Cast")
    out["mask_rowsum"] = mask.sum(-1).astype(np.int32)
    out["kth"] = ev.get("query_triplets/fastdgcnn/Min")[..., 0]
    for pin, node in taps_of(arch).items():
        out["tap/" + pin] = sample(ev.get("query_triplets/" + node))
    if arch == "epc-net":
        out["tap/vlad_assign"] = sample(ev.get("query_triplets/VLAD/Softmax"))
        out["tap/vlad_flat"] = ev.get("query_triplets/VLAD/l2_normalize_2")[:, ::997].copy()
    if is_training:
        for name, val in G.moving_average_updates(nodes, ev).items():
            out["ema/" + name] = val
    return out, ev


def main():
    if not os.path.isdir(REF):
        raise SystemExit("needs the reference tree at %s" % REF)
    pins = {"meta/N": np.int32(N), "meta/seed_w": np.int32(SEED_W), "meta/seed_pc": np.int32(SEED_PC),
            "meta/global_step": np.int32(GLOBAL_STEP), "meta/epoch": np.float32(EPOCH_FED)}
    for arch in MODELS:
        for tr in (False, True):
            t0 = time.time()
            out, ev = evaluate(arch, tr)
            tag = "%s/%s/" % (arch, "train" if tr else "eval")
            for k, v in out.items():
                pins[tag + k] = np.asarray(v)
            print("%s %s: %d tensors, %.0f s; ops: %s" % (arch, "train" if tr else "eval", len(out), time.time() - t0,
                                                         dict(ev.ops_used)), flush=True)
    path = os.path.join(ROOT, "tests", "golden", "graphdef_pins.npz")
    np.savez_compressed(path, **pins)
    print("wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


if __name__ == "__main__":
    main()
