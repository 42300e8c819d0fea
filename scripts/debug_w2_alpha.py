"""GPU debug: whose gradient of VLAD/cluster_weights2 is off in the bf16 step at 4096 points -- the HIP step's or the oracle's?
Four gradients on one tuple: HIP bf16, HIP f32-accurate, oracle with bf16 rounding points (masks + values pinned to the HIP bf16 step),
oracle exact (masks pinned).  Prints the projection alpha and the relative L2 distance of every pair, for a few tensors."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import helpers as H
from helpers import O
import epcnet_oracle_torch as T
TR, ops, TFU = H.pkg("training"), H.pkg("ops"), H.pkg("utils.tf_util")
dev = torch.device("cuda:0")
n, nneg = int(os.environ.get("N", "4096")), 14
ncl = 1 + 2 + nneg + 1
w0 = O.seeded_weights("epc-net", 4)
pcs = O.synthetic_clouds(ncl, n, 9)
tup = [torch.from_numpy(a).to(dev) for a in (pcs[None, :1], pcs[None, 1:3], pcs[None, 3:3 + nneg], pcs[None, 3 + nneg:])]

def hip(precision):
    st = H.make_store("epc-net", w0, dev)
    ts = TR.TrainStep(dict(H.PARAMS, ARCH="epc-net", BATCH_NUM_QUERIES=1, TRAIN_PRECISION=precision), st, outer=H.OUTER)
    ts.global_step = 3
    grads = {}
    orig = ops.adam_multi
    def spy(ws, ms, vs, gs, *a):
        for w_, g in zip(ws, gs):
            for k, t_ in st.vars.items():
                if t_.data_ptr() == w_.data_ptr():
                    grads[k[len(H.OUTER) + 1:]] = g.detach().double().cpu().numpy().copy()
        return orig(ws, ms, vs, gs, *a)
    ops.adam_multi = spy
    TFU.RELU_MASK_TAPS, TFU.VALUE_TAPS = {}, {}
    try:
        loss, _, _ = ts.step(*tup, epoch=7)
    finally:
        ops.adam_multi = orig
        masks, TFU.RELU_MASK_TAPS = TFU.RELU_MASK_TAPS, None
        pins, TFU.VALUE_TAPS = TFU.VALUE_TAPS, None
    masks = {k[len(H.OUTER) + 1:]: v.cpu().numpy() for k, v in masks.items()}
    pins = {k[len(H.OUTER) + 1:]: v.cpu().numpy() for k, v in pins.items()}
    aux = ts.last_aux
    pins["descriptors"] = torch.cat([aux["q_vec"], aux["pos_vecs"], aux["neg_vecs"], aux["other_neg_vec"]], 1).double().cpu().numpy()
    return grads, masks, pins

g16, masks, pins = hip("bf16")
g32, masks32, _ = hip("bf16x6")
srt = ops.morton_sort(torch.from_numpy(pcs).to(dev)).cpu().numpy()[None]
sp = (srt[:, :1], srt[:, 1:3], srt[:, 3:3 + nneg], srt[:, 3 + nneg:])
o16 = T.train_step(w0, *sp, step=3, epoch=7, arch="epc-net", relu_masks=masks, gemm_rounding="bf16", value_pins=pins)["grads"]
oex = T.train_step(w0, *sp, step=3, epoch=7, arch="epc-net", relu_masks=masks32)["grads"]
al = lambda g, r: float(np.vdot(r.ravel(), g.ravel()) / np.vdot(r.ravel(), r.ravel()))
rl = lambda g, r: float(np.linalg.norm(g.ravel() - r.ravel()) / np.linalg.norm(r.ravel()))
for k in ("VLAD/cluster_weights2", "VLAD/cluster_weights", "VLAD/hidden1_weights", "fastdgcnn/conv5/weights", "VLAD/gating_weights"):
    a, b, c, d = g16[k], g32[k], o16[k].reshape(g16[k].shape), oex[k].reshape(g16[k].shape)
    print("%-26s alpha-1 / relL2:  hip16|orc16 %+.2e / %.2e   hip16|hip32 %+.2e / %.2e   orc16|orcEx %+.2e / %.2e   hip32|orcEx %+.2e / %.2e   hip16|orcEx %+.2e / %.2e"
          % (k, al(a, c) - 1, rl(a, c), al(a, b) - 1, rl(a, b), al(c, d) - 1, rl(c, d), al(b, d) - 1, rl(b, d), al(a, d) - 1, rl(a, d)))
