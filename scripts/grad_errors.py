"""Per-tensor gradient error of one training step against the float64 torch oracle (tests/test_gpu_train_step.py's
configuration), printed as a table: relative L2 error and max error / max |g|."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
import helpers as H
from helpers import O
import epcnet_oracle_torch as T
arch, nneg, n = os.environ.get("ARCH", "epc-net"), int(os.environ.get("NEG", "18")), int(os.environ.get("N", "256"))
dev = torch.device("cuda:0")
w0 = O.seeded_weights(arch, 4)
pcs = O.synthetic_clouds(1 + 2 + nneg + 1, n, 9)
q, pos, neg, oth = pcs[None, :1], pcs[None, 1:3], pcs[None, 3:3 + nneg], pcs[None, 3 + nneg:]
ref = T.train_step(w0, q, pos, neg, oth, step=3, epoch=7, arch=arch)
ref32 = T.train_step(w0, q, pos, neg, oth, step=3, epoch=7, arch=arch, dtype=torch.float32)
st = H.make_store(arch, w0, dev)
TR, ops = H.pkg("training"), H.pkg("ops")
params = dict(H.PARAMS, ARCH=arch, BATCH_NUM_QUERIES=1, TRAIN_PRECISION=os.environ.get("PRECISION", "bf16x6"))
ts = TR.TrainStep(params, st, outer=H.OUTER)
ts.global_step = 3
grads = {}
orig = ops.adam_multi
def spy(ws, ms, vs, gs, *a):
    for w, g in zip(ws, gs):
        for k, t_ in st.vars.items():
            if t_.data_ptr() == w.data_ptr():
                grads[k[len(H.OUTER) + 1:]] = g.detach().cpu().numpy().copy()
    return orig(ws, ms, vs, gs, *a)
ops.adam_multi = spy
to = lambda a: torch.from_numpy(a).to(dev)
loss, _, _ = ts.step(to(q), to(pos), to(neg), to(oth), epoch=7)
print("loss %.8f oracle64 %.8f oracle32 %.8f" % (float(loss), ref["loss"], ref32["loss"]))
print("%-44s %10s %10s   %10s %10s" % ("tensor", "relL2", "max/gmax", "o32 relL2", "o32 max"))
for k, g_ref in ref["grads"].items():
    if k.endswith("/biases"):
        continue
    g = grads[k].reshape(g_ref.shape); g32 = ref32["grads"][k]
    nr, gm = max(np.linalg.norm(g_ref), 1e-30), max(np.abs(g_ref).max(), 1e-30)
    print("%-44s %10.2e %10.2e   %10.2e %10.2e" % (k, np.linalg.norm(g - g_ref) / nr, np.abs(g - g_ref).max() / gm,
                                                  np.linalg.norm(g32 - g_ref) / nr, np.abs(g32 - g_ref).max() / gm))
if os.environ.get("CHANNELS"):
    k = os.environ["CHANNELS"]                       # e.g. fastdgcnn/conv2_b
    gb, gb_ref = grads[k + "/bn/beta"], ref["grads"][k + "/bn/beta"]
    gg, gg_ref = grads[k + "/bn/gamma"], ref["grads"][k + "/bn/gamma"]
    m, v = O.ema_names(k)
    print("channel  dbeta(ours)  dbeta(ref)  dgamma(ours) dgamma(ref)   beta     gamma")
    order = np.argsort(-np.abs(gb - gb_ref))
    for c in order[:12]:
        print("%5d  %11.4e %11.4e  %11.4e %11.4e  %8.4f %8.4f" % (c, gb[c], gb_ref[c], gg[c], gg_ref[c],
              w0[k + "/bn/beta"][c], w0[k + "/bn/gamma"][c]))
