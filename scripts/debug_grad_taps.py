"""Where does the backward error first appear?  Gradients w.r.t. every conv1d OUTPUT (post BN+ReLU) of one training
step, HIP path against the float64 and float32 torch oracles."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import numpy as np, torch
import helpers as H
from helpers import O
import epcnet_oracle_torch as T
arch, nneg, n = "epc-net", 18, 256
dev = torch.device("cuda:0")
w0 = O.seeded_weights(arch, 4)
pcs = O.synthetic_clouds(1 + 2 + nneg + 1, n, 9)
q, pos, neg, oth = pcs[None, :1], pcs[None, 1:3], pcs[None, 3:3 + nneg], pcs[None, 3 + nneg:]

def oracle_taps(dtype):
    taps = {}
    orig = T.TorchOracle.conv1d
    def conv1d(self, x, scope, training, bn_decay):
        y = orig(self, x, scope, training, bn_decay)
        y.retain_grad(); taps[scope] = y
        return y
    T.TorchOracle.conv1d = conv1d
    try:
        T.train_step(w0, q, pos, neg, oth, step=3, epoch=7, arch=arch, dtype=dtype)
    finally:
        T.TorchOracle.conv1d = orig
    return {k: v.grad.detach().double().numpy().reshape(-1, v.shape[-1]) for k, v in taps.items()}, \
           {k: v.detach().double().numpy().reshape(-1, v.shape[-1]) for k, v in taps.items()}

g64, a64 = oracle_taps(torch.float64)
g32, _ = oracle_taps(torch.float32)

st = H.make_store(arch, w0, dev)
TR, tf_util = H.pkg("training"), H.pkg("utils.tf_util")
taps = {}
orig = tf_util.conv1d
def conv1d(inputs, num_output_channels, kernel_size, scope, **kw):
    y = orig(inputs, num_output_channels, kernel_size, scope, **kw)
    y.retain_grad(); taps["fastdgcnn/" + scope] = y
    return y
tf_util.conv1d = conv1d
params = dict(H.PARAMS, ARCH=arch, BATCH_NUM_QUERIES=1)
ts = TR.TrainStep(params, st, outer=H.OUTER)
ts.global_step = 3
to = lambda a: torch.from_numpy(a).to(dev)
ts.step(to(q), to(pos), to(neg), to(oth), epoch=7)
# the HIP path Z-orders the points of every cloud: undo it through the forward values (rows are matched by content)
print("%-24s %12s %12s %12s" % ("output of", "fwd relL2", "grad relL2", "o32 grad"))
for k in g64:
    y = taps[k]; g = y.grad.double().cpu().numpy().reshape(-1, y.shape[-1]); a = y.detach().double().cpu().numpy().reshape(-1, y.shape[-1])
    # match rows cloud by cloud via the sorted first-layer... simpler: compare order-invariant per-cloud sums
    ncl = 1 + 2 + nneg + 1
    A, A64 = a.reshape(ncl, n, -1), a64[k].reshape(ncl, n, -1)
    Gm, G64, G32 = g.reshape(ncl, n, -1), g64[k].reshape(ncl, n, -1), g32[k].reshape(ncl, n, -1)
    # order-invariant statistics: per cloud and channel, sum of grad and sum of grad*activation
    s = lambda G, A_: np.concatenate([G.sum(1).ravel(), (G * A_).sum(1).ravel(), (G * G).sum(1).ravel()])
    e = lambda x, r: np.linalg.norm(x - r) / max(np.linalg.norm(r), 1e-30)
    print("%-24s %12.2e %12.2e %12.2e" % (k[10:], e(np.sort(A, 1), np.sort(A64, 1)), e(s(Gm, A), s(G64, A64)), e(s(G32, A64), s(G64, A64))))
