"""GPU: the fused chain vs the per-layer operators, slice by slice of the concat, and both against float64 torch."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
from helpers import O
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_chain as TC
import epcnet_oracle_torch as T

dev = torch.device("cuda:0")
arch, ncl, n = "epc-net", 3, 256
w = O.seeded_weights(arch, 4)
pc = O.synthetic_clouds(ncl, n, 11)
a = TC._backbone(arch, w, pc, dev, True)
b = TC._backbone(arch, w, pc, dev, False)
# float64 reference of the same backbone on the sorted clouds
ops = H.pkg("ops")
srt = ops.morton_sort(torch.from_numpy(pc).to(dev)).cpu().numpy()
orc = T.TorchOracle(w, arch)
mask = torch.tensor(O.pairwise_distance_mask(srt), dtype=torch.float64)
inp = torch.tensor(srt, dtype=torch.float64)
outs = []
for blk in range(1, 5):
    x = orc.conv1d(inp, "fastdgcnn/conv%d" % blk, True, 0.7)
    xm = torch.matmul(mask, x) / 20.0
    t = orc.conv1d(xm - x, "fastdgcnn/conv%d_a" % blk, True, 0.7)
    t = orc.conv1d(t, "fastdgcnn/conv%d_b" % blk, True, 0.7)
    inp = t + xm
    outs.append(inp)
ref = torch.cat(outs, -1).detach().numpy()
for blk in range(4):
    sl = slice(64 * blk, 64 * blk + 64)
    print("block %d: |ref|max %.3f  chain-ref %.2e  layers-ref %.2e  chain-layers %.2e" % (
        blk + 1, np.abs(ref[..., sl]).max(), np.abs(a[0][..., sl] - ref[..., sl]).max(), np.abs(b[0][..., sl] - ref[..., sl]).max(),
        np.abs(a[0][..., sl] - b[0][..., sl]).max()))
G = torch.randn(tuple(a[0].shape), generator=torch.Generator().manual_seed(3)).double()
loss = (torch.cat(outs, -1) * G).sum()
names = [k for k in orc.trainable if k.startswith("fastdgcnn/") and "/conv5/" not in k]
gr = torch.autograd.grad(loss, [orc.w[k] for k in names], allow_unused=True)
for k, g in zip(names, gr):
    if k.endswith("/biases"):
        continue
    g = g.numpy()
    ga, gb = a[1][H.OUTER + "/" + k].reshape(g.shape), b[1][H.OUTER + "/" + k].reshape(g.shape)
    r = lambda u: np.linalg.norm(u - g) / np.linalg.norm(g)
    print("%-34s chain-ref %.2e  layers-ref %.2e" % (k, r(ga), r(gb)))
