"""Randomised parity sweep (GPU box; needs oracle/): many seeds x cloud kinds x sizes.  kNN lists must be bit-identical
to the oracle's on the Z-ordered cloud, descriptors within 1e-4 (EPC-Net) of the f32 oracle.  Prints one line per case
that fails and a summary; exit code 1 on any failure."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers as H
from helpers import O
dev = torch.device("cuda:0")
tf_util = H.pkg("utils.tf_util"); ops = H.pkg("ops")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
precision = os.environ.get("PRECISION", "fast")   # EPC-Net arithmetic under test (EPC-Net-L has one)
t0 = time.time()
fails = cases = 0
worst = {}
engines = {}
seed = 0
while time.time() - t0 < budget:
    for kind in ("uniform", "lidar", "dup", "lattice"):
        for n in (64, 160, 512, 992, 2048, 4096):
            if time.time() - t0 > budget:
                break
            pc = O.synthetic_clouds(2, n, 1000 + seed, kind)
            x = torch.from_numpy(pc).to(dev)
            srt = ops.morton_sort(x)
            kth, idx, cnt = tf_util.knn_index(srt)
            s_np = srt.cpu().numpy()
            rk, lists = O.knn_lists(s_np)
            ok = np.array_equal(kth.cpu().numpy(), rk)
            cn, ix = cnt.cpu().numpy(), idx.cpu().numpy()
            for b in range(2):
                for i in range(n):
                    l = lists[b][i]
                    if cn[b, i] != len(l) or not np.array_equal(ix[b, i, :min(len(l), 32)], l[:32]):
                        ok = False
                        break
            cases += 1
            if not ok:
                fails += 1
                print("kNN MISMATCH kind=%s n=%d seed=%d" % (kind, n, seed))
            for arch in ("epc-net", "epc-net-l"):
                key = (arch, seed % 3)
                if key not in engines:
                    w = O.seeded_weights(arch, seed % 3)
                    engines[key] = (w, H.make_engine(arch, w, dev, precision=precision)[0])
                w, eng = engines[key]
                # make_engine resets the default store: rebuild per use to stay independent
                eng = H.make_engine(arch, w, dev, precision=precision)[0]
                ref, _ = O.forward(s_np[:, None], w, arch=arch, formulation="lists", lists=lists)
                out = eng.forward(x).cpu().numpy()
                err = float(np.linalg.norm(out - ref.reshape(2, -1), axis=1).max())
                worst[(arch, kind)] = max(worst.get((arch, kind), 0.0), err)
                cases += 1
                if not (err <= 1e-4):
                    fails += 1
                    print("DESCRIPTOR kind=%s n=%d seed=%d arch=%s err=%.3e" % (kind, n, seed, arch, err))
    seed += 1
print("precision %s: cases %d  failures %d  seeds %d" % (precision, cases, fails, seed))
for k, v in sorted(worst.items()):
    print("  worst descriptor error %-10s %-8s %.3e" % (k[0], k[1], v))
sys.exit(1 if fails else 0)
