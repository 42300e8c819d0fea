"""GPU: the backbone chain's forward (and, with BWD=1, forward + backward) as the launch chain and as the persistent launches, timed
through tf_util.proxyconv_backbone at the training tuple's size.  NCL=18 N=4096 PRECISION=bf16|bf16x6."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import helpers as H
from helpers import O
V, tf_util, ops, L = H.pkg("variables"), H.pkg("utils.tf_util"), H.pkg("ops"), H.pkg("lib")
dev = torch.device("cuda:0")
ncl, n = int(os.environ.get("NCL", "18")), int(os.environ.get("N", "4096"))
arch = os.environ.get("ARCH", "epc-net"); nb = 4 if arch == "epc-net" else 2
prec = os.environ.get("PRECISION", "bf16")
bwd = os.environ.get("BWD", "0") == "1"
w = O.seeded_weights(arch, 4); pc = O.synthetic_clouds(ncl, n, 11)
st = H.make_store(arch, w, dev)
x = ops.morton_sort(torch.from_numpy(pc).to(dev))
ops.set_gemm_precision(prec)
names = [k for k in st.trainable if "/fastdgcnn/" in k and "/conv5/" not in k]
for k in names:
    st.vars[k].requires_grad_(True)
graph = ops.KnnGraph(x); graph.transposed(); graph.overflow()
G = torch.randn((ncl, n, 64 * nb), device=dev)

def run():
    for v in st.vars.values():
        v.grad = None
    with V.variable_scope(H.OUTER), V.variable_scope("fastdgcnn"):
        cat = tf_util.proxyconv_backbone(x, graph, 20, nb, bn_decay=0.7, is_training=True)
    if bwd:
        cat.backward(G)
    return cat

def timeit(label):
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(5):
        e0.record()
        for _ in range(10):
            run()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 10)
    print("%-28s %s %dx%d %s: %.1f us per %s (eager, includes host launch gaps)" % (label, arch, ncl, n, prec, best * 1e3, "fwd+bwd" if bwd else "fwd"))

outs = {}
for label, f in (("launch chain", False), ("persistent fwd", True)):
    ops.CHAIN_PERSIST_FWD = f
    outs[label] = run().detach().float().cpu().numpy()
    ops.chain_persist_check()
    timeit(label)
ref = outs["launch chain"]
for k, v in outs.items():
    print("%-28s max |cat - launch chain| = %.3e (scale %.3e), rel L2 %.3e" % (k, np.abs(v - ref).max(), np.abs(ref).max(), np.linalg.norm(v - ref) / np.linalg.norm(ref)))

if os.environ.get("STAMPS", "0") == "1":      # needs a -DPST_STAMPS build of the library (scripts/build_variant.sh, EPCNET_LIB=...)
    ops.CHAIN_PERSIST_FWD = True
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ws = ops.chain_workspace(dev)
    nbytes = ws.numel()
    st_ = ws[nbytes - 512 * 64 * 8:].view(torch.int64).view(512, 64).cpu().numpy()
    P = L.lib().epc_chain_parts(ncl * n)
    st_ = st_[:P]
    nst = int((st_[0] != 0).sum())
    t0 = st_[:, 0].min()
    names = ["start", "S done"]
    for b in range(nb):
        names += ["b%d reduced0" % b, "b%d gathered0" % b, "b%d pooled0" % b, "b%d G done" % b, "b%d posted_a" % b,
                  "b%d reduced_a" % b, "b%d gathered_a" % b, "b%d pooled_a" % b, "b%d M done" % b, "b%d posted_b" % b,
                  "b%d reduced_b" % b, "b%d gathered_b" % b, "b%d pooled_b" % b, "b%d H done" % b, "b%d posted0'" % b]
    print("stamp (10-ns ticks -> us): min / median / max over %d workgroups, relative to the earliest start; delta of medians; [leaders' median]" % P)
    prev = 0.0
    for k in range(nst):
        col = (st_[:, k] - t0) / 100.0
        med = float(np.median(col))
        print("  %-18s %8.2f %8.2f %8.2f   +%6.2f   [%8.2f]" % (names[k] if k < len(names) else "?", col.min(), med, col.max(), med - prev, float(np.median(col[:8]))))
        prev = med
