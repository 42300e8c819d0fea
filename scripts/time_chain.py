"""GPU: time each launch of the fused backbone chain (csrc/train_chain.hip) on its own, at the training tuple's size, with and
without the pooled-statistics prologue -- where do the microseconds of a chain kernel go?  NCL=18 N=4096 by default; PIECES=1 times
the bf16 arithmetic (one bf16 value per operand), the default 3 / 2 the f32-accurate one.  The activations are float32 in both."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
L = bench.pkg("lib"); ops = bench.pkg("ops")
lib = L.lib()
dev = torch.device("cuda:0")
ncl, n = int(os.environ.get("NCL", "18")), int(os.environ.get("N", "4096"))
rows = ncl * n
P = lib.epc_chain_parts(rows)
g = torch.Generator(device=dev); g.manual_seed(0)
rnd = lambda *s: torch.randn(s, generator=g, device=dev)
xyz = ops.morton_sort((torch.rand((ncl, n, 3), generator=g, device=dev) * 2 - 1))
graph = ops.KnnGraph(xyz)
rdeg, roff, rlist = graph.transposed(); ovc, ovl = graph.overflow()
one = os.environ.get("PIECES", "3") == "1"
pf, pb = (1, 1) if one else (3, 2)
z0, za, zb, xm, d, dy, s_, gout, dxx = (rnd(rows, 64) for _ in range(9))
cat = rnd(rows, 256); dcat = rnd(rows, 256); cat16 = cat.to(torch.bfloat16)
W = rnd(64, 64) / 8; bias = rnd(64) * 0.1
gamma, beta = torch.rand(64, generator=g, device=dev) + 0.5, rnd(64) * 0.1
mean, var = torch.zeros(64, device=dev), torch.ones(64, device=dev)
stats = torch.empty(P * 192, device=dev); sums = torch.empty(P * 128, device=dev); psums = torch.empty(P * 128, device=dev)
stats_o = torch.empty(P * 192, device=dev)
dwp = torch.empty(P * 4096, device=dev); dgam, dbet = torch.empty(64, device=dev), torch.empty(64, device=dev)
st = L.current_stream()
L.check(lib.epc_chain_stats(z0.data_ptr(), rows, stats.data_ptr(), st))
L.check(lib.epc_chain_sums(dy.data_ptr(), 64, zb.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-3, rows, sums.data_ptr(), st))
p = lambda t: t.data_ptr() if t is not None else None

def fwd_linear(pool, resid, aout, W_):
    return lambda: L.check(lib.epc_chain_fwd_linear(za.data_ptr(), p(stats if pool else None), None, mean.data_ptr(), var.data_ptr(), gamma.data_ptr(),
        beta.data_ptr(), 1e-3, p(xm if resid else None), (cat.data_ptr() + 256) if aout else None, 256, (cat16.data_ptr() + 128) if aout and one else None, p(W if W_ else None), p(bias if W_ else None),
        p(zb if W_ else None), p(stats_o if W_ else None), rows, pf, st))

def fwd_gather(pool):
    return lambda: L.check(lib.epc_chain_fwd_gather(z0.data_ptr(), p(stats if pool else None), None, mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-3,
        graph.xyz.data_ptr(), graph.idx.data_ptr(), graph.cnt.data_ptr(), graph.kth.data_ptr(), 32, ncl, n, 20, W.data_ptr(), bias.data_ptr(), xm.data_ptr(),
        d.data_ptr(), za.data_ptr(), stats_o.data_ptr(), pf, st))

def bwd_linear(xbn, addend, zp, strided=False):
    return lambda: L.check(lib.epc_chain_bwd_linear(dcat.data_ptr() if strided else dy.data_ptr(), 256 if strided else 64, zb.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-3,
        sums.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), W.data_ptr(), cat.data_ptr() if strided else za.data_ptr(), 256 if strided else 64,
        *( (mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr()) if xbn else (None, None, None, None) ),
        s_.data_ptr(), (dcat.data_ptr() + 256 if strided else gout.data_ptr()) if addend else None, 256 if strided else 64, dwp.data_ptr(),
        p(za if zp else None), *( (mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr()) if zp else (None, None, None, None) ),
        p(psums if zp else None), rows, pb, st))

def bwd_gather():
    return lambda: L.check(lib.epc_chain_bwd_gather(s_.data_ptr(), gout.data_ptr(), 64, rdeg.data_ptr(), roff.data_ptr(), rlist.data_ptr(), ovc.data_ptr(), ovl.data_ptr(),
        graph.xyz.data_ptr(), graph.kth.data_ptr(), ncl, n, 20, z0.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-3,
        psums.data_ptr(), dxx.data_ptr(), st))

cases = [("fwd_linear mid (conv_b)            pooled", fwd_linear(True, False, False, True)),
         ("fwd_linear mid (conv_b)            given ", fwd_linear(False, False, False, True)),
         ("fwd_linear head (+resid, cat, next) pooled", fwd_linear(True, True, True, True)),
         ("fwd_linear head (+resid, cat, next) given ", fwd_linear(False, True, True, True)),
         ("fwd_linear tail (no layer)          pooled", fwd_linear(True, True, True, False)),
         ("fwd_gather                          pooled", fwd_gather(True)),
         ("fwd_gather                          given ", fwd_gather(False)),
         ("bwd_linear conv_b (xbn, zp)", bwd_linear(True, False, True)),
         ("bwd_linear conv_a (addend)", bwd_linear(False, True, False)),
         ("bwd_linear conv0 (strided x, addend, zp)", bwd_linear(False, True, True, True)),
         ("bwd_linear plain", bwd_linear(False, False, False)),
         ("bwd_gather", bwd_gather()),
         ("chain_stats", lambda: L.check(lib.epc_chain_stats(z0.data_ptr(), rows, stats.data_ptr(), st))),
         ("chain_sums", lambda: L.check(lib.epc_chain_sums(dy.data_ptr(), 64, zb.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-3, rows, sums.data_ptr(), st))),
         ("chain_bn_bwd", lambda: L.check(lib.epc_chain_bn_bwd(dy.data_ptr(), z0.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-3, sums.data_ptr(), dgam.data_ptr(), dbet.data_ptr(), rows, dxx.data_ptr(), st))),
         ("old neighbour_mean_diff_fwd", lambda: L.check(lib.epc_neighbour_mean_diff_fwd(z0.data_ptr(), graph.xyz.data_ptr(), graph.idx.data_ptr(), graph.cnt.data_ptr(), graph.kth.data_ptr(), 32, ncl, n, 20, xm.data_ptr(), d.data_ptr(), st))),
         ("old gather_bwd_sum", lambda: L.check(lib.epc_neighbour_mean_diff_bwd_gather_sum(s_.data_ptr(), gout.data_ptr(), graph.xyz.data_ptr(), graph.cnt.data_ptr(), graph.kth.data_ptr(), 32, rdeg.data_ptr(), roff.data_ptr(), rlist.data_ptr(), ncl, n, 20, dxx.data_ptr(), st))),
         ]
only = os.environ.get("ONLY")
for name, fn in cases:
    if only and only not in name:
        continue
    print("%-46s" % name, end=" ", flush=True)
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    K = 50
    e0.record()
    for _ in range(K):
        fn()
    e1.record(); torch.cuda.synchronize()
    print("%7.1f us" % (e0.elapsed_time(e1) / K * 1e3), flush=True)
