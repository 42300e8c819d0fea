"""Tuning aid (GPU box): host-side cost of one-cloud forwards (enqueue vs end-to-end) + cProfile of the Python path."""
import os, sys, time
sys.path.insert(0, "/root/repo")
import torch, bench
E = bench.pkg("engine")
dev = torch.device("cuda:0")
store = bench.build_store("epc-net", dev, 0)
eng = E.InferenceEngine("epc-net", bench.PARAMS, store, outer=bench.OUTER)
x = (torch.rand((1, 4096, 3)) * 2 - 1).to(dev); out = torch.empty((1, 256), device=dev)
for _ in range(20): eng.forward(x, out=out)
torch.cuda.synchronize()
import cProfile, pstats
t0 = time.perf_counter()
for _ in range(300): eng.forward(x, out=out)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("enqueue %.1f us per forward (host), total %.1f us per forward" % ((t1 - t0) / 300 * 1e6, (t2 - t0) / 300 * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(300): eng.forward(x, out=out)
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(8)
