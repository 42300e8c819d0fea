"""Tuning aid (GPU box): one 64-cloud forward replayed as a HIP graph against eager launches (same stream)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
E = bench.pkg("engine")
dev = torch.device("cuda:0")
arch = os.environ.get("ARCH", "epc-net")
B = int(os.environ.get("BATCH", "64"))
K = int(os.environ.get("STEPS", "300"))
store = bench.build_store(arch, dev, 0)
eng = E.InferenceEngine(arch, bench.PARAMS, store, outer=bench.OUTER, micro_batch=B)
xyz = (torch.rand((B, 4096, 3)) * 2 - 1).to(dev)
out = torch.empty((B, 256), device=dev)
for _ in range(5):
    eng.forward(xyz, out=out)
torch.cuda.synchronize()
ref = out.clone()
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    eng.forward(xyz, out=out)
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        eng.forward(xyz, out=out)
torch.cuda.synchronize()
def timed(fn):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
a = timed(lambda: eng.forward(xyz, out=out))
b = timed(g.replay)
print("batch %d: eager %.4f ms/step, graph replay %.4f ms/step; same result: %s" % (B, a, b, torch.equal(out, ref)))
