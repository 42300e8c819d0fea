"""Tuning aid (GPU box): does a second, independent 64-cloud extraction on another HIP stream overlap with the first?
Two engines (own workspaces) on two streams, K forwards each, against one engine doing 2K forwards on one stream.
Usage: EPCNET_LIB=/tmp/x.so python scripts/time_two_streams.py name"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
E = bench.pkg("engine")
dev = torch.device("cuda:0")
arch = os.environ.get("ARCH", "epc-net")
B = int(os.environ.get("BATCH", "64"))
K = int(os.environ.get("STEPS", "100"))
NS = int(os.environ.get("STREAMS", "2"))
store = bench.build_store(arch, dev, 0)
engs = [E.InferenceEngine(arch, bench.PARAMS, store, outer=bench.OUTER, micro_batch=B) for _ in range(NS)]
xyz = [(torch.rand((B, 4096, 3)) * 2 - 1).to(dev) for _ in range(NS)]
outs = [torch.empty((B, 256), device=dev) for _ in range(NS)]
streams = [torch.cuda.Stream() for _ in range(NS)]
for e, x, o in zip(engs, xyz, outs):
    for _ in range(3):
        e.forward(x, out=o)
torch.cuda.synchronize()

def run_single(steps):
    t0 = time.perf_counter()
    for _ in range(steps):
        engs[0].forward(xyz[0], out=outs[0])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

def run_multi(steps):
    t0 = time.perf_counter()
    for i in range(steps):
        k = i % NS
        with torch.cuda.stream(streams[k]):
            engs[k].forward(xyz[k], out=outs[k])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3

run_single(20); run_multi(20)
a = run_single(2 * K)
b = run_multi(2 * K)
print("%-20s batch %d: one stream %.3f ms/step, %d streams %.3f ms/step (x%.3f)" % (sys.argv[1] if len(sys.argv) > 1 else "", B, a, NS, b, a / b))
