"""Tuning aid (GPU box): per-stage HIP-event times of epc_net_forward for 64 x 4096 clouds, no output checks
(ablation builds produce wrong descriptors on purpose).  Usage: EPCNET_LIB=/tmp/x.so python scripts/time_stages.py name"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
E = bench.pkg("engine")
dev = torch.device("cuda:0")
arch = os.environ.get("ARCH", "epc-net")
store = bench.build_store(arch, dev, 0)
eng = E.InferenceEngine(arch, bench.PARAMS, store, outer=bench.OUTER, micro_batch=int(os.environ.get("MICRO", os.environ.get("BATCH", "64"))),
                        precision=os.environ.get("PRECISION") or None)
B = int(os.environ.get("BATCH", "64"))
xyz = (torch.rand((B, 4096, 3)) * 2 - 1).to(dev)
for _ in range(5):
    eng.forward(xyz)
profs = [E.StageProfile() for _ in range(20)]
torch.cuda.synchronize()
for p in profs:
    eng.forward(xyz, profile=p)
torch.cuda.synchronize()
acc = {}
for p in profs:
    for k, v in p.elapsed_ms().items():
        acc[k] = acc.get(k, 0) + v / len(profs)
print("%-26s total=%.3f  " % (sys.argv[1] if len(sys.argv) > 1 else "", sum(acc.values())) + " ".join("%s=%.3f" % kv for kv in acc.items()))
