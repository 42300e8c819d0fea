"""Tuning aid (GPU box): wall-clock ms per epc_net_forward of BATCH x 4096 clouds (no stage profile; HIP events around 50 calls,
best of 5 regions after a warm-up).  Env: ARCH, PRECISION, BATCH, MICRO, INFLIGHT.  Usage: python scripts/time_step_plain.py name"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
E = bench.pkg("engine")
dev = torch.device("cuda:0")
arch = os.environ.get("ARCH", "epc-net")
store = bench.build_store(arch, dev, 0)
B = int(os.environ.get("BATCH", "64"))
eng = E.InferenceEngine(arch, bench.PARAMS, store, outer=bench.OUTER, micro_batch=int(os.environ.get("MICRO", str(B))),
                        in_flight=int(os.environ.get("INFLIGHT", "1")), precision=os.environ.get("PRECISION") or None)
xyz = (torch.rand((B, 4096, 3), generator=torch.Generator().manual_seed(0)) * 2 - 1).to(dev)
for _ in range(200):
    out = eng.forward(xyz)
torch.cuda.synchronize()
ms = []
for _ in range(5):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(100):
        out = eng.forward(xyz)
    b.record()
    torch.cuda.synchronize()
    ms.append(a.elapsed_time(b) / 100)
print("%-28s ms/step min %.4f median %.4f   checksum %.9f" % (sys.argv[1] if len(sys.argv) > 1 else "", min(ms), sorted(ms)[2],
                                                           float(out.double().sum())))
