"""GPU box: which torch (aten) operators still launch kernels inside one eager training step, with the source line that calls them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
import bench
dev = torch.device("cuda:0")
arch = os.environ.get("ARCH", "epc-net")
store = bench.build_store(arch, dev, 0)
TR = bench.pkg("training")
params = dict(bench.PARAMS, ARCH=arch, BATCH_NUM_QUERIES=1, DECAY_STEP=200000, BASE_LEARNING_RATE=5e-5, MARGIN_1=0.5, MARGIN_2=0.2)
ts = TR.TrainStep(params, store, outer=bench.OUTER)
g = torch.Generator().manual_seed(0)
mk = lambda p: (torch.rand((1, p, 4096, 3), generator=g) * 2 - 1).to(dev)
q, pos, ng, oth = mk(1), mk(2), mk(14), mk(1)
for _ in range(3):
    ts.step(q, pos, ng, oth, epoch=0)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
    ts.step(q, pos, ng, oth, epoch=0)
    torch.cuda.synchronize()
ka = prof.key_averages(group_by_input_shape=True, group_by_stack_n=8)
rows = []
for e in ka:
    t = getattr(e, "self_device_time_total", 0) or getattr(e, "self_cuda_time_total", 0)
    if e.key.startswith("aten::") and t > 0:
        here = [f for f in (e.stack or []) if "epc-net_amd" in f]
        rows.append((e.key, e.count, t, str(e.input_shapes)[:60], here[0].strip()[-80:] if here else "?"))
rows.sort(key=lambda r: -r[2])
print("aten operators with self device time in one eager step: %d kinds, %d calls, %.1f us" % (len(rows), sum(r[1] for r in rows), sum(r[2] for r in rows)))
for r in rows:
    print("%-28s x%-2d %7.1f us  %-60s %s" % r)
