"""Time the quadruplet training step (config c3) at full size on the GPU box: 1 + 2 + NEG + 1 clouds x 4096 points."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
neg = int(os.environ.get("NEG", "14"))
dev = torch.device("cuda:0")
arch = os.environ.get("ARCH", "epc-net")
store = bench.build_store(arch, dev, 0)
TR = bench.pkg("training")
params = dict(bench.PARAMS, ARCH=arch, TRAIN_PRECISION=os.environ.get("PRECISION", "bf16x6"), BATCH_NUM_QUERIES=1, DECAY_STEP=200000, BASE_LEARNING_RATE=5e-5, MARGIN_1=0.5, MARGIN_2=0.2)
if os.environ.get("ASSIGN_F16X3", "1") == "0":      # A/B: the assignment product in the six-product bf16 form instead of f16x3
    bench.pkg("ops").F16X3_ASSIGN = None
if "PERSIST" in os.environ:                          # A/B: the backbone chain as persistent launches (1) or as the launch chain (0); 
    bench.pkg("ops").CHAIN_PERSIST_FWD = os.environ["PERSIST"][0] == "1"
if "HIDDEN_TAIL" in os.environ:                      # A/B: the VLAD tail behind the hidden projection as one launch each way (1) or op by op (0)
    bench.pkg("ops").HIDDEN_TAIL = os.environ["HIDDEN_TAIL"] == "1"
if "REPLAYS_PER_SYNC" in os.environ:                 # (the step synchronises its stream every 32 replays: a runtime limit, training.py)
    TR.REPLAYS_PER_SYNC = int(os.environ["REPLAYS_PER_SYNC"])
ts = TR.TrainStep(params, store, outer=bench.OUTER)
g = torch.Generator().manual_seed(0)
mk = lambda p: (torch.rand((1, p, 4096, 3), generator=g) * 2 - 1).to(dev)
q, pos, ng, oth = mk(1), mk(2), mk(neg), mk(1)
use_graph = os.environ.get("GRAPH", "1") == "1"
for _ in range(int(os.environ.get("WARM", "30"))):   # warm-up: graph capture, lazy allocations, and the dispatch stall a fresh process can see
    loss, lr, bd = ts.step(q, pos, ng, oth, epoch=0, graph=use_graph)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = int(os.environ.get("STEPS", "60"))
for k_ in range(K):
    loss, lr, bd = ts.step(q, pos, ng, oth, epoch=0, graph=use_graph)
    if os.environ.get("SYNC_EVERY") and (k_ + 1) % int(os.environ["SYNC_EVERY"]) == 0:
        how = os.environ.get("SYNC_HOW", "device")        # debugging the long-run fault: which kind of wait keeps a long replay loop alive
        if how == "device":
            torch.cuda.synchronize()
        elif how == "stream":
            torch.cuda.current_stream().synchronize()
        elif how == "item":
            float(loss)
        elif how == "print":
            print("host at step", k_ + 1, flush=True)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
ncl = 1 + 2 + neg + 1
print(("train step %s (%s, " + params["TRAIN_PRECISION"] + "): %d clouds, %.2f ms/step, %.1f steps/s, %.0f clouds/s, loss %.4f, ~%.1f TFLOP/s (3x fwd FLOPs), peak mem %.2f GB")
      % (arch, "HIP graph" if use_graph else "eager", ncl, dt * 1e3, 1 / dt, ncl / dt, float(loss), 3 * bench.FLOPS_PER_CLOUD[arch] * ncl / dt / 1e12, torch.cuda.max_memory_allocated() / 2**30))
