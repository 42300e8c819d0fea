"""GPU box: conv5's training forward (73 728 x 256 -> 1024 + batch moments): epc_conv5_train_fwd against the statistics GEMM."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ops = bench.pkg("ops")
dev = torch.device("cuda:0")
rows = int(os.environ.get("ROWS", str(18 * 4096)))
g = torch.Generator().manual_seed(0)
x = torch.randn(rows, 256, generator=g).to(dev)
W = (torch.randn(256, 1024, generator=g) / 16).to(dev)
b = torch.randn(1024, generator=g).to(dev)


def t(fn, n=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


a = t(lambda: ops._conv5_train_fwd(x, W, b))
c = t(lambda: ops._gemm_with_stats(x, W, b, ops.F16X3_CONV5))
print("conv5 training forward, %d rows: own kernel (pack + product + finalize) %.1f us, statistics GEMM %.1f us" % (rows, a, c))
