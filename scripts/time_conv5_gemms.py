import os, sys
sys.path.insert(0, "/root/repo")
import torch, bench
ops = bench.pkg("ops")
dev = torch.device("cuda:0")
R = 18 * 4096
def timed(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x256 = torch.randn(R, 256, device=dev); dy = torch.randn(R, 1024, device=dev); w5 = torch.randn(256, 1024, device=dev)
print(sys.argv[1], "conv5 dW %.0f us" % timed(lambda: ops.gemm(x256, dy, trans_a=True, splitk=ops._splitk_for(256, 1024, R), fast=True, deterministic=True)),
      "dX %.0f us" % timed(lambda: ops.gemm(dy, w5, trans_b=True, fast=True)),
      "fwd6 %.0f us" % timed(lambda: ops.gemm(x256, w5)),
      "fwd+stats f16x3 %.0f us" % timed(lambda: ops._gemm_with_stats(x256, w5, None, ops.F16X3_CONV5)))
