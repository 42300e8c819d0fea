"""Thin-layer (64 -> 64) GEMMs of the training step at 18 x 4096 rows: atomic vs ordered split-K for dW, forward with / without
the statistics epilogue, dX.  HIP-event times per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ops = bench.pkg("ops")
dev = torch.device("cuda:0")
R = int(os.environ.get("ROWS", 18 * 4096))


def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


x = torch.randn(R, 64, device=dev)
dy = torch.randn(R, 64, device=dev)
w = torch.randn(64, 64, device=dev)
b = torch.randn(64, device=dev)
for s in (16, 32, 64, 128, 256):
    print("dW splitk %3d: atomic %.1f us   ordered %.1f us" % (
        s, timed(lambda: ops.gemm(x, dy, trans_a=True, splitk=s, fast=True)),
        timed(lambda: ops.gemm(x, dy, trans_a=True, splitk=s, fast=True, deterministic=True))))
print("dX  %.1f us" % timed(lambda: ops.gemm(dy, w, trans_b=True, fast=True)))
print("fwd %.1f us   fwd + stats %.1f us" % (timed(lambda: ops.gemm(x, w, bias=b)), timed(lambda: ops._gemm_with_stats(x, w, b))))
y = torch.empty_like(x)
print("copy (read + write one tensor) %.1f us" % timed(lambda: y.copy_(x)))
