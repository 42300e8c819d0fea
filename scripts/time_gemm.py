"""Sweep split-K for the GEMM shapes of the training step's backward pass (rows = 18 clouds x 4096 points)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ops = bench.pkg("ops")
dev = torch.device("cuda:0")
R = int(os.environ.get("ROWS", 18 * 4096))


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def sweep(name, A, B, splits, **kw):
    M = A.shape[1] if kw.get("trans_a") else A.shape[0]
    K = A.shape[0] if kw.get("trans_a") else A.shape[1]
    N = B.shape[0] if kw.get("trans_b") else B.shape[1]
    for mode, fast, products in (("bf16", False, 1), ("fast", True, 3), ("full", False, 6)):
        prev = ops.set_gemm_precision("bf16" if mode == "bf16" else "bf16x6")
        row = []
        for s in splits:
            us = timed(lambda: ops.gemm(A, B, splitk=s, fast=fast, **kw))
            row.append("%d:%.0fus(%.0fTF)" % (s, us, 2.0 * M * N * K * products / us / 1e6))
        ops.set_gemm_precision(prev)
        print("%-28s M=%d N=%d K=%d %s  %s" % (name, M, N, K, mode, "  ".join(row)), flush=True)


x256 = torch.randn(R, 256, device=dev)
x64 = torch.randn(R, 64, device=dev)
dy1024 = torch.randn(R, 1024, device=dev)
dy64 = torch.randn(R, 64, device=dev)
w5 = torch.randn(256, 1024, device=dev)
w64 = torch.randn(64, 64, device=dev)
wc = torch.randn(1024, 64, device=dev)
sweep("conv5 dW = x^T dy", x256, dy1024, [16, 32, 48, 64, 96, 128], trans_a=True)
sweep("conv5 dX = dy W^T", dy1024, w5, [1], trans_b=True)
sweep("conv5 fwd = x W", x256, w5, [1])
sweep("layer dW = x^T dy", x64, dy64, [32, 64, 128, 256, 512], trans_a=True)
sweep("layer dX = dy W^T", dy64, w64, [1], trans_b=True)
sweep("assign fwd = f Wc", dy1024, wc, [1])
sweep("assign dWc = f^T da", dy1024, dy64, [16, 32, 64, 128], trans_a=True)
sweep("assign df = da Wc^T", dy64, wc, [1], trans_b=True)
