import os, sys
sys.path.insert(0, "/root/repo")
import torch
import bench
ops = bench.pkg("ops")
dev = torch.device("cuda:0")
R = int(os.environ.get("ROWS", 18 * 4096))
def timed(fn, n=200):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g):
            for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n // 20): g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
x = torch.randn(R, 64, device=dev); w = torch.randn(64, 64, device=dev) / 8; b = torch.randn(64, device=dev)
z, mean, var = ops._gemm_with_stats(x, w, b)
gamma = torch.ones(64, device=dev); beta = torch.zeros(64, device=dev)
print(sys.argv[1:], "linear_stats64 %.1f us" % timed(lambda: ops._gemm_with_stats(x, w, b)))
y = torch.empty_like(x)
print("copy %.1f us" % timed(lambda: y.copy_(x)))
dy = torch.randn(R, 64, device=dev)
xg = x.clone().requires_grad_(True); wg = w.clone().requires_grad_(True); gg = gamma.clone().requires_grad_(True); bg = beta.clone().requires_grad_(True)
def fb():
    yy, m, v = ops.LinearBatchNormTrain.apply(xg, wg, b, gg, bg, 1e-3, 1, False)
    yy.backward(dy)
    xg.grad = None; wg.grad = None; gg.grad = None; bg.grad = None
print("layer fwd+bwd %.1f us" % timed(fb))
