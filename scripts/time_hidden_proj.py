"""GPU box: the grouped hidden projection's three products (csrc/train_hidden.hip) against ops.Linear's tile GEMMs, microseconds per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ops = bench.pkg("ops")
dev = torch.device("cuda:0")
for clouds in (18, 22):
    M, K = 4 * clouds, 16384
    x = torch.randn(M, K, device=dev, requires_grad=True)
    W = (torch.randn(K, 256, device=dev) / 128).requires_grad_(True)
    dy = torch.randn(M, 256, device=dev)
    for prec in ("bf16x6", "bf16"):
        prev = ops.set_gemm_precision(prec)
        for name, fn in (("skinny", lambda: ops.HiddenProjection.apply(x, W)), ("tile GEMM", lambda: ops.Linear.apply(x, W, None))):
            def run(backward):
                y = fn()
                if backward:
                    x.grad = W.grad = None
                    y.backward(dy)
            res = []
            for backward in (False, True):
                for _ in range(3):
                    run(backward)
                graph = torch.cuda.CUDAGraph()       # (replayed: the launch overhead of the Python wrappers is not what is measured)
                s = torch.cuda.Stream()
                s.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(s):
                    run(backward)
                    with torch.cuda.graph(graph, stream=s):
                        run(backward)
                torch.cuda.current_stream().wait_stream(s)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                for _ in range(5):
                    graph.replay()
                e0.record()
                for _ in range(50):
                    graph.replay()
                e1.record()
                torch.cuda.synchronize()
                res.append(e0.elapsed_time(e1) / 50 * 1e3)
            print("%2d clouds %-7s %-10s forward %6.1f us   forward + backward %6.1f us" % (clouds, prec, name, res[0], res[1]), flush=True)
        ops.set_gemm_precision(prev)
