"""Fused 64 -> 64 Linear + BatchNorm backward (epc_linear_bn_bwd64: column sums, the fused pass, the ordered dW sum) at
18 x 4096 rows, called straight through the C ABI so that the GPU, not the host, sets the pace: HIP-event time per call."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
ops = bench.pkg("ops")
L = bench.pkg("lib")
dev = torch.device("cuda:0")
R = int(os.environ.get("ROWS", 18 * 4096))
t = lambda *s: torch.randn(*s, device=dev)
dy, z, x, W = t(R, 64), t(R, 64), t(R, 64), t(64, 64) / 8
mean, var, gamma, beta = z.mean(0), z.var(0, unbiased=False), torch.ones(64, device=dev), torch.zeros(64, device=dev)
dx, dW, dg, db = torch.empty_like(x), torch.empty_like(W), torch.empty(64, device=dev), torch.empty(64, device=dev)
ws, n = ops._ws(R, 64, dev)
pf = L.lib().epc_linear_bn_bwd64_partial_floats(R)
part = torch.empty(pf, device=dev)
st = L.current_stream()
call = lambda: L.check(L.lib().epc_linear_bn_bwd64(dy.data_ptr(), z.data_ptr(), x.data_ptr(), W.data_ptr(), mean.data_ptr(),
                                                   var.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-3, 1, R, dx.data_ptr(),
                                                   dW.data_ptr(), dg.data_ptr(), db.data_ptr(), ws.data_ptr(), n, part.data_ptr(),
                                                   part.numel(), st))
for _ in range(10):
    call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(100):
    call()
e1.record()
torch.cuda.synchronize()
print("%s: epc_linear_bn_bwd64 %.1f us per call (3 launches)" % (sys.argv[1] if len(sys.argv) > 1 else "", e0.elapsed_time(e1) / 100 * 1e3))
