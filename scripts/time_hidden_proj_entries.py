import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
L = bench.pkg("lib"); lib = L.lib()
dev = torch.device("cuda:0")
M, K = 72, 16384
x = torch.randn(M, K, device=dev); W = torch.randn(K, 256, device=dev) / 128; dy = torch.randn(M, 256, device=dev)
dx = torch.empty_like(x); dW = torch.empty_like(W); y = torch.empty(M, 256, device=dev)
sc = torch.empty(lib.epc_hidden_proj_scratch_bytes(M, K), dtype=torch.uint8, device=dev)
st = L.current_stream()
def t(name, fn, reps=50):
    for _ in range(5): L.check(fn())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-20s %6.1f us" % (name, e0.elapsed_time(e1) / reps * 1e3), flush=True)
for rep in range(2):
    for P in (1, 2, 3):
        t("fwd P=%d" % P, lambda: lib.epc_hidden_proj_fwd(x.data_ptr(), W.data_ptr(), M, K, P, y.data_ptr(), sc.data_ptr(), sc.numel(), st))
        t("dX  P=%d" % P, lambda: lib.epc_hidden_proj_bwd(x.data_ptr(), W.data_ptr(), dy.data_ptr(), M, K, P, dx.data_ptr(), None, st))
        t("dW  P=%d" % P, lambda: lib.epc_hidden_proj_bwd(x.data_ptr(), W.data_ptr(), dy.data_ptr(), M, K, P, None, dW.data_ptr(), st))
