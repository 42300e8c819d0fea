"""GPU: the VLAD tail behind the hidden projection (loupe.py:323-331 + :61-101) as the fused node (ops.HiddenTail: one launch each way) and as
the per-op path, forward + backward, timed with HIP events over graph-free back-to-back calls.  B=22 G=4 O=256 by default."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
import helpers as H
ops, L = H.pkg("ops"), H.pkg("lib")
dev = torch.device("cuda:0")
B, G, O = int(os.environ.get("B", "22")), int(os.environ.get("G", "4")), int(os.environ.get("O", "256"))
ops.set_gemm_precision(os.environ.get("PRECISION", "bf16"))
g = torch.Generator().manual_seed(0)
h = torch.randn((B * G, O), generator=g).to(dev)
g1, b1, g2, b2 = (torch.rand(O, generator=g).to(dev) + 0.5 for _ in range(4))
Wg = (torch.randn((O, O), generator=g) / 16).to(dev)
dout = torch.randn((B, O), generator=g).to(dev)
lib = L.lib()
vec = lambda: torch.empty(O, device=dev)
mat = lambda: torch.empty((B, O), device=dev)
m1, v1, v1u, m2, v2, v2u, v, gl, out = vec(), vec(), vec(), vec(), vec(), vec(), mat(), mat(), mat()
dh, dWg, dg1, db1, dg2, db2 = torch.empty_like(h), torch.empty_like(Wg), vec(), vec(), vec(), vec()
st = L.current_stream()
fwd = lambda: L.check(lib.epc_hidden_tail_fwd(h.data_ptr(), B, G, O, g1.data_ptr(), b1.data_ptr(), Wg.data_ptr(), g2.data_ptr(), b2.data_ptr(), 1e-3,
                                              1.0, 1.0, m1.data_ptr(), v1.data_ptr(), v1u.data_ptr(), v.data_ptr(), gl.data_ptr(), m2.data_ptr(), v2.data_ptr(),
                                              v2u.data_ptr(), out.data_ptr(), st))
bwd = lambda: L.check(lib.epc_hidden_tail_bwd(dout.data_ptr(), h.data_ptr(), B, G, O, g1.data_ptr(), m1.data_ptr(), v1.data_ptr(), v.data_ptr(), gl.data_ptr(),
                                              Wg.data_ptr(), g2.data_ptr(), b2.data_ptr(), m2.data_ptr(), v2.data_ptr(), 1e-3, dh.data_ptr(), dg1.data_ptr(),
                                              db1.data_ptr(), dWg.data_ptr(), dg2.data_ptr(), db2.data_ptr(), st))
for name, fn in (("epc_hidden_tail_fwd", fwd), ("epc_hidden_tail_bwd", bwd)):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        fn()
    e1.record(); torch.cuda.synchronize()
    print("%-24s B=%d G=%d O=%d %s: %.1f us per launch (back to back)" % (name, B, G, O, ops._GEMM_PRECISION, e0.elapsed_time(e1) * 5))
