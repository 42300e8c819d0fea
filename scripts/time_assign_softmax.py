"""GPU box: the soft assignment's element-wise launches (epc_assign_softmax_fwd / _bwd) through ops.Conv5VladHead's pieces is awkward to
isolate; this times the C entry points on random tensors at the training tuple's size (NEG + 4 clouds x 4096 points)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
L = bench.pkg("lib"); lib = L.lib()
dev = torch.device("cuda:0")
B, N = int(os.environ.get("NEG", "14")) + 4, 4096
R = B * N
z = torch.randn(R, 64, device=dev); a = torch.empty_like(z); da = torch.randn(R, 64, device=dev); dz = torch.empty_like(z)
mean, var = torch.zeros(64, device=dev), torch.ones(64, device=dev)
gamma, beta = torch.ones(64, device=dev), torch.zeros(64, device=dev)
a_sum = torch.empty(B * 64, device=dev); dsum = torch.randn(B, 64, device=dev)
parts = torch.empty(lib.epc_cloud_colsum64_partial_floats(B), device=dev)
dg, db, trow = torch.empty(64, device=dev), torch.empty(64, device=dev), torch.empty(R, device=dev)
ws = torch.empty(1 << 24, dtype=torch.uint8, device=dev)
st = L.current_stream()
def t(name, fn, reps=50):
    for _ in range(5): L.check(fn())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-28s %6.1f us" % (name, e0.elapsed_time(e1) / reps * 1e3), flush=True)
t("assign_softmax_fwd (+finish)", lambda: lib.epc_assign_softmax_fwd(z.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-3, B, N, a.data_ptr(), a_sum.data_ptr(), parts.data_ptr(), parts.numel(), st))
t("assign_softmax_bwd (all)", lambda: lib.epc_assign_softmax_bwd(da.data_ptr(), dsum.data_ptr(), a.data_ptr(), z.data_ptr(), mean.data_ptr(), var.data_ptr(), gamma.data_ptr(), beta.data_ptr(), 1e-3, B, N, dz.data_ptr(), dg.data_ptr(), db.data_ptr(), trow.data_ptr(), ws.data_ptr(), ws.numel(), st))
