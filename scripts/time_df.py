"""GPU: the VLAD feature-gradient product (epc_vlad_df) against the generic GEMM form, 18 x 4096 rows."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
L = bench.pkg("lib"); ops = bench.pkg("ops")
lib = L.lib(); dev = torch.device("cuda:0")
B, N, F = 18, 4096, 1024
g = torch.Generator(device=dev); g.manual_seed(0)
a = torch.rand((B * N, 64), generator=g, device=dev); dz = torch.randn((B * N, 64), generator=g, device=dev)
dvlad = torch.randn((B, F, 64), generator=g, device=dev); Wc = torch.randn((F, 64), generator=g, device=dev)
df = torch.empty((B * N, F), device=dev)
nb = lib.epc_vlad_df_packed_bytes(B, F); packed = torch.empty(nb, dtype=torch.uint8, device=dev)
st = L.current_stream()
def new():
    L.check(lib.epc_vlad_df(a.data_ptr(), dz.data_ptr(), dvlad.data_ptr(), Wc.data_ptr(), B, N, F, 2, packed.data_ptr(), nb, df.data_ptr(), st))
def old():
    lhs = torch.cat((a, dz), dim=1).view(B, N, 128)
    rhs = torch.cat((dvlad.transpose(1, 2), Wc.t().unsqueeze(0).expand(B, 64, F)), dim=1)
    return ops.gemm(lhs, rhs, fast=True).view(B * N, F)
ref = old(); new(); torch.cuda.synchronize()
print("max diff vs generic form: %.3e of %.3e" % (float((df - ref).abs().max()), float(ref.abs().max())))
for name, fn in (("epc_vlad_df", new), ("cat + generic gemm", old)):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): fn()
    e1.record(); torch.cuda.synchronize()
    print("%-20s %7.1f us" % (name, e0.elapsed_time(e1) / 20 * 1e3))
