"""GPU box, diagnostic build only (scripts/build_variant.sh gstamps train_ops.hip "-DGEMM_STAMPS"): where a wave of the split
GEMM kernel spends its cycles, for the three conv5 products of the training step.
EPCNET_LIB=build_variants/lib_gstamps.so python scripts/gemm_stamps.py"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
ops = bench.pkg("ops"); L = bench.pkg("lib")
dev = torch.device("cuda:0")
R = 18 * 4096
x256 = torch.randn(R, 256, device=dev); dy = torch.randn(R, 1024, device=dev); w5 = torch.randn(256, 1024, device=dev)
lib = ctypes.CDLL(L.LIB_PATH)
names = ["store", "barrier1", "fetch", "mfma", "barrier2", "epilogue", "total", "loop"]


def report(name, fn, waves):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    buf = np.zeros((32768, 8), dtype=np.uint32)
    assert lib.epc_debug_gemm_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
    s = buf[:min(waves, 32768)].astype(np.float64)
    tot = np.median(s[:, 6])
    print(name, "waves", waves, " median cycles per wave; share of total")
    for i, nm in enumerate(names):
        print("  %-9s %9.0f  %5.1f %%" % (nm, np.median(s[:, i]), 100 * np.median(s[:, i]) / tot))


report("conv5 dX = dz W^T (M 73728, N 256, K 1024)", lambda: ops.gemm(dy, w5, trans_b=True, fast=True), 2 * 576 * 4)
report("conv5 fwd f16x3 + stats (M 73728, N 1024, K 256)", lambda: ops._gemm_with_stats(x256, w5, None, ops.F16X3_CONV5), 8 * 576 * 4)
