import ctypes, os, sys
import numpy as np, torch
here = os.path.dirname(os.path.abspath(__file__))
lib = ctypes.CDLL(os.path.join(here, "mfma_fp8_probe.so"))
lib.run_probe.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
# small integers are exact in e4m3 (|v| <= 8 with 3 mantissa bits: integers up to 16 exact)
A = rng.integers(-4, 5, (32, 64)).astype(np.float32)
B = rng.integers(-4, 5, (64, 32)).astype(np.float32)
ref = A @ B
for sa, sb in ((0, 0), (0x7f7f7f7f, 0x7f7f7f7f), (127, 127), (0x80808080 - 0x100000000, 0x7f7f7f7f), (126, 127)):
    a, b = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    d = torch.zeros((32, 32), device=dev)
    rc = lib.run_probe(a.data_ptr(), b.data_ptr(), d.data_ptr(), sa, sb, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    D = d.cpu().numpy()
    ratio = D[ref != 0] / ref[ref != 0]
    print("scale_a=%#x scale_b=%#x rc=%d: max|D-ref|=%.3g  ratio D/ref: min %.4g max %.4g" % (sa & 0xffffffff, sb & 0xffffffff, rc, np.abs(D - ref).max(), ratio.min(), ratio.max()))
