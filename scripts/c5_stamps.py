"""GPU box, diagnostic build only (scripts/build_variant.sh stamps conv5_f32.hip "-DC5_STAMPS"): where a conv5_f32 wave's
cycles go.  EPCNET_LIB=build_variants/lib_stamps.so python scripts/c5_stamps.py [arch]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
E = bench.pkg("engine"); L = bench.pkg("lib")
arch = sys.argv[1] if len(sys.argv) > 1 else "epc-net"
B = 64 if arch == "epc-net" else 256
dev = torch.device("cuda:0")
store = bench.build_store(arch, dev, 0)
eng = E.InferenceEngine(arch, bench.PARAMS, store, outer=bench.OUTER, micro_batch=B, precision="f32")
xyz = (torch.rand((B, 4096, 3)) * 2 - 1).to(dev)
for _ in range(20):
    eng.forward(xyz)
torch.cuda.synchronize()
lib = ctypes.CDLL(L.LIB_PATH)
n_waves = min(16384, B * 4096 // 32)
buf = np.zeros((16384, 8), dtype=np.uint32)
assert lib.epc_debug_c5_stamps(buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes)) == 0
s = buf[:n_waves].astype(np.float64)
names = ["prologue", "dma_issue", "mfma", "epilogue", "wait", "barrier", "final", "total"]
print(arch, "waves", n_waves, "median cycles per wave (sum over 32 chunks); share of total")
tot = np.median(s[:, 7])
for i, nm in enumerate(names):
    print("  %-10s %9.0f  %5.1f %%   (p10 %8.0f  p90 %8.0f)" % (nm, np.median(s[:, i]), 100 * np.median(s[:, i]) / tot, np.percentile(s[:, i], 10), np.percentile(s[:, i], 90)))
w0 = s[0::8]; w4 = s[4::8]
print("  waves 0 of the workgroups: mfma %.0f epi %.0f wait %.0f bar %.0f | waves 4: mfma %.0f epi %.0f wait %.0f bar %.0f" %
      (np.median(w0[:, 2]), np.median(w0[:, 3]), np.median(w0[:, 4]), np.median(w0[:, 5]), np.median(w4[:, 2]), np.median(w4[:, 3]), np.median(w4[:, 4]), np.median(w4[:, 5])))
