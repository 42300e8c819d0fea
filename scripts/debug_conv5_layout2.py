import os, sys, ctypes
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import helpers as H
from helpers import O
dev = torch.device("cuda:0")
pc = O.synthetic_clouds(3, 256, 5, "uniform")
w = O.seeded_weights("epc-net", 1)
ref, st = O.forward(pc[:, None], w, arch="epc-net")
eng, _ = H.make_engine("epc-net", w, dev, precision="f32")
L = H.pkg("lib"); lib = L.lib()
xyz = torch.from_numpy(pc).to(dev)
got = H.run_stages(eng, xyz)
cfg = eng.cfg_for(256); packed = eng.packed(cfg)
off5 = packed.data_ptr() + lib.epc_net_packed_offset(ctypes.byref(cfg), 5)
M = 768
cat = got["cat"]
featf = torch.zeros((M // 32, 32, 3, 64, 16), dtype=torch.uint8, device=dev)
rnorm = torch.empty((M,), device=dev); assign = torch.empty((M, 64), device=dev)
assignf = torch.empty((M // 32, 2, 2, 2, 64, 8), dtype=torch.bfloat16, device=dev); apart = torch.empty((M // 32, 64), device=dev)
L.check(lib.epc_conv5_assign_f32_fwd(cat.data_ptr(), 256, off5, M, featf.data_ptr(), rnorm.data_ptr(), assign.data_ptr(), assignf.data_ptr(), apart.data_ptr(), L.current_stream()))
torch.cuda.synchronize()
by = featf[0, 0].permute(1, 0, 2).reshape(64, 16, 3).to(torch.int32)       # (lane, value, byte) of tile 0 chunk 0
bits = (by[..., 0] << 8) | (by[..., 1] << 16) | (by[..., 2] << 24)
vals = bits.view(torch.float32).cpu().numpy()                                # (64 lanes, 16 values)
b = st.taps["fastdgcnn/conv5"].reshape(-1, 1024)[:32, :32]                   # true [point][channel] of tile 0 chunk 0
for lane in (0, 1, 16, 17, 32, 48, 63):
    row = []
    for v in range(16):
        x = vals[lane, v]
        hits = np.argwhere(np.abs(b - x) < 2e-5 * max(abs(x), 1e-3)) if x != 0 else []
        row.append("%s" % (["p%d.c%d" % (h[0], h[1]) for h in hits][:2] if len(hits) else ("0" if x == 0 else "?")))
    print("lane", lane, row)
