"""GPU box: which (point, channel) classes of conv5_f32's feat disagree with the oracle (layout debugging aid)."""
import os, sys
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
got = H.run_stages(eng, torch.from_numpy(pc).to(dev))
a = got["feat"].cpu().numpy().reshape(-1, 1024)
b = st.taps["fastdgcnn/conv5"].reshape(-1, 1024)
bad = np.abs(a - b) > 1e-4 * np.abs(b).max()
print("bad fraction", bad.mean())
pt, ch = np.nonzero(bad)
for name, v, mod in (("p", (pt % 32) // 16, 2), ("li", pt % 16, 16), ("tile", pt // 32, 24), ("c", ch // 32, 32), ("g2", (ch % 32) // 16, 2),
                     ("q", (ch % 16) // 4, 4), ("r", ch % 4, 4)):
    print(name, np.bincount(v, minlength=mod))
# is a bad value found elsewhere in the same point's row / the same channel's column?
for k in range(min(5, len(pt))):
    i, j = pt[k], ch[k]
    row_hit = np.nonzero(np.abs(b[i] - a[i, j]) < 1e-5)[0]
    col_hit = np.nonzero(np.abs(b[:, j] - a[i, j]) < 1e-5)[0]
    print("bad (pt %d, ch %d) got %.6f want %.6f; equals ref at channels %s of this point, points %s of this channel" %
          (i, j, a[i, j], b[i, j], row_hit[:6], col_hit[:6]))
for what in ("assign", "vlad"):
    x = got[what].cpu().numpy().reshape(-1)
    y = (st.taps["vlad_assign"] if what == "assign" else st.taps["vlad_raw"]).reshape(-1)
    print(what, "rel max err", np.abs(x - y).max() / np.abs(y).max())
# mapping got-channel -> true-channel inside chunk 0, using all points of tile 0 (values must match on every point)
for c in (0, 1):
    m = []
    for gi in range(32):
        best = [ti for ti in range(32) if np.abs(a[:32, 32 * c + gi] - b[:32, 32 * c + ti]).max() < 1e-4]
        m.append(best[0] if len(best) == 1 else (-1 if not best else -2))
    print("chunk", c, "got->true", m)
# the same per point: does point i of got equal some other point of ref (channel 0..31)?
pm = []
for i in range(32):
    best = [t for t in range(32) if np.abs(a[i, :1024] - b[t, :1024]).max() < 1e-4]
    pm.append(best[0] if len(best) == 1 else -1)
print("point got->true", pm)
