"""GPU box: the kernels of one Oxford-scale retrieval call (9 200 x 2 760 x 256, k = 25), for rocprofv3 --kernel-trace."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
R = bench.pkg("retrieval")
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(31)
unit = lambda n: torch.nn.functional.normalize(torch.randn((n, 256), generator=g), dim=1).to(dev)
db, q = unit(9200), unit(2760)
for _ in range(30):
    idx = R.knn_search(db, q, 25)
torch.cuda.synchronize()
print("done", tuple(idx.shape) if hasattr(idx, "shape") else len(idx))
