import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import helpers as H
from helpers import O
ops, L = H.pkg("ops"), H.pkg("lib")
dev = torch.device("cuda:0")
ncl, n = int(os.environ.get("NCL", "18")), 4096
x = ops.morton_sort(torch.from_numpy(O.synthetic_clouds(ncl, n, 17)).to(dev))
g = ops.KnnGraph(x)
rdeg, roff, cursor, rlist, oc, ol = g._build_transposed()
lib = L.lib(); st = L.current_stream()
def run():
    L.check(lib.epc_knn_transpose(g.idx.data_ptr(), g.cnt.data_ptr(), 32, ncl, n, rdeg.data_ptr(), roff.data_ptr(), cursor.data_ptr(), rlist.data_ptr(), st))
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50): run()
e1.record(); torch.cuda.synchronize()
print("epc_knn_transpose %d x %d: %.1f us per call (3 launches back to back)" % (ncl, n, e0.elapsed_time(e1) * 20))
