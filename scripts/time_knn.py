"""Time epc_knn_topk of a given library build on 64 Z-ordered 4096-point clouds (tuning aid, GPU box only)."""
import ctypes, sys, torch
lib = ctypes.CDLL(sys.argv[1])
P = ctypes.c_void_p
lib.epc_knn_topk.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P, P, P]
lib.epc_morton_sort.argtypes = [P, ctypes.c_int, ctypes.c_int, P, P, P]
dev = torch.device("cuda:0")
import os
B, N = int(os.environ.get("BATCH", "64")), 4096
for kind in ("uniform", "unsorted"):
    g = torch.Generator().manual_seed(0)
    xyz = (torch.rand((B, N, 3), generator=g) * 2 - 1).to(dev)
    srt = torch.empty_like(xyz)
    st = torch.cuda.current_stream().cuda_stream
    if kind == "uniform":
        assert lib.epc_morton_sort(xyz.data_ptr(), B, N, srt.data_ptr(), None, st) == 0
    else:
        srt = xyz
    idx = torch.empty((B, N, 32), dtype=torch.int32, device=dev)
    cnt = torch.empty((B, N), dtype=torch.int32, device=dev)
    kth = torch.empty((B, N), dtype=torch.float32, device=dev)
    for _ in range(3):
        assert lib.epc_knn_topk(srt.data_ptr(), B, N, 32, idx.data_ptr(), cnt.data_ptr(), kth.data_ptr(), st) == 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.epc_knn_topk(srt.data_ptr(), B, N, 32, idx.data_ptr(), cnt.data_ptr(), kth.data_ptr(), st)
    e1.record(); torch.cuda.synchronize()
    print("%-24s %-9s %.3f ms / %d clouds   (cnt sum %d)" % (sys.argv[2], kind, e0.elapsed_time(e1) / 10, B, int(cnt.sum())))
    if hasattr(lib, "epc_debug_knn_stats"):
        st8 = (ctypes.c_ulonglong * 8)()
        lib.epc_debug_knn_stats(st8, 1)
        lib.epc_knn_topk(srt.data_ptr(), B, N, 32, idx.data_ptr(), cnt.data_ptr(), kth.data_ptr(), st); torch.cuda.synchronize()
        lib.epc_debug_knn_stats(st8, 1)
        waves = B * N / 64
        print("    per wave: counters 0..5 = %.1f %.1f | %.1f %.2f | %.1f %.3f   (culled kernel: tiles tested, scanned | batches w/ insert, network passes | "
              "p2 tiles, batches w/ emit;  collect kernel: tiles tested, scanned | hit batches, compactions | select-loop entries, fallback waves)"
              % tuple(st8[i] / waves for i in range(6)))
