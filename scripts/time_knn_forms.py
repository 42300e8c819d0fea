"""GPU box: epc_knn_topk in its one-lane and four-lane forms (epc_knn_topk_form) at several batch sizes of Hilbert-ordered
4096-point clouds; checks that the two forms write identical outputs."""
import ctypes, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.environ.get("EPCNET_LIB") or os.path.join(ROOT, "epc-net_amd", "libepcnet_hip.so"))
P = ctypes.c_void_p
lib.epc_knn_topk_form.argtypes = [P, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P, P, ctypes.c_int, P]
lib.epc_morton_sort.argtypes = [P, ctypes.c_int, ctypes.c_int, P, P, P]
dev = torch.device("cuda:0")
N = 4096
st = torch.cuda.current_stream().cuda_stream
for B in [int(a) for a in sys.argv[1:]] or [18, 22, 64, 256]:
    g = torch.Generator().manual_seed(0)
    xyz = (torch.rand((B, N, 3), generator=g) * 2 - 1).to(dev)
    srt = torch.empty_like(xyz)
    assert lib.epc_morton_sort(xyz.data_ptr(), B, N, srt.data_ptr(), None, st) == 0
    res = {}
    for form in (0, 1):
        idx = torch.zeros((B, N, 32), dtype=torch.int32, device=dev)
        cnt = torch.zeros((B, N), dtype=torch.int32, device=dev)
        kth = torch.zeros((B, N), dtype=torch.float32, device=dev)
        for _ in range(3):
            assert lib.epc_knn_topk_form(srt.data_ptr(), B, N, 32, idx.data_ptr(), cnt.data_ptr(), kth.data_ptr(), form, st) == 0
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            lib.epc_knn_topk_form(srt.data_ptr(), B, N, 32, idx.data_ptr(), cnt.data_ptr(), kth.data_ptr(), form, st)
        e1.record(); torch.cuda.synchronize()
        res[form] = (e0.elapsed_time(e1) / 20, idx, cnt, kth)
    same = all(torch.equal(res[0][k], res[1][k]) for k in (1, 2, 3))
    print("knn %3d clouds: one lane %.3f ms, four lanes %.3f ms, identical outputs: %s" % (B, res[0][0], res[1][0], same), flush=True)
    if hasattr(lib, "epc_debug_knn_stats"):
        for form in (0, 1):
            st8 = (ctypes.c_ulonglong * 8)()
            lib.epc_debug_knn_stats(st8, 1)
            lib.epc_knn_topk_form(srt.data_ptr(), B, N, 32, idx.data_ptr(), cnt.data_ptr(), kth.data_ptr(), form, st); torch.cuda.synchronize()
            lib.epc_debug_knn_stats(st8, 1)
            waves = B * N / (64 if form == 0 else 16)
            print("    form %d per wave: votes %.1f, tiles scanned %.1f | hit batches %.1f, network passes %.1f | p2 tiles %.1f, emit batches %.1f"
                  % ((form,) + tuple(st8[i] / waves for i in range(6))))
