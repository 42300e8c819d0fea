"""GPU box: the streamed head's kernels one by one at the training tuple's size (NEG + 4 clouds x 4096 points), both arithmetics,
through the C ABI on random tensors: microseconds per call (HIP events over 20 calls) and the bytes each call has to move."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
L = bench.pkg("lib")
lib = L.lib()
dev = torch.device("cuda:0")
B, N = int(os.environ.get("NEG", "14")) + 4, 4096
R = B * N
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
cat, W5, b5 = rnd(R, 256), rnd(256, 1024) * 0.08, rnd(1024) * 0.1
g5, bt5, Wc = 1 + 0.2 * rnd(1024), 0.3 * rnd(1024), rnd(1024, 64) * 0.2
dvlad, C, rn, trow = rnd(B, 1024, 64), torch.rand(R, 64, generator=g).to(dev), torch.rand(R, generator=g).to(dev) + 0.5, rnd(R)
mean5, var5 = torch.zeros(1024, device=dev), torch.ones(1024, device=dev)
sc = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
st = L.current_stream()
EPS = 1e-3


def t(name, nbytes, fn, reps=20):
    for _ in range(3):
        L.check(fn())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print("%-34s %7.1f us   %6.0f MB   %5.2f TB/s" % (name, us, nbytes / 1e6, nbytes / us / 1e6), flush=True)


bn = (mean5.data_ptr(), var5.data_ptr(), g5.data_ptr(), bt5.data_ptr(), EPS)
for mode in os.environ.get("MODES", "bf16,f32").split(","):
    h16 = mode == "bf16"
    w = 2 if h16 else 4
    z5 = torch.empty((R, 1024), dtype=torch.bfloat16 if h16 else torch.float32, device=dev)
    du = torch.empty_like(z5)
    m5, v5, za, rno, mc, vc = (torch.empty(n, device=dev) for n in (1024, 1024, R * 64, R, 64, 64))
    out64, vl, dWc, dcat, dW5, sums = (torch.empty(n, device=dev) for n in (R * 64, B * 65536, 65536, R * 256, 262144, 2048))
    cat16 = cat.to(torch.bfloat16)
    print("---- %s head, %d clouds x %d points" % (mode, B, N))
    if h16:
        t("conv5 fwd (bf16 cat)", R * (256 * 2 + 1024 * 2), lambda: lib.epc_h16_conv5_fwd(cat16.data_ptr(), 1, W5.data_ptr(), b5.data_ptr(), R, z5.data_ptr(), m5.data_ptr(), v5.data_ptr(), sc.data_ptr(), sc.numel(), st))
        t("conv5 fwd (f32 cat)", R * (256 * 4 + 1024 * 2), lambda: lib.epc_h16_conv5_fwd(cat.data_ptr(), 0, W5.data_ptr(), b5.data_ptr(), R, z5.data_ptr(), m5.data_ptr(), v5.data_ptr(), sc.data_ptr(), sc.numel(), st))
        assign, colg, dx = lib.epc_h16_assign, lib.epc_h16_colgemm, lib.epc_h16_conv5_dx
    else:
        t("conv5 fwd", R * (256 * 4 + 1024 * 4), lambda: lib.epc_h32_conv5_fwd(cat.data_ptr(), W5.data_ptr(), b5.data_ptr(), R, z5.data_ptr(), m5.data_ptr(), v5.data_ptr(), sc.data_ptr(), sc.numel(), st))
        assign, colg, dx = lib.epc_h32_assign, lib.epc_h32_colgemm, lib.epc_h32_conv5_dx
    t("assign fwd (z5 -> za, rn, moments)", R * (1024 * w + 64 * 4), lambda: assign(z5.data_ptr(), *bn, Wc.data_ptr(), 0, B, N, za.data_ptr(), rno.data_ptr(), mc.data_ptr(), vc.data_ptr(), sc.data_ptr(), sc.numel(), st))
    t("assign bwd (z5, dvlad -> da)", R * (1024 * w + 64 * 4), lambda: assign(z5.data_ptr(), *bn, dvlad.data_ptr(), 1, B, N, out64.data_ptr(), None, None, None, sc.data_ptr(), sc.numel(), st))
    t("aggregate (z5, a -> vlad)", R * (1024 * w + 64 * 4), lambda: colg(z5.data_ptr(), *bn, C.data_ptr(), rn.data_ptr(), B, N, 1, vl.data_ptr(), sc.data_ptr(), sc.numel(), st))
    t("dWc (z5, dz -> dWc)", R * (1024 * w + 64 * 4), lambda: colg(z5.data_ptr(), *bn, C.data_ptr(), rn.data_ptr(), B, N, 0, dWc.data_ptr(), sc.data_ptr(), sc.numel(), st))
    if h16:
        t("df tail (a, dz, z5 -> du, sums)", R * (2 * 1024 * w + 2 * 64 * 4), lambda: lib.epc_h16_df_tail(C.data_ptr(), C.data_ptr(), dvlad.data_ptr(), Wc.data_ptr(), B, N, z5.data_ptr(), rn.data_ptr(), trow.data_ptr(), *bn, du.data_ptr(), sums.data_ptr(), sc.data_ptr(), sc.numel(), st))
        t("bn bwd apply (du, z5 -> dz5)", R * 3 * 1024 * w, lambda: lib.epc_h16_bn_bwd_apply(du.data_ptr(), z5.data_ptr(), *bn, sums.data_ptr(), sums.data_ptr() + 4096, R, du.data_ptr(), st))
        t("dW5 (cat16, dz5 -> dW5)", R * (256 * 2 + 1024 * w), lambda: lib.epc_h16_conv5_dw(cat16.data_ptr(), 1, du.data_ptr(), R, dW5.data_ptr(), sc.data_ptr(), sc.numel(), st))
    t("dx (dz5 -> dcat)", R * (1024 * w + 256 * 4), lambda: dx(du.data_ptr(), W5.data_ptr(), R, dcat.data_ptr(), sc.data_ptr(), sc.numel(), st))
    dxbn = lib.epc_h16_conv5_dx_bn if h16 else lib.epc_h32_conv5_dx_bn
    t("dx + bn bwd (du, z5 -> dz5, dcat)", R * (3 * 1024 * w + 256 * 4), lambda: dxbn(du.data_ptr(), z5.data_ptr(), mean5.data_ptr(), var5.data_ptr(), g5.data_ptr(), EPS, sums.data_ptr(), sums.data_ptr() + 4096, W5.data_ptr(), R, du.data_ptr(), dcat.data_ptr(), sc.data_ptr(), sc.numel(), st))
    if not h16:
        t("dW5 (cat, dz5 -> dW5)", R * (256 * 4 + 1024 * w), lambda: lib.epc_h32_conv5_dw(cat.data_ptr(), du.data_ptr(), R, dW5.data_ptr(), sc.data_ptr(), sc.numel(), st))
