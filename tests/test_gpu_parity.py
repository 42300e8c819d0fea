"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars: bit-exact for the kNN graph (integer / selection work); descriptor L2 error <= 1e-4 in fp32
(BASELINE.json north_star) -- measured error is ~1e-6, the fp32-vs-fp64 budget of the oracle itself is ~2e-7.
"""
import ctypes

import numpy as np
import pytest
import torch

import helpers as H
from helpers import O

pytestmark = pytest.mark.gpu

DESC_TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X; there is no CPU fallback"
    return torch.device("cuda:0")


@pytest.mark.parametrize("kind,n,seed", [("uniform", 64, 0), ("uniform", 256, 1), ("uniform", 4096, 2),
                                         ("lidar", 4096, 0), ("lattice", 512, 0), ("dup", 256, 0), ("zeros", 128, 0),
                                         ("uniform", 96, 3)])
def test_knn_bit_exact(dev, kind, n, seed):
    tf_util = H.pkg("utils.tf_util")
    pc = O.synthetic_clouds(2, n, seed, kind)
    kth_ref, lists = O.knn_lists(pc)
    kth, idx, cnt = tf_util.knn_index(torch.from_numpy(pc).to(dev))
    kth, idx, cnt = kth.cpu().numpy(), idx.cpu().numpy(), cnt.cpu().numpy()
    assert np.array_equal(kth, kth_ref), "kth differs (== on float32: bit-exact up to the sign of zero)"
    for b in range(pc.shape[0]):
        for i in range(n):
            ref = lists[b][i]
            assert cnt[b, i] == len(ref)
            m = min(len(ref), idx.shape[-1])
            assert np.array_equal(idx[b, i, :m], ref[:m])


@pytest.mark.parametrize("kind,n", [("uniform", 192), ("dup", 128), ("lidar", 2048)])
def test_knn_wide_grid_form_is_bit_exact(dev, kind, n):
    """Grids of two or more workgroups per CU (here 640 / 520 clouds; EPC-Net-L's batch 256 x 4096 in production) run the kNN
    kernel's 4-candidates-per-vote instantiation, two workgroups to a CU: thresholds, counts and lists must be those of the
    oracle -- i.e. of the 8-wide form the small grids of the other tests run -- bit for bit, conv1 fused or not."""
    L = H.pkg("lib")
    lib = L.lib()
    nc = 640 if n < 1024 else 260          # (260 clouds x 2 workgroups = 520 workgroups on 256 CUs)
    pc = O.synthetic_clouds(nc, n, 11, kind)
    xyz = torch.from_numpy(pc).to(dev)
    idx = torch.zeros((nc, n, 32), dtype=torch.int32, device=dev)
    cnt = torch.zeros((nc, n), dtype=torch.int32, device=dev)
    kth = torch.zeros((nc, n), device=dev)
    L.check(lib.epc_knn_topk(xyz.data_ptr(), nc, n, 32, idx.data_ptr(), cnt.data_ptr(), kth.data_ptr(), L.current_stream()))
    torch.cuda.synchronize()
    kth_c, idx_c, cnt_c = kth.cpu().numpy(), idx.cpu().numpy(), cnt.cpu().numpy()
    check = range(nc) if n < 1024 else (0, 1, nc // 2, nc - 1)
    kth_ref, lists = O.knn_lists(pc[list(check)])
    for bi, b in enumerate(check):
        assert np.array_equal(kth_c[b], kth_ref[bi])
        for i in range(n):
            ref = lists[bi][i]
            assert cnt_c[b, i] == len(ref)
            m = min(len(ref), 32)
            assert np.array_equal(idx_c[b, i, :m], ref[:m])
    # the fused conv1 form with 2-byte lists on the same grid
    eng, _ = H.make_engine("epc-net-l", O.seeded_weights("epc-net-l", 0), dev)
    cfg = eng.cfg_for(n)
    pk = eng.packed(cfg).data_ptr() + lib.epc_net_packed_offset(ctypes.byref(cfg), 0)
    idx2 = torch.zeros((nc, n, 32), dtype=torch.int16, device=dev)
    cnt2, kth2 = torch.zeros_like(cnt), torch.zeros_like(kth)
    x1 = torch.zeros((nc * n, 64), device=dev)
    status = torch.zeros((nc,), dtype=torch.int32, device=dev)
    L.check(lib.epc_knn_topk_conv1(xyz.data_ptr(), nc, n, 32, idx2.data_ptr(), 1, cnt2.data_ptr(), kth2.data_ptr(), pk,
                                   x1.data_ptr(), None, status.data_ptr(), L.current_stream()))
    torch.cuda.synchronize()
    m = torch.arange(32, device=dev)[None, None, :] < cnt.clamp(max=32)[..., None]
    assert torch.equal(cnt, cnt2) and torch.equal(kth, kth2) and torch.equal(idx * m, idx2.int() * m)


def _morton_sort(pc, dev):
    L = H.pkg("lib")
    t = torch.from_numpy(pc).to(dev)
    out = torch.empty_like(t)
    perm = torch.empty(pc.shape[:2], dtype=torch.int32, device=dev)
    L.check(L.lib().epc_morton_sort(t.data_ptr(), pc.shape[0], pc.shape[1], out.data_ptr(), perm.data_ptr(),
                                    L.current_stream()))
    torch.cuda.synchronize()
    return out.cpu().numpy(), perm.cpu().numpy()


def _hilbert30(q):
    """30-bit Hilbert-curve index of 10-bit cells (Skilling's transpose algorithm), numpy restatement of sort.hip."""
    X = [q[:, 0].astype(np.uint32).copy(), q[:, 1].astype(np.uint32).copy(), q[:, 2].astype(np.uint32).copy()]
    Q = 512
    while Q > 1:
        P = np.uint32(Q - 1)
        for i in range(3):
            m = (X[i] & np.uint32(Q)) != 0
            X[0] = np.where(m, X[0] ^ P, X[0])
            t = np.where(~m, (X[0] ^ X[i]) & P, 0).astype(np.uint32)
            X[0] = X[0] ^ t
            X[i] = X[i] ^ t
        Q >>= 1
    X[1] = X[1] ^ X[0]
    X[2] = X[2] ^ X[1]
    t = np.zeros_like(X[0])
    Q = 512
    while Q > 1:
        t = np.where((X[2] & np.uint32(Q)) != 0, t ^ np.uint32(Q - 1), t)
        Q >>= 1
    code = np.zeros(len(t), dtype=np.uint64)
    for bit in range(10):
        for d in range(3):          # X[0] carries the most significant bit of every triple
            code |= (((X[d] ^ t) >> np.uint32(bit)) & np.uint32(1)).astype(np.uint64) << np.uint64(3 * bit + (2 - d))
    return code


@pytest.mark.parametrize("kind,n", [("uniform", 4096), ("lidar", 4096), ("uniform", 100), ("zeros", 64), ("dup", 256)])
def test_spatial_sort_is_a_hilbert_order_permutation(dev, kind, n):
    pc = O.synthetic_clouds(2, n, 3, kind)
    srt, perm = _morton_sort(pc, dev)
    for b in range(2):
        assert np.array_equal(np.sort(perm[b]), np.arange(n))
        assert np.array_equal(srt[b], pc[b][perm[b]])
        lo, hi = pc[b].min(0), pc[b].max(0)
        ext = hi - lo
        scale = np.where(ext > 0, np.float32(1023.0) / np.where(ext > 0, ext, 1), 0).astype(np.float32)
        q = np.clip(((srt[b] - lo) * scale), 0, 1023).astype(np.uint32)
        code = _hilbert30(q)
        npow2 = 1 << int(np.ceil(np.log2(max(n, 2))))
        if npow2 == 4096:    # sort.hip: 32-bit keys = top 20 bits of the code | 12-bit index
            key = ((code >> np.uint64(10)) << np.uint64(12)) | perm[b].astype(np.uint64)
        else:
            key = (code << np.uint64(32)) | perm[b].astype(np.uint64)
        assert np.all(np.diff(key.astype(np.int64)) > 0)


@pytest.mark.parametrize("kind,n,seed", [("uniform", 4096, 0), ("lidar", 4096, 1), ("uniform", 1024, 2),
                                         ("lattice", 4096, 0), ("dup", 2048, 0), ("uniform", 8192, 4)])
def test_knn_bit_exact_on_sorted_clouds(dev, kind, n, seed):
    """Same bit-exact bar on Z-ordered clouds, where the bounding-box culling actually skips tiles."""
    tf_util = H.pkg("utils.tf_util")
    pc, _ = _morton_sort(O.synthetic_clouds(1, n, seed, kind), dev)
    kth_ref, lists = O.knn_lists(pc)
    kth, idx, cnt = tf_util.knn_index(torch.from_numpy(pc).to(dev))
    kth, idx, cnt = kth.cpu().numpy(), idx.cpu().numpy(), cnt.cpu().numpy()
    assert np.array_equal(kth, kth_ref)
    for i in range(n):
        ref = lists[0][i]
        assert cnt[0, i] == len(ref)
        m = min(len(ref), idx.shape[-1])
        assert np.array_equal(idx[0, i, :m], ref[:m])


def test_knn_streaming_kernel_large_n(dev):
    """N > 8192 takes the streaming (un-culled) kernel."""
    tf_util = H.pkg("utils.tf_util")
    pc = O.synthetic_clouds(1, 8224, 5)
    kth_ref, lists = O.knn_lists(pc)
    kth, idx, cnt = tf_util.knn_index(torch.from_numpy(pc).to(dev))
    assert np.array_equal(kth.cpu().numpy(), kth_ref)
    assert np.array_equal(cnt.cpu().numpy()[0], np.array([len(l) for l in lists[0]]))


@pytest.mark.parametrize("kind,n", [("uniform", 256), ("lattice", 512), ("zeros", 64)])
def test_knn_mask_matches_reference_form(dev, kind, n):
    """pairwise_distance_mask returns the dense 0/1 mask the reference builds (utils/tf_util.py:647-666)."""
    tf_util = H.pkg("utils.tf_util")
    pc = O.synthetic_clouds(2, n, 0, kind)
    mask = tf_util.pairwise_distance_mask(torch.from_numpy(pc).to(dev), k=20).cpu().numpy()
    assert np.array_equal(mask, O.pairwise_distance_mask(pc))


# (arch, precision) pairs: EPC-Net in both arithmetics of include/epcnet.h, EPC-Net-L (always f32-equivalent)
ARCH_PREC = [("epc-net", "fast"), ("epc-net", "f32"), ("epc-net-l", "f32")]


def _stage_compare(arch, pc, seed, dev, precision="fast"):
    w = O.seeded_weights(arch, seed)
    ref, st = O.forward(pc[:, None], w, arch=arch)
    eng, _ = H.make_engine(arch, w, dev, precision=precision)
    got = H.run_stages(eng, torch.from_numpy(pc).to(dev))
    return ref.reshape(pc.shape[0], -1), st, got, eng


@pytest.mark.parametrize("arch,prec", ARCH_PREC)
@pytest.mark.parametrize("kind,n", [("uniform", 256), ("lidar", 512), ("dup", 128), ("zeros", 64)])
def test_stages_against_oracle(dev, arch, prec, kind, n):
    """Every stage boundary against the oracle's taps.  In EPC_PRECISION_FAST EPC-Net's block chain stores fp16 rows and
    feeds one fp16 value per activation to the MFMA (include/epcnet.h), so its block tolerances are that format's
    precision (2^-11 of the largest value, plus the propagated part); EPC_PRECISION_F32 and EPC-Net-L run the f32 /
    split-bf16 kernels (same gather, index and overflow logic) and are held to 2e-5."""
    pc = O.synthetic_clouds(3, n, 5, kind)
    ref, st, got, eng = _stage_compare(arch, pc, 1, dev, prec)
    nblocks = 4 if arch == "epc-net" else 2
    fast = prec == "fast"
    btol = 1e-3 if fast else 2e-5

    def close(a, b, tol, what):
        a = a.float().cpu().numpy() if torch.is_tensor(a) else a
        err = np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
        assert err <= tol, "%s: relative max error %.3e > %.1e" % (what, err, tol)

    close(got["xs"][0], st.taps["fastdgcnn/conv1"], 2.0 ** -11 if fast else 1e-5, "conv1")
    for b in range(1, nblocks + 1):
        close(got["cat"][..., 64 * (b - 1):64 * b], st.taps["block%d" % b], btol, "block%d" % b)
        if b < nblocks:
            close(got["xs"][b], st.taps["fastdgcnn/conv%d" % (b + 1)], btol, "conv%d" % (b + 1))
    if arch == "epc-net":
        # fast: feat is stored as fp16 (11 significant bits) and computed from fp16 inputs
        close(got["feat"], st.taps["fastdgcnn/conv5"], 1e-3 if fast else 2e-5, "conv5 (fragment order)")
        close(got["assign"].reshape(-1, 64), st.taps["vlad_assign"], 1e-4 if fast else 2e-5, "assign")
        close(got["aprime"], got["assign"].cpu().numpy() * got["rnorm"].cpu().numpy()[..., None],
              2.0 ** -11 if fast else 2.0 ** -16, "assign fragments (fp16 / bf16 hi + lo)")
        v = got["vlad"].cpu().numpy()      # aggregate - a_sum * centres (loupe.py:286-292)
        # fast, worst case = the all-zero padding cloud (every point identical: the fp16 rounding of feat does not average out)
        close(v, st.taps["vlad_raw"], (2e-3 if kind == "zeros" else 2.0 ** -11) if fast else 2e-5, "vlad")
        colss = (got["vlad"].double() ** 2).reshape(v.shape[0], 32, 32, 64).sum(2)
        close(got["colss"], colss.cpu().numpy(), 1e-5, "column sums of squares")
    else:
        close(got["pooled"], st.taps["maxpool"], 2e-5, "maxpool")
    err = np.linalg.norm(got["desc"].cpu().numpy() - ref, axis=1).max()
    print("descriptor L2 error %s/%s %s n=%d: %.3e" % (arch, prec, kind, n, err))
    assert int(got["status"].abs().sum()) == 0
    assert err <= DESC_TOL, "descriptor L2 error %.3e" % err      # one bar for every cloud and both arithmetics


@pytest.mark.parametrize("arch,prec", ARCH_PREC + [("epc-net", None)])
def test_forward_api_full_size(dev, arch, prec):
    """The drop-in call: MODEL.forward(point_cloud (B,P,N,3), is_training=False, params=...) at N = 4096; the arithmetic is
    the YAML-level key PRECISION (absent: the package default, f32-equivalent like the reference's float32 graph)."""
    V = H.pkg("variables")
    pc = O.synthetic_clouds(3, 4096, 11).reshape(1, 3, 4096, 3)
    w = O.seeded_weights(arch, 3)
    ref, _ = O.forward(pc, w, arch=arch)
    H.make_store(arch, w, dev)
    M = H.pkg("models." + arch)
    with V.variable_scope(H.OUTER):
        x = M.placeholder_inputs(1, 3, 4096, 3)
        x.copy_(torch.from_numpy(pc))
        out = M.forward(x, False, bn_decay=None, params=dict(H.PARAMS, PRECISION=prec) if prec else H.PARAMS)
    assert tuple(out.shape) == (1, 3, 256)
    out = out.cpu().numpy()
    assert np.allclose(np.linalg.norm(out, axis=-1), 1.0, atol=1e-5)
    err = np.linalg.norm(out - ref, axis=-1).max()
    print("full-size descriptor L2 error %s/%s: %.3e" % (arch, prec, err))
    assert err <= DESC_TOL, "descriptor L2 error %.3e" % err


@pytest.mark.parametrize("prec", ["fast", "f32"])
@pytest.mark.parametrize("arch,nc,n,kind,micro", [
    ("epc-net", 1, 8192, "uniform", 0),      # largest cloud the LDS-resident kNN / one-workgroup sort take
    ("epc-net", 5, 96, "uniform", 2),        # N not a multiple of 64 / 128 / 256: tail tiles in every kernel, ragged micro-batches
    ("epc-net-l", 3, 160, "lidar", 0),       # max-pool path with workgroups that straddle clouds (per-wave atomics)
    ("epc-net-l", 2, 8192, "uniform", 1),
    ("epc-net", 2, 32, "dup", 0),            # smallest legal cloud: every point is every point's neighbour
    ("epc-net", 1, 8192 + 64, "uniform", 0), # beyond the LDS kNN kernel: streaming kNN, separate conv1 launch
])
def test_edge_shapes(dev, arch, nc, n, kind, micro, prec):
    if arch == "epc-net-l" and prec == "fast":
        pytest.skip("EPC-Net-L has one arithmetic")
    pc = O.synthetic_clouds(nc, n, 17, kind)
    w = O.seeded_weights(arch, 6)
    _, lists = O.knn_lists(pc)
    ref, _ = O.forward(pc[:, None], w, arch=arch, formulation="lists", lists=lists)
    eng, _ = H.make_engine(arch, w, dev, micro_batch=micro, precision=prec)
    out = eng.forward(torch.from_numpy(pc).to(dev)).cpu().numpy()
    err = np.linalg.norm(out - ref.reshape(nc, -1), axis=1).max()
    print("edge shape %s/%s %dx%d %s: descriptor L2 error %.3e" % (arch, prec, nc, n, kind, err))
    assert err <= DESC_TOL


@pytest.mark.parametrize("prec", ["fast", "f32"])
def test_micro_batching_is_invisible(dev, prec):
    w = O.seeded_weights("epc-net", 0)
    pc = torch.from_numpy(O.synthetic_clouds(5, 256, 2)).to(dev)
    a = H.make_engine("epc-net", w, dev, precision=prec)[0].forward(pc).cpu()
    b = H.make_engine("epc-net", w, dev, micro_batch=2, precision=prec)[0].forward(pc).cpu()
    assert torch.equal(a, b)


@pytest.mark.parametrize("arch,prec", ARCH_PREC)
def test_overlapped_passes_and_submit_match_single_stream(dev, arch, prec):
    """epc_net_forward_overlapped (passes dealt over several HIP streams) and InferenceEngine.submit (independent
    batches in flight) return bit-identical descriptors to the one-stream path."""
    E = H.pkg("engine")
    w = O.seeded_weights(arch, 0)
    pc = torch.from_numpy(O.synthetic_clouds(7, 256, 3)).to(dev)
    st = H.make_store(arch, w, dev)
    serial = E.InferenceEngine(arch, H.PARAMS, st, outer=H.OUTER, micro_batch=2, in_flight=1, precision=prec).forward(pc)
    for lanes in (2, 3, 8):
        eng = E.InferenceEngine(arch, H.PARAMS, st, outer=H.OUTER, micro_batch=2, in_flight=lanes, precision=prec)
        for _ in range(2):                                  # second call: lanes and workspace are reused
            assert torch.equal(eng.forward(pc), serial)
    eng = E.InferenceEngine(arch, H.PARAMS, st, outer=H.OUTER, in_flight=2, precision=prec)
    pending = [eng.submit(pc[i:i + 2]) for i in range(0, 6, 2)] + [eng.submit(pc[6:7])]
    eng.drain()
    got = torch.cat([o for o, _ in pending])
    assert torch.equal(got, serial)
    o, ev = eng.submit(pc[:3])
    ev.synchronize()
    assert torch.equal(o, serial[:3])


def test_epc_net_l_batch_256_runs_as_two_halves_in_flight_with_the_same_bits(dev):
    """InferenceEngine.forward's default for EPC-Net-L at 256 clouds x 4096 points (BASELINE.json configs[3]): two 128-cloud halves on two
    HIP streams (engine.L_HALVES_FROM) -- bit-identical to the one-stream pass of the whole batch, call after call."""
    E = H.pkg("engine")
    w = O.seeded_weights("epc-net-l", 0)
    st = H.make_store("epc-net-l", w, dev)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    pc = torch.rand((256, 4096, 3), generator=g, device=dev) * 2 - 1
    serial = E.InferenceEngine("epc-net-l", H.PARAMS, st, outer=H.OUTER, in_flight=1).forward(pc)
    eng = E.InferenceEngine("epc-net-l", H.PARAMS, st, outer=H.OUTER)
    assert eng.in_flight == 2
    for _ in range(3):
        assert torch.equal(eng.forward(pc), serial)
    assert torch.equal(eng.forward(pc[:200]), serial[:200])         # below the threshold: one pass on the caller's stream
    torch.cuda.synchronize()


@pytest.mark.parametrize("nc,n", [(3, 4096), (2, 96), (1, 8192 + 64)])
def test_knn_conv1_fused_launch_is_bit_identical(dev, nc, n):
    """epc_knn_topk_conv1 (one launch) against epc_knn_topk + epc_conv1_fwd: lists, counts, thresholds and both row
    formats of conv1 (f32 for EPC-Net-L, fp16 for EPC-Net) must match bit for bit."""
    L = H.pkg("lib")
    E = H.pkg("engine")
    lib = L.lib()
    eng, _ = H.make_engine("epc-net", O.seeded_weights("epc-net", 0), dev)
    cfg = eng.cfg_for(n)
    pk = eng.packed(cfg).data_ptr() + lib.epc_net_packed_offset(ctypes.byref(cfg), 0)
    status = torch.zeros((nc,), dtype=torch.int32, device=dev)
    xyz = torch.from_numpy(O.synthetic_clouds(nc, n, 5)).to(dev)
    st = L.current_stream()

    def buffers():
        return (torch.zeros((nc, n, 32), dtype=torch.int32, device=dev), torch.zeros((nc, n), dtype=torch.int32, device=dev),
                torch.zeros((nc, n), device=dev), torch.zeros((nc * n, 64), device=dev),
                torch.zeros((nc * n, 64), dtype=torch.float16, device=dev))
    i0, c0, k0, x0, h0 = buffers()
    L.check(lib.epc_knn_topk(xyz.data_ptr(), nc, n, 32, i0.data_ptr(), c0.data_ptr(), k0.data_ptr(), st))
    L.check(lib.epc_conv1_fwd(xyz.data_ptr(), pk, nc * n, x0.data_ptr(), h0.data_ptr(), st))
    i1, c1, k1, x1, h1 = buffers()
    L.check(lib.epc_knn_topk_conv1(xyz.data_ptr(), nc, n, 32, i1.data_ptr(), 0, c1.data_ptr(), k1.data_ptr(), pk,
                                   x1.data_ptr(), h1.data_ptr(), status.data_ptr(), st))
    torch.cuda.synchronize()
    assert torch.equal(c0, c1) and torch.equal(k0, k1)
    m = torch.arange(32, device=dev)[None, None, :] < c0.clamp(max=32)[..., None]
    assert torch.equal(i0 * m, i1 * m)
    if n <= 8192:    # the pipeline's 2-byte lists hold the same entries
        i2 = torch.zeros((nc, n, 32), dtype=torch.int16, device=dev)
        _, c2, k2, x2, h2 = buffers()
        L.check(lib.epc_knn_topk_conv1(xyz.data_ptr(), nc, n, 32, i2.data_ptr(), 1, c2.data_ptr(), k2.data_ptr(), pk,
                                       x2.data_ptr(), h2.data_ptr(), status.data_ptr(), st))
        torch.cuda.synchronize()
        assert torch.equal(c0, c2) and torch.equal(k0, k2) and torch.equal(i0 * m, i2.int() * m)
    else:
        rc = lib.epc_knn_topk_conv1(xyz.data_ptr(), nc, n, 32, i1.data_ptr(), 1, c1.data_ptr(), k1.data_ptr(), pk,
                                    x1.data_ptr(), h1.data_ptr(), status.data_ptr(), st)
        assert rc == -1                      # 2-byte lists only with the LDS kernel (n <= 8192)
    assert torch.equal(x0, x1) and torch.equal(h0.view(torch.int16), h1.view(torch.int16))
    assert float(x1.abs().sum()) > 0 and int(status.abs().sum()) == 0


def test_overlapped_rejects_short_workspace(dev):
    L = H.pkg("lib")
    E = H.pkg("engine")
    eng, _ = H.make_engine("epc-net", O.seeded_weights("epc-net", 0), dev, micro_batch=2)
    cfg = eng.cfg_for(64)
    packed = eng.packed(cfg)
    one = L.lib().epc_net_workspace_bytes(ctypes.byref(cfg), 4)
    ws = torch.empty(one, dtype=torch.uint8, device=dev)       # two lanes need 2 x one
    xyz = torch.zeros((4, 64, 3), device=dev)
    out = torch.empty((4, 256), device=dev)
    aux = torch.cuda.Stream()
    arr = (ctypes.c_void_p * 1)(aux.cuda_stream)
    rc = L.lib().epc_net_forward_overlapped(ctypes.byref(cfg), packed.data_ptr(), xyz.data_ptr(), 4, out.data_ptr(),
                                            ws.data_ptr(), ws.numel(), L.current_stream(), arr, 1)
    assert rc == -2 and b"lanes" in L.lib().epc_last_error()
    rc = L.lib().epc_net_forward_overlapped(ctypes.byref(cfg), packed.data_ptr(), xyz.data_ptr(), 4, out.data_ptr(),
                                            ws.data_ptr(), ws.numel(), L.current_stream(), None, 0)
    assert rc == 0                                              # no auxiliary streams: plain serial passes
    torch.cuda.synchronize()


@pytest.mark.parametrize("prec", ["fast", "f32"])
def test_permutation_invariance(dev, prec):
    """Permuting a cloud's points leaves the descriptor unchanged up to fp32 summation order."""
    w = O.seeded_weights("epc-net", 0)
    eng, _ = H.make_engine("epc-net", w, dev, precision=prec)
    pc = O.synthetic_clouds(1, 4096, 7)
    perm = np.random.RandomState(0).permutation(4096)
    a = eng.forward(torch.from_numpy(pc).to(dev)).cpu().numpy()
    b = eng.forward(torch.from_numpy(np.ascontiguousarray(pc[:, perm])).to(dev)).cpu().numpy()
    assert np.linalg.norm(a - b) <= 1e-5


def test_pairwise_topk(dev):
    L = H.pkg("lib")
    rng = np.random.RandomState(0)
    db = rng.randn(500, 256).astype(np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    db[17] = db[3]  # exact tie: lower index first
    q = rng.randn(50, 256).astype(np.float32)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    q[0] = db[3]
    dist_ref, idx_ref = O.knn_bruteforce(db, q, 25)
    tdb, tq = torch.from_numpy(db).to(dev), torch.from_numpy(q).to(dev)
    idx = torch.empty((50, 25), dtype=torch.int32, device=dev)
    dist = torch.empty((50, 25), dtype=torch.float32, device=dev)
    L.check(L.lib().epc_pairwise_topk(tdb.data_ptr(), 500, tq.data_ptr(), 50, 256, 25, idx.data_ptr(),
                                      dist.data_ptr(), L.current_stream()))
    torch.cuda.synchronize()
    assert np.array_equal(idx.cpu().numpy(), idx_ref)
    assert np.allclose(dist.cpu().numpy(), dist_ref, atol=1e-5)


def _topk_both(L, db, q, k, dev):
    """(idx, dist) of the LDS form and of the workspace (MFMA + exact re-rank) form."""
    tdb, tq = torch.from_numpy(db).to(dev), torch.from_numpy(q).to(dev)
    out = []
    for ws_form in (False, True):
        idx = torch.full((len(q), k), -7, dtype=torch.int32, device=dev)
        dist = torch.full((len(q), k), -7.0, dtype=torch.float32, device=dev)
        if ws_form:
            need = L.lib().epc_pairwise_topk_workspace_bytes(len(db), len(q))
            ws = torch.empty(need, dtype=torch.uint8, device=dev)
            L.check(L.lib().epc_pairwise_topk_ws(tdb.data_ptr(), len(db), tq.data_ptr(), len(q), db.shape[1], k,
                                                 idx.data_ptr(), dist.data_ptr(), ws.data_ptr(), need, L.current_stream()))
        elif len(db) * 4 + db.shape[1] * 4 <= 160 * 1024:
            L.check(L.lib().epc_pairwise_topk(tdb.data_ptr(), len(db), tq.data_ptr(), len(q), db.shape[1], k, idx.data_ptr(),
                                              dist.data_ptr(), L.current_stream()))
        else:
            out.append(None)
            continue
        torch.cuda.synchronize()
        out.append((idx.cpu().numpy(), dist.cpu().numpy()))
    return out


@pytest.mark.parametrize("n_db,n_q,kind", [(500, 50, "unit"), (9200, 300, "unit"), (40000, 64, "unit"), (1000, 40, "clumped"),
                                           (300, 20, "identical"), (130, 7, "scaled")])
def test_pairwise_topk_workspace_form_is_the_exact_answer(dev, n_db, n_q, kind):
    """epc_pairwise_topk_ws (f32-MFMA pairwise matrix + exact re-rank + proof / fallback) against the LDS form (bit-identical
    indices AND distances) and the oracle's brute force: G6-sized, Oxford-sized, 40 000 rows (past the LDS form's cap),
    descriptors in tight clumps (near-ties by the dozen: the fallback path), all-identical rows (ties -> lower index), and
    un-normalised rows of very different norms."""
    L = H.pkg("lib")
    rng = np.random.RandomState(n_db)
    db = rng.randn(n_db, 256).astype(np.float32)
    if kind == "clumped":          # 20 centres, members 1e-4 apart: far more than k + 8 rows inside any rounding band
        db = (db[:20][rng.randint(0, 20, n_db)] + 1e-4 * rng.randn(n_db, 256)).astype(np.float32)
    if kind == "identical":
        db[:] = db[0]
    if kind != "scaled":
        db /= np.linalg.norm(db, axis=1, keepdims=True)
    else:
        db *= np.exp(rng.uniform(-3, 3, size=(n_db, 1))).astype(np.float32)
    q = db[rng.randint(0, n_db, n_q)] + (0.3 if kind != "clumped" else 1e-4) * rng.randn(n_q, 256).astype(np.float32)
    if kind != "scaled":
        q /= np.linalg.norm(q, axis=1, keepdims=True)
    q = np.ascontiguousarray(q, dtype=np.float32)
    q[0] = db[min(3, n_db - 1)]                       # an exact match: distance 0, not sqrt(rounding)
    lds, ws = _topk_both(L, db, q, 25, dev)
    if lds is not None:
        assert np.array_equal(ws[0], lds[0]) and np.array_equal(ws[1], lds[1])
    assert ws[1][0, 0] == 0.0 and (np.diff(ws[1], axis=1) >= 0).all()
    if kind in ("unit", "scaled"):
        dist_ref, idx_ref = O.knn_bruteforce(db, q, 25)
        assert np.array_equal(ws[0], idx_ref)
        assert np.allclose(ws[1], dist_ref, rtol=1e-5, atol=1e-5)
    if kind == "identical":
        assert np.array_equal(ws[0], np.tile(np.arange(25, dtype=np.int32), (n_q, 1)))


def test_pairwise_topk_nan_rows_and_queries(dev):
    """flagged clouds have NaN descriptors by design.  A NaN query has no neighbour at a finite distance
    -> (-1, +Inf) in every slot, from BOTH forms, with no store outside the buffers (canaries around idx / dist / workspace);
    NaN database rows are never selected; a query with fewer than k finite rows gets the finite ones then (-1, +Inf)."""
    L = H.pkg("lib")
    R = H.pkg("retrieval")
    rng = np.random.RandomState(11)
    n_db, n_q, k = 700, 40, 25
    db = rng.randn(n_db, 256).astype(np.float32)
    db /= np.linalg.norm(db, axis=1, keepdims=True)
    q = db[rng.randint(0, n_db, n_q)] + 0.3 * rng.randn(n_q, 256).astype(np.float32)
    q = np.ascontiguousarray(q / np.linalg.norm(q, axis=1, keepdims=True), dtype=np.float32)
    db[5] = np.nan
    db[77, 3] = np.inf
    q[2] = np.nan
    q[9, 100] = np.nan
    good_db = np.isfinite(db).all(axis=1)
    lds, ws = _topk_both(L, db, q, k, dev)
    for idx, dist in (lds, ws):
        for i in (2, 9):
            assert (idx[i] == -1).all() and np.isinf(dist[i]).all()
        ok = [i for i in range(n_q) if i not in (2, 9)]
        dref, iref = O.knn_bruteforce(db[good_db], q[ok], k)
        remap = np.nonzero(good_db)[0]
        assert np.array_equal(idx[ok], remap[iref])
        assert np.allclose(dist[ok], dref, rtol=1e-5, atol=1e-5)
    assert np.array_equal(lds[0], ws[0]) and np.array_equal(lds[1], ws[1])
    # fewer than k finite rows: 30 rows, 11 of them NaN (row 5 from above), k = 25 -> 19 neighbours then (-1, +Inf)
    db2 = db[:30].copy()
    db2[10:20] = np.nan
    nf = int(np.isfinite(db2).all(axis=1).sum())
    assert nf == 19
    lds2, ws2 = _topk_both(L, db2, q[:5], k, dev)
    for idx, dist in (lds2, ws2):
        for i in (0, 1, 3, 4):
            assert (idx[i, :nf] >= 0).all() and (idx[i, nf:] == -1).all() and np.isinf(dist[i, nf:]).all()
            assert not set(idx[i, :nf].tolist()) & (set(range(10, 20)) | {5})
        assert (idx[2] == -1).all()
    # canaries: the workspace form must not write past its buffers when a query is NaN (the old fallback stored row[0x7fffffff])
    tdb, tq = torch.from_numpy(db).to(dev), torch.from_numpy(q).to(dev)
    need = L.lib().epc_pairwise_topk_workspace_bytes(n_db, n_q)
    arena = torch.full((need + 8192,), 0x5A, dtype=torch.uint8, device=dev)
    out = torch.full((n_q * k + 64,), -7, dtype=torch.int32, device=dev)
    dst = torch.full((n_q * k + 64,), -7.0, dtype=torch.float32, device=dev)
    L.check(L.lib().epc_pairwise_topk_ws(tdb.data_ptr(), n_db, tq.data_ptr(), n_q, 256, k, out.data_ptr(), dst.data_ptr(),
                                         arena.data_ptr() + 4096, need, L.current_stream()))
    torch.cuda.synchronize()
    assert (arena[:4096] == 0x5A).all() and (arena[4096 + need:] == 0x5A).all()
    assert (out[n_q * k:] == -7).all() and (dst[n_q * k:] == -7.0).all()
    # the host bookkeeping skips the -1 slots (a NaN query is evaluated and scores no hit, like a KDTree answer of far rows)
    truth = [[int(v)] for v in rng.randint(0, n_db, n_q)]
    rec, sim, opr = R.recall_from_indices(ws[0], np.nan_to_num(db), np.nan_to_num(q), truth)
    assert np.isfinite(rec).all()


def test_pairwise_topk_large_norm_near_ties(dev):
    """un-normalised descriptors whose near-ties sit far from the origin.  The proof's rounding bound must
    scale with the norms involved (dim, |q|^2, the database's largest |d|^2): the workspace form has to return the LDS form's
    lists bit for bit, i.e. flag every query the GEMM's cancellation could have mis-ranked."""
    L = H.pkg("lib")
    rng = np.random.RandomState(21)
    n_db, n_q, k = 3000, 64, 25
    base = rng.randn(1, 256).astype(np.float32) * 40.0                    # |d|^2 ~ 4e5: f32 spacing of the norms ~ 0.03
    db = (base + rng.randn(n_db, 256).astype(np.float32) * 0.05).astype(np.float32)   # pairwise d2 ~ 1.3, spread ~ 0.1
    db[1500:] = rng.randn(1500, 256).astype(np.float32)                   # plus ordinary rows near the origin
    q = np.ascontiguousarray(db[rng.randint(0, 1500, n_q)] + 0.02 * rng.randn(n_q, 256).astype(np.float32), dtype=np.float32)
    q[0] = db[3]
    lds, ws = _topk_both(L, db, q, k, dev)
    assert np.array_equal(ws[0], lds[0]) and np.array_equal(ws[1], lds[1])
    # and a single outlier row of huge norm must not break the others' answers (the norm-free half of the bound)
    db2 = rng.randn(2000, 256).astype(np.float32)
    db2 /= np.linalg.norm(db2, axis=1, keepdims=True)
    db2[123] *= 1e6
    q2 = np.ascontiguousarray(db2[rng.randint(0, 2000, 32)] + 0.3 * rng.randn(32, 256).astype(np.float32), dtype=np.float32)
    lds2, ws2 = _topk_both(L, db2, q2, k, dev)
    assert np.array_equal(ws2[0], lds2[0]) and np.array_equal(ws2[1], lds2[1])


def test_evaluate_protocol_single_rank(dev):
    """get_latent_vectors + get_recall/evaluate_runs (evaluate.py:293-332, 351-537) on synthetic runs: GPU neighbour
    search + host bookkeeping must reproduce the oracle's recall numbers on the same descriptors."""
    R, D = H.pkg("retrieval"), H.pkg("distributed")
    w = O.seeded_weights("epc-net-l", 0)
    eng, _ = H.make_engine("epc-net-l", w, dev)
    rng = np.random.RandomState(3)
    runs = [O.synthetic_clouds(40 + 5 * r, 256, 50 + r) for r in range(3)]
    vecs = [R.get_latent_vectors(eng, c, batch_size=16, device=dev) for c in runs]
    for v, c in zip(vecs, runs):
        assert v.shape == (len(c), 256) and np.allclose(np.linalg.norm(v, axis=1), 1, atol=1e-5)
    truth = {(m, n): [list(rng.choice(len(runs[m]), size=rng.randint(0, 4), replace=False)) for _ in range(len(runs[n]))]
             for m in range(3) for n in range(3)}
    res = R.evaluate_runs(vecs, vecs, lambda m, n: truth[(m, n)], device=dev)
    rec = np.zeros(25); opr = []; sim = []
    for m in range(3):
        for n in range(3):
            if m != n:
                r_, s_, o_ = O.get_recall(vecs[m], vecs[n], truth[(m, n)])
                rec += r_; opr.append(o_); sim.extend(s_)
    assert np.allclose(res["ave_recall"], rec / 6) and np.isclose(res["ave_one_percent_recall"], np.mean(opr))
    assert np.isclose(res["average_similarity"], np.mean(sim), atol=1e-6)
    # world_size 1 degenerates cleanly
    idx = D.sharded_knn(torch.from_numpy(vecs[0]).to(dev), len(vecs[0]), torch.from_numpy(vecs[1]).to(dev),
                        len(vecs[1]), 25, R.knn_search)
    assert np.array_equal(idx, O.knn_bruteforce(vecs[0], vecs[1], 25)[1])


def test_last_status_refuses_after_a_multi_lane_call(dev):
    """a forward dealt over the engine's lanes (num_clouds > micro_batch, in_flight > 1) keeps one workspace slice per
    lane, so there are no "status words of the last pass" to report: last_status must raise, not return another call's words."""
    L = H.pkg("lib")
    w = O.seeded_weights("epc-net-l", 0)
    eng, _ = H.make_engine("epc-net-l", w, dev, micro_batch=4, in_flight=2)
    pc = torch.from_numpy(O.synthetic_clouds(12, 256, 3)).to(dev)
    one = eng.forward(pc[:4])
    assert eng.last_status(4) == [0, 0, 0, 0]
    many = eng.forward(pc)                                   # 3 passes over 2 lanes
    assert torch.equal(many[:4], one)
    with pytest.raises(L.EpcNetError):
        eng.last_status(12)
    eng.forward(pc[4:8])
    assert eng.last_status(4) == [0, 0, 0, 0]


def test_errors_are_loud(dev):
    L = H.pkg("lib")
    w = O.seeded_weights("epc-net", 0)
    eng, _ = H.make_engine("epc-net", w, dev)
    with pytest.raises(L.EpcNetError):
        eng.forward(torch.zeros((1, 100, 3), device=dev))      # N not a multiple of 32
    with pytest.raises(L.EpcNetError):
        eng.forward(torch.zeros((1, 64, 3)))                   # CPU tensor: no fallback
    with pytest.raises(L.EpcNetError):
        eng.forward(torch.zeros((1, 64, 4), device=dev))       # INPUT_DIM != 3


def test_plain_c_caller(dev):
    """The C ABI from a C99 program (examples/epcnet_forward.c): packs a seeded variable table, runs epc_net_forward,
    checks unit-norm descriptors and bit-identical results for a repeated cloud; exit code 0 = all checks passed."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "examples")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([os.path.join(root, "examples", "epcnet_forward"), "5"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "5 clouds x 4096 points" in r.stdout


@pytest.mark.parametrize("prec", ["fast", "f32"])
def test_full_size_properties(dev, prec):
    """BASELINE.json configs[1] at full size (64 x 4096 x 3): properties that need no oracle run of that size.
    (a) unit-norm, finite descriptors; (b) a cloud's descriptor does not depend on its batch (bit-identical alone, in a
    batch of 64, and at another position); (d) four sampled clouds of the batch against the oracle; (c) kNN lists of sampled queries: ascending, self included, every listed j
    satisfies a_ij >= kth and the count equals |{j : a_ij >= kth}| with a_ij evaluated by the oracle's formula."""
    w = O.seeded_weights("epc-net", 0)
    eng, _ = H.make_engine("epc-net", w, dev, precision=prec)
    pc = O.synthetic_clouds(64, 4096, 123)
    x = torch.from_numpy(pc).to(dev)
    out = eng.forward(x)
    assert bool(torch.isfinite(out).all()) and float((out.norm(dim=1) - 1).abs().max()) < 1e-5
    alone = eng.forward(x[17:18])
    assert torch.equal(alone[0], out[17])
    rolled = eng.forward(torch.roll(x, shifts=5, dims=0))
    assert torch.equal(rolled[22], out[17])
    # (d) four sampled clouds of the 64-batch against the oracle (the whole batch would take the numpy oracle minutes)
    sel = [0, 17, 40, 63]
    ref, _ = O.forward(pc[sel][:, None], w, arch="epc-net")
    err = np.linalg.norm(out[sel].cpu().numpy() - ref.reshape(4, -1), axis=1).max()
    print("EPC-Net 64 x 4096 (%s): descriptor L2 error on 4 sampled clouds of the batch %.3e" % (prec, err))
    assert err <= DESC_TOL
    ops = H.pkg("ops")
    tf_util = H.pkg("utils.tf_util")
    srt = ops.morton_sort(x[:4])
    kth, idx, cnt = tf_util.knn_index(srt)
    s_np, kth, idx, cnt = srt.cpu().numpy(), kth.cpu().numpy(), idx.cpu().numpy(), cnt.cpu().numpy()
    rng = np.random.RandomState(0)
    for b in range(4):
        a = O.neg_sq_dist(s_np[b:b + 1])[0]                      # (4096, 4096) in the reference's association
        for i in rng.choice(4096, size=64, replace=False):
            sel = np.nonzero(a[i] >= kth[b, i])[0]
            assert cnt[b, i] == len(sel) >= 20
            lst = idx[b, i, :min(cnt[b, i], 32)]
            assert np.array_equal(lst, sel[:32]) and i in sel
            assert kth[b, i] == np.sort(a[i])[-20]


def test_epc_net_l_batch_256_full_size(dev):
    """BASELINE.json configs[3]: EPC-Net-L at batch 256 x 4096 x 3 (models/epc-net-l.py:29-102; the library's default
    micro-batch for this architecture).  (a) finite unit-norm descriptors; (b) a cloud's descriptor does not depend on its
    batch: bit-identical alone, in the batch of 256 and at another position of it; (c) four sampled clouds against the oracle."""
    w = O.seeded_weights("epc-net-l", 0)
    eng, _ = H.make_engine("epc-net-l", w, dev)
    pc = O.synthetic_clouds(256, 4096, 321)
    x = torch.from_numpy(pc).to(dev)
    out = eng.forward(x)
    assert tuple(out.shape) == (256, 256)
    assert bool(torch.isfinite(out).all()) and float((out.norm(dim=1) - 1).abs().max()) < 1e-5
    assert eng.last_status(256) == [0] * 256
    for i in (0, 131, 255):
        assert torch.equal(eng.forward(x[i:i + 1])[0], out[i])
    rolled = eng.forward(torch.roll(x, shifts=37, dims=0))
    assert torch.equal(rolled[(131 + 37) % 256], out[131]) and torch.equal(rolled[36], out[255])
    sel = [3, 77, 190, 255]
    ref, _ = O.forward(pc[sel][:, None], w, arch="epc-net-l")
    err = np.linalg.norm(out[sel].cpu().numpy() - ref.reshape(4, -1), axis=1).max()
    print("EPC-Net-L 256 x 4096: descriptor L2 error on 4 sampled clouds %.3e" % err)
    assert err <= DESC_TOL
