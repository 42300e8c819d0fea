"""The two forms of the LDS kNN kernel -- one lane per query (knn_topk_culled_kernel) and four lanes per query
(knn_topk_quad_kernel, the default) -- against the oracle's lists (utils/tf_util.py:647-666 in index form): thresholds,
counts and list entries bit for bit, on both, for ordinary, tied, padded and ragged clouds, sorted or not, conv1 fused or not.
The form is an explicit argument of the test entry points epc_knn_topk_form / epc_knn_topk_conv1_form (include/epcnet.h)."""
import ctypes

import numpy as np
import pytest
import torch

import helpers as H
from helpers import O

pytestmark = pytest.mark.gpu


@pytest.fixture
def dev():
    return torch.device("cuda:0")


@pytest.fixture(params=[0, 1], ids=["one_lane", "four_lanes"])
def form(request):
    return request.param


def _check(pc, kth, idx, cnt):
    kth_ref, lists = O.knn_lists(pc)
    assert np.array_equal(kth, kth_ref)
    for b in range(pc.shape[0]):
        for i in range(pc.shape[1]):
            ref = lists[b][i]
            assert cnt[b, i] == len(ref), (b, i)
            m = min(len(ref), idx.shape[-1])
            assert np.array_equal(idx[b, i, :m], ref[:m]), (b, i)


@pytest.mark.parametrize("kind,n,sort", [("uniform", 4096, True), ("lidar", 2048, True), ("uniform", 992, False), ("dup", 512, True),
                                         ("lattice", 512, False), ("zeros", 64, False), ("uniform", 20, False), ("uniform", 100, True),
                                         ("lidar", 1000, True), ("dup", 4096, True), ("uniform", 8192, True), ("lidar", 6000, False)])
def test_both_forms_give_the_oracle_lists(dev, form, kind, n, sort):
    ops = H.pkg("ops")
    tf_util = H.pkg("utils.tf_util")
    pc = O.synthetic_clouds(3, n, 7, kind)
    x = torch.from_numpy(pc).to(dev)
    if sort and n % 32 == 0:
        x = ops.morton_sort(x)
    kth, idx, cnt = tf_util.knn_index(x, form=form)
    _check(x.cpu().numpy(), kth.cpu().numpy(), idx.cpu().numpy(), cnt.cpu().numpy())


def test_forms_agree_with_conv1_fused_and_short_lists(dev):
    """epc_knn_topk_conv1 with 2-byte lists (the fused pipeline's format): both forms leave identical lists, counts, thresholds and
    conv1 rows; a cloud with a NaN coordinate gets the same padded rows and status bit."""
    L = H.pkg("lib")
    lib = L.lib()
    n, nc = 1024, 5
    eng, _ = H.make_engine("epc-net", O.seeded_weights("epc-net", 0), dev)
    cfg = eng.cfg_for(n)
    pk = eng.packed(cfg).data_ptr() + lib.epc_net_packed_offset(ctypes.byref(cfg), 0)
    pc = O.synthetic_clouds(nc, n, 3, "lidar")
    pc[2, 17, 1] = np.nan
    xyz = H.pkg("ops").morton_sort(torch.from_numpy(np.nan_to_num(pc)).to(dev))
    xyz[2, 17, 1] = float("nan")
    out = {}
    for f in (0, 1):
        idx = torch.zeros((nc, n, 32), dtype=torch.int16, device=dev)
        cnt = torch.zeros((nc, n), dtype=torch.int32, device=dev)
        kth = torch.zeros((nc, n), device=dev)
        x32 = torch.zeros((nc * n, 64), device=dev)
        x16 = torch.zeros((nc * n, 64), dtype=torch.float16, device=dev)
        status = torch.zeros((nc,), dtype=torch.int32, device=dev)
        L.check(lib.epc_knn_topk_conv1_form(xyz.data_ptr(), nc, n, 32, idx.data_ptr(), 1, cnt.data_ptr(), kth.data_ptr(), pk,
                                            x32.data_ptr(), x16.data_ptr(), status.data_ptr(), f, L.current_stream()))
        torch.cuda.synchronize()
        out[f] = [t.cpu().numpy() for t in (idx, cnt, kth, x32, x16, status)]
    a, b = out[0], out[1]
    keep = [0, 1, 3, 4]
    assert np.array_equal(a[5], b[5]) and a[5][2] != 0 and not a[5][keep].any()
    for u, v in zip(a[:3], b[:3]):
        assert np.array_equal(u[keep], v[keep])
    assert np.array_equal(a[3], b[3], equal_nan=True) and np.array_equal(a[4], b[4], equal_nan=True)    # conv1 rows
    # the poisoned cloud (its descriptor is NaN, EPC_STATUS_NONFINITE_INPUT): what a row holds depends on the order in which NaN
    # comparisons fail, but its first 20 entries are valid row numbers in both forms
    for o in (a, b):
        first = o[0][2].astype(np.int64)[:, :20] & 0xFFFF
        assert first.min() >= 0 and first.max() < n
    # and the un-poisoned clouds are the oracle's
    _check(xyz.cpu().numpy()[keep], a[2][keep], a[0][keep].astype(np.int64) & 0xFFFF, a[1][keep])
