"""CPU tests of the host layer: C-ABI exports, naming contract, checkpoint reader, loud failure without a GPU."""
import importlib
import json
import os
import re
import shutil

import numpy as np
import pytest
import torch

import helpers as H
from helpers import O, ROOT

GOLD = os.path.join(ROOT, "tests", "golden")
TABLES = json.load(open(os.path.join(GOLD, "ckpt_tables.json")))


def test_library_exports_every_declared_symbol():
    L = H.pkg("lib")
    header = open(os.path.join(ROOT, "include", "epcnet.h")).read()
    declared = set(re.findall(r"\b(epc_[a-z0-9_]+)\s*\(", header))
    assert declared == set(L.EXPORTS), declared ^ set(L.EXPORTS)
    for name in declared:
        assert hasattr(L.lib(), name), name
    assert L.lib().epc_version() >= 100
    # ... and the prose may not name entry points that do not exist (the header spoke of an
    # `epc_net_forward_status`): every epc_* identifier in the header, comments included, and in the documents a binder reads, is an
    # exported symbol or one of the header's types
    types = {"epc_cfg", "epc_status", "epc_profile", "epc_chain_fwd_args", "epc_chain_fwd_block"}
    for doc in ("include/epcnet.h", "INTEGRATION.md", "DESIGN.md", "README.md", "epc-net_amd/lib.py", "epc-net_amd/engine.py"):
        text = open(os.path.join(ROOT, doc)).read()
        unknown = {n for n in re.findall(r"\bepc_[a-z0-9_]+\b", text) if n not in declared and n not in types}
        assert not unknown, "%s names entry points the library does not have: %s" % (doc, sorted(unknown))
    # the precision prose describes the arithmetic the kernels run (csrc/common.h split8_f16s, csrc/conv5_f32.hip)
    assert "split-fp16" in header and "3-byte" in header and "epc_net_forward_status" not in header


def test_cfg_struct_layout_and_sizes():
    import ctypes
    L, E = H.pkg("lib"), H.pkg("engine")
    assert ctypes.sizeof(L.EpcCfg) == 36               # nine int32 fields (include/epcnet.h: epc_cfg, `precision` last)
    assert E.make_cfg("epc-net", 4096, H.PARAMS).precision == L.EPC_PRECISION_F32      # the package default: the reference's class
    assert E.make_cfg("epc-net", 4096, dict(H.PARAMS, PRECISION="fast")).precision == L.EPC_PRECISION_FAST
    with pytest.raises(ValueError):
        E.make_cfg("epc-net", 4096, dict(H.PARAMS, PRECISION="fp8"))
    f32 = E.make_cfg("epc-net", 4096, H.PARAMS, precision="f32")
    cfg = E.make_cfg("epc-net", 4096, H.PARAMS, precision="fast")
    # the f32-equivalent path keeps f32 tensors between its stages (and `feat` as 3-byte values against the fast path's fp16):
    # 1.6 x the workspace of the fp16 path
    ratio = L.lib().epc_net_workspace_bytes(ctypes.byref(f32), 64) / L.lib().epc_net_workspace_bytes(ctypes.byref(cfg), 64)
    assert 1.5 < ratio < 1.7
    assert L.lib().epc_net_packed_bytes(ctypes.byref(f32)) == L.lib().epc_net_packed_bytes(ctypes.byref(cfg))
    badp = E.make_cfg("epc-net", 4096, H.PARAMS)
    badp.precision = 7
    assert L.lib().epc_net_packed_bytes(ctypes.byref(badp)) == 0
    nbytes = L.lib().epc_net_packed_bytes(ctypes.byref(cfg))
    assert nbytes >= 4704832 * 4 - 4 * 10000          # folded weights: ~ the trainable matrices
    offs = [L.lib().epc_net_packed_offset(ctypes.byref(cfg), s) for s in range(7)]
    assert offs == sorted(offs) and offs[0] == 0 and all(o % 256 == 0 for o in offs)
    assert L.lib().epc_net_workspace_bytes(ctypes.byref(cfg), 64) > 64 * 4096 * 1024 * 2   # holds the fp16 conv5 map
    bad = E.make_cfg("epc-net", 4100, H.PARAMS)       # N not a multiple of 32
    assert L.lib().epc_net_packed_bytes(ctypes.byref(bad)) == 0


@pytest.mark.parametrize("arch,count", [("epc-net", 4704832), ("epc-net-l", 418880)])
def test_state_dict_equals_reference_checkpoint_names(arch, count):
    st = H.make_store(arch, O.seeded_weights(arch, 0), "cpu")
    ref = {e["name"]: tuple(e["shape"]) for e in TABLES[arch]["entries"]
           if not e["name"].endswith(("/Adam", "/Adam_1")) and e["name"] not in ("Variable", "beta1_power", "beta2_power")}
    assert {k: tuple(v.shape) for k, v in st.state_dict().items()} == ref
    assert st.num_trainable_params() == count


def test_reference_initialisers():
    V = H.pkg("variables")
    st = V.reset_default_store(device="cpu", seed=1)
    M = H.pkg("models.epc-net")
    with V.variable_scope("query_triplets"):
        M.declare_variables(H.PARAMS, 4096)
    w = st.vars["query_triplets/fastdgcnn/conv5/weights"]
    lim = (6.0 / (256 + 1024)) ** 0.5                                # xavier uniform, utils/tf_util.py:42
    assert float(w.abs().max()) <= lim and float(w.abs().max()) > 0.95 * lim
    assert float(st.vars["query_triplets/fastdgcnn/conv1/biases"].abs().max()) == 0.0
    assert float(st.vars["query_triplets/VLAD/bn/moving_variance"].min()) == 1.0
    h = st.vars["query_triplets/VLAD/hidden1_weights"]
    assert abs(float(h.std()) - 1 / 8.0) < 2e-3                      # N(0, 1/sqrt(cluster_size)), loupe.py:316


@pytest.mark.parametrize("tag,n", [("epc-net", 221), ("epc-net-l", 115), ("epc-net-l-d_student", 113),
                                   ("epc-net-l-d_teacher", 221)])
def test_tf_bundle_index_reader_on_reference_files(tag, n):
    tfb = H.pkg("tf_bundle")
    entries = tfb.read_index(os.path.join(GOLD, tag + ".ckpt.index"))
    assert len(entries) == n
    want = {e["name"]: (e["dtype"], tuple(e["shape"]), e["offset"], e["size"]) for e in TABLES[tag]["entries"]}
    got = {e.name: (e.dtype.name, e.shape, e.offset, e.size) for e in entries.values()}
    assert got == want


def test_checkpoint_round_trip_and_missing_payload(tmp_path):
    """Seeded weights written through the REFERENCE's index table load back into the store; a missing .data shard
    raises like evaluate.py:264-266 refuses to run."""
    tfb = H.pkg("tf_bundle")
    prefix = str(tmp_path / "model_epoch22_iter18101.ckpt")
    shutil.copyfile(os.path.join(GOLD, "epc-net.ckpt.index"), prefix + ".index")
    st = H.make_store("epc-net", O.seeded_weights("epc-net", 0, mode="init"), "cpu")
    with pytest.raises(FileNotFoundError):
        st.load_checkpoint(prefix)
    entries = tfb.read_index(prefix + ".index")
    w = O.seeded_weights("epc-net", 5)
    tensors = {}
    for name, e in entries.items():
        rel = name[len("query_triplets/"):] if name.startswith("query_triplets/") else None
        tensors[name] = w[rel] if rel in w else np.zeros(e.shape, dtype=e.dtype)
    tfb.write_checkpoint_payload(prefix, tensors)
    assert os.path.getsize(tfb.data_path_for(prefix)) == 56476940            # SURVEY.md 8c payload size
    # the index is the REFERENCE's: its entries carry the checksums of the trained weights, which this synthetic payload
    # cannot match -- the loader must notice (TensorFlow's BundleReader verifies them too), and load when told not to check
    with pytest.raises(ValueError, match="checksum"):
        st.load_checkpoint(prefix)
    unused = st.load_checkpoint(prefix, verify_crc=False)
    assert "beta1_power" in unused and any(u.endswith("/Adam") for u in unused)
    for k, v in w.items():
        assert np.array_equal(st.vars["query_triplets/" + k].numpy(), v)


def test_forward_fails_loudly_without_gpu():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    L, V = H.pkg("lib"), H.pkg("variables")
    V.reset_default_store(device="cpu", seed=0)
    M = H.pkg("models.epc-net")
    x = M.placeholder_inputs(1, 1, 64, 3)
    with V.variable_scope("query_triplets"), pytest.raises(L.EpcNetError):
        M.forward(x, False, params=H.PARAMS)


def test_submit_fails_loudly_without_gpu_and_micro_batch_mirror():
    """The throughput entry point has no CPU fallback either; lib.micro_batch_of mirrors micro_batch() of pipeline.hip
    (the workspace the library asks for is per pass, so it stops growing at the micro-batch)."""
    import ctypes
    L, E, V = H.pkg("lib"), H.pkg("engine"), H.pkg("variables")
    for arch, default in (("epc-net", 64), ("epc-net-l", 256)):
        cfg = E.make_cfg(arch, 4096, H.PARAMS)
        assert [L.micro_batch_of(cfg, n) for n in (1, default - 1, default, 10 * default)] == [1, default - 1, default, default]
        ws = [L.lib().epc_net_workspace_bytes(ctypes.byref(cfg), n) for n in (default - 1, default, 10 * default)]
        assert ws[0] < ws[1] == ws[2]
        cfg2 = E.make_cfg(arch, 4096, H.PARAMS, micro_batch=8)
        assert L.micro_batch_of(cfg2, 100) == 8 and L.micro_batch_of(cfg2, 3) == 3
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    st = V.reset_default_store(device="cpu", seed=0)
    eng = E.InferenceEngine("epc-net", H.PARAMS, st, in_flight=2)
    with pytest.raises(L.EpcNetError):
        eng.submit(torch.zeros(1, 64, 3))
    with pytest.raises(L.EpcNetError):
        eng.submit(torch.zeros(1, 64, 4))          # INPUT_DIM != 3 is rejected before any device work
    eng.drain()                                    # nothing submitted: a no-op


def test_forward_argument_errors():
    V = H.pkg("variables")
    V.reset_default_store(device="cpu", seed=0)
    M = H.pkg("models.epc-net")
    with pytest.raises(TypeError):
        M.forward(torch.zeros(1, 1, 64, 3), False)                        # params is required (epc-net.py:37)
    with pytest.raises(ValueError):
        M.forward(torch.zeros(1, 1, 64, 13), False, params=H.PARAMS)      # INPUT_DIM mismatch


def test_losses_match_oracle_and_reference_error_behaviour():
    M = H.pkg("models.epc-net")
    rng = np.random.RandomState(0)
    def unit(*s):
        a = rng.randn(*s); return a / np.linalg.norm(a, axis=-1, keepdims=True)
    q, pos, neg, oth = unit(3, 1, 256), unit(3, 2, 256), unit(3, 14, 256), unit(3, 1, 256)
    tq, tp, tn, to = (torch.from_numpy(a) for a in (q, pos, neg, oth))
    assert float(M.lazy_quadruplet_loss(tq, tp, tn, to, 0.5, 0.2)) == pytest.approx(O.lazy_quadruplet_loss(q, pos, neg, oth, 0.5, 0.2), rel=1e-12)
    assert float(M.quadruplet_loss(tq, tp, tn, to, 0.5, 0.2)) == pytest.approx(O.quadruplet_loss(q, pos, neg, oth, 0.5, 0.2), rel=1e-12)
    assert float(M.lazy_triplet_loss(tq, tp, tn, 0.5)) == pytest.approx(O.lazy_triplet_loss(q, pos, neg, 0.5), rel=1e-12)
    assert float(M.triplet_loss(tq, tp, tn, 0.5)) == pytest.approx(O.triplet_loss(q, pos, neg, 0.5), rel=1e-12)
    assert float(M.lazy_quadruplet_loss_sm(tq, tp, tn, to, 0.2)) == pytest.approx(O.lazy_quadruplet_loss_sm(q, pos, neg, oth, 0.2), rel=1e-12)
    with pytest.raises(NameError):                                         # models/epc-net.py:205 returns `soft_los`
        M.softmargin_loss(tq, tp, tn)
    with pytest.raises(NameError):
        M.quadruplet_loss_sm(tq, tp, tn, to, 0.2)
    L = importlib.import_module("epc-net_amd.models.epc-net-l")
    assert set(M.LOSS_NAMES) <= set(dir(L))                                # identical loss family in both files


def test_unused_reference_ops_are_named_not_silently_missing():
    tf_util = H.pkg("utils.tf_util")
    for name in ("conv2d", "conv3d", "dropout", "knn", "get_edge_feature", "pairwise_distance", "avg_pool2d"):
        with pytest.raises(NotImplementedError):
            getattr(tf_util, name)()


def test_plain_c_example_builds_against_the_header():
    """examples/epcnet_forward.c is a C99 program that uses the ABI without Python or torch: it must compile against
    include/epcnet.h and link against the built library (no GPU needed for that)."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(["make", "-C", os.path.join(root, "examples"), "clean"], check=True, capture_output=True)
    r = subprocess.run(["make", "-C", os.path.join(root, "examples")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert os.path.exists(os.path.join(root, "examples", "epcnet_forward"))


def test_step_inputs_recognised_as_slices_of_one_buffer():
    """training._joined_along_dim1: the concat of train.py:252 is skipped only for the exact consecutive dim-1 slices."""
    import torch
    T = H.pkg("training")
    joined = torch.arange(2 * 18 * 5 * 3, dtype=torch.float32).reshape(2, 18, 5, 3)
    sizes = [1, 2, 14, 1]
    parts = torch.split(joined, sizes, 1)
    assert torch.equal(T._joined_along_dim1(parts), joined)
    assert T._joined_along_dim1(parts).data_ptr() == joined.data_ptr()
    assert T._joined_along_dim1((parts[0], parts[2], parts[1], parts[3])) is None          # wrong order
    assert torch.equal(T._joined_along_dim1(parts[:3]), joined[:, :17])                   # a prefix is still a slice run
    assert T._joined_along_dim1(tuple(p.clone() for p in parts)) is None                  # separate tensors
    t = torch.arange(18 * 5 * 3, dtype=torch.float32).reshape(18, 5, 3)                   # bench.py's tuples: t[None, a:b]
    q = (t[None, 0:1], t[None, 1:3], t[None, 3:17], t[None, 17:18])
    assert torch.equal(T._joined_along_dim1(q), t[None])
    assert T._joined_along_dim1((t[None, 0:1], t[None, 2:3])) is None                     # a gap
