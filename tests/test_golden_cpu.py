"""The CPU restatement (oracle/) against the committed golden vectors G1-G6 (tests/golden/vectors.npz).

The vectors were produced by scripts/make_golden_vectors.py from the same restatement -- the reference itself cannot
run here -- so what this file pins is that the oracle, its seeded generators and numpy's random stream have not moved
since the vectors were committed.  Integer results must be identical; float results may differ by BLAS summation order.
"""
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_spec = importlib.util.spec_from_file_location("make_golden_vectors",
                                               os.path.join(ROOT, "scripts", "make_golden_vectors.py"))
G = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(G)
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "vectors.npz"))


def _compare(out, rtol=0.0, atol=0.0):
    assert out, "generator produced nothing"
    for key, val in out.items():
        ref = GOLD[key]
        val = np.asarray(val)
        assert val.shape == ref.shape and val.dtype == ref.dtype, key
        if np.issubdtype(ref.dtype, np.integer):
            assert np.array_equal(val, ref), key
        else:
            assert np.allclose(val, ref, rtol=rtol, atol=atol), (key, float(np.abs(val - ref).max()))


def test_fixture_is_complete():
    groups = {k.split("/")[0] for k in GOLD.files}
    assert groups == {"g1", "g2", "g3", "g4", "g5", "g6"}
    assert len([k for k in GOLD.files if k.startswith("g1/") and k.endswith("/cnt")]) == len(G.KNN_CASES)


def test_g1_knn_graph_is_bit_identical():
    out = {}
    G.g1(out)
    _compare(out)                       # elementwise float32 arithmetic and integer selection: no tolerance at all
    zeros = GOLD["g1/zeros_128_0/cnt"]
    assert np.all(zeros == 128)         # all-zero padding cloud: every point is everyone's neighbour
    assert GOLD["g1/lattice_512_0/cnt"].max() > 20 and GOLD["g1/uniform_4096_3/cnt"].max() == 20


def test_g2_proxyconv_block():
    out = {}
    G.g2(out)
    _compare(out, atol=2e-6)


def test_g3_gvlad():
    out = {}
    G.g3(out)
    _compare(out, atol=2e-6)


def test_g4_full_networks_and_fp32_budget():
    out = {}
    G.g4(out)
    _compare(out, atol=2e-6)
    for arch in ("epc-net", "epc-net-l"):
        a, b = GOLD["g4/%s/desc_f32" % arch], GOLD["g4/%s/desc_f64" % arch]
        assert np.allclose(np.linalg.norm(b, axis=1), 1.0, atol=1e-12)
        assert np.linalg.norm(a - b, axis=1).max() <= 5e-6        # the budget the 1e-4 GPU bar sits above


def test_g5_training_step_quantities():
    out = {}
    G.g5(out)                           # also re-runs the central-difference check of the stored gradients
    _compare(out, rtol=1e-7, atol=1e-10)
    assert 0.0 < float(GOLD["g5/loss"]) < 4.7 and float(GOLD["g5/bn_decay"]) == 0.5


def test_g6_retrieval_bookkeeping():
    out = {}
    G.g6(out)
    _compare(out, rtol=1e-12)
    rec = GOLD["g6/recall"]
    assert rec.shape == (25,) and np.all(np.diff(rec) >= 0) and rec[-1] <= 100.0
