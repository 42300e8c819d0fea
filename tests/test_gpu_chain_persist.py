"""GPU tests of the PERSISTENT forward of the backbone chain (csrc/train_chain_persist.hip: one launch, grid-wide barriers instead of
kernel boundaries; models/epc-net.py:66-134 in training mode) against the launch chain it replaces (csrc/train_chain.hip) -- the same
products in the same arithmetic on the same operands; only the summation order of the batch statistics differs (group partials by row
range instead of sixteen strided slices)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import helpers as H
from helpers import O
from test_gpu_chain import _backbone

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _with_persist(on, fn):
    ops = H.pkg("ops")
    prev = ops.CHAIN_PERSIST_FWD
    ops.CHAIN_PERSIST_FWD = on
    try:
        out = fn()
        ops.chain_persist_check()
        return out
    finally:
        ops.CHAIN_PERSIST_FWD = prev


def test_first_launch_in_a_child_process():
    """The very first persistent launches of a session run in a CHILD with a hard time limit: a grid barrier that could not complete
    (a workgroup not resident) must end in the kernel's own bounded spin and an EPC_EHIP, never in a hung device."""
    code = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r); sys.path.insert(0, %r)
import helpers as H
from helpers import O
from test_gpu_chain import _backbone
ops = H.pkg("ops")
dev = torch.device("cuda:0")
assert H.pkg("lib").lib().epc_chain_persist_ok(4 * 256) == 1
w = O.seeded_weights("epc-net", 4); pc = O.synthetic_clouds(4, 256, 5)
ops.CHAIN_PERSIST_FWD = True
a = _backbone("epc-net", w, pc, dev, True)
ops.chain_persist_check()
ops.CHAIN_PERSIST_FWD = False
b = _backbone("epc-net", w, pc, dev, True)
print("max diff", float(np.abs(a[0] - b[0]).max()))
assert np.isfinite(a[0]).all() and np.abs(a[0] - b[0]).max() <= 1e-5 * max(np.abs(b[0]).max(), 1.0)
print("CHILD OK")
""" % (ROOT, os.path.join(ROOT, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "CHILD OK" in r.stdout, (r.stdout[-2000:], r.stderr[-2000:])


@pytest.mark.parametrize("arch,ncl,n,kind,precision", [("epc-net", 3, 256, "uniform", "bf16x6"), ("epc-net-l", 5, 96, "uniform", "bf16x6"),
                                                      ("epc-net", 2, 32, "uniform", "bf16x6"),          # two workgroups of one wave: two groups of one
                                                      ("epc-net", 2, 256, "ties", "bf16x6"), ("epc-net", 18, 4096, "uniform", "bf16x6"),
                                                      ("epc-net", 22, 4096, "uniform", "bf16"), ("epc-net", 7, 1000, "uniform", "bf16")])
def test_persistent_chain_equals_the_launch_chain(dev, arch, ncl, n, kind, precision):
    """Outputs to 1e-5 of their scale, moving statistics to 2e-6, gradients to 1e-4 relative L2 (small sizes; at full size ReLU-mask
    flips between two float32 summation orders move single gradient elements: the bars of test_gpu_chain)."""
    assert H.pkg("lib").lib().epc_chain_persist_ok(ncl * n) == 1
    w = O.seeded_weights(arch, 4)
    pc = O.synthetic_clouds(ncl, n, 11)
    if kind == "ties":
        pc[0, 40:120] = pc[0, 7]
        pc[1] = 0.0
    a = _with_persist(True, lambda: _backbone(arch, w, pc, dev, True, precision=precision))
    b = _with_persist(False, lambda: _backbone(arch, w, pc, dev, True, precision=precision))
    assert np.isfinite(a[0]).all()
    small = n < 4096 and kind == "uniform"
    bar_out, bar_grad = (1e-5, 1e-4) if small else ((5e-5, 2e-2) if kind == "uniform" else (1e-4, 2e-2))
    if precision == "bf16":
        # One bf16 value per operand: the last bit of a batch moment moves rounding boundaries of the operands it normalises, so two
        # correct implementations of this arithmetic differ by single bf16 ulps of single elements (2^-9 of an element, grown through
        # twelve layers).  What is held instead: the two lie equally far from the f32-accurate chain (a systematic error would show
        # there) and much closer to each other than to it.
        c = _with_persist(False, lambda: _backbone(arch, w, pc, dev, True, precision="bf16x6"))
        rel = lambda x, y: np.linalg.norm(x - y) / np.linalg.norm(y)
        dab, dac, dbc = rel(a[0], b[0]), rel(a[0], c[0]), rel(b[0], c[0])
        print("bf16: persistent vs launch %.2e; vs the f32-accurate chain: persistent %.2e, launch %.2e" % (dab, dac, dbc))
        assert abs(dac - dbc) <= 0.1 * dbc and dab <= 0.5 * dbc, (dab, dac, dbc)
        bar_out, bar_grad = 3e-2, 5e-2
    assert np.abs(a[0] - b[0]).max() <= bar_out * max(np.abs(b[0]).max(), 1.0)
    worst = (0.0, "")
    for k, gb in b[1].items():
        ga = a[1][k]
        if gb is None or k.endswith("/biases"):
            assert ga is None or np.abs(ga).max() <= 1e-4
            continue
        rel = np.linalg.norm(ga - gb) / max(np.linalg.norm(gb), 1e-30)
        if precision == "bf16":      # as the outputs: as far from the f32-accurate gradient as the launch chain's, closer to it than to that
            gc = c[1][k]
            rac, rbc = np.linalg.norm(ga - gc) / np.linalg.norm(gc), np.linalg.norm(gb - gc) / np.linalg.norm(gc)
            worst = max(worst, (rel / rbc, k))
            assert rac <= 1.3 * rbc + 1e-3 and rel <= 1.2 * rbc + 1e-3, (k, rel, rac, rbc)
            continue
        worst = max(worst, (rel, k))
        assert rel <= bar_grad, (k, rel)
    for k, vb in b[2].items():
        bar = (2e-6 if kind == "uniform" else 5e-5) if precision != "bf16" else 5e-3      # (bf16: the statistics of bf16-rounded products)
        assert np.abs(a[2][k] - vb).max() <= 2e-6 + bar * np.abs(vb).max(), k
    print("persistent vs launch chain, %s %dx%d (%s, %s): cat max diff %.2e, worst gradient rel L2 (bf16: / the launch chain's distance from f32) %.2e (%s)"
          % (arch, ncl, n, kind, precision, np.abs(a[0] - b[0]).max(), worst[0], worst[1]))


def test_persistent_chain_is_bit_reproducible(dev):
    w = O.seeded_weights("epc-net", 4)
    pc = O.synthetic_clouds(18, 4096, 5)
    a = _with_persist(True, lambda: _backbone("epc-net", w, pc, dev, True))
    b = _with_persist(True, lambda: _backbone("epc-net", w, pc, dev, True))
    assert np.array_equal(a[0], b[0])
    for k in a[1]:
        assert (a[1][k] is None and b[1][k] is None) or np.array_equal(a[1][k], b[1][k]), k
    for k in a[2]:
        assert np.array_equal(a[2][k], b[2][k]), k


def test_abandoned_barrier_is_reported_not_hung(dev):
    """A launch whose barrier cannot complete (here: a spin budget of ONE tick, so the first workgroup that has to wait gives up) sets
    the sticky error word, every workgroup leaves, epc_chain_persist_status returns EPC_EHIP, later launches return at once until the
    reset -- and after the reset the chain runs again and agrees with the launch chain."""
    L, ops = H.pkg("lib"), H.pkg("ops")
    lib = L.lib()
    w = O.seeded_weights("epc-net", 4)
    pc = O.synthetic_clouds(18, 4096, 5)
    ws = ops.chain_workspace(dev)
    words = ws.view(torch.int32)
    prev_ticks, prev_flags = ops.CHAIN_SPIN_TICKS, ops.CHAIN_PERSIST_FWD
    try:
        ops.CHAIN_PERSIST_FWD = True
        ops.CHAIN_SPIN_TICKS = 1
        _backbone("epc-net", w, pc, dev, True)
        torch.cuda.synchronize()
        assert lib.epc_chain_persist_status(ws.data_ptr(), L.current_stream()) == -3
        assert int(words[2112]) == 1
        seq = int(words[0])
        _backbone("epc-net", w, pc, dev, True)          # returns at once (results undefined), the word stays set
        assert lib.epc_chain_persist_status(ws.data_ptr(), L.current_stream()) == -3
        with pytest.raises(L.EpcNetError):
            ops.chain_persist_check()                    # raises and resets
        ops.CHAIN_SPIN_TICKS = prev_ticks
        torch.cuda.synchronize()
        assert int(words[2112]) == 0 and int(words[2048]) == 0 and int(words[0]) == seq + 1
        a = _backbone("epc-net", w, pc, dev, True)
        ops.chain_persist_check()
        ops.CHAIN_PERSIST_FWD = False
        b = _backbone("epc-net", w, pc, dev, True)
        assert np.isfinite(a[0]).all() and np.abs(a[0] - b[0]).max() <= 5e-5 * np.abs(b[0]).max()
    finally:
        ops.CHAIN_SPIN_TICKS = prev_ticks
        ops.CHAIN_PERSIST_FWD = prev_flags


def test_rows_beyond_the_persistent_form_take_the_launch_chain(dev):
    """More than twelve 32-row tiles per workgroup (25 x 4096 rows on 256 CUs: thirteen) is not covered: epc_chain_persist_ok says so and
    ProxyConvChain takes the launches -- same interface, no error."""
    lib = H.pkg("lib").lib()
    assert lib.epc_chain_persist_ok(24 * 4096) == 1 and lib.epc_chain_persist_ok(25 * 4096) == 0
    w = O.seeded_weights("epc-net-l", 4)
    pc = O.synthetic_clouds(25, 4096, 5)
    a = _with_persist(True, lambda: _backbone("epc-net-l", w, pc, dev, True))
    b = _with_persist(False, lambda: _backbone("epc-net-l", w, pc, dev, True))
    assert np.array_equal(a[0], b[0])
