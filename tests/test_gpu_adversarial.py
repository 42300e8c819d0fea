"""Adversarial parity set: both arithmetics of the EPC-Net inference path (include/epcnet.h
EPC_PRECISION_F32 / EPC_PRECISION_FAST) and EPC-Net-L against the float32 oracle at N = 4096, on weights and clouds chosen
to defeat reduced-precision arithmetic.  ONE bar, no exemptions: descriptor L2 error <= 1e-4 (BASELINE.json north_star).

Weights:
  "benign": the seeded trained-like set of the other tests (oracle.seeded_weights) -- the degenerate CLOUDS are the point;
  "mild" / "hard" (oracle.adversarial_weights): Student-t matrices with x50 outlier output channels, gamma log-uniform in
      [0.3, 3] / [0.1, 10], moving statistics CALIBRATED to the layer's real statistics (what a trained checkpoint holds;
      without that the network is ill-conditioned in float32 itself: f32-vs-f64 distance 0.4, nothing to compare with);
      "hard" additionally floors 5 % of every layer's moving variances at ~1e-4 (1/sqrt(var + 1e-3) ~ 30).
Clouds: uniform, LiDAR-like, 25 / 50 / 75 % of the points copies of one point, 25 / 50 % zero padding, all-zero.

Conditioning.  On the degenerate clouds the neighbour mean is (count / 20) * x (the divisor stays 20 while ties select
hundreds of rows, utils/tf_util.py:660-665), which the four blocks compound; with outlier weights float32 ITSELF then drifts:
the oracle's float32 and float64 descriptors are up to 1e-4 apart on some of these cases (`gap` below; seeds are chosen so that
it stays below that).  The bar is therefore applied where it is meaningful and a relative one everywhere:
  * gap <= 3e-5 (float32 is well-posed):  |ours - float32 oracle| <= 1e-4, no exemptions (every benign-weight case is here);
  * always:  |ours - float64 oracle| <= max(1e-4, 4 * gap)  -- never worse than float32 arithmetic itself.

The fast arithmetic has a range (fp16 activations, |W' * 256| <= 65504): outside it the library must REFUSE (EPC_ERANGE at
pack time, NaN descriptor + status bit per cloud at run time) and `forward(check=True)` must return the f32-equivalent
result for the refused clouds.  INSIDE its range it is only as good as its 11-bit activations: on the heavy-tailed weight
sets it returns finite vectors that are 1e-2 off (measured; test_fast_arithmetic_limits_are_what_the_docs_say pins that), which
is why it is an explicit opt-in and the f32-equivalent arithmetic is the package default and the bench headline.
"""
import ctypes

import numpy as np
import pytest
import torch

import helpers as H
from helpers import O

pytestmark = pytest.mark.gpu

DESC_TOL = 1e-4
N = 4096
KINDS = ["uniform", "lidar", "repeat25", "repeat50", "repeat75", "zeropad25", "zeropad50", "zeros"]
LEVELS = {"benign": None, "mild": dict(seed=8, gamma_range=(0.3, 3.0), floor_frac=0.0),
          "hard": dict(seed=7, gamma_range=(0.1, 10.0), floor_frac=0.05)}
WELL_POSED = 3e-5

_cache = {}


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "GPU tests need an MI355X; there is no CPU fallback"
    return torch.device("cuda:0")


def case(arch, level):
    """(weights, clouds (8, N, 3), oracle float32 descriptors, oracle float64 descriptors) -- computed once per module."""
    key = (arch, level)
    if key not in _cache:
        if LEVELS[level] is None:
            w = O.seeded_weights(arch, 11)
        else:
            w = O.adversarial_weights(arch, calibrate_on=O.synthetic_clouds(2, N, 4242), **LEVELS[level])
        pc = np.concatenate([O.synthetic_clouds(1, N, 30 + i, k) for i, k in enumerate(KINDS)], 0)
        with np.errstate(all="ignore"):
            ref, _ = O.forward(pc[:, None], w, arch=arch)
            ref64, _ = O.forward(pc[:, None], w, arch=arch, dtype=np.float64)
        ref, ref64 = ref.reshape(len(KINDS), -1), ref64.reshape(len(KINDS), -1)
        assert np.isfinite(ref).all()
        gap = np.linalg.norm(ref - ref64, axis=1)
        assert gap.max() < 1.5e-4, gap                  # (seeds chosen so; see "Conditioning")
        if LEVELS[level] is None:
            assert gap.max() <= WELL_POSED, gap         # benign weights: every cloud is a plain 1e-4 case
        assert (gap <= WELL_POSED).sum() >= 2, gap      # at least the uniform and LiDAR-like clouds
        _cache[key] = (w, pc, ref, ref64)
    return _cache[key]


def errors(out, ref, ref64):
    """(error vs the float32 oracle, error vs the float64 oracle, the oracle's own float32-float64 gap) per cloud."""
    return (np.linalg.norm(out - ref, axis=1), np.linalg.norm(out - ref64, axis=1), np.linalg.norm(ref - ref64, axis=1))


def within_bar(e32, e64, gap):
    return (e32 <= DESC_TOL if gap <= WELL_POSED else True) and e64 <= max(DESC_TOL, 4 * gap)


def report(tag, e32, e64, gap, flags=None):
    print(tag, " ".join("%s %s" % (k, "flagged" if (flags is not None and flags[i]) else
                                   "%.1e%s" % (e32[i], "" if gap[i] <= WELL_POSED else "(gap %.0e, vs f64 %.1e)" % (gap[i], e64[i])))
                        for i, k in enumerate(KINDS)))


@pytest.mark.parametrize("level", list(LEVELS))
@pytest.mark.parametrize("arch", ["epc-net", "epc-net-l"])
def test_f32_equivalent_arithmetic_everywhere(dev, arch, level):
    w, pc, ref, ref64 = case(arch, level)
    eng, _ = H.make_engine(arch, w, dev, precision="f32")
    out = eng.forward(torch.from_numpy(pc).to(dev)).cpu().numpy()
    e32, e64, gap = errors(out, ref, ref64)
    report("adversarial %s/f32 %s:" % (arch, level), e32, e64, gap)
    assert np.isfinite(out).all() and np.allclose(np.linalg.norm(out, axis=1), 1, atol=1e-5)
    # BOTH criteria on every cloud of the committed seeds: within 1e-4 of the float32 oracle -- no waiver for
    # the ill-conditioned clouds, whose float32 oracle itself sits up to 1.5e-4 from the float64 result -- and within
    # max(1e-4, 4 x that gap) of the float64 oracle.
    assert (e32 <= DESC_TOL).all(), e32
    assert all(b <= max(DESC_TOL, 4 * g) for b, g in zip(e64, gap)), (e64, gap)
    assert eng.last_status(len(pc)) == [0] * len(pc)


@pytest.mark.parametrize("level", ["benign", "hard"])
def test_fast_arithmetic_is_right_or_refuses(dev, level):
    """EPC_PRECISION_FAST on ordinary weights (and on the set that drives it out of range): every cloud is either within the
    bar or flagged (NaN descriptor + status bit); a weight set that does not fit fp16 is refused at pack time."""
    L = H.pkg("lib")
    w, pc, ref, ref64 = case("epc-net", level)
    eng, _ = H.make_engine("epc-net", w, dev, precision="fast")
    try:
        out = eng.forward(torch.from_numpy(pc).to(dev), check=False).cpu().numpy()
    except L.EpcNetError as e:
        assert e.status == L.EPC_ERANGE and "EPC_PRECISION_F32" in str(e)
        print("adversarial epc-net/fast %s: refused at pack time (%s)" % (level, e))
        return
    status = eng.last_status(len(pc))
    e32, e64, gap = errors(np.nan_to_num(out), ref, ref64)
    report("adversarial epc-net/fast %s:" % level, e32, e64, gap, status)
    # Ordinary clouds must be TAKEN (with ordinary or mildly adversarial weights).  The clumped / padded 4096-point clouds are
    # refused even with benign weights: 1024+ identical points make the neighbour mean (count / 20) * x = 51 x .. 205 x, four
    # blocks compound it past 65504 -- the fast arithmetic's documented limit (include/epcnet.h); forward(check=True) re-runs them in f32.
    if level != "hard":
        assert status[0] == 0 and status[1] == 0
    for i, k in enumerate(KINDS):
        if status[i]:
            assert status[i] == L.EPC_STATUS_FP16_RANGE and np.isnan(out[i]).all(), k
        else:
            assert np.isfinite(out[i]).all() and within_bar(e32[i], e64[i], gap[i]), "%s: %.3e / %.3e" % (k, e32[i], e64[i])


@pytest.mark.parametrize("level", ["benign", "hard"])
def test_fast_with_check_reextracts_refused_clouds(dev, level):
    """forward(check=True): the clouds the fast arithmetic refuses (here: the clumped / padded ones, or all of them) come back
    in the f32-equivalent arithmetic."""
    w, pc, ref, ref64 = case("epc-net", level)
    eng, _ = H.make_engine("epc-net", w, dev, precision="fast")
    out = eng.forward(torch.from_numpy(pc).to(dev), check=True).cpu().numpy()
    e32, e64, gap = errors(out, ref, ref64)
    report("adversarial epc-net/fast+check %s:" % level, e32, e64, gap)
    assert np.isfinite(out).all() and all(within_bar(*t) for t in zip(e32, e64, gap))


def test_fast_arithmetic_limits_are_what_the_docs_say(dev):
    """Heavy-tailed weights that stay inside fp16's range ("mild"): the fast arithmetic is NOT within the bar on ordinary
    clouds (1e-2 on the uniform cloud when this was written) -- the documented reason it is opt-in.  If a later fast kernel
    passes here, this test should be turned into a parity test and the documentation changed."""
    w, pc, ref, ref64 = case("epc-net", "mild")
    eng, _ = H.make_engine("epc-net", w, dev, precision="fast")
    out = eng.forward(torch.from_numpy(pc[:2]).to(dev)).cpu().numpy()
    err = np.linalg.norm(np.nan_to_num(out) - ref[:2], axis=1)
    print("fast arithmetic on mildly adversarial weights: uniform %.1e lidar %.1e (bar 1e-4)" % (err[0], err[1]))
    assert np.isfinite(out).all()            # in range: nothing is flagged, the vectors are finite
    assert err.max() > DESC_TOL


def test_uncalibrated_outlier_weights_never_give_inf(dev):
    """Weights whose moving variances are NOT calibrated (variance down to 1e-4 with gamma up to 10 on un-normalised
    activations): float32 itself is ill-conditioned there, so there is no parity bar -- but the fast arithmetic must refuse
    (EPC_ERANGE: |W' * 256| > 65504) and the f32-equivalent path must stay finite and unit-norm."""
    L = H.pkg("lib")
    w = O.adversarial_weights("epc-net", 3, calibrate_on=None)
    pc = torch.from_numpy(O.synthetic_clouds(2, 512, 1)).to(dev)
    eng, _ = H.make_engine("epc-net", w, dev, precision="fast")
    with pytest.raises(L.EpcNetError) as ei:
        eng.forward(pc)
    assert ei.value.status == L.EPC_ERANGE
    eng, _ = H.make_engine("epc-net", w, dev, precision="f32")
    out = eng.forward(pc)
    assert bool(torch.isfinite(out).all()) and float((out.norm(dim=1) - 1).abs().max()) < 1e-4


@pytest.mark.parametrize("arch,prec", [("epc-net", "fast"), ("epc-net", "f32"), ("epc-net-l", "f32")])
def test_non_finite_coordinates_poison_only_their_cloud(dev, arch, prec):
    """A NaN / Inf coordinate: the reference's graph returns NaN for that cloud (every a_ij involving the point is NaN,
    utils/tf_util.py:651-656) and the other clouds of the batch are untouched (inference BN uses stored statistics).  The
    kernels must not fault (short neighbour lists are padded) and must flag the cloud."""
    L = H.pkg("lib")
    w = O.seeded_weights(arch, 0)
    pc = O.synthetic_clouds(4, 1024, 9)
    bad = pc.copy()
    bad[1, 17, 2] = np.nan
    bad[3, 1000, 0] = np.inf
    eng, _ = H.make_engine(arch, w, dev, precision=prec)
    clean = eng.forward(torch.from_numpy(pc).to(dev)).clone()
    out = eng.forward(torch.from_numpy(bad).to(dev))
    status = eng.last_status(4)     # (an Inf coordinate also drives conv1 out of fp16's range: both bits in fast precision)
    assert [s & L.EPC_STATUS_NONFINITE_INPUT for s in status] == [0, 1, 0, 1] and status[0] == status[2] == 0
    assert bool(torch.isnan(out[1]).all()) and bool(torch.isnan(out[3]).all())
    assert torch.equal(out[0], clean[0]) and torch.equal(out[2], clean[2])


def test_fp16_range_flag_on_large_activations(dev):
    """Benign weights, coordinates scaled by 1e6: conv1's output leaves fp16 -- the fast path flags the cloud (status bit,
    NaN descriptor), the f32-equivalent path and forward(check=True) return the oracle's descriptor."""
    L = H.pkg("lib")
    w = O.seeded_weights("epc-net", 1)
    pc = O.synthetic_clouds(2, 512, 3)
    pc[1] *= 1e6
    ref, _ = O.forward(pc[:, None], w, arch="epc-net")
    ref = ref.reshape(2, -1)
    x = torch.from_numpy(pc).to(dev)
    fast, _ = H.make_engine("epc-net", w, dev, precision="fast")
    out = fast.forward(x, check=False)
    assert fast.last_status(2) == [0, L.EPC_STATUS_FP16_RANGE]
    assert bool(torch.isnan(out[1]).all()) and np.linalg.norm(out[0].cpu().numpy() - ref[0]) <= DESC_TOL
    got = fast.forward(x, check=True).cpu().numpy()                      # re-extracts the flagged cloud in f32
    assert np.linalg.norm(got - ref, axis=1).max() <= DESC_TOL
    eng, _ = H.make_engine("epc-net", w, dev, precision="f32")
    got = eng.forward(x).cpu().numpy()
    assert np.linalg.norm(got - ref, axis=1).max() <= DESC_TOL
