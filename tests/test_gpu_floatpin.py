"""The float path of the encoder's cost decisions, pinned (SURVEY §7 hard part 3).

Every decision of a11 compares double sums of terms count * -log2(count * (1/total)) (CEntropyEstimator::calc_logs,
utils.h:800-810; CEntropy, utils.h:706-757).  Additions and multiplications are IEEE (library built with -ffp-contract=off,
like the x86-64 reference), so the one place the device could differ from the reference's glibc is log2 itself.  Here the
device's values (cl_estimator_logs: same translation unit, flags and expression as k_estimator / gap_stats) are compared
BIT FOR BIT with host glibc over every pair total <= 4096 and 10^7 sampled pairs up to the estimator's bound 2^20, through
digests committed by tests/golden/make_floatpin.py (generated in the build container)."""
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import make_floatpin as FP  # noqa: E402

FIX = json.load(open(os.path.join(HERE, "golden", "floatpin.json")))


def test_host_libm_matches_fixture_and_known_answers():
    """CPU: the oracle's host form reproduces the committed digests on this machine, and is exact where exactness is known."""
    from oracle import pyoracle as O
    for b in (0, 1, 31, 63):
        c, t = FP.dense_block(b)
        assert FP.digest(O.estimator_logs(c, t)) == FIX["dense"][b]
    c, t = FP.sampled_block(3)
    assert FP.digest(O.estimator_logs(c, t)) == FIX["sampled"][3]
    # 1/2^j is exact, count = 2^i: log2 must be the exact integer i - j
    i, j = np.meshgrid(np.arange(0, 21), np.arange(0, 21))
    m = i <= j
    got = O.estimator_logs((1 << i[m]).astype(np.uint32), (1 << j[m]).astype(np.uint32))
    assert np.array_equal(got, (j[m] - i[m]).astype(np.float64))
    assert O.estimator_logs(np.array([0], np.uint32), np.array([7], np.uint32))[0] == 0.0


def _device(ctx, c, t):
    import torch
    dc = torch.from_numpy(c.view(np.int32)).to(ctx.device)
    dt = torch.from_numpy(t.view(np.int32)).to(ctx.device)
    return ctx.estimator_logs(dc, dt).cpu().numpy()


def _explain(ctx, c, t, dev):
    from oracle import pyoracle as O
    host = O.estimator_logs(c, t)
    bad = np.nonzero(dev.view(np.uint64) != host.view(np.uint64))[0]
    ulp = np.abs(dev.view(np.int64)[bad] - host.view(np.int64)[bad])
    ex = [(int(c[k]), int(t[k]), float(host[k]).hex(), float(dev[k]).hex()) for k in bad[:5]]
    return f"{len(bad)} of {len(c)} values differ from host glibc (max {int(ulp.max()) if len(bad) else 0} ulp); first: {ex}"


@pytest.mark.gpu
def test_device_log2_equals_host_glibc_all_totals_to_4096(ctx):
    for b in range(FP.DENSE_MAX // FP.DENSE_BLOCK):
        c, t = FP.dense_block(b)
        dev = _device(ctx, c, t)
        assert FP.digest(dev) == FIX["dense"][b], f"totals {64 * b + 1}..{64 * b + 64}: " + _explain(ctx, c, t, dev)


@pytest.mark.gpu
def test_device_log2_equals_host_glibc_sampled_to_2p20(ctx):
    for b in range(FP.N_SAMPLED // FP.SAMPLED_BLOCK):
        c, t = FP.sampled_block(b)
        dev = _device(ctx, c, t)
        assert FP.digest(dev) == FIX["sampled"][b], f"sampled block {b}: " + _explain(ctx, c, t, dev)


@pytest.mark.gpu
def test_device_log2_exact_powers_and_empty_counter(ctx):
    i, j = np.meshgrid(np.arange(0, 21), np.arange(0, 21))
    m = i <= j
    got = _device(ctx, (1 << i[m]).astype(np.uint32), (1 << j[m]).astype(np.uint32))
    assert np.array_equal(got, (j[m] - i[m]).astype(np.float64))
    assert _device(ctx, np.array([0, 0], np.uint32), np.array([7, 1], np.uint32)).tolist() == [0.0, 0.0]
