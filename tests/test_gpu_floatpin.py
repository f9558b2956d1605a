"""The float path of the encoder's cost decisions, pinned (SURVEY §7 hard part 3).

Every decision of a11 compares double sums of terms count * -log2(count * (1/total)) (CEntropyEstimator::calc_logs,
utils.h:800-810; CEntropy, utils.h:706-757).  Additions and multiplications are IEEE (library built with -ffp-contract=off,
like the x86-64 reference), so the one place the device could differ from the reference's glibc is log2 itself — and the
device's own log2 does differ (1 ulp on 2.7 % of these arguments, measured).  The library therefore evaluates glibc's
published log2 algorithm itself (colord_amd/csrc/log2_glibc.hpp, constants from this image's libm.a).  Pinned here:
  CPU: the host build of that header == this machine's libm, bit for bit, on 10^7 doubles incl. the branch boundaries;
  GPU: the device's values (cl_estimator_logs: same translation unit, flags and expression as k_estimator / gap_stats)
       == host glibc over every pair total <= 4096 and 10^7 sampled pairs up to the estimator's bound 2^20, through digests
       committed by tests/golden/make_floatpin.py (generated in the build container)."""
import ctypes as C
import subprocess
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


@pytest.fixture(scope="module")
def restated(tmp_path_factory):
    """Host build of the product's log2_glibc.hpp (tests/tools/log2_host.cpp)."""
    so = str(tmp_path_factory.mktemp("log2") / "liblog2_host.so")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC", os.path.join(HERE, "tools", "log2_host.cpp"), "-o", so])
    L = C.CDLL(so)
    L.log2_restated.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
    L.estimator_logs_restated.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
    return L


def test_restated_glibc_log2_equals_libm_bit_for_bit(restated):
    """The algorithm the device runs, compiled for the host, against the libm the reference calls."""
    import math
    rng = np.random.default_rng(5)
    n = 2_500_000
    lo, hi = 1.0 - float.fromhex("0x1.5b51p-5"), 1.0 + float.fromhex("0x1.6ab2p-5")          # where glibc switches polynomials
    x = np.concatenate([np.exp(rng.uniform(-60, 60, n)), 1 + rng.uniform(-0.06, 0.06, n), rng.uniform(1e-7, 1.0, n),
                        lo + rng.uniform(-1e-5, 1e-5, n // 2), hi + rng.uniform(-1e-5, 1e-5, n // 2),
                        np.array([1.0, lo, hi, np.nextafter(lo, 0), np.nextafter(hi, 2), 0.5, 2.0, 2.0 ** -20, 1 / 3, float.fromhex("0x1.6p-1"), float.fromhex("0x1.6p0")])])
    got = np.empty(len(x))
    restated.log2_restated(x.ctypes.data, len(x), got.ctypes.data)
    # libm through the oracle's helper: -log2(count * (1/total)) with total = 1 gives -log2(count); direct values below
    sample = rng.choice(len(x), 200_000, replace=False)
    ref = np.array([math.log2(v) for v in x[sample]])                  # CPython's math.log2 is libm's log2
    assert np.array_equal(got[sample].view(np.uint64), ref.view(np.uint64))
    from oracle import pyoracle as O
    for b in (0, 17, 63):
        c, t = FP.dense_block(b)
        r = np.empty(len(c))
        restated.estimator_logs_restated(c.ctypes.data, t.ctypes.data, len(c), r.ctypes.data)
        assert np.array_equal(r.view(np.uint64), O.estimator_logs(c, t).view(np.uint64))
        assert FP.digest(r) == FIX["dense"][b]
    for b in range(FP.N_SAMPLED // FP.SAMPLED_BLOCK):
        c, t = FP.sampled_block(b)
        r = np.empty(len(c))
        restated.estimator_logs_restated(c.ctypes.data, t.ctypes.data, len(c), r.ctypes.data)
        assert FP.digest(r) == FIX["sampled"][b]


def test_log2_table_is_this_images_libm():
    """The committed constants are the ones in the image's libm.a (regenerate with tools/gen_glibc_log2_table.py)."""
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import gen_glibc_log2_table as G
    try:
        vals, _ = G.table_from_libm()
    except Exception as e:                                             # no static libm on this machine
        pytest.skip(f"libm.a not available: {e}")
    txt = open(os.path.join(os.path.dirname(HERE), "colord_amd", "csrc", "glibc_log2_table.inc")).read()
    import re
    have = [float.fromhex(h) for h in re.findall(r"-?0x[0-9a-f.]+p[+-]\d+", txt)]
    assert have == vals


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
