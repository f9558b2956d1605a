#!/usr/bin/env python3
"""The device-wide exclusive scans through the library's sort (its digit histograms are scanned by them) and directly through the
candidate stage are covered by the suite; this checks the look-back scan alone on awkward sizes through cl_sort_u64 (sizes around
tile multiples, 1 .. 40 M keys) against numpy."""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from colord_amd.device import Context, _check
ctx = Context(0)
rng = np.random.default_rng(1)
for n in (1, 2, 255, 4095, 4096, 4097, 65536 + 17, 1_000_003, 40_000_000):
    k = rng.integers(0, 1 << 40, n, dtype=np.uint64)
    d = torch.from_numpy(k.view(np.int64)).to(ctx.device)
    _check(ctx, ctx.lib.cl_sort_u64(ctx.h, d.data_ptr(), n, 0, 40))
    assert np.array_equal(d.cpu().numpy().view(np.uint64), np.sort(k)), n
print("scan / sort sizes ok")
