"""The two forms of the synthetic-input generator (csrc/synth/ontsim_core.h) are bit-identical: what bench.py generates in
HBM is what the host form writes as FASTQ for the reference CPU path."""
import numpy as np
import pytest
from colord_amd import ontsim

pytestmark = pytest.mark.gpu


def test_device_generator_equals_host_generator(ctx):
    t = ontsim.ReadTable(seed=5, genome_len=800_000, target_bases=12_000_000)
    hc, ho, hq = ontsim.host_reads(t)
    dc, do, dq = ontsim.device_reads(t, ctx.device)
    assert np.array_equal(do.cpu().numpy(), ho)
    assert np.array_equal(dc.cpu().numpy(), hc) and np.array_equal(dq.cpu().numpy(), hq)
    # a sub-range generated on its own is the same bytes (chunks of bench.py)
    a, b = t.n_reads // 3, 2 * t.n_reads // 3
    sc, so, sq = ontsim.device_reads(t, ctx.device, a, b)
    assert np.array_equal(sc.cpu().numpy(), hc[ho[a]:ho[b]]) and np.array_equal(sq.cpu().numpy(), hq[ho[a]:ho[b]])
