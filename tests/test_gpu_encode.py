"""GPU parity suite for the edit-script encoder (a10-a12): the tuple streams cl_encode_reads produces from the
reference's candidate lists must equal, byte for byte, the streams tapped from the unmodified reference
(golden es.bin) — gap alignment with edlib's tie-breaking, indel canonicalisation, static and adaptive (per reader
pack) cost decisions, recursion into alternative references, tuple run-length rules."""
import numpy as np
import pytest
import torch
from util import golden
from test_gpu_dna import ref_subset

pytestmark = pytest.mark.gpu

# presets (arg_parse.cpp:89-408 / SURVEY App. B): min part length to consider an alternative read, max recursion
PRESET_BY_LEVEL = {1: (64, 3), 2: (48, 5), 3: (48, 6)}


def gpu_streams(ctx, g, pack_bounds=None):
    rs = g.reads
    c = g.p("c")
    has_n = rs.has_n()
    accept = g.accept.astype(bool) & ~has_n
    reads = ctx.pack_readset(rs)
    refs = ctx.pack_readset(ref_subset(rs, accept))
    cand = np.full((rs.n_reads, c), 0xffffffff, np.uint32)
    cn = np.zeros(rs.n_reads, np.uint32)
    for i, e in enumerate(g.cands):
        cn[i] = len(e["refs"])
        cand[i, :cn[i]] = e["refs"]
    anc = ctx.anchor_candidates(reads, refs, torch.from_numpy(cand.view(np.int32)).to(ctx.device), torch.from_numpy(cn.view(np.int32)).to(ctx.device), g.p("a"))
    min_alt, max_rec = PRESET_BY_LEVEL[g.p("level")]
    es, off, nt = ctx.encode_reads(reads, refs, anc, g.p("a"), min_alt, max_rec, 1.0, rs.pack_bounds() if pack_bounds is None else pack_bounds)
    es, off, nt = es.cpu().numpy(), off.cpu().numpy(), nt.cpu().numpy()
    anc.free(); refs.free(); reads.free()
    return es, off, nt


@pytest.mark.parametrize("cfg", ["c3_clr_ratio", "s6m_ont", "s3m_ont_n_ratio", "c1_ont_default", "c6_ont_org"])
def test_tuple_streams_equal_reference(ctx, cfg):
    g = golden(cfg)
    es, off, nt = gpu_streams(ctx, g)
    n_es = 0
    bad = []
    for i in range(g.reads.n_reads):
        got = es[off[i]:off[i + 1]].tobytes()
        if nt[i] != g.es[i][1] or got != g.es[i][2]:
            bad.append(i)
        n_es += len(got) > 0 and got[0] >> 4 == 10
    assert not bad, f"{len(bad)} of {g.reads.n_reads} reads differ, first {bad[:10]}"
    if cfg in ("c3_clr_ratio", "s6m_ont", "s3m_ont_n_ratio"):
        assert n_es > 10                                    # the edit-script path is really exercised
