"""CPU suite: oracle DNA coder (a14 + a16 framing) against the reference's own `dna` stream payloads
(byte-exact), fed with the reference's tuple streams (golden es.bin) and reference-read set."""
import hashlib
import pytest
from oracle import pyoracle as O
from util import PLAIN_CONFIGS, golden


def dna_parts(g):
    rs = g.reads
    dc = O.DnaCoder(g.p("c"), g.p("level"), g.p("n_pseudo"))
    has_n = rs.has_n()
    for i in range(rs.n_reads):
        if g.accept[i] and not has_n[i]:
            dc.add_ref(rs.read(i))
    bounds = rs.pack_bounds()
    parts = []
    for pi in range(len(bounds) - 1):
        for i in range(bounds[pi], bounds[pi + 1]):
            dc.encode(g.es[i][2], g.es[i][1])
        parts.append((int(bounds[pi + 1] - bounds[pi]), dc.finish_part()))
    return parts


@pytest.mark.parametrize("cfg", PLAIN_CONFIGS)
def test_dna_stream_is_byte_identical_to_reference(cfg):
    g = golden(cfg)
    got = [[n, len(p), hashlib.sha256(p).hexdigest()] for n, p in dna_parts(g)]
    assert got == g.spec["streams"]["dna"]["parts"]


def test_tuple_stream_golden_is_consistent():
    # es.bin sanity (App. A layout): tuple counts and plain reads reproduce the bases
    import numpy as np
    g = golden("c1_ont_default")
    for i in (0, 17, 99):
        pack, nt, raw = g.es[i]
        assert raw[0] >> 4 == 9 and nt == len(raw)          # start_plain + one byte per base
        assert np.array_equal(np.frombuffer(raw[1:], np.uint8) & 0xf, g.reads.read(i))
