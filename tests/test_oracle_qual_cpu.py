"""CPU suite: oracle range coder + quality coder (a13, a15, a16 framing) against the reference's own
`qual` stream payloads (byte-exact) and a decode round trip that must equal the reference's decompressed
qualities (the .quan files for lossy modes)."""
import gzip
import hashlib
import os
import numpy as np
import pytest
from oracle import pyoracle as O
from util import ALL_CONFIGS, DATA, golden


def encode_all(g):
    rs = g.reads
    mode, src, level = g.p("qual_mode"), g.p("source"), g.p("level")
    qc = O.QualCoder(True, mode, src, level)
    bounds = rs.pack_bounds()
    parts = []
    for pi in range(len(bounds) - 1):
        for i in range(bounds[pi], bounds[pi + 1]):
            flags = O.es_flags(g.es[i][2], len(rs.read(i))) if level > 1 else None
            qc.encode(rs.read(i), rs.qual(i), flags)
        parts.append(qc.finish_part())
    return parts


@pytest.mark.parametrize("cfg", ALL_CONFIGS)
def test_qual_stream_is_byte_identical_to_reference(cfg):
    g = golden(cfg)
    parts = encode_all(g)
    exp = g.spec["streams"]["qual"]["parts"]
    assert [[0, len(p), hashlib.sha256(p).hexdigest()] for p in parts] == exp


@pytest.mark.parametrize("cfg", ["c1_ont_default", "c2_hifi_org", "c7_hifi_balanced", "s3m_ont_n_ratio"])
def test_qual_decode_round_trip(cfg):
    g = golden(cfg)
    rs = g.reads
    mode, src, level = g.p("qual_mode"), g.p("source"), g.p("level")
    parts = encode_all(g)
    dec = O.QualCoder(False, mode, src, level)
    bounds = rs.pack_bounds()
    out = []
    for pi, part in enumerate(parts):
        dec.set_input(part)
        for i in range(bounds[pi], bounds[pi + 1]):
            flags = O.es_flags(g.es[i][2], len(rs.read(i))) if level > 1 else None
            out.append(dec.decode(rs.read(i), flags))
    got = np.concatenate(out)
    if mode == 0:                                   # -q org: lossless
        assert np.array_equal(got, rs.quals)
    quan = os.path.join(DATA, (g.spec.get("input") or "") + ".quan.gz")
    if g.spec.get("input") and os.path.exists(quan) and mode != 0 and "balanced" not in cfg and "default" in cfg:
        from colord_amd.fastq import read_fastx
        exp = read_fastx(quan)                      # the reference CI's expected decompressed output
        assert np.array_equal(got, exp.quals)
    # quantised modes: decoded values stay inside the bin of the original value
    if mode in (1, 2, 3):
        fwd = np.asarray(O.QUAL_DEFAULTS[mode][0])
        assert np.array_equal(np.searchsorted(fwd, got.astype(np.int64) - 33, side="right"),
                              np.searchsorted(fwd, rs.quals.astype(np.int64) - 33, side="right"))


def test_range_coder_known_answer():
    # a 2-symbol stream through a fresh binary model: bytes fixed by the coder arithmetic (sub_rc.h:83-100,203-210)
    qc = O.QualCoder(True, O.QM["fix2"], 0, 1)
    q = np.array([33 + 3, 33 + 20, 33 + 20, 33 + 1], np.uint8)
    qc.encode(np.array([0, 1, 2, 3], np.uint8), q)
    p = qc.finish_part()
    assert len(p) == 8
    dec = O.QualCoder(False, O.QM["fix2"], 0, 1)
    dec.set_input(p)
    assert list(dec.decode(np.array([0, 1, 2, 3], np.uint8))) == [33 + 1, 33 + 13, 33 + 13, 33 + 1]
