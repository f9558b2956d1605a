"""Shared helpers for the parity tests: golden-vector loading and input reconstruction."""
from __future__ import annotations
import gzip
import json
import os
import struct
import functools
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
DATA = os.path.join(ROOT, "tests", "data")

ALL_CONFIGS = sorted(d for d in os.listdir(GOLD) if os.path.exists(os.path.join(GOLD, d, "streams.json")))
# configs without a reference genome (pseudo reads need the genome cutter)
PLAIN_CONFIGS = [c for c in ALL_CONFIGS if "genome" not in c]


class Golden:
    def __init__(self, cfg):
        self.cfg = cfg
        self.dir = os.path.join(GOLD, cfg)
        self.spec = json.load(open(os.path.join(self.dir, "streams.json")))
        self.params = {}
        for line in open(os.path.join(self.dir, "params.txt")):
            k, v = line.split()
            self.params[k] = float(v) if k == "sparse_exp" else int(v)

    def p(self, k):
        return self.params[k]

    @functools.cached_property
    def kept(self):
        a = np.fromfile(os.path.join(self.dir, "kept.bin"), dtype=np.dtype([("k", "<u8"), ("c", "<u4")]))
        o = np.argsort(a["k"], kind="stable")
        return a["k"][o].copy(), a["c"][o].copy()

    @functools.cached_property
    def accept(self):
        return np.fromfile(os.path.join(self.dir, "accept.bin"), dtype=np.uint8)

    @functools.cached_property
    def cands(self):
        b = gzip.open(os.path.join(self.dir, "cands.bin.gz"), "rb").read()
        p, out = 0, []
        while p < len(b):
            rid, has_n, ln, n = struct.unpack_from("<IBII", b, p)
            p += 13
            refs = list(struct.unpack_from("<%dI" % n, b, p))
            p += 4 * n
            com = []
            for _ in range(n):
                (m,) = struct.unpack_from("<I", b, p)
                p += 4
                com.append(np.frombuffer(b, np.uint64, m, p).copy())
                p += 8 * m
            out.append(dict(read_id=rid, has_n=bool(has_n), len=ln, refs=refs, common=com))
        return out

    @functools.cached_property
    def es(self):
        b = gzip.open(os.path.join(self.dir, "es.bin.gz"), "rb").read()
        p, out = 0, []
        while p < len(b):
            pack, nt, nb = struct.unpack_from("<III", b, p)
            p += 12
            out.append((pack, nt, b[p:p + nb]))
            p += nb
        return out

    @functools.cached_property
    def reads(self):
        from colord_amd.fastq import read_fastx
        from colord_amd.synth import make_reads
        if self.spec.get("synth"):
            return make_reads(**self.spec["synth"])
        return read_fastx(os.path.join(DATA, self.spec["input"] + ".gz"))


@functools.lru_cache(maxsize=None)
def golden(cfg) -> Golden:
    return Golden(cfg)
