#!/usr/bin/env python3
"""Debugging aid: tuple streams of a golden configuration by the wave-per-read emission against the golden streams; prints the
tuples around the first difference of the first few differing reads.  Usage: tools/emit_diff.py <cfg>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from colord_amd.device import Context
from util import golden
from test_gpu_encode import gpu_streams
NAMES = ["ins", "del", "M", "sub", "ANCHOR", "SKIP", "ALT", "MAIN", "plain", "sP", "sES", "sPN"]
def decode(b):
    out, i = [], 0
    while i < len(b):
        t = b[i] >> 4
        if t in (4, 5): out.append((i, NAMES[t], ((b[i] & 15) << 24) | (b[i+1] << 16) | (b[i+2] << 8) | b[i+3])); i += 4
        elif t in (6, 10): out.append((i, NAMES[t], (b[i] & 15, int.from_bytes(b[i+1:i+5], "big")))); i += 5
        else: out.append((i, NAMES[t], b[i] & 15)); i += 1
    return out
def rle(t):
    out = []
    for _, n, v in t:
        if out and out[-1][0] == (n, v) and n in ("ins", "del", "M", "sub"): out[-1][1] += 1
        else: out.append([(n, v), 1])
    return [f"{n}{'' if n in ('del', 'M', 'MAIN') else v}x{c}" if c > 1 else f"{n}{'' if n in ('del', 'M', 'MAIN') else v}" for (n, v), c in out]
cfg = sys.argv[1]
g = golden(cfg); ctx = Context(0)
es, off, nt = gpu_streams(ctx, g)
shown = 0
for i in range(g.reads.n_reads):
    got = es[off[i]:off[i + 1]].tobytes(); exp = g.es[i][2]
    if got == exp and nt[i] == g.es[i][1]: continue
    k = next((j for j in range(min(len(got), len(exp))) if got[j] != exp[j]), min(len(got), len(exp)))
    print(f"read {i}: got {len(got)} B / {nt[i]} tuples, expected {len(exp)} B / {g.es[i][1]} tuples, first difference at byte {k}")
    dg, de = decode(got), decode(exp)
    jg = next((j for j, t in enumerate(dg) if t[0] >= k), len(dg)); je = next((j for j, t in enumerate(de) if t[0] >= k), len(de))
    print("   got     :", " ".join(rle(dg[max(0, jg - 12):jg + 14])))
    print("   expected:", " ".join(rle(de[max(0, je - 12):je + 14])))
    shown += 1
    if shown == 5: break
