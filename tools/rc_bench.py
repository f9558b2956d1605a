#!/usr/bin/env python3
"""Standalone timing of the interval coder (k_range_code, csrc/rc_dev.hpp) through the quality coder's C ABI: synthetic ONT-like
qualities, 4-avg level 1, parts of `part` symbols (default 4 Mi: the reference's cut, one dependent chain of that length per lane).
Prints the kernel's time per launch, ns per symbol of the longest chain and a digest of the payload (the bytes must not change
when the kernel does).  Usage: tools/rc_bench.py [bases] [part_symbols]"""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from colord_amd.device import Context
n_bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 300_000_000
part = int(float(sys.argv[2])) if len(sys.argv) > 2 else 4 << 20
rlen = 20_000
n_reads = n_bases // rlen
ctx = Context(0, timing=True)
dev = ctx.device
g = torch.Generator(device=dev); g.manual_seed(7)
# a slowly varying level + noise, clipped to Phred 1..40 ('!' + q)
lvl = torch.randint(5, 30, (n_reads * (rlen // 100),), device=dev, generator=g).repeat_interleave(100)
q = (lvl + torch.randint(-4, 5, (n_reads * rlen,), device=dev, generator=g)).clamp(1, 40).to(torch.uint8) + 33
codes = torch.randint(0, 4, (n_reads * rlen,), device=dev, generator=g, dtype=torch.uint8)
off = torch.arange(n_reads + 1, device=dev, dtype=torch.int64) * rlen
reads = ctx.pack_reads(codes, off)
per = max(1, part // (rlen + 1))
pb = np.unique(np.concatenate([np.arange(0, n_reads, per), [n_reads]])).astype(np.uint32)
qc = ctx.qual_coder(mode=2, source=0, level=1, fwd=(7, 14, 26))
for it in range(2):
    payload, sizes = qc.encode(reads, q, off, pb)
    torch.cuda.synchronize()
    ms, k, byt = ctx.acc.pop("k_range_code", (0.0, 0, 0.0))[:3]
    ctx.acc.clear()
    longest = int(np.diff(pb).max()) * rlen
    print(f"pass {it}: {len(pb) - 1} parts of <= {longest} symbols; k_range_code {ms:.1f} ms in {k} launches -> {ms * 1e6 / max(k, 1) / longest:.1f} ns per symbol of the chain; "
          f"payload {payload.numel()} B sha256 {hashlib.sha256(payload.cpu().numpy().tobytes()).hexdigest()[:16]}", flush=True)
