#!/usr/bin/env python3
"""Does a smaller batch make the quality coder's scatter cheaper?  The model stage sorts a batch by context and writes every symbol's
(cum, freq, total) to its place in stream order: one scattered 8-byte write per symbol over the whole batch's triples (8 B x symbols).
The model state is carried from batch to batch, so the same 1 G symbols can be coded as 1, 4, 16 or 64 batches (triples region
8 GB .. 128 MB, the last within the 256-MB memory-side cache) with identical bytes.  Prints wall time and the model kernels' times.
Usage: tools/qual_batch.py [bases]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from colord_amd.device import Context
n_bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
part = 65536
rlen = 20_000
n_reads = n_bases // rlen
ctx = Context(0, timing=True)
dev = ctx.device
g = torch.Generator(device=dev); g.manual_seed(7)
lvl = torch.randint(5, 30, (n_reads * (rlen // 100),), device=dev, generator=g).repeat_interleave(100)
q = (lvl + torch.randint(-4, 5, (n_reads * rlen,), device=dev, generator=g)).clamp(1, 40).to(torch.uint8) + 33
codes = torch.randint(0, 4, (n_reads * rlen,), device=dev, generator=g, dtype=torch.uint8)
del lvl
per = max(1, part // (rlen + 1))
for nb in [1, 4, 16, 64, 1]:
    qc = ctx.qual_coder(mode=2, source=0, level=1, fwd=(7, 14, 26))
    h = hashlib.sha256(); total = 0
    ctx.acc.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    rb = (n_reads + nb - 1) // nb
    rb = (rb + per - 1) // per * per                                          # batches end at part bounds
    outs = []
    for r0 in range(0, n_reads, rb):
        r1 = min(n_reads, r0 + rb)
        off = torch.arange(r1 - r0 + 1, device=dev, dtype=torch.int64) * rlen
        reads = ctx.pack_reads(codes[r0 * rlen:r1 * rlen], off)
        pb = np.unique(np.concatenate([np.arange(0, r1 - r0, per), [r1 - r0]])).astype(np.uint32)
        payload, sizes = qc.encode(reads, q[r0 * rlen:r1 * rlen], off, pb)
        outs.append((payload, sizes)); del reads
    torch.cuda.synchronize(); t = time.perf_counter() - t0
    for payload, sizes in outs: h.update(payload.cpu().numpy().tobytes()); total += payload.numel()
    top = sorted(((v[0], n, v[1]) for n, v in ctx.acc.items()), reverse=True)[:6]
    print(f"{nb:3d} batches of {rb * rlen / 1e6:7.1f} M symbols: {t * 1e3:7.1f} ms; {total} B sha {h.hexdigest()[:12]}; " + ", ".join(f"{n} {m:.0f} ms/{k}" for m, n, k in top), flush=True)
    del qc, outs
