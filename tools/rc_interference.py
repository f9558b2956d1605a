#!/usr/bin/env python3
"""What kind of neighbour slows the interval coder (k_range_code, csrc/rc_dev.hpp)?  The quality coder runs at the headline's shape
(1 G symbols, parts of 64 Ki) on its own stream while torch's stream is kept busy with ONE kind of background work:
  none | copy (streaming 16-GB copies) | gather (random 8-byte reads) | scatter (random 8-byte writes) | gemm (fp32 matmul: MFMA + LDS) | alu (sin/cos chains on an L2-sized array)
Prints k_range_code's time per launch under each.  Usage: tools/rc_interference.py [bases] [part_symbols]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from colord_amd.device import Context
n_bases = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_000_000_000
part = int(float(sys.argv[2])) if len(sys.argv) > 2 else 65536
rlen = 20_000
n_reads = n_bases // rlen
ctx = Context(0, timing=True)
dev = ctx.device
g = torch.Generator(device=dev); g.manual_seed(7)
lvl = torch.randint(5, 30, (n_reads * (rlen // 100),), device=dev, generator=g).repeat_interleave(100)
q = (lvl + torch.randint(-4, 5, (n_reads * rlen,), device=dev, generator=g)).clamp(1, 40).to(torch.uint8) + 33
codes = torch.randint(0, 4, (n_reads * rlen,), device=dev, generator=g, dtype=torch.uint8)
off = torch.arange(n_reads + 1, device=dev, dtype=torch.int64) * rlen
reads = ctx.pack_reads(codes, off)
per = max(1, part // (rlen + 1))
pb = np.unique(np.concatenate([np.arange(0, n_reads, per), [n_reads]])).astype(np.uint32)
qc = ctx.qual_coder(mode=2, source=0, level=1, fwd=(7, 14, 26))
del lvl, codes
N = 1 << 31                                                                  # 16 GB of int64
big_a = torch.empty(N, dtype=torch.int64, device=dev); big_b = torch.empty(N, dtype=torch.int64, device=dev)
idx = torch.randint(0, N, (1 << 28,), device=dev, generator=g)
vals = torch.empty(1 << 28, dtype=torch.int64, device=dev)
ma = torch.randn(8192, 8192, device=dev); mb = torch.randn(8192, 8192, device=dev); mc = torch.empty(8192, 8192, device=dev)
small = torch.randn(1 << 22, device=dev)
def bg(kind):
    if kind == "copy": big_b.copy_(big_a)
    elif kind == "gather": torch.index_select(big_a, 0, idx, out=vals)
    elif kind == "scatter": big_b.index_copy_(0, idx, vals)
    elif kind == "gemm": torch.matmul(ma, mb, out=mc)
    elif kind == "alu":
        for _ in range(8): small.sin_().cos_()
side = torch.cuda.Stream(device=dev)
qc.encode(reads, q, off, pb); torch.cuda.synchronize(); ctx.acc.clear()
for kind in ["none", "copy", "gather", "scatter", "gemm", "alu", "none"]:
    # how long does one background op take by itself
    t_one = 0.0
    if kind != "none":
        with torch.cuda.stream(side):
            bg(kind); side.synchronize(); t0 = time.perf_counter(); bg(kind); side.synchronize(); t_one = time.perf_counter() - t0
    n_bg = 0 if kind == "none" else max(4, int(2.5 / max(t_one, 1e-4)))       # about 2.5 s of it queued behind the coder's back
    with torch.cuda.stream(side):
        for _ in range(n_bg): bg(kind)
    t0 = time.perf_counter()
    payload, sizes = qc.encode(reads, q, off, pb)
    t_enc = time.perf_counter() - t0
    left = not side.query()
    torch.cuda.synchronize()
    ms, k, byt = ctx.acc.pop("k_range_code", (0.0, 0, 0.0))[:3]
    other = sorted(((v[0], n) for n, v in ctx.acc.items()), reverse=True)[:3]
    ctx.acc.clear()
    print(f"{kind:8s}: k_range_code {ms / max(k, 1):7.1f} ms/launch ({k} launches); encode call {t_enc * 1e3:7.1f} ms; background op {t_one * 1e3:6.1f} ms x {n_bg}, still running at the end: {left}; "
          f"next: {', '.join(f'{n} {m:.0f}' for m, n in other)}", flush=True)
