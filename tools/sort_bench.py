#!/usr/bin/env python3
"""Standalone timing of the in-tree radix sort (csrc/sort.hip) through the C ABI: (u64 key, u32 value) pairs and u64 keys alone,
uniform and skewed digits.  Usage: tools/sort_bench.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from colord_amd.device import Context
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 600_000_000
ctx = Context(0)
dev = ctx.device
g = torch.Generator(device=dev); g.manual_seed(1)
def make(kind):
    if kind == "uniform":
        c = torch.randint(0, 1 << 24, (n,), device=dev, generator=g, dtype=torch.int64)
    else:   # skewed: 90 % of the keys in 2^15 contexts (a dense family), hot ones among them
        c = torch.randint(0, 1 << 15, (n,), device=dev, generator=g, dtype=torch.int64)
        c = torch.minimum(c, torch.randint(0, 1 << 15, (n,), device=dev, generator=g, dtype=torch.int64)) + (5 << 20)
    return (c << 16) | torch.randint(0, 8, (n,), device=dev, generator=g, dtype=torch.int64)
for kind in ("uniform", "skewed"):
    for with_v in (True, False):
        for bits in (24, 20, 17, 16):
            k = make(kind); v = torch.arange(n, device=dev, dtype=torch.int32) if with_v else None
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ctx.sort_u64(k, v, 16, 16 + bits)
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
            passes = (bits + 7) // 8 if (bits + 7) // 8 == (bits + 9) // 10 else (bits + 9) // 10
            byt = n * passes * ((8 + 2 * 8) + (8 if with_v else 0))
            m = (1 << bits) - 1; ok = bool((((k[1:] >> 16) & m) >= ((k[:-1] >> 16) & m)).all())
            print(f"{kind:8s} pairs={with_v!s:5s} bits={bits}: {dt * 1e3:8.1f} ms  {passes} passes  {dt * 1e3 / passes:7.1f} ms/pass  {byt / dt / 1e12:5.2f} TB/s algorithmic  sorted={ok}", flush=True)
            del k, v
