#!/usr/bin/env python3
"""Times the library's radix sort alone: n u64 keys of `bits` random bits with a u32 payload, wall time of `reps` sorts (the call returns when the sort is through).
Usage: tools/sort_time.py [n] [bits] [reps]   (COLORD_HIP_LIBRARY picks another build of the library)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from colord_amd.device import Context
n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 1_100_000_000
bits = int(sys.argv[2]) if len(sys.argv) > 2 else 32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = Context(0)
g = torch.Generator(device=ctx.device).manual_seed(1)
src = torch.randint(0, 1 << bits, (n,), generator=g, dtype=torch.int64, device=ctx.device)
vals = torch.arange(n, dtype=torch.int32, device=ctx.device)
ts = []
for r in range(reps + 1):
    k = src.clone(); v = vals.clone(); torch.cuda.synchronize()
    t0 = time.time()
    ctx.sort_u64(k, v, 0, bits)                      # (returns when the sort is through: sort.hip waits for its stream)
    ts.append((time.time() - t0) * 1e3)
print(f"n = {n}, {bits} bits, with payload: {min(ts[1:]):.1f} ms best, {sorted(ts[1:])[len(ts[1:]) // 2]:.1f} ms median of {reps} ({os.environ.get('COLORD_HIP_LIBRARY', 'in-tree library')})")
assert bool((k[1:] >= k[:-1]).all())
