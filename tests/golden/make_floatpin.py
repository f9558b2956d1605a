#!/usr/bin/env python3
"""Generates tests/golden/floatpin.json: SHA-256 digests of the HOST libm values of the encoder's decision logarithm
-log2(count * (1/total)) (calc_logs, /root/reference/src/colord/utils.h:800-810) over
  (a) every pair 1 <= count <= total <= 4096, in blocks of 64 totals, and
  (b) 10^7 sampled pairs with total <= 2^20 (the estimator's rescale bound, utils.h:779-781), in blocks of 10^6.
The values come from oracle/floatpin.c (gcc + this container's glibc — the libm the reference binary oracle/_ref/colord is
linked against).  Run in the build container:  python tests/golden/make_floatpin.py
The GPU test recomputes the pairs (floatpin_pairs below is deterministic integer arithmetic), evaluates them with
cl_estimator_logs on the device and compares digests; on a mismatch it reports the differing pairs."""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

DENSE_MAX, DENSE_BLOCK = 4096, 64
N_SAMPLED, SAMPLED_BLOCK, SAMPLED_MAX = 10_000_000, 1_000_000, 1 << 20


def dense_block(b: int):
    """All (count, total) with total in [64 b + 1, 64 b + 64], count in [1, total]."""
    tot = np.arange(DENSE_BLOCK * b + 1, DENSE_BLOCK * (b + 1) + 1, dtype=np.uint32)
    total = np.repeat(tot, tot)
    count = (np.arange(len(total), dtype=np.int64) - np.repeat(np.cumsum(tot.astype(np.int64)) - tot, tot) + 1).astype(np.uint32)
    return count, total


def _splitmix(x: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def sampled_block(b: int):
    """Block b of the sampled pairs: total in [1, 2^20] (half of them log-uniform), count in [1, total]; counter-based."""
    i = np.arange(b * SAMPLED_BLOCK, (b + 1) * SAMPLED_BLOCK, dtype=np.uint64)
    h1, h2 = _splitmix(i * np.uint64(2)), _splitmix(i * np.uint64(2) + np.uint64(1))
    uni = (h1 % np.uint64(SAMPLED_MAX)) + np.uint64(1)
    bits = (h1 >> np.uint64(40)) % np.uint64(21)                       # log-uniform: a magnitude, then the low bits
    logu = np.minimum(((np.uint64(1) << bits) | (h1 & ((np.uint64(1) << bits) - np.uint64(1)))), np.uint64(SAMPLED_MAX))
    total = np.where((h1 >> np.uint64(63)) == 1, uni, logu).astype(np.uint32)
    count = ((h2 % total.astype(np.uint64)) + np.uint64(1)).astype(np.uint32)
    return count, total


def digest(values: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(values, dtype=np.float64).tobytes()).hexdigest()


def main():
    from oracle import pyoracle as O
    out = {"what": "sha256 of float64 -log2(count*(1/total)), host glibc via oracle/floatpin.c", "dense": [], "sampled": []}
    for b in range(DENSE_MAX // DENSE_BLOCK):
        c, t = dense_block(b)
        out["dense"].append(digest(O.estimator_logs(c, t)))
    for b in range(N_SAMPLED // SAMPLED_BLOCK):
        c, t = sampled_block(b)
        assert c.min() >= 1 and (c <= t).all() and t.max() <= SAMPLED_MAX
        out["sampled"].append(digest(O.estimator_logs(c, t)))
    import platform
    out["libc"] = " ".join(platform.libc_ver())
    json.dump(out, open(os.path.join(HERE, "floatpin.json"), "w"), indent=0)
    print("wrote floatpin.json:", len(out["dense"]), "dense blocks,", len(out["sampled"]), "sampled blocks, libc", out["libc"])


if __name__ == "__main__":
    main()
