"""Seed-reproducible synthetic ONT-like reads (host generator; SURVEY.md §8d recipe).

genome: G uniform random bases; read length clip(lognormal(ln 20000 - sigma^2, sigma=0.7), 200, 200000)
(N50 ~ 20 kb), uniform start, fair strand; per-base errors 2 % deletion, 3 % substitution, 2 % insertion;
qualities i.i.d. from {Q4,Q10,Q20,Q34} = '%+5C' with p = (.1,.2,.4,.3); header
``read_<n> ch=<n%512> start_time=2020-01-01T00:00:<n%60>Z``; '+' line empty.
"""
from __future__ import annotations
import numpy as np
from .fastq import ReadSet

_QV = np.frombuffer(b"%+5C", dtype=np.uint8)
_COMP = np.array([3, 2, 1, 0], dtype=np.uint8)


def make_reads(seed: int, genome_len: int, target_bases: int, mean_scale: float = 20000.0,
               sigma: float = 0.7, n_frac: float = 0.0, max_len: int = 200000, err=(0.02, 0.03, 0.02)) -> ReadSet:
    """err: per-base (deletion, substitution, insertion) rates — the ONT-like default, or e.g. (0.001, 0.001, 0.001) for HiFi-like reads."""
    e_del, e_sub, e_ins = err
    rng = np.random.default_rng(seed)
    genome = rng.integers(0, 4, genome_len, dtype=np.uint8)
    mu = np.log(mean_scale) - sigma * sigma
    seqs, quals, headers = [], [], []
    tot = 0
    n = 0
    while tot < target_bases:
        ln = int(min(max(rng.lognormal(mu, sigma), 200), min(max_len, genome_len - 1)))
        st = int(rng.integers(0, genome_len - ln))
        s = genome[st:st + ln]
        if rng.random() < 0.5:
            s = _COMP[s[::-1]]
        r = rng.random(ln)
        sub = (r >= e_del) & (r < e_del + e_sub)
        s = s.copy()
        s[sub] = (s[sub] + rng.integers(1, 4, int(sub.sum()))) % 4
        s = s[r >= e_del]
        ins = np.nonzero(rng.random(len(s)) < e_ins)[0]
        s = np.insert(s, ins, rng.integers(0, 4, len(ins)).astype(np.uint8))
        if n_frac > 0 and rng.random() < n_frac:          # sprinkle a few N into some reads
            pos = rng.integers(0, len(s), max(1, len(s) // 2000))
            s[pos] = 4
        q = _QV[rng.choice(4, len(s), p=[0.1, 0.2, 0.4, 0.3])]
        seqs.append(s)
        quals.append(q)
        headers.append(b"read_%d ch=%d start_time=2020-01-01T00:00:%02dZ" % (n, n % 512, n % 60))
        tot += len(s)
        n += 1
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    return ReadSet(np.concatenate(seqs), offsets, np.concatenate(quals), headers, [False] * n, True)


def make_giant_gap_reads(seed: int, n_backbones: int = 6, backbone_len: int = 120_000, coverage: int = 22, n_chimeras: int = 24,
                         junk=(30_000, 60_000), err=(0.02, 0.03, 0.02)) -> ReadSet:
    """Reads that make the edit-script encoder align GIANT gaps (tens of thousands of symbols on both sides), which random sampling
    of a genome yields only a few times per Gbase: `n_backbones` regions of a random genome, each covered `coverage` times by reads of
    nearly the whole region (so their k-mers pass the count filter and the early ones become reference reads with long tails), then
    CHIMERAS — 15-25 kb of a region, a stretch of unrelated sequence as long as `junk`, and for every other one 15 kb from further
    right in the region: an inner gap (unrelated x the reference's own stretch) or a flank (unrelated x the reference's tail)."""
    rng = np.random.default_rng(seed)
    e_del, e_sub, e_ins = err
    genome = rng.integers(0, 4, n_backbones * (backbone_len + 50_000), dtype=np.uint8)

    def noisy(s):
        r = rng.random(len(s))
        s = s.copy()
        sub = (r >= e_del) & (r < e_del + e_sub)
        s[sub] = (s[sub] + rng.integers(1, 4, int(sub.sum()))) % 4
        s = s[r >= e_del]
        ins = np.nonzero(rng.random(len(s)) < e_ins)[0]
        return np.insert(s, ins, rng.integers(0, 4, len(ins)).astype(np.uint8))

    seqs = []
    for b in range(n_backbones):
        base = b * (backbone_len + 50_000)
        for c in range(coverage):
            st = base + int(rng.integers(0, 20_000)); ln = backbone_len - int(rng.integers(0, 30_000))
            s = noisy(genome[st:st + ln])
            seqs.append(_COMP[s[::-1]] if rng.random() < 0.5 else s)
    for c in range(n_chimeras):
        b = c % n_backbones
        base = b * (backbone_len + 50_000) + int(rng.integers(0, 15_000))
        l1 = int(rng.integers(15_000, 25_000)); j = int(rng.integers(junk[0], junk[1]))
        parts = [noisy(genome[base:base + l1]), rng.integers(0, 4, j, dtype=np.uint8)]
        if c % 2:
            parts.append(noisy(genome[base + l1 + j:base + l1 + j + 15_000]))
        s = np.concatenate(parts)
        seqs.append(_COMP[s[::-1]] if c % 3 == 0 else s)
    quals = [_QV[rng.choice(4, len(s), p=[0.1, 0.2, 0.4, 0.3])] for s in seqs]
    headers = [b"read_%d ch=%d start_time=2020-01-01T00:00:%02dZ" % (i, i % 512, i % 60) for i in range(len(seqs))]
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    return ReadSet(np.concatenate(seqs), offsets, np.concatenate(quals), headers, [False] * len(seqs), True)
