"""Device-side synthetic ONT-like read generator (torch ops; benchmark/test INPUT only, not the hot path).

Same statistical recipe as colord_amd.synth: random genome, lognormal read lengths with
N50 ~ 20 kb, random strand, 2 % deletions, 3 % substitutions, 2 % insertions, qualities i.i.d. from
'%+5C' with p = (.1,.2,.4,.3).  Generated directly in HBM so the benchmark's timed region never sees
PCIe or a parser.  (Not bit-identical to the numpy generator: different RNG streams.)
"""
from __future__ import annotations
import math
import torch


def make_reads_device(device, seed: int, genome_len: int, target_bases: int, mean_scale: float = 20000.0,
                      sigma: float = 0.7, chunk_bases: int = 64_000_000, with_quals: bool = False, read_seed: int | None = None):
    g = torch.Generator(device=device).manual_seed(seed)
    genome = torch.randint(0, 4, (genome_len,), generator=g, device=device, dtype=torch.uint8)
    if read_seed is not None:                  # same genome on every rank, different reads
        g = torch.Generator(device=device).manual_seed(read_seed)
    mu = math.log(mean_scale) - sigma * sigma
    mean_len = math.exp(mu + sigma * sigma / 2)
    out_codes, out_lens, out_quals = [], [], []
    total = 0
    while total < target_bases:
        want = min(chunk_bases, target_bases - total)
        n = max(1, int(want / (mean_len * 0.98)) + 1)
        ln = torch.exp(torch.randn(n, generator=g, device=device) * sigma + mu).clamp_(200, min(200000, genome_len - 1)).long()
        start = (torch.rand(n, generator=g, device=device, dtype=torch.float64) * (genome_len - ln).double()).long()
        rev = torch.rand(n, generator=g, device=device) < 0.5
        src_off = torch.cumsum(ln, 0) - ln
        tot_src = int(ln.sum().item())
        rid = torch.repeat_interleave(torch.arange(n, device=device), ln)
        local = torch.arange(tot_src, device=device) - src_off[rid]
        gpos = torch.where(rev[rid], start[rid] + ln[rid] - 1 - local, start[rid] + local)
        b = genome[gpos]
        b = torch.where(rev[rid], 3 - b, b)
        r = torch.rand(tot_src, generator=g, device=device)
        sub = (r >= 0.02) & (r < 0.05)
        b = torch.where(sub, (b + torch.randint(1, 4, (tot_src,), generator=g, device=device, dtype=torch.uint8)) % 4, b)
        keep = r >= 0.02
        ins = keep & (torch.rand(tot_src, generator=g, device=device) < 0.02)
        emit = keep.long() + ins.long()
        opos = torch.cumsum(emit, 0) - emit
        n_out = int(emit.sum().item())
        codes = torch.empty(n_out, dtype=torch.uint8, device=device)
        codes[opos[keep]] = b[keep]
        codes[opos[ins] + 1] = torch.randint(0, 4, (int(ins.sum().item()),), generator=g, device=device, dtype=torch.uint8)
        per_read = torch.zeros(n, dtype=torch.long, device=device).index_add_(0, rid, emit)
        out_codes.append(codes)
        out_lens.append(per_read)
        if with_quals:
            u = torch.rand(n_out, generator=g, device=device)
            qidx = torch.bucketize(u, torch.tensor([0.1, 0.3, 0.7], device=device))
            out_quals.append(torch.tensor(list(b"%+5C"), dtype=torch.uint8, device=device)[qidx])
        total += n_out
        del rid, local, gpos, b, r, sub, keep, ins, emit, opos
    codes = torch.cat(out_codes)
    lens = torch.cat(out_lens)
    lens = lens[lens > 0] if bool((lens == 0).any()) else lens
    offsets = torch.cat([torch.zeros(1, dtype=torch.long, device=device), torch.cumsum(lens, 0)])
    if with_quals:
        return codes, offsets, torch.cat(out_quals)
    return codes, offsets
