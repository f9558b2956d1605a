"""Element indices above 2^31 and byte offsets above 2^32 inside single launches (the round-4 bug — a v_readlane of the low half of a 64-bit
offset sign-extended at 2.4 GB of coder output — was found in a bench, by luck): the in-tree radix sort on 2.2 G (u64 key, u32 value) pairs
through the C ABI, checked by properties — sorted on the key bits, stable (the values are the original positions: ascending inside every run of
equal keys), and the same multiset (wrapping sums of keys, of values and of key x value).  The chunk-sized counterpart for the walks, the
emission and the model stages is tools/big_offset_check.py (one 2.2-Gbase chunk against four: equal streams; profiles/r05_big_offset_check.txt)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_radix_sort_of_2_2_g_pairs(ctx):
    n = 2_200_000_000
    dev = ctx.device
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 110e9:
        pytest.skip("needs 110 GB of free device memory")
    g = torch.Generator(device=dev); g.manual_seed(5)
    keys = torch.empty(n, dtype=torch.int64, device=dev)
    for a in range(0, n, 1 << 28):                       # (in pieces: randint's own temporaries stay small)
        b = min(n, a + (1 << 28))
        keys[a:b] = torch.randint(0, 1 << 20, (b - a,), device=dev, generator=g, dtype=torch.int64) << 13
    vals = torch.arange(n, device=dev, dtype=torch.int64).to(torch.int32)          # positions 0 .. n - 1 (wrap above 2^31: compared as unsigned below)
    def sums(k, v):
        s1 = s2 = s3 = 0
        for a in range(0, n, 1 << 28):
            kk, vv = k[a:a + (1 << 28)], v[a:a + (1 << 28)].to(torch.int64) & 0xffffffff
            s1 += int(kk.sum().item()); s2 += int(vv.sum().item()); s3 = (s3 + int((kk * vv).sum().item())) & ((1 << 64) - 1)
        return s1, s2, s3
    before = sums(keys, vals)
    ctx.sort_u64(keys, vals, 13, 33)
    torch.cuda.synchronize()
    assert sums(keys, vals) == before
    for a in range(0, n - 1, 1 << 28):
        b = min(n - 1, a + (1 << 28))
        k0, k1 = keys[a:b], keys[a + 1:b + 1]
        assert bool((k1 >= k0).all()), f"not sorted in [{a}, {b})"
        v0, v1 = vals[a:b].to(torch.int64) & 0xffffffff, vals[a + 1:b + 1].to(torch.int64) & 0xffffffff
        assert bool(((k1 > k0) | (v1 > v0)).all()), f"not stable in [{a}, {b})"
