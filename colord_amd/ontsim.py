"""Synthetic ONT reads of SURVEY.md section 8d in two BIT-IDENTICAL forms (csrc/synth/ontsim_core.h: every base and quality
is a hash of (seed, read, position), no RNG state):
  * host   — FASTQ text for the reference CPU path (`write_fastq`) or arrays (`host_reads`);
  * device — base codes + quality bytes generated directly in HBM (`device_reads`), chunk by chunk.
The read table (start, source length, strand per read) is drawn once on the host with numpy and shared by both forms.
Benchmark / test INPUT tooling: nothing here is on the compress path.
"""
from __future__ import annotations
import ctypes as C
import math
import os
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def _lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libontsim.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} is missing: run `make -C colord_amd/csrc`")
        L = C.CDLL(path)
        P, U64, U32 = C.c_void_p, C.c_uint64, C.c_uint32
        L.os_dev_lengths.argtypes = [U64, U64, P, P, P, U64, U32, P, P]
        L.os_dev_fill.argtypes = [U64, U64, P, P, P, U64, U32, P, P, P, P]
        L.os_host_reads.argtypes = [U64, U64, P, P, P, U64, U32, P, P, U64, P]
        L.os_host_fastq.argtypes = [C.c_char_p, C.c_int, U64, U64, P, P, P, U64, U32, C.POINTER(U64)]
        for f in (L.os_dev_lengths, L.os_dev_fill, L.os_host_reads, L.os_host_fastq):
            f.restype = C.c_int
        _LIB = L
    return _LIB


class ReadTable:
    """start / len_src / strand of every read; lengths clip(lognormal(ln 20000 - sigma^2, sigma = 0.7), 200, 200000): N50 ~ 20 kb."""

    def __init__(self, seed: int, genome_len: int, target_bases: int, mean_scale: float = 20000.0, sigma: float = 0.7, max_len: int = 200000):
        rng = np.random.default_rng(seed)
        mu = math.log(mean_scale) - sigma * sigma
        mean_len = math.exp(mu + sigma * sigma / 2)
        max_len = min(max_len, genome_len - 1)
        lens = np.empty(0, np.int64)
        while lens.sum() < target_bases:
            n = int((target_bases - lens.sum()) / mean_len * 1.02) + 16
            lens = np.concatenate([lens, np.clip(rng.lognormal(mu, sigma, n), 200, max_len).astype(np.int64)])
        n = int(np.searchsorted(np.cumsum(lens), target_bases)) + 1
        self.len_src = np.ascontiguousarray(lens[:n].astype(np.uint32))
        self.start = np.ascontiguousarray((rng.random(n) * (genome_len - self.len_src.astype(np.float64))).astype(np.uint64))
        self.strand = np.ascontiguousarray((rng.random(n) < 0.5).astype(np.uint8))
        self.seed, self.gseed, self.genome_len = int(seed), int(seed) * 0x9e3779b97f4a7c15 % (1 << 64) ^ 0xabcdef, int(genome_len)

    @property
    def n_reads(self):
        return len(self.len_src)

    def cuts(self, bases_per_chunk: float):
        """Read ranges of about bases_per_chunk source bases each."""
        acc = np.cumsum(self.len_src.astype(np.int64))
        out, base = [0], 0
        while out[-1] < self.n_reads:
            i = int(np.searchsorted(acc, base + bases_per_chunk, side="left")) + 1
            i = min(max(i, out[-1] + 1), self.n_reads)
            out.append(i)
            base = int(acc[i - 1])
        return out


def host_reads(t: ReadTable, r0: int = 0, r1: int | None = None, with_quals: bool = True):
    """(codes uint8 0..3, offsets int64 [n+1], quals uint8 or None) of reads r0..r1, generated on the host."""
    r1 = t.n_reads if r1 is None else r1
    n = r1 - r0
    cap = int(t.len_src[r0:r1].astype(np.int64).sum()) * 2 + 64
    codes = np.empty(cap, np.uint8)
    quals = np.empty(cap, np.uint8) if with_quals else None
    off = np.zeros(n + 1, np.uint64)
    rc = _lib().os_host_reads(t.seed, t.gseed, t.start[r0:].ctypes.data, t.len_src[r0:].ctypes.data, t.strand[r0:].ctypes.data, r0, n,
                              codes.ctypes.data, quals.ctypes.data if with_quals else None, cap, off.ctypes.data)
    if rc:
        raise RuntimeError("os_host_reads failed")
    tot = int(off[-1])
    return codes[:tot].copy(), off.astype(np.int64), (quals[:tot].copy() if with_quals else None)


def write_fastq(t: ReadTable, path: str, r0: int = 0, r1: int | None = None) -> int:
    """FASTQ of reads r0..r1 (headers `@read_<n> ch=<n%512> start_time=2020-01-01T00:00:<n%60>Z`); returns the number of bases."""
    r1 = t.n_reads if r1 is None else r1
    nb = C.c_uint64(0)
    rc = _lib().os_host_fastq(path.encode(), 0, t.seed, t.gseed, t.start[r0:].ctypes.data, t.len_src[r0:].ctypes.data, t.strand[r0:].ctypes.data, r0, r1 - r0, C.byref(nb))
    if rc:
        raise RuntimeError(f"os_host_fastq({path}) failed: {rc}")
    return int(nb.value)


def device_reads(t: ReadTable, device, r0: int = 0, r1: int | None = None, with_quals: bool = True):
    """(codes, offsets int64 [n+1], quals or None) as torch tensors on `device`, generated there."""
    import torch
    r1 = t.n_reads if r1 is None else r1
    n = r1 - r0
    L = _lib()
    d_start = torch.from_numpy(t.start[r0:r1].view(np.int64)).to(device)
    d_len = torch.from_numpy(t.len_src[r0:r1].view(np.int32)).to(device)
    d_strand = torch.from_numpy(t.strand[r0:r1]).to(device)
    out_len = torch.empty(n, dtype=torch.int32, device=device)
    stream = torch.cuda.current_stream(device).cuda_stream
    if L.os_dev_lengths(t.seed, t.gseed, d_start.data_ptr(), d_len.data_ptr(), d_strand.data_ptr(), r0, n, out_len.data_ptr(), stream):
        raise RuntimeError("os_dev_lengths failed")
    off = torch.zeros(n + 1, dtype=torch.int64, device=device)
    torch.cumsum(out_len.long(), 0, out=off[1:])
    tot = int(off[-1].item())
    codes = torch.empty(tot, dtype=torch.uint8, device=device)
    quals = torch.empty(tot, dtype=torch.uint8, device=device) if with_quals else None
    if L.os_dev_fill(t.seed, t.gseed, d_start.data_ptr(), d_len.data_ptr(), d_strand.data_ptr(), r0, n, off.data_ptr(), codes.data_ptr(),
                     quals.data_ptr() if with_quals else None, stream):
        raise RuntimeError("os_dev_fill failed")
    torch.cuda.synchronize(device)
    return codes, off, quals
