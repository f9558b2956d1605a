"""Host-side FASTQ/FASTA reading with the reference reader's semantics (plumbing, not the hot path).

Follows src/colord/in_reads.cpp:24-42 (ACGTN -> 0..4, anything else is an error), :62-77 / :104-112
(a pack closes once sum(len+1) >= 4 Mi, the +1 being the reference's guard byte) and :114-226
(multi-line FASTA, CRLF tolerance).
"""
from __future__ import annotations
import gzip
from dataclasses import dataclass, field
import numpy as np

READS_PACK_SIZE = 2 << 21          # defs.h:45

_CODE = np.full(256, 255, dtype=np.uint8)
for _c, _v in zip(b"ACGTN", range(5)):
    _CODE[_c] = _v                       # upper case only: the lower-case entries of SymbToBinMap are commented out (utils.h:472-475)


@dataclass
class ReadSet:
    """Reads in file order: concatenated base codes (1 B/base, 0..4) + offsets, qualities, headers."""
    bases: np.ndarray                      # uint8 codes
    offsets: np.ndarray                    # int64, n_reads+1
    quals: np.ndarray | None               # uint8 ASCII, same offsets (None for FASTA)
    headers: list = field(default_factory=list)   # bytes, without '@'/'>'
    plus_eq: list = field(default_factory=list)   # bool: '+' line repeats the header
    is_fastq: bool = True

    @property
    def n_reads(self) -> int:
        return len(self.offsets) - 1

    def read(self, i: int) -> np.ndarray:
        return self.bases[self.offsets[i]:self.offsets[i + 1]]

    def qual(self, i: int) -> np.ndarray:
        return self.quals[self.offsets[i]:self.offsets[i + 1]]

    def has_n(self) -> np.ndarray:
        isn = (self.bases == 4).astype(np.int64)
        cs = np.concatenate([[0], np.cumsum(isn)])
        return (cs[self.offsets[1:]] - cs[self.offsets[:-1]]) > 0

    def pack_bounds(self) -> np.ndarray:
        """Read index boundaries of the reference's packs (in_reads.cpp:62-77)."""
        lens = np.diff(self.offsets) + 1
        bounds = [0]
        acc = 0
        for i, l in enumerate(lens):
            acc += int(l)
            if acc >= READS_PACK_SIZE:
                bounds.append(i + 1)
                acc = 0
        if bounds[-1] != self.n_reads:
            bounds.append(self.n_reads)
        return np.asarray(bounds, dtype=np.int64)


def _open(path):
    with open(path, "rb") as f:
        magic = f.read(2)
    return gzip.open(path, "rb") if magic == b"\x1f\x8b" else open(path, "rb")


def read_fastx(path: str) -> ReadSet:
    with _open(path) as f:
        data = f.read()
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    lines = [l[:-1] if l.endswith(b"\r") else l for l in lines]
    if not lines:
        return ReadSet(np.zeros(0, np.uint8), np.zeros(1, np.int64), np.zeros(0, np.uint8))
    is_fastq = lines[0][:1] == b"@"
    seqs, quals, headers, plus_eq = [], [], [], []
    if is_fastq:
        if len(lines) % 4:
            raise ValueError("truncated FASTQ")
        for i in range(0, len(lines), 4):
            h, s, p, q = lines[i:i + 4]
            headers.append(h[1:])
            if len(p) > 1 and p[1:] != h[1:]:
                raise ValueError("quality header not empty but different than read header")
            plus_eq.append(len(p) > 1)
            seqs.append(s)
            quals.append(q)
    else:
        cur = None
        for l in lines:
            if l[:1] == b">":
                if cur is not None:
                    seqs.append(b"".join(cur))
                headers.append(l[1:])
                plus_eq.append(False)
                cur = []
            else:
                cur.append(l)
        if cur is not None:
            seqs.append(b"".join(cur))
    lens = np.fromiter((len(s) for s in seqs), dtype=np.int64, count=len(seqs))
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    bases = _CODE[np.frombuffer(b"".join(seqs), dtype=np.uint8)]
    if (bases == 255).any():
        raise ValueError("Only ACGTN symbols supported inside a read")
    q = np.frombuffer(b"".join(quals), dtype=np.uint8).copy() if is_fastq else None
    if is_fastq and len(q) != len(bases):
        raise ValueError("quality length differs from read length")
    return ReadSet(bases, offsets, q, headers, plus_eq, is_fastq)


def read_genome(path: str):
    """Multi-FASTA (plain or gzip) of a reference genome -> (codes uint8 0..3 back to back, offsets uint64 n + 1): the reader of
    CReferenceGenome (reference_genome.cpp:106-196, reference_genome.h:50-55; csrc/cli/genome_io.hpp is the C++ form): header
    lines start a new sequence, only A C G T of either case are kept.  A line starting with '>' right after a header line
    belongs to the sequence (the reference's state machine looks for headers only after sequence lines)."""
    with _open(path) as f:
        data = np.frombuffer(f.read(), dtype=np.uint8)
    if not data.size:
        raise ValueError(f"file {path} is empty")
    if data[0] != ord(">"):
        raise ValueError("wrong reference genome file format, multi fasta expected")
    eol = (data == 10) | (data == 13)
    first_of_line = np.empty(data.size, dtype=bool)
    first_of_line[0] = True
    first_of_line[1:] = eol[:-1] & ~eol[1:]
    cand = np.flatnonzero(first_of_line & (data == ord(">")))
    eol_pos = np.flatnonzero(eol)
    lut = np.full(256, 4, dtype=np.uint8)
    for i, ch in enumerate("ACGT"):
        lut[ord(ch)] = lut[ord(ch.lower())] = i
    codes = lut[data]
    keep = codes < 4
    starts = []
    prev_end = -1                                   # end (first EOL) of the last accepted header line
    for h in cand.tolist():
        if prev_end >= 0 and not (~eol[prev_end:h]).any():
            continue                                # only line ends since the last header: this line is sequence data
        e = int(eol_pos[np.searchsorted(eol_pos, h)]) if eol_pos.size and eol_pos[-1] > h else data.size
        keep[h:e] = False
        starts.append(h)
        prev_end = e
    kept_before = np.concatenate([[0], np.cumsum(keep, dtype=np.int64)])
    off = np.array([int(kept_before[h]) for h in starts] + [int(kept_before[-1])], dtype=np.uint64)
    return np.ascontiguousarray(codes[keep]), off


def genome_pseudo_reads(codes: np.ndarray, off: np.ndarray, read_len: int, overlap: int):
    """Overlapping pieces of the genome's sequences (reference_genome.cpp:391-419): (codes, offsets) of the pseudo reads."""
    if read_len <= overlap:
        raise ValueError("reference genome: pseudo-read length does not exceed the overlap")
    pieces, lens = [], []
    for s in range(len(off) - 1):
        b, n = int(off[s]), int(off[s + 1] - off[s])
        for start in range(0, n, read_len - overlap):
            e = min(start + read_len, n)
            pieces.append(codes[b + start:b + e]); lens.append(e - start)
    out = np.concatenate(pieces) if pieces else np.zeros(0, np.uint8)
    return out, np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)


def _record_start(f, pos: int, size: int) -> int:
    """Offset of the first FASTQ record that starts at or after byte `pos` of a plain 4-line FASTQ: a line that starts with '@'
    and whose next-but-one line starts with '+' (a quality line may start with '@' too; it is followed by a header, not by a
    '+' line two lines on... unless that quality line starts with '+' as well, which the check of the line after rules out)."""
    if pos <= 0:
        return 0
    if pos >= size:
        return size
    # the window grows until it holds the first line end and twelve line starts behind it (or the file ends): records of ultra-long
    # reads are megabytes — a window without a line end must not make a mid-line byte a line start, and one with fewer than four
    # lines must not end the share at the end of the file
    want = 1 << 22
    while True:
        f.seek(pos - 1)
        buf = f.read(want)
        eof = pos - 1 + len(buf) >= size
        first = buf.find(b"\n")
        starts = []
        if first >= 0:
            i = first + 1                          # first line start at or after pos
            while i < len(buf) and len(starts) < 12:
                starts.append(i)
                j = buf.find(b"\n", i)
                if j < 0:
                    break
                i = j + 1
        if len(starts) >= 12 or eof:
            break
        want *= 4
    for a in range(len(starts) - 3):
        l0, l2 = buf[starts[a]:starts[a] + 1], buf[starts[a + 2]:starts[a + 2] + 1]
        l1 = buf[starts[a + 1]:starts[a + 1] + 1]
        if l0 == b"@" and l2 == b"+" and l1 not in (b"@", b"+", b""):
            # the sequence line's length must be the quality line's (rules out a quality line read as a header)
            seq_len = starts[a + 2] - starts[a + 1]
            q_end = buf.find(b"\n", starts[a + 3])
            if q_end < 0 or (q_end + 1 - starts[a + 3]) == seq_len:
                return pos - 1 + starts[a]
    return size


def read_fastx_range(path: str, rank: int, world: int):
    """Rank `rank`'s share of a PLAIN (not gzipped) 4-line FASTQ: the records that start in its byte range — no rank reads, parses
    or holds the whole file.  Returns (ReadSet, file size) or None when the file is not of that kind (gzip, FASTA, multi-line
    records): then the caller reads everything and keeps its share."""
    import os
    with open(path, "rb") as f:
        magic = f.read(2)
        if magic[:2] == b"\x1f\x8b" or magic[:1] != b"@":
            return None
        size = os.path.getsize(path)
        a = _record_start(f, size * rank // world, size)
        b = _record_start(f, size * (rank + 1) // world, size) if rank + 1 < world else size
        f.seek(a)
        data = f.read(b - a)
    lines = data.split(b"\n")
    if lines and lines[-1] == b"":
        lines.pop()
    lines = [l[:-1] if l.endswith(b"\r") else l for l in lines]
    if len(lines) % 4:
        return None                                   # (multi-line records or a cut inside a record: not this reader's kind)
    seqs, quals, headers, plus_eq = [], [], [], []
    for i in range(0, len(lines), 4):
        h, s_, p_, q = lines[i:i + 4]
        if h[:1] != b"@" or p_[:1] != b"+" or len(s_) != len(q):
            return None
        if len(p_) > 1 and p_[1:] != h[1:]:
            raise ValueError("quality header not empty but different than read header")
        headers.append(h[1:]); plus_eq.append(len(p_) > 1); seqs.append(s_); quals.append(q)
    lens = np.fromiter((len(x) for x in seqs), dtype=np.int64, count=len(seqs))
    offsets = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    bases = _CODE[np.frombuffer(b"".join(seqs), dtype=np.uint8)] if seqs else np.zeros(0, np.uint8)
    if (bases == 255).any():
        raise ValueError("Only ACGTN symbols supported inside a read")
    qv = np.frombuffer(b"".join(quals), dtype=np.uint8).copy() if seqs else np.zeros(0, np.uint8)
    return ReadSet(bases, offsets, qv, headers, plus_eq, True), size


def write_fastq(path: str, rs: ReadSet, quals: np.ndarray | None = None) -> None:
    lut = np.frombuffer(b"ACGTN", dtype=np.uint8)
    q = rs.quals if quals is None else quals
    with open(path, "wb") as f:
        for i in range(rs.n_reads):
            a, b = rs.offsets[i], rs.offsets[i + 1]
            f.write(b"@" + rs.headers[i] + b"\n")
            f.write(lut[rs.bases[a:b]].tobytes())
            f.write(b"\n+" + (rs.headers[i] if rs.plus_eq[i] else b"") + b"\n")
            f.write(q[a:b].tobytes())
            f.write(b"\n")
