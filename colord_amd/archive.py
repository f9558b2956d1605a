"""CoLoRd archive container (host I/O; format of src/colord/archive.cpp:92-114,170-236,268-283).

part  = varint(metadata) + payload;  varint(x) = 1 byte n (number of significant bytes) + n bytes BE.
file  = parts ..., footer, u64-LE footer size.  footer = varint(n_streams), per stream: name\\0,
        varint(n_parts), varint(raw_size), per part varint(offset), varint(size).
"""
from __future__ import annotations
import struct
from dataclasses import dataclass, field


def _varint(x: int) -> bytes:
    n = (x.bit_length() + 7) // 8
    return bytes([n]) + x.to_bytes(n, "big")


class _Cur:
    def __init__(self, buf, pos=0):
        self.buf, self.pos = buf, pos

    def varint(self):
        n = self.buf[self.pos]
        v = int.from_bytes(self.buf[self.pos + 1:self.pos + 1 + n], "big")
        self.pos += 1 + n
        return v

    def cstr(self):
        e = self.buf.index(b"\0", self.pos)
        s = self.buf[self.pos:e]
        self.pos = e + 1
        return s.decode()


@dataclass
class Stream:
    name: str
    raw_size: int = 0
    parts: list = field(default_factory=list)      # (metadata, payload bytes)


def read_archive(path: str) -> dict:
    buf = open(path, "rb").read()
    (fsz,) = struct.unpack("<Q", buf[-8:])
    cur = _Cur(buf, len(buf) - 8 - fsz)
    out = {}
    for _ in range(cur.varint()):
        st = Stream(cur.cstr())
        n_parts = cur.varint()
        st.raw_size = cur.varint()
        locs = [(cur.varint(), cur.varint()) for _ in range(n_parts)]
        for off, size in locs:
            pc = _Cur(buf, off)
            meta = pc.varint()
            st.parts.append((meta, bytes(buf[pc.pos:pc.pos + size])))
        out[st.name] = st
    return out


def write_archive(path: str, streams: list) -> None:
    """streams: list of Stream; parts are written stream by stream (the reader finds them by name)."""
    with open(path, "wb") as f:
        off = 0
        locs = []
        for st in streams:
            l = []
            for meta, payload in st.parts:
                l.append((off, len(payload)))
                v = _varint(meta)
                f.write(v)
                f.write(payload)
                off += len(v) + len(payload)
            locs.append(l)
        foot = bytearray(_varint(len(streams)))
        for st, l in zip(streams, locs):
            foot += st.name.encode() + b"\0" + _varint(len(l)) + _varint(st.raw_size)
            for o, s in l:
                foot += _varint(o) + _varint(s)
        f.write(foot)
        f.write(struct.pack("<Q", len(foot)))
