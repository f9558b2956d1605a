"""Debugging aid: run the host build of the GPU encoder logic (tests/tools/encode_host.hip) on a golden config and
report the reads whose tuple stream differs from the reference's.  Usage: python tests/tools/encode_host_check.py CFG"""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from oracle import pyoracle as O
from util import golden

PRESET_BY_LEVEL = {1: (64, 3), 2: (48, 5), 3: (48, 6)}


def pack(reads_list):
    lens = np.array([len(r) for r in reads_list], np.uint32)
    words = (lens.astype(np.int64) + 31) // 32
    woff = np.concatenate([[0], np.cumsum(words)]).astype(np.uint64)
    packed = np.zeros(int(woff[-1]) + 1, np.uint64)
    inv = np.zeros(int(woff[-1]) + 1, np.uint32)
    for i, r in enumerate(reads_list):
        r = np.asarray(r, np.uint64)
        n = len(r)
        pad = np.zeros(int(words[i]) * 32, np.uint64)
        pad[:n] = r & 3
        isn = np.zeros(int(words[i]) * 32, np.uint32)
        isn[:n] = (r > 3)
        isn[n:] = 1
        sh = (62 - 2 * (np.arange(32, dtype=np.uint64)))
        packed[int(woff[i]):int(woff[i]) + int(words[i])] = (pad.reshape(-1, 32) << sh).sum(axis=1, dtype=np.uint64)
        ish = (31 - np.arange(32, dtype=np.uint32))
        inv[int(woff[i]):int(woff[i]) + int(words[i])] = (isn.reshape(-1, 32) << ish).sum(axis=1, dtype=np.uint32)
    return packed, woff, lens, inv


def main(cfg, force_large=0):
    g = golden(cfg)
    rs = g.reads
    c = g.p("c")
    has_n = rs.has_n()
    accept = g.accept.astype(bool) & ~has_n
    all_reads = [rs.read(i) for i in range(rs.n_reads)]
    rp, rw, rl, rinv = pack(all_reads)
    fp, fw, fl, _ = pack([all_reads[i] for i in range(rs.n_reads) if accept[i]])
    enc = O.Encoder(g.p("a"), g.p("k"), g.p("f"), g.p("source"))
    for i in range(rs.n_reads):
        if accept[i]:
            enc.add_ref(all_reads[i])
    n_c = np.zeros(rs.n_reads, np.uint32)
    cand = np.zeros((rs.n_reads, c, 4), np.uint32)
    coff = np.zeros(rs.n_reads * c + 1, np.uint64)
    data = []
    for i in range(rs.n_reads):
        exp = [] if has_n[i] else enc.candidates(all_reads[i], g.cands[i]["refs"])
        n_c[i] = len(exp)
        for j in range(c):
            coff[i * c + j] = len(data)
            if j < len(exp):
                rid, rev, tot, anchors = exp[j]
                cand[i, j] = (rid, rev, tot, len(anchors))
                data.extend(anchors)
    coff[-1] = len(data)
    data = np.array(data, np.uint32).reshape(-1, 3) if data else np.zeros((1, 3), np.uint32)
    lib = C.CDLL("/tmp/libenc_host.so")
    min_alt, max_rec = PRESET_BY_LEVEL[g.p("level")]
    pb = np.asarray(rs.pack_bounds(), np.uint32)
    cap = int(rl.sum()) + 64 * rs.n_reads
    out = np.zeros(cap, np.uint8); off = np.zeros(rs.n_reads + 1, np.uint64); nt = np.zeros(rs.n_reads, np.uint32); why = np.zeros(32, np.uint32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    hn = has_n.astype(np.uint8)
    lib.dbg_encode.restype = C.c_int
    rc = lib.dbg_encode(P(rp), P(rw), P(rl), P(rinv), P(hn), C.c_uint32(rs.n_reads), P(fp), P(fw), P(fl), P(n_c), P(cand), P(coff), P(data),
                        C.c_uint32(c), C.c_uint32(g.p("a")), C.c_uint32(min_alt), C.c_uint32(max_rec), C.c_double(1.0), P(pb), C.c_uint32(len(pb) - 1),
                        C.c_uint64(1024 << 20), C.c_uint32(force_large), P(out), C.c_uint64(cap), P(off), P(nt), P(why))
    print("rc", rc, "gaps per class [trivial, nb1..4, large]", list(why[:6]), "frames per level", [int(x) for x in why[8:18] if x])
    bad = []
    for i in range(rs.n_reads):
        got = out[int(off[i]):int(off[i + 1])].tobytes()
        if nt[i] != g.es[i][1] or got != g.es[i][2]:
            bad.append(i)
    print(cfg, "reads", rs.n_reads, "differ", len(bad), bad[:20])
    for i in bad[:3]:
        got = out[int(off[i]):int(off[i + 1])].tobytes(); exp = g.es[i][2]
        k = next((j for j in range(min(len(got), len(exp))) if got[j] != exp[j]), min(len(got), len(exp)))
        print(" read", i, "len", rl[i], "got", len(got), "exp", len(exp), "first diff at byte", k, got[max(0, k - 8):k + 8].hex(), exp[max(0, k - 8):k + 8].hex())
    return len(bad)


if __name__ == "__main__":
    sys.exit(1 if main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 0) else 0)
