#!/usr/bin/env python3
"""Regenerates tests/golden/* from the UNMODIFIED reference (run in the build container only).

Needs oracle/_ref/ref_dump and oracle/_ref/colord (make -C oracle ref).  For every configuration it
runs the tap harness (oracle/ref_harness/ref_dump.cpp) and the plain reference binary, checks that both
archives carry identical stream payloads, and stores:
  <cfg>/params.txt  <cfg>/kept.bin  <cfg>/accept.bin  <cfg>/cands.bin.gz  <cfg>/es.bin.gz
  <cfg>/streams.json   {stream: {"parts": [[meta, size, sha256]...]}}  (payloads themselves for small ones)
Synthetic inputs are regenerated from colord_amd.synth (seeded) — only their recipe is stored.
"""
import gzip, hashlib, json, os, shutil, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from colord_amd.archive import read_archive
from colord_amd.fastq import write_fastq
from colord_amd.synth import make_reads

REF = os.path.join(ROOT, "oracle", "_ref")
DATA = os.path.join(ROOT, "tests", "data")
OUT = os.path.join(ROOT, "tests", "golden")

CONFIGS = {
    # BASELINE.json configs 1-4 (the reference's own CI cases, .github/workflows/main.yml:35-81)
    "c1_ont_default": dict(args=["compress-ont"], input="M.bovis.fastq"),
    "c2_hifi_org": dict(args=["compress-pbhifi", "-q", "org"], input="D.melanogaster.fastq"),
    "c3_clr_ratio": dict(args=["compress-pbraw", "-p", "ratio"], input="A.thaliana.fastq"),
    "c4_ont_genome": dict(args=["compress-ont", "-s"], input="M.bovis.fastq", genome="M.bovis-reference.fna"),
    "c6_ont_org": dict(args=["compress-ont", "-q", "org"], input="M.bovis.fastq"),
    "c7_hifi_balanced": dict(args=["compress-pbhifi", "--priority", "balanced"], input="D.melanogaster.fastq"),
    # multi-pack synthetic (2 packs), default ONT preset; and one with N-containing reads, ratio preset
    "s6m_ont": dict(args=["compress-ont"], synth=dict(seed=1, genome_len=200_000, target_bases=6_000_000)),
    "s3m_ont_n_ratio": dict(args=["compress-ont", "-p", "ratio"], synth=dict(seed=7, genome_len=100_000, target_bases=3_000_000, n_frac=0.2, mean_scale=8000.0)),
    "s5m_hifi": dict(args=["compress-pbhifi"], synth=dict(seed=3, genome_len=150_000, target_bases=5_000_000, mean_scale=12000.0)),
    # the parameters the reference picks for config 5's 50 Gbases (compression.cpp:84-88): 50-bit k-mers, 44-bit m-mers
    "s6m_ont_k25": dict(args=["compress-ont", "-k", "25", "-a", "22"], synth=dict(seed=11, genome_len=200_000, target_bases=6_000_000)),
    "s4m_ont_k23_balanced": dict(args=["compress-ont", "-k", "23", "-a", "21", "-p", "balanced"], synth=dict(seed=12, genome_len=120_000, target_bases=4_000_000, mean_scale=9000.0)),
}


def gunzip(name, dst):
    with gzip.open(os.path.join(DATA, name + ".gz"), "rb") as f, open(dst, "wb") as g:
        shutil.copyfileobj(f, g)


def streams_of(path):
    out = {}
    for name, st in read_archive(path).items():
        if name == "info":
            continue
        out[name] = {"parts": [[m, len(p), hashlib.sha256(p).hexdigest()] for m, p in st.parts]}
    return out


def main(only=None):
    for cfg, spec in CONFIGS.items():
        if only and cfg not in only:
            continue
        with tempfile.TemporaryDirectory() as tmp:
            if "synth" in spec:
                inp = os.path.join(tmp, cfg + ".fastq")
                write_fastq(inp, make_reads(**spec["synth"]))
            else:
                inp = os.path.join(tmp, spec["input"])
                gunzip(spec["input"], inp)
            args = list(spec["args"])
            if "genome" in spec:
                g = os.path.join(tmp, spec["genome"])
                gunzip(spec["genome"], g)
                args += ["-G", g]
            dump = os.path.join(tmp, "dump")
            env = dict(os.environ, COLORD_DUMP_DIR=dump)
            subprocess.check_call([os.path.join(REF, "ref_dump")] + args + ["-t", "8", inp, os.path.join(tmp, "tap.colord")], env=env,
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            subprocess.check_call([os.path.join(REF, "colord")] + args + ["-t", "8", inp, os.path.join(tmp, "ref.colord")],
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            s_tap, s_ref = streams_of(os.path.join(tmp, "tap.colord")), streams_of(os.path.join(tmp, "ref.colord"))
            assert s_tap == s_ref, f"{cfg}: tap harness changes the archive"
            # round trip through the reference decompressor
            dargs = [os.path.join(REF, "colord"), "decompress"] + (["-G", g] if "genome" in spec else []) + [os.path.join(tmp, "ref.colord"), os.path.join(tmp, "rt.fastq")]
            subprocess.check_call(dargs, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            d = os.path.join(OUT, cfg)
            os.makedirs(d, exist_ok=True)
            for f in ("params.txt", "kept.bin", "accept.bin"):
                shutil.copy(os.path.join(dump, f), os.path.join(d, f))
            for f in ("cands.bin", "es.bin"):
                with open(os.path.join(dump, f), "rb") as i, gzip.GzipFile(os.path.join(d, f + ".gz"), "wb", 9, mtime=0) as o:
                    shutil.copyfileobj(i, o)
            arch = read_archive(os.path.join(tmp, "ref.colord"))
            with gzip.GzipFile(os.path.join(d, "meta.bin.gz"), "wb", 9, mtime=0) as o:
                o.write(arch["meta"].parts[0][1])
            with open(os.path.join(tmp, "rt.fastq"), "rb") as f:
                rt_sha = hashlib.sha256(f.read()).hexdigest()
            json.dump({"args": spec["args"], "input": spec.get("input"), "genome": spec.get("genome"), "synth": spec.get("synth"),
                       "streams": s_ref, "decompressed_sha256": rt_sha}, open(os.path.join(d, "streams.json"), "w"), indent=1, sort_keys=True)
            print(cfg, {k: (len(v["parts"]), sum(p[1] for p in v["parts"])) for k, v in s_ref.items()})


if __name__ == "__main__":
    main(set(sys.argv[1:]))
