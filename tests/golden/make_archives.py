#!/usr/bin/env python3
"""Decoder fixtures (run in the build container only): archives written by the UNMODIFIED reference (oracle/_ref/colord) and the
SHA-256 of what its own `decompress` returns for them.  tests/test_decode_cpu.py decodes the same archives with
`colord_hip decompress` (cl_dna_decode_part / cl_qual_decode_part / cl_id_decode_part) and compares.

  full inputs   c1 / c2 / c3 of BASELINE.json (all three sequencing modes, levels 1-3, sparse and all-reads reference sets)
  M.bovis[:24]  every quality mode at level 2 (per-base class flags on), both lossy header modes, level 3 with -q org
"""
import gzip, hashlib, json, os, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = os.path.join(ROOT, "oracle", "_ref", "colord")
DATA = os.path.join(ROOT, "tests", "data")
OUT = os.path.join(ROOT, "tests", "golden", "archives")

CASES = {
    "c1_ont_default": ("M.bovis.fastq", 0, ["compress-ont"]),
    "c2_hifi_org": ("D.melanogaster.fastq", 0, ["compress-pbhifi", "-q", "org"]),
    "c3_clr_ratio": ("A.thaliana.fastq", 0, ["compress-pbraw", "-p", "ratio"]),
}
for q in ("org", "5-avg", "4-avg", "2-avg", "5-fix", "4-fix", "2-fix", "avg", "none"):
    CASES[f"bovis24_q_{q}_balanced"] = ("M.bovis.fastq", 24, ["compress-ont", "-q", q, "-p", "balanced"])
CASES["bovis24_org_ratio"] = ("M.bovis.fastq", 24, ["compress-ont", "-q", "org", "-p", "ratio"])
CASES["bovis24_header_main"] = ("M.bovis.fastq", 24, ["compress-ont", "-i", "main"])
CASES["bovis24_header_none"] = ("M.bovis.fastq", 24, ["compress-ont", "-i", "none"])


# reference-genome mode (config 4 of BASELINE.json): the genome stored in the archive (-s) and only its checksum (no -s)
GENOME = "M.bovis-reference.fna"
CASES["c4_ont_genome_stored"] = ("M.bovis.fastq", 0, ["compress-ont", "-G", GENOME, "-s"])
CASES["c4_ont_genome_external"] = ("M.bovis.fastq", 0, ["compress-ont", "-G", GENOME])


def main():
    os.makedirs(OUT, exist_ok=True)
    exp = {}
    with tempfile.TemporaryDirectory() as tmp:
        for name, (inp, n_reads, args) in CASES.items():
            fq = os.path.join(tmp, name + ".fastq")
            lines = gzip.open(os.path.join(DATA, inp + ".gz"), "rb").read().split(b"\n")
            open(fq, "wb").write(b"\n".join(lines[:4 * n_reads] if n_reads else lines[:-1]) + b"\n")
            arc = os.path.join(OUT, name + ".colord")
            genome = os.path.join(tmp, GENOME)
            if GENOME in args and not os.path.exists(genome):
                open(genome, "wb").write(gzip.open(os.path.join(DATA, GENOME + ".gz"), "rb").read())
            run_args = [genome if a == GENOME else a for a in args]
            subprocess.check_call([REF] + run_args + ["-t", "4", fq, arc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            out = os.path.join(tmp, name + ".out")
            subprocess.check_call([REF, "decompress"] + (["-G", genome] if name.endswith("_external") else []) + [arc, out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            exp[name] = {"args": args, "input": inp, "first_reads": n_reads, "decompressed_sha256": hashlib.sha256(open(out, "rb").read()).hexdigest(),
                         "archive_bytes": os.path.getsize(arc)}
            print(name, exp[name]["archive_bytes"])
    json.dump(exp, open(os.path.join(OUT, "expected.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
