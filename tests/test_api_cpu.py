"""CPU suite for the public C++ reader API (include/colord_api.h mirrors the reference's src/API/colord_api.h:27-103).

A program written against the API — tests/tools/api_dump.cpp, and the reference's own src/API_example/api_example.cpp compiled
UNCHANGED against this build's header and library when the reference tree is present — must print, for archives written by the
unmodified reference (tests/golden/archives), exactly the FASTQ the reference's `decompress` returns, and report the archive's
info fields.  Decoding is host code: no GPU."""
import hashlib
import json
import os
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARC = os.path.join(ROOT, "tests", "golden", "archives")
EXP = json.load(open(os.path.join(ARC, "expected.json")))
LIBDIR = os.path.join(ROOT, "colord_amd")
REF_EXAMPLE = "/root/reference/src/API_example/api_example.cpp"


def build(src, out):
    subprocess.check_call(["g++", "-O1", "-std=c++17", src, "-I", os.path.join(ROOT, "include"), "-L", LIBDIR, "-lcolord_hip_api", "-lcolord_hip", "-lz", "-lpthread",
                           f"-Wl,-rpath,{LIBDIR}", "-Wl,-rpath,/opt/rocm/lib", "-o", out])
    return out


@pytest.fixture(scope="module")
def api_dump(tmp_path_factory):
    assert os.path.exists(os.path.join(LIBDIR, "libcolord_hip_api.a")), "run make -C colord_amd/csrc (or __graft_entry__.build())"
    return build(os.path.join(ROOT, "tests", "tools", "api_dump.cpp"), str(tmp_path_factory.mktemp("api") / "api_dump"))


@pytest.mark.parametrize("name", sorted(EXP))
def test_api_records_equal_the_reference_decompressor(api_dump, name):
    extra = [os.path.join(ROOT, "tests", "data", "M.bovis-reference.fna.gz")] if name.endswith("_external") else []
    r = subprocess.run([api_dump, os.path.join(ARC, name + ".colord")] + extra, capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    assert hashlib.sha256(r.stdout).hexdigest() == EXP[name]["decompressed_sha256"]


def test_api_info_fields(api_dump):
    r = subprocess.run([api_dump, os.path.join(ARC, "c1_ont_default.colord")], capture_output=True, text=True)
    assert r.returncode == 0
    for line in ("is fastq: true", "colord archive version: 1.2.", "total reads: 100", "total bases: 449286", "compression level: 1", "reads source: Oxford Nanopore",
                 "quality compression mode: Quad average", "header compression mode: Original", "records: 100"):
        assert line in r.stderr, line
    r = subprocess.run([api_dump, os.path.join(ARC, "c2_hifi_org.colord")], capture_output=True, text=True)
    assert "reads source: PacBio HiFi" in r.stderr and "quality compression mode: Original" in r.stderr and "compression level: 2" in r.stderr
    r = subprocess.run([api_dump, os.path.join(ARC, "bovis24_q_4-fix_balanced.colord")], capture_output=True, text=True)
    assert "quality compression mode: Quad threshold" in r.stderr and "quality reverse thresholds: " in r.stderr


def test_api_errors_are_exceptions(api_dump, tmp_path):
    r = subprocess.run([api_dump, str(tmp_path / "missing.colord")], capture_output=True, text=True)
    assert r.returncode == 1 and "Error: cannot open archive" in r.stderr
    bad = tmp_path / "bad.colord"
    bad.write_bytes(b"not an archive at all")
    r = subprocess.run([api_dump, str(bad)], capture_output=True, text=True)
    assert r.returncode == 1 and "Error:" in r.stderr


@pytest.mark.skipif(not os.path.exists(REF_EXAMPLE), reason="reference tree not present")
def test_the_references_own_api_example_builds_and_runs_against_this_library(tmp_path):
    exe = build(REF_EXAMPLE, str(tmp_path / "api_example"))
    for name in ("c1_ont_default", "c3_clr_ratio"):
        r = subprocess.run([exe, os.path.join(ARC, name + ".colord")], capture_output=True)
        assert r.returncode == 0, r.stderr.decode()
        assert hashlib.sha256(r.stdout).hexdigest() == EXP[name]["decompressed_sha256"]
        assert b"Database info:" in r.stderr
