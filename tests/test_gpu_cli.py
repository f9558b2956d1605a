"""The command-line compressor (colord_amd/colord_hip: host C++ over the C ABI) against the unmodified reference:
its archive must be decoded by the reference's `colord decompress` to exactly what the reference decodes from its own
archive of the same FASTQ, and every stream except `info` (time stamp, command line) must be byte-identical."""
import hashlib
import os
import subprocess
import pytest
from colord_amd import archive as AR
from colord_amd.fastq import write_fastq
from colord_amd.synth import make_reads

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "colord")
CLI = os.path.join(ROOT, "colord_amd", "colord_hip")


def sha(path):
    return hashlib.sha256(open(path, "rb").read()).hexdigest()


@pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(CLI)), reason="needs oracle/_ref/colord and colord_amd/colord_hip")
@pytest.mark.parametrize("mode,prio,seed", [("compress-ont", None, 5), ("compress-ont", "balanced", 6), ("compress-pbhifi", None, 7), ("compress-pbraw", "ratio", 8)])
def test_cli_archive_decoded_by_reference(tmp_path, mode, prio, seed):
    rs = make_reads(seed=seed, genome_len=100_000, target_bases=5_000_000, mean_scale=6000.0)
    fq = str(tmp_path / "in.fastq")
    write_fastq(fq, rs)
    extra = ["-p", prio] if prio else []
    ref_arc, ref_out, my_arc, my_out = (str(tmp_path / x) for x in ("ref.colord", "ref.fastq", "gpu.colord", "gpu.fastq"))
    subprocess.check_call([REF, mode, "-t", "4"] + extra + [fq, ref_arc], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([REF, "decompress", ref_arc, ref_out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    subprocess.check_call([CLI, mode] + extra + [fq, my_arc])
    subprocess.check_call([REF, "decompress", my_arc, my_out], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    assert sha(my_out) == sha(ref_out)
    a, b = AR.read_archive(ref_arc), AR.read_archive(my_arc)
    assert set(a) == set(b)
    for name in a:
        if name != "info":
            assert [(m, hashlib.sha256(p).hexdigest()) for m, p in a[name].parts] == [(m, hashlib.sha256(p).hexdigest()) for m, p in b[name].parts], name
