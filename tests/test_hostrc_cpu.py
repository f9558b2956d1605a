"""The host decoder's interval arithmetic and model family (colord_amd/csrc/host_coder.hpp: dense context index, unrolled / blocked symbol
search, products instead of the second division, loop-free renormalisation) against the oracle's decoder (oracle/rc.h, oracle/rc.c — the
restatement of sub_rc.h:216-392 and rc.h:34-764 that this suite pins to the reference's golden streams): 4 * 10^6 random renormalisation
states, and streams of 300 000 symbols for alphabets of 2 .. 256 symbols with and without exclusions, dense and hashed contexts."""
import os
import shutil
import subprocess
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (shutil.which("g++") and shutil.which("gcc")), reason="needs gcc / g++")
def test_host_decoder_equals_the_oracle_decoder(tmp_path):
    obj, exe = str(tmp_path / "orc.o"), str(tmp_path / "hostrc_test")
    subprocess.check_call(["gcc", "-O2", "-c", os.path.join(ROOT, "oracle", "rc.c"), "-o", obj])
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(ROOT, "tests", "tools", "hostrc_test.cpp"), obj, "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), r.stdout[-2000:] + r.stderr[-2000:]
