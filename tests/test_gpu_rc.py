"""The interval coder's kernel alone (k_range_code, csrc/rc_dev.hpp) against the oracle's interval coder (oracle/rc.h, the restatement of
sub_rc.h:72-100,203-210 that the CPU suite pins to the reference's streams) on random triples — totals over the whole 21-bit range and at the extremes, ragged groups, an empty
part: sizes, bytes, and nothing written beyond a part's size (the kernel stores whole unaligned words that later stores overwrite).
The coder goldens (test_gpu_qual.py, test_gpu_dna.py) cover the kernel on the models' real statistics; this covers the corners of
the division by reciprocal and of the renormalisation that real models do not reach."""
import os
import shutil
import subprocess
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="needs hipcc")
def test_range_code_kernel_equals_the_oracle_coder(tmp_path):
    exe = str(tmp_path / "rc_kernel_test")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "tools", "rc_kernel_test.hip"), "-o", exe])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok:" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
