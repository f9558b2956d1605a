"""CPU suite: the C-ABI library builds, loads and exports exactly what include/colord_hip.h declares."""
import os
import re
import subprocess
import pytest
from colord_amd import _native as N

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    hdr = open(os.path.join(ROOT, "include", "colord_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(cl_[a-z0-9_]+)\s*\(", hdr))
    return sorted(names - {"cl_status"})          # `cl_status (*callback)(...)` members of cl_exchange are not entry points


def test_library_exports_every_declared_symbol():
    lib = N.load()
    names = header_functions()
    assert len(names) >= 40
    for n in names:
        assert hasattr(lib, n), f"{n} declared in colord_hip.h but not exported"
    assert sorted(N.exported_names()) == names, "ctypes binding and header disagree"


def test_library_contains_gfx950_code_object():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", N.LIB_PATH], capture_output=True, text=True)
    assert "gfx950" in (out.stdout + out.stderr)


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from colord_amd.device import Context
    with pytest.raises(N.ColordHipError):
        Context(0)


def test_host_acceptor_entry_point_matches_golden():
    # cl_ref_accept is a host function of the ABI (sequential RNG stream), so it is checkable without a GPU
    import ctypes as C
    import numpy as np
    from util import PLAIN_CONFIGS, golden
    lib = N.load()
    for cfg in PLAIN_CONFIGS:
        g = golden(cfg)
        if not g.p("sparse"):
            continue
        out = np.zeros(g.p("n_reads") + g.p("n_pseudo"), np.uint8)
        assert lib.cl_ref_accept(g.p("n_reads"), g.p("n_pseudo"), g.p("sparse_range"), g.p("sparse_exp"), out.ctypes.data) == 0
        assert np.array_equal(out, g.accept)
