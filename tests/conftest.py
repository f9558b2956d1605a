import os
import sys
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


@pytest.fixture(scope="session")
def ctx():
    """One HIP context for the whole GPU session; fails loudly when the library or the GPU is missing."""
    from colord_amd.device import Context
    c = Context(0)
    yield c
    c.close()


def pytest_runtest_logreport(report):
    """Every failure's full text goes to gpurun_out/test_failures.txt as well (a summary line on a terminal is cut at the first 80 columns;
    a rare failure on the GPU box must leave its message behind)."""
    if report.failed:
        try:
            d = os.path.join(ROOT, "gpurun_out")
            os.makedirs(d, exist_ok=True)
            with open(os.path.join(d, "test_failures.txt"), "a") as f:
                f.write(f"==== {report.nodeid} [{report.when}]\n{report.longreprtext}\n")
        except OSError:
            pass
