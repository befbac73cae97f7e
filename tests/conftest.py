import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "www24-rat_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


# The CPU suite runs the kernels through a lane-accurate host emulation (one OS thread per GPU thread): seconds to minutes per test.
# Emulator runs that only repeat a `-m gpu` test at another shape are skipped unless RAT_CPU_FULL=1 — the complete matrix runs on the
# MI355X (tests/test_gpu_*.py), every kernel and every host path keeps at least one emulator run here.
import pytest  # noqa: E402

FULL_CPU = os.environ.get("RAT_CPU_FULL") == "1"
gpu_twin = pytest.mark.skipif(not FULL_CPU, reason="emulator repeat of a -m gpu test at another shape; RAT_CPU_FULL=1 runs it")


def twin(*values, **kw):
    """pytest.param(...) carrying the gpu_twin skip"""
    return pytest.param(*values, marks=gpu_twin, **kw)
