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


@pytest.fixture()
def knob():
    """knob(lib, name, value): set a diagnostic knob of a loaded kernel library (rat_debug_set_knob — the library reads RAT_* variables
    once, at load, never per launch) for the rest of the test; the previous value comes back afterwards."""
    undo = []

    def set_knob(lib, name, value):
        import ctypes
        fn = lib.cdll.rat_debug_set_knob
        fn.restype, fn.argtypes = ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]
        old = fn(name.encode(), int(value))
        assert old != -2 ** 31, "unknown knob %r" % name
        undo.append((fn, name.encode(), old))
    yield set_knob
    for fn, name, old in reversed(undo):
        fn(name, old)
