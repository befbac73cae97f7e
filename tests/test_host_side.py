"""Host-side logic that needs neither GPU nor kernels: config loading (reference YAML layout, including the shipped
`configs/datasets/` location the reference itself fails to search), the vectorised batch source, metrics, the C-ABI
surface of the built library, and the no-fallback rule."""
import ctypes
import os
import re
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = os.path.join(ROOT, "tests", "fixtures_cfg")


def test_load_config_merges_base_expid_and_dataset():
    from rat_amd.config import load_config
    params = load_config(os.path.join(CFG, "RAT_m2", "demo"), "RAT_m2_demo")
    assert params["model_id"] == "RAT_m2_demo" and params["model"] == "RAT_m2"
    assert params["model_root"] == "./_demo_models/"                       # from Base
    assert params["retrieval_configs"]["topK"] == 3                        # from ../../datasets/*.yaml
    with pytest.raises(ValueError):
        load_config(os.path.join(CFG, "RAT_m2", "demo"), "nope")


def test_batch_source_matches_reference_getitem():
    """data_generator.py:66-78: row i -> concat(darray[i], pool[retr_indices[i]]); -1 indexes the LAST pool row."""
    from rat_amd.data import RetrievalBatches
    rs = np.random.RandomState(0)
    n, L, K = 23, 4, 3
    data = np.concatenate([rs.randint(0, 9, size=(n, L)), rs.randint(0, 2, size=(n, 1))], axis=1).astype(np.float64)
    idx = rs.randint(-1, n, size=(n, K))
    vals, lens = rs.rand(n, K), rs.randint(0, K + 1, size=n)
    src = RetrievalBatches(data, data, idx, vals, lens, batch_size=5)
    assert len(src) == 5
    seen = 0
    for X, y, v, l in src:
        for b in range(X.shape[0]):
            i = seen + b
            ref = np.concatenate([data[i][None], data[idx[i]]])            # numpy semantics, negative index included
            np.testing.assert_array_equal(X[b].numpy(), ref[:, :-1].astype(np.int32))
            np.testing.assert_array_equal(y[b].numpy(), ref[:, -1].astype(np.float32))
        np.testing.assert_allclose(v.numpy(), vals[seen:seen + X.shape[0]].astype(np.float32))
        seen += X.shape[0]
    assert seen == n


def test_metrics_match_sklearn():
    from rat_amd.metrics import auc_score, log_loss
    sk = pytest.importorskip("sklearn.metrics")
    rs = np.random.RandomState(1)
    y = rs.randint(0, 2, size=500)
    p = np.round(rs.rand(500), 2)                                          # ties on purpose
    assert abs(auc_score(y, p) - sk.roc_auc_score(y, p)) < 1e-12
    assert abs(log_loss(y, p) - sk.log_loss(y, np.clip(p, 1e-7, 1 - 1e-7))) < 1e-12


def test_library_exports_every_declared_symbol():
    """include/rat_hip.h <-> librat_hip.so <-> the ctypes table agree (no compute calls: there is no GPU here)."""
    from rat_amd._lib import DEFAULT_LIB, EXPORTED_SYMBOLS
    header = open(os.path.join(ROOT, "include", "rat_hip.h")).read()
    declared = set(re.findall(r"\b(rat_[a-z0-9_]+)\s*\(", header))
    assert declared == set(EXPORTED_SYMBOLS), declared ^ set(EXPORTED_SYMBOLS)
    if not os.path.exists(DEFAULT_LIB):
        import importlib.util
        spec = importlib.util.spec_from_file_location("rat_build", os.path.join(ROOT, "www24-rat_amd", "build.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        mod.build()
    lib = ctypes.CDLL(DEFAULT_LIB)
    for sym in declared:
        assert hasattr(lib, sym), sym
    from rat_amd._lib import ABI_VERSION
    assert lib.rat_version() == ABI_VERSION == 9


def test_no_silent_fallback_when_library_is_missing(tmp_path, monkeypatch):
    from rat_amd import _lib
    monkeypatch.setenv("RAT_HIP_LIBRARY", str(tmp_path / "missing.so"))
    with pytest.raises(_lib.RatError, match="no fallback"):
        _lib.RatLib()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "www24-rat_amd", "rat_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("# oracle", ""), fn


def test_compat_registers_fuxictr_namespace():
    import sys
    from rat_amd import compat
    saved = {k: v for k, v in sys.modules.items() if k.startswith("fuxictr")}
    try:
        for k in list(saved):
            del sys.modules[k]
        assert compat.install() in ("registered", "patched")
        import fuxictr
        from fuxictr.pytorch import models
        assert fuxictr.__version__.startswith("1.2")                       # run_expid.py:14-15 of the reference
        assert getattr(models, "RAT_m2").__name__ == "RAT_m2"
    finally:
        for k in [k for k in sys.modules if k.startswith("fuxictr")]:
            del sys.modules[k]
        sys.modules.update(saved)


def test_cpu_baseline_runs_both_thread_settings_and_bounds_the_all_cores_attempt(monkeypatch):
    """bench.py's cpu_baseline: the 32-thread run in-process, the all-cores run in a time-boxed CPU-only child (a 256-thread host needs
    357 s per step at the full batch).  Tiny workload, an 'os.cpu_count() = 64' host: both complete; with a zero budget the child is ended
    and the entry says so."""
    import importlib
    sys.path.insert(0, ROOT)
    bench = importlib.import_module("bench")
    from rat_amd import synthetic
    monkeypatch.setattr(bench.os, "cpu_count", lambda: 64)
    spec = synthetic.WORKLOADS["tiny"]
    out = bench.cpu_baseline("tiny", spec, 32, 1000, timed_steps=1, all_cores_budget_s=120.0)
    assert [r["threads"] for r in out["runs"]] == [32, 64] and all(r["value"] for r in out["runs"])
    assert out["value"] == max(r["value"] for r in out["runs"]) and out["kind"] == "port" and out["sample_batch"] == 32
    out = bench.cpu_baseline("tiny", spec, 32, 1000, timed_steps=1, all_cores_budget_s=0.01)
    assert out["runs"][1]["value"] is None and "ended" in out["runs"][1]["note"] and out["value"] == out["runs"][0]["value"]
