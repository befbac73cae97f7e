"""End to end on the MI355X (-m gpu): encoded splits on disk WITHOUT retrieval files -> run_expid computes the top-K retrieval on
the device (rat_bm25_topk, <X>-fold for the training split, training data as the pool for valid / test — the reference's
DataGenerator / h5_generator rules), stores retrieval_{K}_{split}.npz next to the data, keeps the dataset in HBM and trains."""
import json
import os
import shutil
import sys

import numpy as np
import pytest
import yaml

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_run_expid_precomputes_retrieval_and_trains(tmp_path, monkeypatch):
    sys.path.insert(0, ROOT)
    import run_expid
    from rat_amd import data as rat_data, synthetic
    spec = synthetic.WORKLOADS["tiny"]
    fm = synthetic.feature_map_for("demo_x1_retrieval", spec)
    data_dir = tmp_path / "data" / "demo_x1_retrieval"
    data_dir.mkdir(parents=True)
    sizes = {"train": 3000, "valid": 400, "test": 400}
    for i, (split, n) in enumerate(sizes.items()):
        arr, _, _, _ = rat_data.synthetic_split(fm, n, 3, seed=10 + i)
        np.savez_compressed(data_dir / (split + ".npz"), data=arr)
    fm.save(str(data_dir / "feature_map.json"))
    cfg_dir = tmp_path / "cfg" / "RAT_m2" / "demo"
    cfg_dir.mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "tests", "fixtures_cfg", "RAT_m2", "demo", "model_config.yaml"), cfg_dir / "model_config.yaml")
    ds = {"demo_x1_retrieval": {"data_format": "npz", "data_root": str(tmp_path / "data") + "/",
                                "retrieval_configs": {"used_cols": list(fm.feature_specs)[:3], "exact_match_cols": [], "split_type": "10-fold",
                                                      "label_wise": False, "pre_retrieval": True, "qry_batch_size": 1000,
                                                      "db_chunk_size": 50000, "device": "cuda:0", "topK": 3}}}
    (cfg_dir / "dataset_config.yaml").write_text(yaml.safe_dump(ds))
    monkeypatch.chdir(tmp_path)
    run_expid.main(["--config", str(cfg_dir), "--expid", "RAT_m2_demo", "--gpu", "0", "--epochs", "1"])
    n_train = sizes["train"]
    fold = int(np.ceil(n_train / 10))
    for split, n in sizes.items():
        r = np.load(data_dir / ("retrieval_3_%s.npz" % split))
        assert r["indices"].shape == (n, 3) and r["values"].shape == (n, 3) and r["lens"].shape == (n,)
        assert (r["indices"] >= -1).all() and (r["indices"] < n_train).all()
        assert (np.diff(r["values"], axis=1) <= 0).all()
    tr = np.load(data_dir / "retrieval_3_train.npz")["indices"]
    own = np.arange(n_train)[:, None] // fold
    assert ((tr // fold) != own).all(), "a training row retrieved a neighbour from its own fold"
    csvs = [f for f in os.listdir(tmp_path / "_demo_models" / "demo_x1_retrieval") if f.endswith(".csv")]
    assert csvs, "no result line written"
    line = open(tmp_path / "_demo_models" / "demo_x1_retrieval" / csvs[0]).read()
    assert "AUC" in line and "logloss" in line


def test_training_learns_on_synthetic_data(tmp_path, monkeypatch):
    """run_expid --synthetic: labels depend on the ids (rat_amd.data.synthetic_split), so two epochs through the HIP path must lift
    the validation AUC well above chance (0.70 measured) — optimizer, checkpointing, evaluation and the result line included."""
    import re
    sys.path.insert(0, ROOT)
    import run_expid
    monkeypatch.chdir(tmp_path)
    run_expid.main(["--config", os.path.join(ROOT, "tests", "fixtures_cfg", "RAT_m2", "demo"), "--expid", "RAT_m2_demo", "--gpu", "0",
                    "--synthetic", "20000", "--epochs", "2"])
    out_dir = tmp_path / "_demo_models" / "demo_x1_retrieval"
    line = open(out_dir / [f for f in os.listdir(out_dir) if f.endswith(".csv")][0]).read()
    aucs = [float(x) for x in re.findall(r"AUC: ([0-9.]+)", line)]
    assert aucs and min(aucs) > 0.6, line
