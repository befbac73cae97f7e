#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REAL reference.

Runs only in the build container (needs /root/reference).  The reference is
imported read-only with empty stand-in modules for ``h5py`` and ``dgl`` (neither
is touched on the RAT_m2 path); nothing of it is copied here.
The fixtures hold DATA only: the case description (config + seeds), and the
reference's outputs (y_pred, loss, gradient and post-Adam-step summaries,
initial weights under ``seed_everything``).  Inputs and weights are regenerated
from ``numpy.random.RandomState`` seeds by ``tests/golden_cases.py`` on both
sides, so the files stay small.

    python tests/golden/make_golden.py        # rewrites tests/golden/*.npz
"""
import os
import sys
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np
import torch

import golden_cases as gc


def import_reference():
    for name in ["h5py", "dgl", "dgl.function", "dgl.nn", "dgl.nn.functional"]:
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["dgl.nn.functional"].edge_softmax = None
    sys.path.insert(0, "/root/reference")
    from fuxictr.pytorch import models                 # noqa: E402
    from fuxictr.features import FeatureMap            # noqa: E402
    from fuxictr.pytorch.torch_utils import seed_everything  # noqa: E402
    return models, FeatureMap, seed_everything


def build_reference_model(case, models, FeatureMap, seed_everything, seed=None):
    fm = FeatureMap(case["name"], "/tmp/rat_golden")
    fm.feature_specs = gc.feature_specs(case)
    fm.num_fields = len(fm.feature_specs)
    if seed is not None:
        seed_everything(seed)
    kw = gc.model_kwargs(case)
    kw["model_root"] = "/tmp/rat_golden/models"
    return getattr(models, case.get("model", "RAT_m2"))(fm, **kw)       # like run_expid.py:75


def run_case(case, models, FeatureMap, seed_everything):
    out = {}
    # ---- (1) initial weights under the reference's own init rules
    model = build_reference_model(case, models, FeatureMap, seed_everything, seed=case["init_seed"])
    sd = model.state_dict()
    for k, v in sd.items():
        if k.startswith("query_proj"):
            continue
        gc.put_summary(out, "init/" + k, v.detach().numpy(), full_limit=case["full_limit"])
    out["param_count"] = np.int64(sum(p.numel() for p in model.parameters() if p.requires_grad))

    # ---- (2) deterministic weights + inputs, eval forward
    weights = gc.make_weights(case, {k: tuple(v.shape) for k, v in sd.items()})
    model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    X, y, rv, rl = gc.make_inputs(case)
    batch = (torch.from_numpy(X), torch.from_numpy(y), torch.from_numpy(rv), torch.from_numpy(rl))
    model.eval()
    with torch.no_grad():
        ev = model.forward(batch)
    out["eval/y_pred"] = ev["y_pred"].numpy().astype(np.float32)
    out["eval/y_true"] = ev["y_true"].numpy().astype(np.float32)

    # ---- (3) two training iterations exactly as BaseModel.train_one_epoch does them
    model.train()
    stash = {}
    inner_forward = model.forward

    def recording_forward(inputs):
        stash["ret"] = inner_forward(inputs)
        return stash["ret"]
    model.forward = recording_forward          # get_total_loss() drops y_pred; keep it
    for step in (1, 2):
        model.optimizer.zero_grad()
        loss = model.get_total_loss(batch)
        out["train%d/y_pred" % step] = stash["ret"]["y_pred"].detach().numpy().astype(np.float32)
        loss.backward()
        out["train%d/loss" % step] = np.float64(loss.item())
        for k, p in model.named_parameters():
            if k.startswith("query_proj"):
                assert p.grad is None
                continue
            gc.put_summary(out, "train%d/grad/%s" % (step, k), p.grad.numpy(), full_limit=case["full_limit"])
        gnorm = torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
        out["train%d/gnorm" % step] = np.float64(float(gnorm))
        model.optimizer.step()
        for k, v in model.state_dict().items():
            if k.startswith("query_proj"):
                continue
            gc.put_summary(out, "train%d/post/%s" % (step, k), v.detach().numpy().astype(np.float64)
                           if v.dtype == torch.int64 else v.detach().numpy(), full_limit=case["full_limit"])
    # ---- (4) eval after training (uses the updated running stats)
    model.eval()
    with torch.no_grad():
        ev = model.forward(batch)
    out["eval_after/y_pred"] = ev["y_pred"].numpy().astype(np.float32)
    if case.get("task", "binary_classification") != "binary_classification":
        return out                                            # (AUC / logloss are classification metrics)
    from sklearn.metrics import roc_auc_score, log_loss
    yt = out["eval/y_true"].reshape(-1).astype(np.float64)
    yp = np.clip(out["eval/y_pred"].reshape(-1).astype(np.float64), 1e-7, 1 - 1e-7)
    if 0 < yt.sum() < len(yt):
        out["eval/auc"] = np.float64(roc_auc_score(yt, out["eval/y_pred"].reshape(-1).astype(np.float64)))
    out["eval/logloss"] = np.float64(log_loss(yt, yp))
    return out


def main():
    models, FeatureMap, seed_everything = import_reference()
    os.makedirs("/tmp/rat_golden/models", exist_ok=True)
    torch.set_num_threads(1)          # one thread -> reproducible reduction order
    for case in gc.CASES:
        only = sys.argv[1:]
        if only and case["name"] not in only:
            continue
        out = run_case(case, models, FeatureMap, seed_everything)
        path = os.path.join(HERE, case["name"] + ".npz")
        np.savez_compressed(path, **out)
        print("%-16s %4d arrays  %7.1f KB  params=%d" % (case["name"], len(out), os.path.getsize(path) / 1024,
                                                        int(out["param_count"])))
    # known-answer parameter counts of the three shipped configs (exps/RAT_m2/*/*.log)
    counts = {}
    for kc in gc.KNOWN_COUNT_CASES:
        m = build_reference_model(kc, models, FeatureMap, seed_everything, seed=1)
        counts[kc["name"]] = np.int64(sum(p.numel() for p in m.parameters() if p.requires_grad))
        print("count", kc["name"], int(counts[kc["name"]]), "expected", kc["expected_params"])
        assert int(counts[kc["name"]]) == kc["expected_params"]
    np.savez_compressed(os.path.join(HERE, "param_counts.npz"), **counts)


if __name__ == "__main__":
    main()
