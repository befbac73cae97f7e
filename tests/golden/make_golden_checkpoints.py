#!/usr/bin/env python3
"""Generate `.model` checkpoint fixtures with the REAL reference classes (on-disk formats).

Runs only in the build container (needs /root/reference, imported read-only exactly like make_golden.py).  For one small
case per model variant the reference model is built, given the case's deterministic weights, trained for the two iterations
of make_golden.py (so BatchNorm's running statistics and `num_batches_tracked` are no longer their initial values), and
written with the reference's OWN `BaseModel.save_weights` (base_model.py:275-276: `torch.save(self.state_dict(), path)`).
The fixture is that file — tensor data keyed by the reference's state_dict names — plus the reference's eval-mode
predictions under exactly those weights, which make_golden.py already records as `eval_after/y_pred`; this script
re-derives them and asserts they equal the committed golden arrays, so the two fixtures cannot drift apart.

    python tests/golden/make_golden_checkpoints.py     # rewrites tests/golden/ckpt_*.model
"""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import numpy as np
import torch

import golden_cases as gc
import make_golden as mg

CHECKPOINT_CASES = ["tiny_seq_bn", "m0_tiny_seq", "m1_tiny_seq", "m3_tiny_seq"]


def main():
    models, FeatureMap, seed_everything = mg.import_reference()
    os.makedirs("/tmp/rat_golden/models", exist_ok=True)
    torch.set_num_threads(1)
    for name in CHECKPOINT_CASES:
        case = gc.case_by_name(name)
        model = mg.build_reference_model(case, models, FeatureMap, seed_everything, seed=case["init_seed"])
        sd = model.state_dict()
        weights = gc.make_weights(case, {k: tuple(v.shape) for k, v in sd.items()})
        model.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
        X, y, rv, rl = gc.make_inputs(case)
        batch = (torch.from_numpy(X), torch.from_numpy(y), torch.from_numpy(rv), torch.from_numpy(rl))
        model.train()
        for _ in (1, 2):                                   # BaseModel.train_one_epoch's iteration (base_model.py:220-226)
            model.optimizer.zero_grad()
            loss = model.get_total_loss(batch)
            loss.backward()
            torch.nn.utils.clip_grad_norm_(model.parameters(), 10.0)
            model.optimizer.step()
        path = os.path.join(HERE, "ckpt_%s.model" % name)
        model.save_weights(path)                           # the reference's own writer
        model.eval()
        with torch.no_grad():
            pred = model.forward(batch)["y_pred"].numpy().astype(np.float32)
        golden = np.load(os.path.join(HERE, name + ".npz"))
        assert np.array_equal(pred, golden["eval_after/y_pred"]), name
        print("%-14s %6.1f KB  %d tensors" % (name, os.path.getsize(path) / 1024, len(model.state_dict())))


if __name__ == "__main__":
    main()
