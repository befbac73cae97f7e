#!/usr/bin/env python3
"""Diagnostic: a few hundred training steps of the product's default configuration (fused iteration, hipGraph replays, dead-token
pruning) at the north-star shape — loss finite and falling on rotating batches, weights finite, device memory flat after the graphs
exist.  `python tools/soak.py [steps] [workload] [model]` (model: RAT_m2 (default) / RAT_m0 / RAT_m1 / RAT_m3).  Not a test (takes ~20 s of GPU); run on the GPU box."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "www24-rat_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
from rat_amd import models, synthetic  # noqa: E402
from rat_amd.base_model import seed_everything  # noqa: E402


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
    name = sys.argv[2] if len(sys.argv) > 2 else "synthetic_F20_V1M_K10_d64_B4096"
    spec = synthetic.WORKLOADS[name]
    fm = synthetic.feature_map_for(name, spec)
    seed_everything(2021)
    variant = sys.argv[3] if len(sys.argv) > 3 else "RAT_m2"
    model = getattr(models, variant)(fm, **synthetic.model_kwargs(spec, gpu=0))
    batches = [synthetic.make_batch(spec, fm, seed=100 + i, device=model.device, as_float64=False) for i in range(8)]
    model.train()
    losses, mem = [], []
    for s in range(steps):
        losses.append(model.train_step(batches[s % len(batches)]))
        if s in (9, steps - 1):
            torch.cuda.synchronize()
            mem.append(torch.cuda.memory_allocated())
    losses = [float(v) for v in losses]
    assert all(v == v and abs(v) < 1e6 for v in losses), "non-finite loss"
    assert bool(torch.isfinite(model._flat).all()), "non-finite weights"
    model.check_id_errors()
    first, last = sum(losses[:8]) / 8, sum(losses[-8:]) / 8
    replayed = any(e[1] for e in model._step_graphs.values())
    print(variant + " %s: %d steps, loss %.5f -> %.5f (means of 8), graph replays: %s, pruning: %s, memory after 10 steps %.1f MB, at the end %.1f MB"
          % (name, steps, first, last, replayed, model.prune_dead_tokens, mem[0] / 2 ** 20, mem[1] / 2 ** 20))
    assert last < first, "the loss did not fall on 8 rotating batches"
    assert mem[1] <= mem[0] * 1.001 + (1 << 20), "device memory grew during the run"


if __name__ == "__main__":
    main()
