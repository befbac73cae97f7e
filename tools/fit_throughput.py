#!/usr/bin/env python3
"""Diagnostic: throughput of the product's TRAINING LOOP (BaseModel.fit_generator over device-resident retrieval batches — what
run_expid.py runs) against the bare training step bench.py measures, at a bench workload's shape.  The difference is what the loop
adds per step: the epoch's permutation, the row-id upload, rat_batch_assemble, the running-loss add.
`python tools/fit_throughput.py [workload] [steps]`."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "www24-rat_amd")):
    sys.path.insert(0, p)

import torch  # noqa: E402
from rat_amd import data as rd  # noqa: E402
from rat_amd import models, synthetic  # noqa: E402
from rat_amd.base_model import seed_everything  # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "synthetic_F20_V1M_K10_d64_B4096"
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    spec = synthetic.WORKLOADS[name]
    fm = synthetic.feature_map_for(name, spec)
    seed_everything(2021)
    model = models.RAT_m2(fm, **synthetic.model_kwargs(spec, gpu=0))
    B, K = spec["batch"], spec["K"]
    data, idx, val, lens = rd.synthetic_split(fm, B * steps, K, seed=3)
    gen = rd.RetrievalBatches(data, data, idx, val, lens, B, shuffle=True, seed=0).to_device(model.device)
    model.fit_generator(gen, epochs=1)                      # warm-up epoch: caches, graph capture
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.fit_generator(gen, epochs=1)
    torch.cuda.synchronize()
    loop = (time.perf_counter() - t0) / steps
    batches = list(gen)[:8]
    model.train()
    for b in batches:
        model.train_step(b)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        model.train_step(batches[s % len(batches)])
    torch.cuda.synchronize()
    bare = (time.perf_counter() - t0) / steps
    print("%s: fit_generator %.3f ms per step (%.0f samples/s); bare train_step %.3f ms; the loop adds %.3f ms"
          % (name, loop * 1e3, B / loop, bare * 1e3, (loop - bare) * 1e3))


if __name__ == "__main__":
    main()
