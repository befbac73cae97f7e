"""Data parallelism without a cluster: 2 ranks over gloo on CPU (kernels through the host-emulation build), each taking
half of a batch, must reproduce the single-process full-batch training step (BatchNorm off: batch statistics are the one
coupling that data parallelism does not reproduce exactly — DESIGN.md §multi-GPU)."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _setup_paths():
    root = os.path.dirname(HERE)
    for p in (root, HERE, os.path.join(root, "www24-rat_amd"), os.path.join(HERE, "emu")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _make(case_name):
    import golden_cases as gc
    import model_cases as mc
    case = dict(gc.case_by_name(case_name))
    case["batch_norm"] = False
    model = mc.build_model(case, gpu=-1, seed=1)
    mc.load_weights(model, case)
    return case, model, mc.batch_of(case)


def _worker(rank, world, port, case_name, emu_path, out_dir):
    _setup_paths()
    import rat_amd._lib as L
    L._default = L.RatLib(emu_path)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case, model, batch = _make(case_name)
    per = batch[0].shape[0] // world
    shard = tuple(t[rank * per:(rank + 1) * per] for t in batch)
    model.train()
    for _ in range(2):
        loss = model.train_step(shard)
    torch.save({"flat": model._flat.clone(), "loss": loss}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_step_equals_full_batch_step():
    _setup_paths()
    import build_emu
    import rat_amd._lib as L
    emu_path = build_emu.build()
    old = L._default
    L._default = L.RatLib(emu_path)
    try:
        case, model, batch = _make("tiny_seq_bn")
        model.train()
        for _ in range(2):
            full_loss = model.train_step(batch)
        ref = model._flat.clone()
    finally:
        L._default = old
    port = 29500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as out_dir:
        mp.spawn(_worker, args=(2, port, "tiny_seq_bn", emu_path, out_dir), nprocs=2, join=True)
        r0 = torch.load(os.path.join(out_dir, "rank0.pt"))
        r1 = torch.load(os.path.join(out_dir, "rank1.pt"))
    assert torch.equal(r0["flat"], r1["flat"]), "replicas diverged"
    np.testing.assert_allclose(r0["flat"].numpy(), ref.numpy(), rtol=2e-4, atol=2e-6)
    # each rank reports (local BCE + reg)/world; their sum is the full-batch loss
    assert abs(float(r0["loss"] + r1["loss"]) - float(full_loss)) < 1e-5
