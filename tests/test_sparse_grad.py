"""Row-sparse / deterministic embedding-gradient path on CPU: the kernel sources through the host emulation (tests/emu), plus the
2-rank gloo exchange of (row ids, gradient rows).  The GPU twin is tests/test_gpu_sparse.py."""
import os
import sys
import tempfile

import numpy as np
import pytest
from conftest import gpu_twin
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "emu"))

import sparse_cases as sc  # noqa: E402
from rat_amd._lib import RatLib  # noqa: E402


@pytest.fixture(scope="module")
def emu():
    import build_emu
    return RatLib(build_emu.build())


@pytest.fixture()
def emu_default(emu):
    import rat_amd._lib as L
    old = L._default
    L._default = emu
    yield emu
    L._default = old


@pytest.mark.parametrize("d", [8, 64])
def test_sorted_segmented_reduce(emu, d):
    sc.check_sorted_reduce(emu, "cpu", d)


def test_scalar_reduce_for_the_wide_tables(emu):
    sc.check_scalar_reduce(emu, "cpu")


def test_merge_of_gathered_row_lists_and_row_adam(emu):
    sc.check_merge_rows(emu, "cpu")


@gpu_twin
def test_sorted_mode_equals_atomic_mode_and_is_reproducible(emu_default):
    sc.check_model_sorted_equals_atomic("tiny_seq_bn", gpu=-1)


def test_out_of_vocabulary_ids_are_treated_alike_by_every_gradient_path(emu_default):
    sc.check_model_sorted_equals_atomic("tiny_seq_bn", gpu=-1, bad_ids=True)


def test_sparse_training_equals_dense_training(emu_default):
    sc.check_model_sparse_training("tiny_seq_bn", gpu=-1, steps=2)


def _worker(rank, world, port, emu_path, out_dir):
    for p in (os.path.dirname(HERE), HERE, os.path.join(os.path.dirname(HERE), "www24-rat_amd"), os.path.join(HERE, "emu")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import rat_amd._lib as L
    import sparse_cases as sc2
    L._default = L.RatLib(emu_path)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case, model, batch = sc2._model("tiny_seq_bn", -1, "sparse", embedding_regularizer=0.0, batch_norm=False)
    per = batch[0].shape[0] // world
    shard = tuple(t[rank * per:(rank + 1) * per] for t in batch)
    model.train()
    for _ in range(2):
        loss = model.train_step(shard)
    torch.save({"flat": model._flat.clone(), "loss": loss}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sparse_exchange_equals_single_process(emu_default):
    """C2 of SURVEY.md §8e: all-gather of (row ids, gradient rows) + local deterministic merge == the full batch on one process"""
    import build_emu
    case, model, batch = sc._model("tiny_seq_bn", -1, "sparse", embedding_regularizer=0.0, batch_norm=False)
    model.train()
    for _ in range(2):
        full_loss = model.train_step(batch)
    ref = model._flat.clone()
    port = 30500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as out_dir:
        mp.spawn(_worker, args=(2, port, build_emu.build(), out_dir), nprocs=2, join=True)
        r0 = torch.load(os.path.join(out_dir, "rank0.pt"))
        r1 = torch.load(os.path.join(out_dir, "rank1.pt"))
    assert torch.equal(r0["flat"], r1["flat"]), "replicas diverged"
    np.testing.assert_allclose(r0["flat"].numpy(), ref.numpy(), rtol=2e-4, atol=2e-6)
    assert abs(float(r0["loss"] + r1["loss"]) - float(full_loss)) < 1e-5


def _owner_worker(rank, world, port, emu_path, out_dir):
    for p in (os.path.dirname(HERE), HERE, os.path.join(os.path.dirname(HERE), "www24-rat_amd"), os.path.join(HERE, "emu")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import rat_amd._lib as L
    import sparse_cases as sc2
    from rat_amd import ops
    L._default = L.RatLib(emu_path)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _case, model, _batch = sc2._model("tiny_seq_bn", -1, "sparse", embedding_regularizer=0.0, batch_norm=False)
    total_rows, width, cap = 41, 4, 16
    # hand-made local lists (sorted, unique) that stress the partition: (0) every pair in owner 0's range, rank 1 holds nothing;
    # (1) the same rows on both ranks; (2) ragged, rows on both sides of the range edge, the last row of the table included
    lists = {0: ([0, 3, 7, 20], []), 1: ([2, 5, 25, 40], [2, 5, 25, 40]), 2: ([1, 19, 20, 21, 39], [0, 20, 40])}
    out = {}
    for key, per_rank in lists.items():
        mine = per_rank[rank]
        rows = torch.full((cap,), 12345, dtype=torch.int32)          # garbage behind `count`
        rows[:len(mine)] = torch.tensor(mine, dtype=torch.int32)
        g = torch.Generator().manual_seed(100 * key + rank)
        grads = torch.randn(cap, width, generator=g)
        count = torch.tensor([len(mine)], dtype=torch.int32)
        dense = {}
        for owner in (True, False):
            model.owner_exchange = owner
            merged = model._merge_sparse((rows, grads, count, width, total_rows, 0))
            d = torch.zeros(total_rows * width)
            model._scatter_merged(d, merged, width)
            dense[owner] = d
            n = sum(int(r[2]) for r in merged["records"])
            assert n == len(set(per_rank[0]) | set(per_rank[1])), (key, owner, n)
        assert torch.equal(dense[True], dense[False]), key           # same sums in the same (rank) order: bit-identical
        out[key] = dense[True]
    torch.save(out, os.path.join(out_dir, "owner%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_owner_partitioned_exchange_on_ragged_lists(emu_default):
    """the all-to-all form of the row-list exchange (round 4) against the all-gather form on hand-made lists: an owner that receives
    nothing, a rank that sends nothing, rows on the range edge, garbage behind `count` — both forms must give the same bits on both ranks"""
    import build_emu
    port = 31500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as out_dir:
        mp.spawn(_owner_worker, args=(2, port, build_emu.build(), out_dir), nprocs=2, join=True)
        r0 = torch.load(os.path.join(out_dir, "owner0.pt"))
        r1 = torch.load(os.path.join(out_dir, "owner1.pt"))
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k
        assert float(r0[k].abs().sum()) > 0
