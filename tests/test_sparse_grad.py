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
    _case, model, _batch = sc2._model("tiny_seq_bn", -1, "atomic", batch_norm=False)
    d = model._cfg["d"]
    assert d % 4 == 0 and model._cfg["use_wide"]
    rows_a_total, rows_b_total = model._n_feat // d, model._n_tab - model._n_feat
    per_a, per_b = -(-rows_a_total // world), -(-rows_b_total // world)
    last_a, last_b, cap = rows_a_total - 1, rows_b_total - 1, 16
    # hand-made local lists (sorted, unique) that stress the partition: (0) every pair in owner 0's range, rank 1 holds nothing;
    # (1) the same rows on both ranks, the last row of each family included; (2) ragged, rows on both sides of the range edge
    lists = {0: (([0, 1, per_a - 1], []), ([0, per_b - 1], [])),
             1: (([2, 5, per_a, last_a], [2, 5, per_a, last_a]), ([1, last_b], [1, last_b])),
             2: (([1, per_a - 1, per_a, per_a + 1, last_a], [0, per_a, last_a]), ([per_b - 1, per_b], [0, per_b, last_b]))}
    out = {}

    def make(mine, width, seed):
        rows = torch.full((cap,), 12345, dtype=torch.int32)          # garbage behind `count`
        rows[:len(mine)] = torch.tensor(mine, dtype=torch.int32)
        grads = torch.randn(cap, width, generator=torch.Generator().manual_seed(seed))
        return rows, grads, torch.tensor([len(mine)], dtype=torch.int32)
    n_grad = model._flat.numel()
    for key, (fam_a, fam_b) in lists.items():
        ra, ga, ca = make(fam_a[rank], d, 100 * key + rank)
        rb, gb, cb = make(fam_b[rank], 1, 100 * key + rank + 50)
        parts = [(ra, ga, ca, d, rows_a_total, 0), (rb, gb, cb, 1, rows_b_total, model._n_feat)]
        label = torch.randn(model._n_emb - model._n_tab, generator=torch.Generator().manual_seed(7 * key + rank))
        dense = {}
        for owner in (True, False):
            g = torch.zeros(n_grad)
            g[model._n_tab:model._n_emb] = label
            if owner:
                # what _owner_prepare publishes from the batch's plans, from the hand-made lists here
                cnt = torch.zeros(2 * world, dtype=torch.int32)
                for f, (mine, per) in enumerate(((fam_a[rank], per_a), (fam_b[rank], per_b))):
                    for r_ in mine:
                        cnt[f * world + min(r_ // per, world - 1)] += 1
                model._owner_publish(cnt, (None, None), (per_a, per_b))
                assert model._exchange_lists_owner(g, parts) is None
                stats = model._owner_stats
                assert stats["sent"] == len(fam_a[rank]) + len(fam_b[rank]) and stats["collectives"] == 3
            else:
                for part in parts:
                    rows, grads, count, width, _t, base = model._merge_sparse(part)
                    ops.scatter_rows(g[base:], rows, grads, count, width, lib=model._lib)
                    assert int(count) == len(set(lists[key][0 if width == d else 1][0]) | set(lists[key][0 if width == d else 1][1]))
                dist.all_reduce(g[model._n_tab:model._n_emb])
            dense[owner] = g
        assert torch.equal(dense[True], dense[False]), key           # same sums in the same (rank) order: bit-identical
        out[key] = dense[True]
    torch.save(out, os.path.join(out_dir, "owner%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_owner_partitioned_exchange_on_ragged_lists(emu_default):
    """the owner form of the row-list exchange (count matrix -> one packed all-to-all -> owner merge -> one packed all-gather ->
    rat_owner_scatter, both table families and the label table's partial gradient in the same buffers) against the all-gather form on
    hand-made lists: an owner that receives nothing, a rank that sends nothing, rows on the range edge, the last row of a family,
    garbage behind `count` — both forms must give the same bits on both ranks"""
    import build_emu
    port = 31500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as out_dir:
        mp.spawn(_owner_worker, args=(2, port, build_emu.build(), out_dir), nprocs=2, join=True)
        r0 = torch.load(os.path.join(out_dir, "owner0.pt"))
        r1 = torch.load(os.path.join(out_dir, "owner1.pt"))
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k
        assert float(r0[k].abs().sum()) > 0


def _owner_worker_random(rank, world, port, emu_path, out_dir):
    for p in (os.path.dirname(HERE), HERE, os.path.join(os.path.dirname(HERE), "www24-rat_amd"), os.path.join(HERE, "emu")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import rat_amd._lib as L
    import sparse_cases as sc2
    from rat_amd import ops
    L._default = L.RatLib(emu_path)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    _case, model, _batch = sc2._model("tiny_seq_bn", -1, "atomic", batch_norm=False)
    d = model._cfg["d"]
    rows_a_total, rows_b_total = model._n_feat // d, model._n_tab - model._n_feat
    per_a, per_b = -(-rows_a_total // world), -(-rows_b_total // world)
    assert rows_a_total % world != 0 or rows_b_total % world != 0        # the last owner's range is shorter than the others'
    out = {}
    for trial in range(3):
        rs = np.random.RandomState(1000 * trial + rank)
        mine_a = sorted(rs.choice(rows_a_total, size=rs.randint(0, min(12, rows_a_total)), replace=False).tolist())
        mine_b = sorted(rs.choice(rows_b_total, size=rs.randint(0, min(9, rows_b_total)), replace=False).tolist())
        cap_a, cap_b = 16, 12

        def make(mine, width, cap, seed):
            rows = torch.full((cap,), 777777, dtype=torch.int32)
            rows[:len(mine)] = torch.tensor(mine, dtype=torch.int32)
            return rows, torch.randn(cap, width, generator=torch.Generator().manual_seed(seed)), torch.tensor([len(mine)], dtype=torch.int32)
        ra, ga, ca = make(mine_a, d, cap_a, 31 * trial + rank)
        rb, gb, cb = make(mine_b, 1, cap_b, 57 * trial + rank)
        parts = [(ra, ga, ca, d, rows_a_total, 0), (rb, gb, cb, 1, rows_b_total, model._n_feat)]
        label = torch.randn(model._n_emb - model._n_tab, generator=torch.Generator().manual_seed(5 * trial + rank))
        dense = {}
        for owner in (True, False):
            g = torch.zeros(model._flat.numel())
            g[model._n_tab:model._n_emb] = label
            if owner:
                cnt = torch.zeros(2 * world, dtype=torch.int32)
                for f, (mine, per) in enumerate(((mine_a, per_a), (mine_b, per_b))):
                    for r_ in mine:
                        cnt[f * world + r_ // per] += 1
                model._owner_publish(cnt, (None, None), (per_a, per_b))
                model._exchange_lists_owner(g, parts)
            else:
                for part in parts:
                    rows, grads, count, width, _t, base = model._merge_sparse(part)
                    ops.scatter_rows(g[base:], rows, grads, count, width, lib=model._lib)
                # (an all-reduce over more than two ranks sums in an unspecified order; the owner form sums the label partials in
                # rank order — the reference is formed the same way)
                parts_l = [torch.empty_like(label) for _ in range(world)]
                dist.all_gather(parts_l, label)
                acc = torch.zeros_like(label)
                for t_ in parts_l:
                    acc = acc + t_
                g[model._n_tab:model._n_emb] = acc
            dense[owner] = g
        assert torch.equal(dense[True], dense[False]), trial
        out[trial] = dense[True]
    torch.save(out, os.path.join(out_dir, "rnd%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_owner_partitioned_exchange_four_ranks_random_lists(emu_default):
    """four ranks (a world that does not divide the row counts: the last owner's range is shorter), random ragged lists — empty ones
    included — of both table families: owner form == all-gather form bit for bit, identical on every rank"""
    import build_emu
    port = 23500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as out_dir:
        mp.spawn(_owner_worker_random, args=(4, port, build_emu.build(), out_dir), nprocs=4, join=True)
        res = [torch.load(os.path.join(out_dir, "rnd%d.pt" % r)) for r in range(4)]
    for k in res[0]:
        assert all(torch.equal(res[0][k], res[r][k]) for r in range(1, 4)), k
