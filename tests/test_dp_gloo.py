"""Data parallelism without a cluster: 2 ranks over gloo on CPU (kernels through the host-emulation build), each taking
half of a batch, must reproduce the single-process full-batch training step — with BatchNorm OFF (no cross-sample coupling
at all) and with BatchNorm ON, where the SyncBN exchange (rat_bn_local_stats -> all-gather -> rat_bn_relu_fwd_sync, and the
matching backward) makes the replicas normalise with the GLOBAL batch statistics, i.e. what the reference's single-device
BatchNorm1d sees (deep.py:128-132).  The case has embedding_regularizer = 0.01: the table gradients travel as
all-gathered (row ids, gradient rows) lists, the replica-identical lambda*W term is applied locally by the fused optimizer
(base_model.py:79-94)."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import twin

HERE = os.path.dirname(os.path.abspath(__file__))


def _setup_paths():
    root = os.path.dirname(HERE)
    for p in (root, HERE, os.path.join(root, "www24-rat_amd"), os.path.join(HERE, "emu")):
        if p not in sys.path:
            sys.path.insert(0, p)


def _make(case_name, batch_norm=False, rows=0):
    """rows > 0: the case's batch repeated / cut to that many samples (a world of 8 needs at least 8)"""
    import golden_cases as gc
    import model_cases as mc
    case = dict(gc.case_by_name(case_name))
    case["batch_norm"] = batch_norm
    model = mc.build_model(case, gpu=-1, seed=1)
    mc.load_weights(model, case)
    batch = mc.batch_of(case)
    if rows:
        reps = -(-rows // batch[0].shape[0])
        batch = tuple(torch.cat([t] * reps)[:rows] for t in batch)
    return case, model, batch


def _worker(rank, world, port, case_name, emu_path, out_dir, batch_norm=False, row_lists=True, owner=True, rows=0):
    _setup_paths()
    import rat_amd._lib as L
    L._default = L.RatLib(emu_path)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case, model, batch = _make(case_name, batch_norm, rows)
    per = batch[0].shape[0] // world
    shard = tuple(t[rank * per:(rank + 1) * per] for t in batch)
    model.train()
    if world == 1:
        model.dp_single_rank = True           # one rank through the data-parallel path (what tests/test_gpu_rccl.py does with RCCL)
        assert model._dp()
    merges, inner, inner_owner = [0], model._merge_sparse, model._exchange_lists_owner

    def counting_merge(part):                 # the all-gather form: one merge per table family
        merges[0] += 1
        return inner(part)

    def counting_owner(g, lists):             # the owner form: both families in one exchange
        merges[0] += len(lists)
        return inner_owner(g, lists)
    model._merge_sparse = counting_merge
    model._exchange_lists_owner = counting_owner
    model.row_list_exchange = row_lists       # (the traffic rule would pick the dense all-reduce for this toy vocabulary)
    model.owner_exchange = owner              # all-to-all to the rows' owners (default) / all-gather at capacity + merge of the union
    assert model._cfg["lam_emb"] > 0 and model._grad_mode == "atomic"
    for _ in range(2):
        loss = model.train_step(shard)
    buffers = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k}
    torch.save({"flat": model._flat.clone(), "loss": loss, "buffers": buffers, "merges": merges[0],
                "owner_stats": model.__dict__.get("_owner_stats")},
               os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("batch_norm,row_lists,owner,case_name", [
    twin(False, True, True, "tiny_seq_bn", id="bn_off"), pytest.param(True, True, True, "tiny_seq_bn", id="sync_bn"),
    twin(True, True, False, "tiny_seq_bn", id="sync_bn_gather_at_capacity"), pytest.param(True, False, True, "tiny_seq_bn", id="sync_bn_dense_tables"),
    # round 6: a VARIANT under data parallelism — RAT_m3 at the Tmall head geometry (16 heads of width 20 at d = 10: the one-launch head-group
    # kernels), SyncBN; table rows of 40 bytes cannot travel as row lists (the sort / reduce kernels need 16-byte rows): dense all-reduce
    pytest.param(True, False, True, "m3_tmall_real_heads", id="RAT_m3_tmall_heads_sync_bn")])
def test_two_rank_step_equals_full_batch_step(batch_norm, row_lists, owner, case_name):
    _setup_paths()
    import build_emu
    import rat_amd._lib as L
    emu_path = build_emu.build()
    old = L._default
    L._default = L.RatLib(emu_path)
    try:
        case, model, batch = _make(case_name, batch_norm)
        model.train()
        for _ in range(2):
            full_loss = model.train_step(batch)
        ref = model._flat.clone()
        # biases of a Linear that feeds BatchNorm have a TRUE gradient of 0: what reaches Adam is rounding noise, which Adam turns
        # into +-lr steps whose sign depends on the summation order (documented for the reference itself in
        # tests/test_oracle_golden.py::noise_gradient_tensors) — excluded from the replica-vs-full-batch comparison
        import model_cases as mc
        keep = torch.ones_like(ref, dtype=torch.bool)
        for name in (mc.noise_tensors(model) if batch_norm else ()):
            o = model._offsets[name]
            keep[o:o + model._params[name].numel()] = False
        ref_buffers = {k: v.clone() for k, v in model.state_dict().items() if "running_" in k}
    finally:
        L._default = old
    port = 29500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as out_dir:
        mp.spawn(_worker, args=(2, port, case_name, emu_path, out_dir, batch_norm, row_lists, owner), nprocs=2, join=True)
        r0 = torch.load(os.path.join(out_dir, "rank0.pt"))
        r1 = torch.load(os.path.join(out_dir, "rank1.pt"))
    assert torch.equal(r0["flat"], r1["flat"]), "replicas diverged"
    if row_lists and owner:                          # the owner-partitioned exchange ran: every pair went to exactly one owner
        s0, s1 = r0["owner_stats"], r1["owner_stats"]
        assert s0 is not None and s1 is not None and s0["sent"] + s1["sent"] == s0["received"] + s1["received"] > 0
        assert s0["collectives"] == 3             # one all-to-all + one all-gather behind the count matrix's all-gather
    else:
        assert r0["owner_stats"] is None
    # embedding_regularizer = 0.01 > 0 (dense semantics: lambda*W on every row) and yet the table gradients travelled as row lists
    # (SURVEY.md §8e C2): 2 steps x 2 table families (feature tables, LR tables) merged on every rank, no dense table all-reduce
    # — or, with row_lists off, as one dense all-reduce of the table block (what the traffic rule picks when every rank brings a full batch)
    assert r0["merges"] == r1["merges"] == (4 if row_lists else 0)
    np.testing.assert_allclose(r0["flat"][keep].numpy(), ref[keep].numpy(), rtol=2e-4, atol=2e-6)
    assert bool(ref_buffers) == batch_norm
    for k, v in ref_buffers.items():                 # running statistics: the global batch's, identical on both ranks
        assert torch.equal(r0["buffers"][k], r1["buffers"][k]), k
        # running_mean follows the (noise-stepped, see above) bias of the Linear in front: after step 1 that bias differs by up to
        # 2*lr between the two runs and shifts the batch mean of step 2 by as much (momentum 0.1 -> 2e-4); variances are unaffected
        atol = 3e-4 if k.endswith("running_mean") else 1e-6
        np.testing.assert_allclose(r0["buffers"][k].numpy(), v.numpy(), rtol=2e-5, atol=atol)
    # each rank reports (local BCE + reg)/world; their sum is the full-batch loss
    assert abs(float(r0["loss"] + r1["loss"]) - float(full_loss)) < 1e-5


@pytest.mark.parametrize("row_lists", [twin(True, id="row_lists"), twin(False, id="dense_tables")])
def test_one_rank_through_the_data_parallel_path_equals_the_plain_step(row_lists):
    """`dp_single_rank`: a process group of ONE rank issues every collective of the step (SyncBN all-gathers, the async dense-net
    all-reduce, the row-list all-gathers + merge or the dense table all-reduce) and must land where the plain step lands — the
    CPU twin of tests/test_gpu_rccl.py, which runs the same thing on RCCL."""
    _setup_paths()
    import build_emu
    import rat_amd._lib as L
    emu_path = build_emu.build()
    old = L._default
    L._default = L.RatLib(emu_path)
    try:
        case, model, batch = _make("tiny_seq_bn", True)
        assert not model._dp()
        model.train()
        for _ in range(2):
            full_loss = model.train_step(batch)
        ref = model._flat.clone()
        import model_cases as mc
        keep = torch.ones_like(ref, dtype=torch.bool)
        for name in mc.noise_tensors(model):
            o = model._offsets[name]
            keep[o:o + model._params[name].numel()] = False
    finally:
        L._default = old
    port = 27500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as out_dir:
        mp.spawn(_worker, args=(1, port, "tiny_seq_bn", emu_path, out_dir, True, row_lists), nprocs=1, join=True)
        r0 = torch.load(os.path.join(out_dir, "rank0.pt"))
    assert r0["merges"] == (4 if row_lists else 0)
    np.testing.assert_allclose(r0["flat"][keep].numpy(), ref[keep].numpy(), rtol=2e-4, atol=2e-6)
    assert abs(float(r0["loss"]) - float(full_loss)) < 1e-5


def test_eight_rank_step_equals_full_batch_step():
    """The world size the north star names: 8 ranks (gloo, host-emulated kernels), ONE sample each, SyncBN, the owner-partitioned
    exchange with eight owners (row ranges that do not divide evenly, owners that receive nothing) == the single-process step on the
    8-sample batch; all eight replicas bit-identical."""
    _setup_paths()
    import build_emu
    import rat_amd._lib as L
    emu_path = build_emu.build()
    old = L._default
    L._default = L.RatLib(emu_path)
    world = 8
    try:
        case, model, batch = _make("tiny_seq_bn", True, rows=world)
        model.train()
        for _ in range(2):
            full_loss = model.train_step(batch)
        ref = model._flat.clone()
        import model_cases as mc
        keep = torch.ones_like(ref, dtype=torch.bool)
        for name in mc.noise_tensors(model):
            o = model._offsets[name]
            keep[o:o + model._params[name].numel()] = False
    finally:
        L._default = old
    port = 21500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as out_dir:
        mp.spawn(_worker, args=(world, port, "tiny_seq_bn", emu_path, out_dir, True, True, True, world), nprocs=world, join=True)
        res = [torch.load(os.path.join(out_dir, "rank%d.pt" % r)) for r in range(world)]
    for r in res[1:]:
        assert torch.equal(res[0]["flat"], r["flat"]), "replicas diverged"
    stats = [r["owner_stats"] for r in res]
    assert all(s_ is not None and s_["collectives"] == 3 for s_ in stats)
    assert sum(s_["sent"] for s_ in stats) == sum(s_["received"] for s_ in stats) > 0
    np.testing.assert_allclose(res[0]["flat"][keep].numpy(), ref[keep].numpy(), rtol=2e-4, atol=2e-6)
    assert abs(sum(float(r["loss"]) for r in res) - float(full_loss)) < 1e-5
