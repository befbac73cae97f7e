"""Host logic of the training loop that the reference's driver relies on (fuxictr/pytorch/models/base_model.py:144-211):
validation cadence, early stopping / lr decay, checkpoint writes under data parallelism, and the out-of-vocabulary id check
that replaces nn.Embedding's IndexError.  Kernels run through the host emulation (tests/emu); no GPU."""
import logging
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


@pytest.fixture()
def emu_model():
    _setup_paths()
    import build_emu
    import golden_cases as gc
    import model_cases as mc
    import rat_amd._lib as L
    old = L._default
    L._default = L.RatLib(build_emu.build())
    try:
        case = gc.case_by_name("tiny_seq_bn")
        model = mc.build_model(case, gpu=-1, seed=1)
        yield case, model, mc.batch_of(case)
    finally:
        L._default = old


def test_out_of_vocabulary_ids_raise_like_nn_embedding(emu_model):
    case, model, batch = emu_model
    model.eval()
    with torch.no_grad():
        model.forward(batch)
    model.check_id_errors()                                  # clean batch: nothing to report
    X = batch[0].clone()
    X[0, 1, 0] = 7.0                                          # field "a" has vocab 7: ids 0..6
    with torch.no_grad():
        model.forward((X,) + tuple(batch[1:]))
    with pytest.raises(IndexError, match="1 feature id"):
        model.check_id_errors()
    model.check_id_errors()                                  # counters were reset by the report
    y = batch[1].clone()
    y[2, 1] = 3.0                                             # a retrieved label outside {0, 1}
    with torch.no_grad():
        model.forward((batch[0], y) + tuple(batch[2:]))
    with pytest.raises(IndexError, match="1 label id"):
        model.check_id_errors()


def test_label_wise_retrieval_is_rejected_on_the_device_path():
    _setup_paths()
    from rat_amd.data import RetrievalBatches
    rs = np.random.RandomState(0)
    n, L, K = 9, 3, 2
    data = np.concatenate([rs.randint(0, 5, size=(n, L)), rs.randint(0, 2, size=(n, 1))], axis=1).astype(np.float64)
    src = RetrievalBatches(data, data, rs.randint(0, n, size=(n, 2 * K)), rs.rand(n, 2 * K), rs.randint(0, K, size=(n, 2)), batch_size=4)
    with pytest.raises(AssertionError, match="label-wise"):
        src.to_device("cpu")


def test_validation_cadence_early_stop_and_lr_decay(emu_model, caplog):
    """checkpoint_and_earlystop (base_model.py:160-179): improvement resets the counter and saves; a non-improving value decays
    the lr and, after `patience` of them, stops — with the reference's log lines."""
    case, model, batch = emu_model
    with tempfile.TemporaryDirectory() as d:
        model.checkpoint = os.path.join(d, "m.model")
        model._best_metric, model._stopping_steps, model._stop_training = -np.inf, 0, False
        model._every_x_epochs, model._patience = 1, 2
        lr0 = model.optimizer.param_groups[0]["lr"]
        with caplog.at_level(logging.INFO):
            model.checkpoint_and_earlystop(1, {"AUC": 0.7})
            assert os.path.exists(model.checkpoint) and model._best_metric == 0.7 and model._stopping_steps == 0
            model.checkpoint_and_earlystop(2, {"AUC": 0.7 + 5e-7})          # within min_delta: not an improvement
            assert model._stopping_steps == 1 and not model._stop_training
            assert abs(model.optimizer.param_groups[0]["lr"] - 0.1 * lr0) < 1e-12
            model.checkpoint_and_earlystop(3, {"AUC": 0.6})
            assert model._stopping_steps == 2 and model._stop_training
        text = caplog.text
        for line in ("Save best model: monitor(max): 0.700000", "Monitor(max) STOP: 0.600000 !", "Reduce learning rate on plateau: 0.000010",
                     "Early stopping at epoch=3"):
            assert line in text, line
        model._batches_per_epoch, model._every_x_batches = 10, 5
        assert [b for b in range(10) if model._validation_due(b)] == [4, 9]


def _rank_worker(rank, world, port, emu_path, out_dir):
    _setup_paths()
    import golden_cases as gc
    import model_cases as mc
    import rat_amd._lib as L
    L._default = L.RatLib(emu_path)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    model = mc.build_model(gc.case_by_name("tiny_seq_bn"), gpu=-1, seed=1)
    model.checkpoint = os.path.join(out_dir, "shared.model")
    model._best_metric, model._stopping_steps, model._stop_training = -np.inf, 0, False
    model._every_x_epochs, model._patience = 1, 1
    with torch.no_grad():
        model.fc.bias.fill_(float(rank + 1))                  # replicas differ on purpose: the file must hold rank 0's state
    # the ranks see DIFFERENT validation values (different shards): rank 0's decides for everybody
    model.checkpoint_and_earlystop(1, {"AUC": 0.8 if rank == 0 else 0.1})
    first = (model._best_metric, model._stopping_steps)
    model.checkpoint_and_earlystop(2, {"AUC": 0.5 if rank == 0 else 0.99})
    torch.save({"first": first, "second": (model._best_metric, model._stopping_steps, model._stop_training,
                                            model.optimizer.param_groups[0]["lr"])}, os.path.join(out_dir, "r%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_ranks_agree_on_early_stopping_and_only_rank0_writes():
    _setup_paths()
    import build_emu
    emu_path = build_emu.build()
    port = 31500 + (os.getpid() % 2000)
    with tempfile.TemporaryDirectory() as out_dir:
        mp.spawn(_rank_worker, args=(2, port, emu_path, out_dir), nprocs=2, join=True)
        r0, r1 = torch.load(os.path.join(out_dir, "r0.pt")), torch.load(os.path.join(out_dir, "r1.pt"))
        assert r0 == r1
        assert r0["first"] == (0.8, 0) and r0["second"][:3] == (0.8, 1, True) and abs(r0["second"][3] - 1e-4) < 1e-12
        state = torch.load(os.path.join(out_dir, "shared.model"))
        assert float(state["fc.bias"][0]) == 1.0              # rank 0's replica


def test_a_failed_iteration_takes_its_clock_tick_back(emu_model):
    """rat_step_begin advances the optimizer's clock and BatchNorm's num_batches_tracked before forward / backward run; an
    exception in between must not leave them one ahead with no update applied."""
    case, model, batch = emu_model
    model.train()
    model.train_step(batch)
    opt = model.optimizer
    step, counts = opt._step, model._bn_counts.clone()
    flat = model._flat.clone()
    inner = model._run_backward

    def boom(*a, **k):
        raise RuntimeError("injected")
    model._run_backward = boom
    with pytest.raises(RuntimeError, match="injected"):
        model.train_step(batch)
    model._run_backward = inner
    assert opt._step == step and torch.equal(model._bn_counts, counts) and torch.equal(model._flat, flat)
    model.train_step(batch)                                   # and the next good step is step + 1 on host and device
    assert opt._step == step + 1 and int(opt._step_dev[0]) == step + 1
    assert torch.equal(model._bn_counts, counts + 1)


def test_last_grad_norm_follows_the_last_step_whichever_form(emu_model):
    case, model, batch = emu_model
    model.train()
    model.train_step(batch)                                   # fused form
    fused = model.optimizer.last_grad_norm()
    assert fused is not None and fused > 0
    model.optimizer.zero_grad()
    (model.get_total_loss(batch) * 3.0).backward()            # autograd form, a different gradient
    model.optimizer.clip_and_step(10.0)
    plain = model.optimizer.last_grad_norm()
    assert plain is not None and abs(plain - fused) > 1e-3 * fused
    model.optimizer.zero_grad()
    model.get_total_loss(batch).backward()
    model.optimizer.clip_and_step(None)                       # a step without clipping computes no norm
    assert model.optimizer.last_grad_norm() is None


def test_sparse_rows_with_a_non_adam_optimizer_is_refused_at_construction():
    _setup_paths()
    import build_emu
    import golden_cases as gc
    import model_cases as mc
    import rat_amd._lib as L
    old = L._default
    L._default = L.RatLib(build_emu.build())
    try:
        case = dict(gc.case_by_name("tiny_seq_bn"), embedding_regularizer=0.0, optimizer="SGD")
        with pytest.raises(NotImplementedError, match="Adam only"):
            mc.build_model(case, gpu=-1, seed=1, embedding_grad="sparse")
        mc.build_model(dict(case, optimizer="adam"), gpu=-1, seed=1, embedding_grad="sparse")
    finally:
        L._default = old


def test_dropout_generator_state_travels_with_the_optimizer_state():
    """the device-side mask generator's (base seed, counter) is resume state — a restored run continues the mask sequence"""
    _setup_paths()
    import build_emu
    import golden_cases as gc
    import model_cases as mc
    import rat_amd._lib as L
    old = L._default
    L._default = L.RatLib(build_emu.build())
    try:
        case = gc.case_by_name("tiny_seq_bn")
        # (embedding_grad="sorted": the bit-reproducible table gradients — with fp32 atomics two runs of the same step may differ in the
        #  last bit, on the emulator depending on how the OS schedules its threads; this test compares two runs bit for bit)
        a = mc.build_model(case, gpu=-1, seed=1, emb_dropout=0.2, embedding_grad="sorted")
        batch = mc.batch_of(case)
        a.train()
        assert a.dropout_state() is None
        for _ in range(2):
            a.train_step(batch)
        st = a.dropout_state()
        assert st["counter"] == 2
        sd = a.optimizer.state_dict()
        assert sd["rat_dropout"] == st
        b = mc.build_model(case, gpu=-1, seed=1, emb_dropout=0.2, embedding_grad="sorted")
        b.load_state_dict(a.state_dict())
        b.optimizer.load_state_dict({k: (v.clone() if torch.is_tensor(v) else v) for k, v in sd.items()})
        b.train()
        la, lb = a.train_step(batch), b.train_step(batch)     # third step of both: same weights, same moments, same masks
        assert float(la) == float(lb) and torch.equal(a._flat, b._flat)
        assert b.dropout_state() == {"base": st["base"], "counter": 3}
    finally:
        L._default = old


def test_waiting_ranks_give_up_when_rank0_fails_or_takes_too_long(tmp_path, monkeypatch):
    """the pollers of run_expid's retrieval pre-computation must not spin forever"""
    _setup_paths()
    import run_expid
    monkeypatch.setattr(dist, "barrier", lambda *a, **k: None)
    target, marker = str(tmp_path / "retrieval_3_train.npz"), str(tmp_path / "retrieval_3_train.npz.failed")
    with pytest.raises(SystemExit, match="gave up waiting"):
        run_expid._wait_for_file(target, leader=False, poll_s=0.01, failed_marker=marker, max_wait_s=0.05)
    open(marker, "w").write("MemoryError: top-K\n")
    with pytest.raises(SystemExit, match="MemoryError: top-K"):
        run_expid._wait_for_file(target, leader=False, poll_s=0.01, failed_marker=marker, max_wait_s=5)
    # a marker older than this process is an EARLIER job's: it must not end this one
    os.utime(marker, (run_expid._PROCESS_START - 3600, run_expid._PROCESS_START - 3600))
    with pytest.raises(SystemExit, match="gave up waiting"):
        run_expid._wait_for_file(target, leader=False, poll_s=0.01, failed_marker=marker, max_wait_s=0.05)
    open(target, "w").write("x")
    run_expid._wait_for_file(target, leader=False, poll_s=0.01, failed_marker=marker, max_wait_s=5)       # the file wins
    run_expid._wait_for_file(target, leader=True)


def _eval_worker(rank, world, port, emu_path, out_dir):
    _setup_paths()
    import golden_cases as gc
    import model_cases as mc
    import rat_amd._lib as L
    from rat_amd import data as rd
    L._default = L.RatLib(emu_path)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    case = gc.case_by_name("tiny_seq_bn")
    model = mc.build_model(case, gpu=-1, seed=1)
    mc.load_weights(model, case)
    fm_rows = 37                                               # 37 rows in batches of 8: ragged shards (8 = 3 + 3 + 2) and a 5-row tail (2 + 2 + 1)
    X, y, rv, rl = mc.batch_of(case)
    reps = -(-fm_rows // X.shape[0])
    ids = torch.cat([X[:, 0, :]] * reps)[:fm_rows].numpy()
    labels = (torch.arange(fm_rows) % 2).double().numpy()[:, None]
    data = np.concatenate([ids, labels], axis=1)
    K = X.shape[1] - 1
    retr = np.stack([(np.arange(fm_rows) + 1 + k) % fm_rows for k in range(K)], axis=1)
    gen = rd.RetrievalBatches(data, data, retr, np.zeros((fm_rows, K)), np.full(fm_rows, K), 8)
    seen = []
    inner = model.forward

    def counting(batch):
        seen.append(int(batch[0].shape[0]))
        return inner(batch)
    model.forward = counting
    sharded = model.evaluate_generator(gen)
    n_sharded = sum(seen)
    assert gen.shard == (0, 1) and not gen.keep_all            # the source is handed back as it came
    model.shard_evaluation = False
    seen.clear()
    full = model.evaluate_generator(gen)
    torch.save({"sharded": sharded, "full": full, "rows_sharded": n_sharded, "rows_full": sum(seen)}, os.path.join(out_dir, "e%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_evaluation_is_sharded_by_rank_and_every_rank_gets_the_global_metrics():
    """evaluate_generator (base_model.py:232-247) under data parallelism — each rank forwards its slice of every
    batch (ragged, nothing dropped), the predictions are all-gathered, every rank computes the same metrics as an unsharded pass"""
    _setup_paths()
    import build_emu
    emu_path = build_emu.build()
    port = 25500 + (os.getpid() % 2000)
    world = 3
    with tempfile.TemporaryDirectory() as out_dir:
        mp.spawn(_eval_worker, args=(world, port, emu_path, out_dir), nprocs=world, join=True)
        res = [torch.load(os.path.join(out_dir, "e%d.pt" % r)) for r in range(world)]
    assert sum(r["rows_sharded"] for r in res) == 37 and all(r["rows_full"] == 37 for r in res)
    assert max(r["rows_sharded"] for r in res) <= 15           # 4 x ceil(8 / 3) + ceil(5 / 3): a third of the work, not all of it
    for r in res:
        assert r["sharded"].keys() == res[0]["full"].keys()
        for k in r["sharded"]:
            assert abs(r["sharded"][k] - res[0]["full"][k]) < 1e-9, (k, r["sharded"], res[0]["full"])
