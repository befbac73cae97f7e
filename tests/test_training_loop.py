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
