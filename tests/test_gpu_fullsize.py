"""Parity at BASELINE.json's FULL north-star configuration (configs[1]: F=20, 1M-row vocab, K=10, d=64, B=4096, depth 4) — run on
the MI355X box with -m gpu.  The oracle needs minutes for 4096 samples, so the full batch is checked through
size-independent properties plus a direct oracle comparison on a slice of the batch (the model is per-sample independent in
eval mode, so a slice of the full-size prediction must equal the oracle on that slice)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

NAME = "synthetic_F20_V1M_K10_d64_B4096"


VARIANT = {"RAT_m2": "m2", "RAT_m1": "m1", "RAT_m3": "m3", "RAT_m0": "m0"}


@pytest.fixture(scope="module", params=["RAT_m2", "RAT_m1", "RAT_m3", "RAT_m0"])
def setup(request):
    from rat_amd import models, synthetic
    from rat_amd.base_model import seed_everything
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    spec = synthetic.WORKLOADS[NAME]
    fm = synthetic.feature_map_for(NAME, spec)
    seed_everything(2021)
    model = getattr(models, request.param)(fm, **synthetic.model_kwargs(spec, gpu=0))
    # the reference initialises tables with std 1e-4: scale them up so that attention is far from uniform and every path matters
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith("embedding_layer.") and p.shape[-1] == spec["d"]:      # feature tables only (the label table is N(0,1) already)
                p.mul_(3000.0)
    batch = synthetic.make_batch(spec, fm, seed=7)             # host float64 4-tuple, like the reference loader
    return spec, fm, model, batch, request.param


def _predict(model, batch):
    model.eval()
    with torch.no_grad():
        return model.forward(batch)["y_pred"].reshape(-1).double().cpu()


def test_slice_of_full_batch_matches_oracle(setup):
    from oracle import rat_m2_oracle as orc
    spec, fm, model, batch, which = setup
    full = _predict(model, batch)
    assert full.shape[0] == spec["batch"] and bool(torch.isfinite(full).all())
    assert float(full.std()) > 1e-3, "degenerate predictions would make the comparison vacuous"
    rows = torch.arange(0, spec["batch"], spec["batch"] // 48)[:48]
    cfg = orc.Config(fields=orc.fields_from_specs(fm.feature_specs), embedding_dim=spec["d"], num_heads=spec["num_heads"],
                     dim_head=spec["dim_head"], depth=spec["depth"], scale_dim=spec["scale_dim"],
                     dnn_hidden_units=tuple(spec["dnn_hidden_units"]), batch_norm=spec["batch_norm"], use_wide=spec["use_wide"],
                     variant=VARIANT[which])
    w = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    ref = orc.forward(w, batch[0][rows], batch[1][rows], cfg, training=False)
    ref = (ref[0] if isinstance(ref, (tuple, list)) else ref).reshape(-1).double()
    np.testing.assert_allclose(full[rows].numpy(), ref.numpy(), rtol=0, atol=2e-6)      # tolerance of DESIGN.md §2 (y_pred 2e-6 abs)
    import margins
    margins.record("test_gpu_fullsize.slice_of_full_batch", "synthetic_F20_V1M_K10_d64_B4096/" + which, "y_pred, absolute",
                   float((full[rows] - ref).abs().max()), 2e-6, arith=model.arith)


def test_prediction_is_per_sample_and_chunking_independent(setup):
    """eval-mode predictions of a sub-batch equal the same rows of the full batch (different chunking / tile occupancy)"""
    spec, fm, model, batch, which = setup
    full = _predict(model, batch)
    for lo, n in ((0, 1), (5, 37), (1000, 1027)):
        sub = tuple(t[lo:lo + n] for t in batch)
        np.testing.assert_allclose(_predict(model, sub).numpy(), full[lo:lo + n].numpy(), rtol=0, atol=1e-6)


def test_prediction_invariant_to_retrieved_order(setup):
    """the K retrieved samples form a set: permuting them (with their labels) must not change the target's prediction"""
    spec, fm, model, batch, which = setup
    X, y, rv, rl = batch
    perm = torch.cat([torch.zeros(1, dtype=torch.long), 1 + torch.randperm(spec["K"], generator=torch.Generator().manual_seed(3))])
    a = _predict(model, batch)
    b = _predict(model, (X[:, perm], y[:, perm], rv, rl))
    np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=0, atol=2e-6)


def test_forward_is_deterministic(setup):
    spec, fm, model, batch, which = setup
    assert torch.equal(_predict(model, batch), _predict(model, batch))


def test_full_batch_gradient_is_mean_of_half_batch_gradients(setup):
    """loss = mean BCE (+ batch-independent L2): with BatchNorm in eval mode the gradient of the full batch is the average of
    the two half-batch gradients — checks the whole backward path (atomics, slabs, split-K) at the full size."""
    from rat_amd import models, synthetic
    from rat_amd.base_model import seed_everything
    spec, fm, _, batch, which = setup
    spec2 = dict(spec, batch_norm=False)
    seed_everything(2021)
    model = getattr(models, which)(fm, **synthetic.model_kwargs(spec2, gpu=0, embedding_regularizer=0.0))
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith("embedding_layer.") and p.shape[-1] == spec["d"]:
                p.mul_(3000.0)
    model.train()

    def grads(b):
        model.optimizer.zero_grad()
        model.get_total_loss(b).backward()
        torch.cuda.synchronize()
        return {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}

    h = spec["batch"] // 2
    g_full = grads(batch)
    g_a = grads(tuple(t[:h] for t in batch))
    g_b = grads(tuple(t[h:] for t in batch))
    for n, g in g_full.items():
        avg = 0.5 * (g_a[n] + g_b[n])
        scale = float(g.abs().max()) + 1e-12
        err = (g - avg).abs() / scale
        # fp32 summation order differs (atomics, slabs, split-K): gradient tolerance of DESIGN.md §2 is 3e-4 rel.  A handful of
        # elements may exceed it legitimately: the head GEMMs pick a different k-split for M = 2048 than for M = 4096, a hidden
        # unit sitting within an ulp of 0 then lands on the other side of the ReLU and that ONE sample's DNN-branch gradient
        # row changes (measured: ~5 of 4.9 M units) — bounded here by count and size instead of being hidden by a loose tolerance
        assert float((err > 5e-4).float().mean()) < 1e-4, (n, float(err.max()))
        assert float(err.max()) < 2e-2, (n, float(err.max()))


def test_auc_and_logloss_match_the_oracle_within_1e4(setup):
    """north_star: "AUC/Logloss matching the reference within 1e-4 on identical inputs" — metrics of the HIP path's predictions
    against the (reference-pinned) oracle's on 512 samples of the full-size batch, through the product's own metric code
    (rat_amd.metrics = fuxictr/metrics.py:20-36: sklearn roc_auc_score, log_loss with eps 1e-7)."""
    from oracle import rat_m2_oracle as orc
    from rat_amd.metrics import evaluate_metrics
    spec, fm, model, batch, which = setup
    rows = torch.arange(0, spec["batch"], spec["batch"] // 512)[:512]
    full = _predict(model, batch)
    cfg = orc.Config(fields=orc.fields_from_specs(fm.feature_specs), embedding_dim=spec["d"], num_heads=spec["num_heads"],
                     dim_head=spec["dim_head"], depth=spec["depth"], scale_dim=spec["scale_dim"],
                     dnn_hidden_units=tuple(spec["dnn_hidden_units"]), batch_norm=spec["batch_norm"], use_wide=spec["use_wide"],
                     variant=VARIANT[which])
    w = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref = orc.forward(w, batch[0][rows], batch[1][rows], cfg, training=False).reshape(-1).double().numpy()
    y_true = batch[1][rows, 0].numpy()
    assert 0 < y_true.sum() < len(y_true)
    mine = evaluate_metrics(y_true, full[rows].numpy(), ["AUC", "logloss"])
    want = evaluate_metrics(y_true, ref, ["AUC", "logloss"])
    assert abs(mine["AUC"] - want["AUC"]) < 1e-4 and abs(mine["logloss"] - want["logloss"]) < 1e-4, (mine, want)
    import margins
    for k in ("AUC", "logloss"):
        margins.record("test_gpu_fullsize.auc_and_logloss", "synthetic_F20_V1M_K10_d64_B4096/" + which, k + ", absolute", abs(mine[k] - want[k]),
                       1e-4, arith=model.arith)


def test_dead_token_pruning_at_the_full_size(setup):
    """BASELINE.json configs[1] at its full size, whole batch of 4096: the product's default step (the last block computes only what
    x[:, 0][:, 0] depends on — RAT_m2.prune_dead_tokens, also RAT_m3) against the same model with every token of every block computed:
    bit-equal eval predictions, the same training loss, every gradient tensor within summation-order rounding of its largest element."""
    spec, fm, model, batch, which = setup
    if which not in ("RAT_m2", "RAT_m3"):
        pytest.skip("the cascaded / joint variants are not pruned")
    assert model.prune_dead_tokens is True, "pruning is the default"
    preds = {}
    for prune in (True, False):                      # (both before any training-mode pass: that one moves BatchNorm's running statistics)
        model.prune_dead_tokens = prune
        preds[prune] = _predict(model, batch)
    assert torch.equal(preds[True], preds[False]), "predictions differ with dead-token pruning"
    res = {}
    for prune in (True, False):
        model.prune_dead_tokens = prune
        model.train()
        model.optimizer.zero_grad()
        loss = model.get_total_loss(batch)
        loss.backward()
        torch.cuda.synchronize()
        res[prune] = (float(loss), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
        model.optimizer.zero_grad()
    model.prune_dead_tokens = True
    (la, ga), (lb, gb) = res[True], res[False]
    assert abs(la - lb) < 3e-6 * max(1.0, abs(lb))        # (the regulariser's sum of 65 M squares goes through atomics: ~10 ulps run to run)
    assert set(ga) == set(gb)
    worst = 0.0
    for k in ga:
        scale = float(gb[k].abs().max()) + 1e-30
        err = float((ga[k] - gb[k]).abs().max()) / scale
        worst = max(worst, err)
        assert err <= 4e-6, (k, err)
    print("%s at B = %d: pruned vs full — predictions bit-equal, worst gradient difference %.1e of the tensor's largest element over %d tensors"
          % (which, spec["batch"], worst, len(ga)))
