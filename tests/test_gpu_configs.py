"""Parity at the shapes of EVERY single-GPU BASELINE.json config — run on the MI355X box with -m gpu.

    configs[0]  mltag_like_K10_d16_B256             (MovieLens-Tag plumbing case: the WHOLE batch against the oracle)
    configs[1]  synthetic_F20_V1M_K10_d64_B4096     (north star; forward / AUC checks live in test_gpu_fullsize.py)
    configs[2]  kkbox_like_F13_K10_d64_B4096        (fused <64,10> kernels at S = 14)
    configs[4]  tmall_like_F8_K30_d64_h32_B4096     (32 x 10 heads at d = 64: GROUPED mode, four 8-head launches of <64,10>, L = 31 / 9)
    + the reference's own KKBox geometry kkbox_real_F13_K5_d40_B4096 (embedding_dim 40: the bf16x3 kernels inside their 64-wide tiles)
    (configs[3] is the 8-GPU / 100 M-row config: its per-rank shape is covered by tests/test_sparse_grad.py)

Two kinds of checks per workload, both against the reference-pinned oracle (oracle/rat_m2_oracle.py) with FULL-SIZE parameters
(the real vocabularies, depth 4, DNN head):
  * forward at the full batch size: sampled rows of the full-batch prediction vs the oracle on those rows (eval mode; the model is
    per-sample independent there), 2e-6 — the tolerance of DESIGN.md §2 — plus AUC / logloss within 1e-4 on 512 rows;
  * BACKWARD: loss and EVERY parameter gradient of a slice of the batch against `orc.loss_and_grads` (BatchNorm off), the slice
    sized so that every attention / FFN work-group loops over >= 4 chunks — i.e. the persistent MFMA weight-gradient accumulators,
    the per-work-group slabs and their fixed-order reduction are compared with the oracle, not only with themselves.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

# (workload, samples of the gradient slice)
CASES = [
    ("mltag_like_K10_d16_B256", 256),
    ("synthetic_F20_V1M_K10_d64_B4096", 384),
    ("kkbox_like_F13_K10_d64_B4096", 384),
    ("tmall_like_F8_K30_d64_h32_B4096", 384),
    ("kkbox_real_F13_K5_d40_B4096", 768),
]
GRAD_RTOL = 3e-4          # gradients 3e-4 relative (to the tensor's largest element) — the ceiling; per workload below
# Per-workload gates = <= 5 x the worst error MEASURED on the MI355X box (tests/margins.py -> profiles/round6/r6_parity_margins.txt,
# table in DESIGN.md §2); a workload without an entry keeps the ceiling.
GRAD_GATES = {"mltag_like_K10_d16_B256": 1e-5,               # measured worst 1.9e-6 (f32)
              "synthetic_F20_V1M_K10_d64_B4096": 1e-5,       # 2.0e-6 (bf16x3), 1.8e-6 (f32)
              "kkbox_like_F13_K10_d64_B4096": 5e-6,          # 9.7e-7, 8.5e-7
              "tmall_like_F8_K30_d64_h32_B4096": 1e-5,       # 1.8e-6, 2.0e-6
              "kkbox_real_F13_K5_d40_B4096": 4.5e-5}         # 8.9e-6, 8.8e-6 (to_qkv.weight of the d = 40 layer: 40-wide rows, smaller maxima)
GRAD_GATES_BN = {"synthetic_F20_V1M_K10_d64_B4096": 7.5e-6,  # 1.5e-6
                 "tmall_like_F8_K30_d64_h32_B4096": 1e-5}    # 1.9e-6
LOSS_GATE = 1.5e-6                                           # absolute, x max(1, |loss|): the worst seen is 3.6e-7 on losses of 0.6-0.7


def _record(name, model, kind, loss, ref_loss, worst):
    import margins
    k, (err, scale) = max(worst.items(), key=lambda kv: kv[1][0])
    gates = GRAD_GATES_BN if kind.endswith("batchnorm") else GRAD_GATES
    margins.record("test_gpu_configs." + kind, name, "gradient, relative to the tensor's largest element", err, gates.get(name, GRAD_RTOL),
                   arith=model.arith, where=k)
    margins.record("test_gpu_configs." + kind, name, "loss, absolute", abs(float(loss) - float(ref_loss)),
                   LOSS_GATE * max(1.0, abs(float(ref_loss))), arith=model.arith)


def _oracle_cfg(orc, spec, fm, **over):
    kw = dict(fields=orc.fields_from_specs(fm.feature_specs), embedding_dim=spec["d"], num_heads=spec["num_heads"],
              dim_head=spec["dim_head"], depth=spec["depth"], scale_dim=spec["scale_dim"],
              dnn_hidden_units=tuple(spec["dnn_hidden_units"]), batch_norm=spec["batch_norm"], use_wide=spec["use_wide"])
    kw.update(over)
    return orc.Config(**kw)


def _build(name, **spec_over):
    from rat_amd import models, synthetic
    from rat_amd.base_model import seed_everything
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    spec = dict(synthetic.WORKLOADS[name], **spec_over)
    fm = synthetic.feature_map_for(name, spec)
    seed_everything(2021)
    model = models.RAT_m2(fm, **synthetic.model_kwargs(spec, gpu=0, embedding_regularizer=0.0))
    # the reference initialises tables with std 1e-4: scale them up so that attention is far from uniform and every path matters
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith("embedding_layer.") and p.shape[-1] == spec["d"]:
                p.mul_(3000.0)
    batch = synthetic.make_batch(spec, fm, seed=7)             # host float64 4-tuple, like the reference loader
    return spec, fm, model, batch


@pytest.fixture(scope="module", params=[c[0] for c in CASES if c[0] != "synthetic_F20_V1M_K10_d64_B4096"])
def fwd_setup(request):
    return (request.param,) + _build(request.param)


def _predict(model, batch):
    model.eval()
    with torch.no_grad():
        return model.forward(batch)["y_pred"].reshape(-1).double().cpu()


def test_full_batch_forward_matches_oracle(fwd_setup):
    from oracle import rat_m2_oracle as orc
    name, spec, fm, model, batch = fwd_setup
    full = _predict(model, batch)
    B = spec["batch"]
    assert full.shape[0] == B and bool(torch.isfinite(full).all())
    assert float(full.std()) > 1e-3, "degenerate predictions would make the comparison vacuous"
    rows = torch.arange(B) if B <= 256 else torch.arange(0, B, B // 48)[:48]
    w = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref = orc.forward(w, batch[0][rows], batch[1][rows], _oracle_cfg(orc, spec, fm), training=False).reshape(-1).double()
    np.testing.assert_allclose(full[rows].numpy(), ref.numpy(), rtol=0, atol=2e-6)
    model.check_id_errors()
    import margins
    margins.record("test_gpu_configs.full_batch_forward", name, "y_pred, absolute", float((full[rows] - ref).abs().max()), 2e-6, arith=model.arith)


def test_auc_and_logloss_match_the_oracle_within_1e4(fwd_setup):
    from oracle import rat_m2_oracle as orc
    from rat_amd.metrics import evaluate_metrics
    name, spec, fm, model, batch = fwd_setup
    B = spec["batch"]
    n = min(B, 512)
    rows = torch.arange(0, B, B // n)[:n]
    full = _predict(model, batch)
    w = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    with torch.no_grad():
        ref = orc.forward(w, batch[0][rows], batch[1][rows], _oracle_cfg(orc, spec, fm), training=False).reshape(-1).double().numpy()
    y_true = batch[1][rows, 0].numpy()
    assert 0 < y_true.sum() < len(y_true)
    mine = evaluate_metrics(y_true, full[rows].numpy(), ["AUC", "logloss"])
    want = evaluate_metrics(y_true, ref, ["AUC", "logloss"])
    assert abs(mine["AUC"] - want["AUC"]) < 1e-4 and abs(mine["logloss"] - want["logloss"]) < 1e-4, (mine, want)
    import margins
    for k in ("AUC", "logloss"):
        margins.record("test_gpu_configs.auc_and_logloss", name, k + ", absolute", abs(mine[k] - want[k]), 1e-4, arith=model.arith)


def _relu_margins(orc, w, X, y, cfg):
    """Per sample: the smallest |pre-activation| over every ReLU input of the DNN head, evaluated in float64 on the target sample's
    field embeddings; with BatchNorm the ReLU input is the NORMALISED value under the training-mode statistics of exactly these
    rows (deep.py:126-141)."""
    w64 = {k: v.double() for k, v in w.items() if k.startswith(("dnn.", "embedding_layer.", "label_embedding_layer."))}
    _, target_fields = orc.build_grid(X, y, w64, cfg)
    a = target_fields.reshape(target_fields.shape[0], -1)
    layers, out_pos = orc.dnn_layout(cfg)
    margin = torch.full((a.shape[0],), float("inf"), dtype=torch.float64)
    for lin, bn in layers:
        z = a @ w64["dnn.dnn.%d.weight" % lin].t() + w64["dnn.dnn.%d.bias" % lin]
        if bn is not None:
            mu = z.mean(dim=0)
            var = ((z - mu) ** 2).mean(dim=0)
            z = (z - mu) / torch.sqrt(var + cfg.bn_eps) * w64["dnn.dnn.%d.weight" % bn] + w64["dnn.dnn.%d.bias" % bn]
        margin = torch.minimum(margin, z.abs().min(dim=1).values)
        a = torch.relu(z)
    return margin


def _relu_safe_rows(orc, w, X, y, cfg, margin=1e-5):
    """Rows whose DNN pre-activations all stay `margin` away from 0 (float64 evaluation of the head on the target sample's field
    embeddings).  A hidden unit within rounding distance of 0 lands on either side of the ReLU depending on the GEMM's summation
    order; ONE such flip changes that sample's whole DNN-branch gradient (every row of the first weight matrix, its table rows) by
    far more than any tolerance — an artefact of the comparison, not of either implementation (the reference itself is not
    reproducible there), so those few samples are left out of the slice."""
    assert not cfg.batch_norm
    return _relu_margins(orc, w, X, y, cfg) > margin


def _relu_safe_slice_with_batchnorm(orc, w, X, y, cfg, nslice, margin=3e-6, tries=60):
    """BatchNorm couples the rows of a slice (the ReLU inputs are normalised with the slice's own statistics), so dropping an unsafe
    row moves every other row's pre-activations by O(1/n) — far more than the margin: "drop and re-check" is a fresh draw every
    time, not a converging iteration.  This therefore SEARCHES: start with rows [0, nslice), and while some row has a normalised
    pre-activation within `margin` of 0 (the fp32 GEMM + statistics error is ~1e-6 there) replace those rows by the next unused
    ones.  About one row in 400 is replaced per round and a round succeeds with probability ~1/3.  -> (row indices, rounds, rows
    replaced in total)"""
    n_all = X.shape[0]
    rows = torch.arange(nslice)
    nxt, replaced = nslice, 0
    for it in range(tries):
        bad = (_relu_margins(orc, w, X[rows], y[rows], cfg) <= margin).nonzero().reshape(-1)
        if bad.numel() == 0:
            return rows, it, replaced
        for b in bad.tolist():
            assert nxt < n_all, "ran out of replacement rows"
            rows[b] = nxt
            nxt += 1
            replaced += 1
    raise AssertionError("no ReLU-safe slice found in %d rounds" % tries)


def _chunks_per_group(n, T, S):
    """chunks per work-group (256 of them) of the fused attention backward, both phases: a chunk = floor(64 / L) sequences"""
    intra = -(-(n * T) // (64 // S)) / 256.0
    cross = -(-(n * S) // (64 // T)) / 256.0
    return intra, cross


def _grad_errors(model, sub, ref_grads, skip=()):
    """one training forward / backward of the slice -> ({parameter: (worst error / the reference tensor's largest element, that
    element)}, loss)"""
    model.train()
    model.optimizer.zero_grad()
    loss = model.get_total_loss(sub)
    loss.backward()
    torch.cuda.synchronize()
    model.check_id_errors()
    worst = {}
    for k, p in model.named_parameters():
        if k.startswith("query_proj"):
            assert p.grad is None
            continue
        if k in skip:
            continue
        got, ref = p.grad.detach().cpu().double(), ref_grads[k].double()
        scale = float(ref.abs().max())
        assert scale > 0, k
        worst[k] = (float((got - ref).abs().max()) / scale, scale)
    return worst, loss.detach()


@pytest.mark.parametrize("name,nslice", CASES, ids=[c[0] for c in CASES])
def test_gradients_of_a_full_size_slice_match_the_oracle(name, nslice):
    from oracle import rat_m2_oracle as orc
    spec, fm, model, batch = _build(name, batch_norm=False)
    T, S = spec["K"] + 1, spec["F"] + 1
    cfg = _oracle_cfg(orc, spec, fm, batch_norm=False, embedding_regularizer=0.0)
    w = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    take = min(spec["batch"], nslice + nslice // 4)                          # head-room for the rows dropped below
    safe = _relu_safe_rows(orc, w, batch[0][:take], batch[1][:take], cfg)
    keep = safe.nonzero().reshape(-1)[:nslice]
    print("%s: %d of the first %d samples are ReLU-unsafe and left out; slice = %d samples"
          % (name, int((~safe).sum()), take, keep.numel()))
    assert keep.numel() >= min(nslice, spec["batch"]) * 0.7, keep.numel()
    if nslice < spec["batch"]:
        intra, cross = _chunks_per_group(int(keep.numel()), T, S)
        assert intra >= 4 and cross >= 4, "slice too small: every work-group must loop over >= 4 chunks (%s)" % ((intra, cross),)
    sub = tuple(t[keep] for t in batch)
    ref_loss, _ref_pred, ref_grads, _ = orc.loss_and_grads(w, sub[0], sub[1], cfg, training=True)
    default = model.arith
    # the default arithmetic last: its result is the one asserted below; the other arithmetic the library offers for this geometry
    # (exact fp32 beside bf16x3) goes through the same comparison first, at the same gate
    for arith in [m for m in model.arith_modes() if m != default] + [default]:
        model.set_arith(arith)
        worst, loss = _grad_errors(model, sub, ref_grads)
        assert abs(float(loss) - float(ref_loss)) < LOSS_GATE * max(1.0, abs(float(ref_loss))), (arith, float(loss), float(ref_loss))
        if arith != default:
            _record(name, model, "full_size_slice", loss, ref_loss, worst)
            gate = GRAD_GATES.get(name, GRAD_RTOL)
            assert all(v[0] < gate for v in worst.values()), (arith, max(worst.items(), key=lambda kv: kv[1][0]))
    _record(name, model, "full_size_slice", loss, ref_loss, worst)
    gate = GRAD_GATES.get(name, GRAD_RTOL)
    bad = {k: v for k, v in worst.items() if not v[0] < gate}
    assert not bad, "gradients outside %g of their tensor's largest element: %s" % (gate, sorted(bad.items(), key=lambda kv: -kv[1][0])[:12])
    assert len(worst) >= 20


@pytest.mark.parametrize("name,nslice", [("synthetic_F20_V1M_K10_d64_B4096", 384), ("tmall_like_F8_K30_d64_h32_B4096", 384)],
                         ids=["synthetic_F20_V1M_K10_d64_B4096", "tmall_like_F8_K30_d64_h32_B4096"])
def test_gradients_of_a_full_size_slice_match_the_oracle_with_batchnorm(name, nslice):
    """The same comparison with `batch_norm=True` — the shipped setting (deep.py:128-132): training-mode statistics of the slice,
    BatchNorm backward through 400-wide (Tmall: 200 / 80) layers, dgamma / dbeta, and the running-statistics update.  Every
    parameter gradient is compared except the biases in front of a BatchNorm (true gradient 0, rounding noise on both sides)."""
    from oracle import rat_m2_oracle as orc
    import model_cases as mc
    spec, fm, model, batch = _build(name, batch_norm=True)
    T, S = spec["K"] + 1, spec["F"] + 1
    cfg = _oracle_cfg(orc, spec, fm, batch_norm=True, embedding_regularizer=0.0)
    w = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    keep, rounds, replaced = _relu_safe_slice_with_batchnorm(orc, w, batch[0], batch[1], cfg, nslice)
    print("%s (BatchNorm on): ReLU-safe slice of %d samples after %d round(s), %d sample(s) replaced" % (name, keep.numel(), rounds, replaced))
    intra, cross = _chunks_per_group(int(keep.numel()), T, S)
    assert intra >= 4 and cross >= 4
    sub = tuple(t[keep] for t in batch)
    model.train()
    model.optimizer.zero_grad()
    loss = model.get_total_loss(sub)
    loss.backward()
    torch.cuda.synchronize()
    model.check_id_errors()
    ref_loss, _ref_pred, ref_grads, bn_state = orc.loss_and_grads(w, sub[0], sub[1], cfg, training=True)
    assert abs(float(loss) - float(ref_loss)) < LOSS_GATE * max(1.0, abs(float(ref_loss))), (float(loss), float(ref_loss))
    noise = mc.noise_tensors(model)
    assert len(noise) == len(spec["dnn_hidden_units"])
    worst = {}
    for k, p in model.named_parameters():
        if k.startswith("query_proj"):
            assert p.grad is None
            continue
        if k in noise:
            continue
        got, ref = p.grad.detach().cpu().double(), ref_grads[k].double()
        scale = float(ref.abs().max())
        assert scale > 0, k
        worst[k] = (float((got - ref).abs().max()) / scale, scale)
    _record(name, model, "full_size_slice_batchnorm", loss, ref_loss, worst)
    gate = GRAD_GATES_BN.get(name, GRAD_RTOL)
    bad = {k: v for k, v in worst.items() if not v[0] < gate}
    assert not bad, "gradients outside %g of their tensor's largest element: %s" % (gate, sorted(bad.items(), key=lambda kv: -kv[1][0])[:12])
    assert sum(1 for k in worst if k.startswith("dnn.")) >= 2 * len(spec["dnn_hidden_units"]) + 2      # W, gamma, beta per layer + out
    sd = model.state_dict()
    for k, v in bn_state.items():                                   # running statistics after this one training forward
        if k.endswith("num_batches_tracked"):
            assert int(sd[k]) == int(v)
        else:
            np.testing.assert_allclose(sd[k].cpu().numpy(), v.numpy(), rtol=3e-5, atol=3e-6, err_msg=k)
