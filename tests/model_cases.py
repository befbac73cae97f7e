"""Model-level parity checks shared by the CPU (emulation) and GPU test files."""
import os

import numpy as np
import torch

import golden_cases as gc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def build_model(case, gpu=-1, seed=None, **overrides):
    from rat_amd.base_model import seed_everything
    from rat_amd.features import FeatureMap
    from rat_amd import models
    fm = FeatureMap.from_specs(case["name"], gc.feature_specs(case))
    kw = gc.model_kwargs(case)
    kw["gpu"] = gpu
    kw["model_root"] = "/tmp/rat_amd_models/"
    kw.update(overrides)
    if seed is not None:
        seed_everything(seed)
    return getattr(models, case.get("model", "RAT_m2"))(fm, **kw)      # resolved by name like run_expid.py:75


def load_weights(model, case):
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    w = gc.make_weights(case, shapes)
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in w.items()})


def batch_of(case):
    X, y, rv, rl = gc.make_inputs(case)
    return (torch.from_numpy(X), torch.from_numpy(y), torch.from_numpy(rv), torch.from_numpy(rl))


def noise_tensors(model):
    """biases of a Linear feeding BatchNorm (true gradient 0; see tests/test_oracle_golden.py)."""
    return {"dnn.dnn.%d.bias" % lin for lin, bn, _ in model._dnn_layers if bn is not None}


def check_init(name, gpu):
    case = gc.case_by_name(name)
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    model = build_model(case, gpu=gpu, seed=case["init_seed"])
    sd = model.state_dict()
    assert int(gold["param_count"]) == model.count_parameters()
    keys = [k for k in sd if not k.startswith("query_proj")]
    assert sorted(keys) == sorted(k[len("init/"):].replace("#summary", "") for k in gold.files if k.startswith("init/"))
    for k in keys:
        gc.check_summary(gold, "init/" + k, sd[k].detach().cpu().numpy(), rtol=0, atol=0)


def check_eval(name, gpu):
    case = gc.case_by_name(name)
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    model = build_model(case, gpu=gpu, seed=1)
    load_weights(model, case)
    model.eval()
    with torch.no_grad():
        out = model.forward(batch_of(case))
    np.testing.assert_allclose(out["y_pred"].cpu().numpy(), gold["eval/y_pred"], rtol=0, atol=2e-6)
    np.testing.assert_array_equal(out["y_true"].cpu().numpy(), gold["eval/y_true"])
    if gpu >= 0:
        import margins
        margins.record("check_eval", name, "y_pred, absolute", float(np.abs(out["y_pred"].cpu().numpy() - gold["eval/y_pred"]).max()), 2e-6,
                       arith=model.arith)


def noise_step(case):
    """how far ONE optimizer step can move a parameter whose true gradient is zero (the biases in front of BatchNorm hold rounding
    noise): Adam / Adagrad normalise the step to +-lr; RMSprop's first steps divide by sqrt((1 - alpha) g^2) = |g| / 10 -> 10 lr; SGD
    moves by lr x noise (nothing)"""
    lr, opt = case.get("learning_rate", 1e-3), case.get("optimizer", "adam")
    return {"adam": lr, "Adagrad": lr, "RMSprop": 10.0 * lr, "SGD": lr * 1e-3}[opt]


def eval_after_atol(case):
    """eval predictions after two training steps with BatchNorm: the running mean has followed the noise-stepped biases (see noise_step);
    a regression head returns the raw logit (no sigmoid's <= 1/4 slope in front of the comparison)"""
    if not case["batch_norm"]:
        return 3e-6
    return 2e-3 * max(1.0, noise_step(case) / 1e-3) * (4.0 if case.get("task") == "regression" else 1.0)


# Gates on the MEASURED quantity: the worst error of a gradient tensor relative to the largest element the fixture
# stores for it, on the MI355X box (profiles/round6/r6_parity_margins.txt; <= 5 x the worst seen, rounded).  The elementwise comparison
# of check_summary (rtol 3e-4 + atol 3e-6 per element) stays as it was; this one is the tighter of the two wherever a tensor has
# elements of very different size.  Default: every case but the three below landed at or under 4.4e-6.
TRAINING_GRAD_GATE = 2.5e-5
TRAINING_GRAD_GATES = {"kkbox_shape": 6e-5,                  # 1.1e-5 (LayerNorm bias of the cross phase)
                       "m1_northstar_shape": 1e-4,           # 2.0e-5 (a table row)
                       "m0_northstar_shape": 3e-4}           # 1.4e-4: the output layer's bias, a 3-sample sum that cancels to ~1e-3 of its terms
LOSS_GATE = 2e-6                                             # x max(1, |loss|): losses of 5-9 (regularised tables) have ulps of 4.8e-7


def check_training(name, gpu):
    case = gc.case_by_name(name)
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    model = build_model(case, gpu=gpu, seed=1)
    load_weights(model, case)
    batch = batch_of(case)
    model.train()
    noise = noise_tensors(model)
    worst_grad, worst_post, worst_loss = (0.0, None), (0.0, None), 0.0
    for step in (1, 2):
        before = {k: v.detach().clone() for k, v in model.state_dict().items()}
        model.optimizer.zero_grad()
        loss = model.get_total_loss(batch)
        assert abs(float(loss.detach()) - float(gold["train%d/loss" % step])) < LOSS_GATE * max(1.0, abs(float(gold["train%d/loss" % step])))
        worst_loss = max(worst_loss, abs(float(loss.detach()) - float(gold["train%d/loss" % step])))
        loss.backward()
        n = 0
        for k, p in model.named_parameters():
            if k.startswith("query_proj"):
                assert p.grad is None
                continue
            # a bias feeding BatchNorm has the exact gradient 0: both sides hold cancellation noise of sum(dz) there, whose size
            # scales with |dz| (1/sqrt(var) of a 3-row batch can be large) — only bound it
            err = gc.check_summary(gold, "train%d/grad/%s" % (step, k), p.grad.detach().cpu().numpy(), rtol=3e-4,
                                   atol=3e-5 if k in noise else 3e-6)
            if k not in noise:
                rel = err / max(gc.summary_scale(gold, "train%d/grad/%s" % (step, k)), 1e-30)
                worst_grad = max(worst_grad, (rel, "step %d %s" % (step, k)), key=lambda t: t[0])
            n += 1
        assert n == sum(1 for k in gold.files if k.startswith("train%d/grad/" % step))
        norm_sq = model.optimizer.clip_and_step(10.0)
        gn = float(torch.sqrt(norm_sq)[0])
        # fp32 on both sides: on m1_northstar_shape the reference's own norm is 7e-6 (relative) away from the float64 oracle's
        assert abs(gn - float(gold["train%d/gnorm" % step])) < 3e-5 * max(1.0, gn)
        for k, v in model.state_dict().items():
            if k.startswith("query_proj"):
                assert torch.equal(v, before[k])
                continue
            if k in noise:
                assert float((v - before[k]).abs().max()) <= 1.0001 * noise_step(case)
                continue
            atol = 3e-6 if not k.endswith("running_mean") else 3e-6 + step * 0.1 * noise_step(case) * 1.01
            err = gc.check_summary(gold, "train%d/post/%s" % (step, k), v.detach().cpu().numpy(), rtol=3e-4, atol=atol)
            if not k.endswith("running_mean"):                # (running_mean carries the noise tensors' bound, see atol above)
                rel = err / max(gc.summary_scale(gold, "train%d/post/%s" % (step, k)), 1e-30)
                worst_post = max(worst_post, (rel, "step %d %s" % (step, k)), key=lambda t: t[0])
    if gpu >= 0:
        import margins
        gate = TRAINING_GRAD_GATES.get(name, TRAINING_GRAD_GATE)
        margins.record("check_training", name, "gradient, relative to the fixture's largest stored element", worst_grad[0], gate, arith=model.arith,
                       where=worst_grad[1])
        margins.record("check_training", name, "post-step weight, relative", worst_post[0], 3e-4, arith=model.arith, where=worst_post[1])
        margins.record("check_training", name, "loss, absolute", worst_loss, LOSS_GATE * max(1.0, abs(float(gold["train2/loss"]))), arith=model.arith)
        assert worst_grad[0] < gate, (name, worst_grad, gate)
    model.eval()
    with torch.no_grad():
        yp = model.forward(batch)["y_pred"].cpu().numpy()
    np.testing.assert_allclose(yp, gold["eval_after/y_pred"], rtol=0, atol=eval_after_atol(case))


def check_wide_heads_group_loop(gpu, batch=3, topk=2, nfields=2, heads=16, depth=1, dropout=0.0):
    """Wide heads at embedding_dim 64 (heads = G x 8): the forward of a layer as ONE launch looping over the head groups
    (rat_attn_fwd_groups) against G launches of the 8-head kernel, on the same weights and batch: loss, predictions and every gradient
    agree to rounding (only the order in which the groups' partial output projections are summed differs), the loss equals the oracle's,
    and the one-launch form is what actually ran."""
    from oracle import rat_m2_oracle as orc
    from rat_amd import ops
    case = dict(gc.case_by_name("northstar_shape"), name="wide_heads_probe", batch=batch, topk=topk, num_heads=heads, depth=depth,
                fields=gc.case_by_name("northstar_shape")["fields"][:nfields], batch_norm=False, embedding_regularizer=0.0)
    out = {}
    calls = []
    real = ops.attn_fwd_groups
    ops.attn_fwd_groups = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        for loop in (True, False):
            model = build_model(case, gpu=gpu, seed=1, dropout=dropout)
            load_weights(model, case)
            assert model.arith == "bf16x3" and model._attn_mode(ops.intra_map(batch, topk + 1, nfields + 1)) == ("grouped", 8)
            model.group_loop = loop
            model.train()
            if dropout:
                model._dropout_state_init(base=4242)                 # the same masks in both forms
            model.optimizer.zero_grad()
            n0 = len(calls)
            tb = tuple(t.to(model.device) for t in batch_of(case))
            loss = model.get_total_loss(tb)
            loss.backward()
            assert (len(calls) - n0 == 2 * depth) if loop else (len(calls) == n0)
            grads = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}
            model.eval()
            with torch.no_grad():
                yp = model.forward(tb)["y_pred"].cpu()
            assert (len(calls) - n0 == 4 * depth) if loop else (len(calls) == n0)         # the inference forward takes it too
            out[loop] = (float(loss.detach()), grads, yp)
    finally:
        ops.attn_fwd_groups = real
    (l1, g1, y1), (l0, g0, y0) = out[True], out[False]
    assert abs(l1 - l0) < 2e-6 and float((y1 - y0).abs().max()) < 2e-6
    assert sorted(g1) == sorted(g0) and len(g1) >= 20
    for k in g1:
        scale = float(g0[k].abs().max()) + 1e-30
        assert float((g1[k] - g0[k]).abs().max()) / scale < 1e-4, k
    if dropout == 0:
        model = build_model(case, gpu=gpu, seed=1)
        load_weights(model, case)
        w = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
        cfg = orc.Config(fields=orc.fields_from_specs(gc.feature_specs(case)), embedding_dim=64, num_heads=heads, dim_head=10, depth=depth,
                         scale_dim=case["scale_dim"], dnn_hidden_units=tuple(case["dnn_hidden_units"]), batch_norm=False,
                         use_wide=case["use_wide"])
        b = batch_of(case)
        ref_loss = orc.loss_and_grads(w, b[0], b[1], cfg, training=True)[0]
        assert abs(l1 - float(ref_loss)) < 2e-6, (l1, float(ref_loss))


CHECKPOINT_CASES = ["tiny_seq_bn", "m0_tiny_seq", "m1_tiny_seq", "m3_tiny_seq"]


def check_variant_against_oracle(gpu, variant, base="northstar_shape", gate=1e-4, **over):
    """a model VARIANT on a golden case's fields / seeds with some hyper-parameters changed (no fixture exists for the combination): loss,
    predictions and every gradient of one training forward / backward against the reference-pinned oracle run on the same weights and batch.
    -> (model, worst gradient error relative to the tensor's largest element)"""
    from oracle import rat_m2_oracle as orc
    case = dict(gc.case_by_name(base), name="variant_probe", model=variant, batch_norm=False, embedding_regularizer=0.0)
    case.update(over)
    model = build_model(case, gpu=gpu, seed=1)
    load_weights(model, case)
    batch = batch_of(case)
    w = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    cfg = orc.Config(fields=orc.fields_from_specs(gc.feature_specs(case)), embedding_dim=case["embedding_dim"], num_heads=case["num_heads"],
                     dim_head=case["dim_head"], depth=case["depth"], scale_dim=case["scale_dim"],
                     dnn_hidden_units=tuple(case["dnn_hidden_units"]), batch_norm=False, use_wide=case["use_wide"], embedding_regularizer=0.0,
                     variant={"RAT_m2": "m2", "RAT_m1": "m1", "RAT_m3": "m3", "RAT_m0": "m0"}[variant])
    ref_loss, ref_pred, ref_grads, _ = orc.loss_and_grads(w, batch[0], batch[1], cfg, training=True)
    model.train()
    model.optimizer.zero_grad()
    loss = model.get_total_loss(batch)
    loss.backward()
    assert abs(float(loss.detach()) - float(ref_loss)) < 2e-6 * max(1.0, abs(float(ref_loss))), (float(loss.detach()), float(ref_loss))
    worst = (0.0, None)
    for k, p in model.named_parameters():
        if k.startswith("query_proj"):
            continue
        ref = ref_grads[k].double()
        err = float((p.grad.detach().cpu().double() - ref).abs().max()) / (float(ref.abs().max()) + 1e-30)
        worst = max(worst, (err, k), key=lambda t: t[0])
    assert worst[0] < gate, worst
    model.eval()
    with torch.no_grad():
        yp = model.forward(batch)["y_pred"].reshape(-1).cpu()
        want = orc.forward(w, batch[0], batch[1], cfg, training=False).reshape(-1)
    assert float((yp - want).abs().max()) < 2e-6
    return model, worst[0]


def check_checkpoint(name, gpu, tmpdir):
    """On-disk format of `.model` files (base_model.py:275-284): tests/golden/ckpt_<case>.model was written by the REFERENCE class's
    own save_weights after two training steps (tests/golden/make_golden_checkpoints.py).  load_weights must take it as is and the
    eval predictions must be the reference's under those weights (`eval_after/y_pred`, 2e-6 — no training on our side, so BatchNorm's
    running statistics are the file's); save_weights must write a file with the same keys, shapes, dtypes and bits."""
    case = gc.case_by_name(name)
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    path = os.path.join(GOLD, "ckpt_%s.model" % name)
    model = build_model(case, gpu=gpu, seed=99)
    model.load_weights(path)
    model.eval()
    with torch.no_grad():
        yp = model.forward(batch_of(case))["y_pred"].cpu().numpy()
    np.testing.assert_allclose(yp, gold["eval_after/y_pred"], rtol=0, atol=2e-6)
    ref = torch.load(path, map_location="cpu")
    out = os.path.join(str(tmpdir), "roundtrip.model")
    model.save_weights(out)
    mine = torch.load(out, map_location="cpu")
    assert list(mine.keys()) == list(ref.keys()), "state_dict keys / order differ from the reference's file"
    for k in ref:
        assert mine[k].dtype == ref[k].dtype and mine[k].shape == ref[k].shape, k
        assert torch.equal(mine[k], ref[k]), k
    # and the reference-side reader: torch.load + load_state_dict(strict) into a fresh model of ours round-trips the weights
    again = build_model(case, gpu=gpu, seed=5)
    again.load_weights(out)
    for k, v in again.state_dict().items():
        assert torch.equal(v.cpu(), ref[k]), k


def check_train_step_api(name, gpu, steps=2, **model_kw):
    """BaseModel.train_step — the fused iteration (no autograd node, regulariser folded into the two-sweep optimizer, on a GPU
    replayed as a hipGraph from the third call of a batch shape on) — against the reference's golden training run: the loss of
    every step and the weights after it (`train<k>/post/*` = after the reference's zero_grad / backward / clip / Adam.step).
    steps > 2: the extra steps are compared with the literal sequence (train_step_reference_order) on a second model."""
    case = gc.case_by_name(name)
    gold = np.load(os.path.join(GOLD, name + ".npz"))
    model = build_model(case, gpu=gpu, seed=1, **model_kw)
    load_weights(model, case)
    batch = batch_of(case)
    model.train()
    noise = noise_tensors(model)
    twin = None
    if steps > 2:
        twin = build_model(case, gpu=gpu, seed=1, **model_kw)
        load_weights(twin, case)
        twin.train()
        twin.fused_step = False
    for step in range(1, steps + 1):
        before = {k: v.detach().clone() for k, v in model.state_dict().items()}
        loss = float(model.train_step(batch))
        if twin is not None:
            ref_loss = float(twin.train_step(batch))
        if step <= 2:
            assert abs(loss - float(gold["train%d/loss" % step])) < 2e-6, (step, loss, float(gold["train%d/loss" % step]))
        else:
            assert abs(loss - ref_loss) < 5e-6, (step, loss, ref_loss)
        for k, v in model.state_dict().items():
            if k.startswith("query_proj"):
                assert torch.equal(v, before[k])
                continue
            if k in noise:
                assert float((v - before[k]).abs().max()) <= 1.0001 * noise_step(case)
                continue
            atol = 3e-6 if not k.endswith("running_mean") else 3e-6 + step * 0.1 * noise_step(case) * 1.01
            if step <= 2:
                gc.check_summary(gold, "train%d/post/%s" % (step, k), v.detach().cpu().numpy(), rtol=3e-4, atol=atol)
            else:
                # Adam steps of rounding-level gradients are sign-like: bound the outliers instead of loosening everything
                a, r = v.detach().cpu().double(), twin.state_dict()[k].detach().cpu().double()
                bad = (a - r).abs() > atol * step + 3e-4 * r.abs()
                assert float(bad.double().mean()) < 1e-3 and float((a - r).abs().max()) <= 2.1e-3 * step, (k, step)
    assert all(p.grad is None for p in model.parameters()), "the fused step leaves no p.grad behind (zero_grad semantics)"
    return model


def check_pruning_equivalence(name, gpu):
    """RAT_m2.prune_dead_tokens (the last block computes only what the class token depends on) against the full computation on the
    same weights and batch: the predictions must be the same numbers (per-sequence / per-token arithmetic does not depend on which
    other rows are computed), the loss too, and every gradient must agree to summation-order rounding — the skipped rows contribute
    exact zeros."""
    case = gc.case_by_name(name)
    out = {}
    for prune in (True, False):
        model = build_model(case, gpu=gpu, seed=1)
        model.prune_dead_tokens = prune
        load_weights(model, case)
        batch = batch_of(case)
        model.eval()
        with torch.no_grad():
            yp = model.forward(batch)["y_pred"].detach().cpu().clone()
        model.train()
        model.optimizer.zero_grad()
        loss = model.get_total_loss(batch)
        loss.backward()
        grads = {k: p.grad.detach().cpu().clone() for k, p in model.named_parameters() if p.grad is not None}
        out[prune] = (yp, float(loss), grads)
    (ya, la, ga), (yb, lb, gb) = out[True], out[False]
    assert torch.equal(ya, yb), "predictions differ with dead-token pruning"
    assert abs(la - lb) < 5e-7 * max(1.0, abs(lb))         # (the loss sums go through atomics: a few ulps of an O(1) number run to run)
    assert set(ga) == set(gb)
    for k in ga:
        scale = float(gb[k].abs().max()) + 1e-30
        assert float((ga[k] - gb[k]).abs().max()) <= 2e-6 * scale + 1e-9, (k, float((ga[k] - gb[k]).abs().max()), scale)
