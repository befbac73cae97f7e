"""A training RUN on the MI355X against the oracle (-m gpu): north_star's "AUC / Logloss matching the reference within 1e-4 on identical
inputs" checked over a trajectory, not one step — WITH A CONTROL.  The datasets are not available offline, so the
data are the structured synthetic split of rat_amd.data.synthetic_split (labels depend on the ids through a logistic rule, neighbours
share the first column's id: the model has something to learn), 80 different batches, 2048 held-out samples, at two geometries
(GEOMETRIES below): the MovieLens-Tag shape of BASELINE.json configs[0] (F = 3, K = 10, d = 16, 2 heads x 10, depth 4, scale 4, DNN
400^3, B = 256) and — round 5 — a d = 64, 8 x 10 heads geometry on which the bf16x3 ENCODER kernels run (the headline's arithmetic).

Five trajectories from the same initial weights on the same batches:
    oracle fp64            the reference's arithmetic in double precision: the "truth" of this run
    oracle fp32            the same in fp32 (torch on the host's cores) — what the reference itself computes
    oracle fp32, 1 thread  the same with another summation order inside torch's kernels: a second fp32 sample
    HIP exact fp32         train_step() with arith="f32" (fp32 MFMA everywhere), fused iteration, hipGraph replays from step 3
    HIP bf16x3             the product's default arithmetic (arith="auto": at d = 16 the DNN head's GEMMs run bf16x3 and the encoder
                           the exact-fp32 kernels; at d = 64 the encoder's projections / FFN run bf16x3 as well)
Per-step losses of two implementations that agree to rounding on every single step (tests/test_gpu_model.py) still separate over a run:
Adam's early steps are sign-like, so a gradient element that rounds to the other side of zero moves its weight by 2 lr.  The control
turns that sentence into a measurement: the fp32 oracle separates from the fp64 oracle at the same rate.  Measured on the MI355X box
(round 4; tools/experiments/trajectory_spread.py repeats every run): per-step loss against fp64 — fp32 oracle rms 2.7-3.2e-4, worst
0.9-1.3e-3; HIP rms / worst no larger; held-out predictions against fp64 — fp32 oracle rms 3.4-3.7e-3, max 1.4-1.8e-2; HIP 2.0-3.6e-3 /
1.0-1.3e-2.  The two scalars north_star names are single numbers with sign cancellations inside: the EIGHT fp32 oracle runs alone
(thread counts 1 .. 16: each another summation order) span |d AUC| 3e-6 .. 2.6e-4 and |d logloss| 4e-5 .. 2.1e-4 around the fp64 run,
repeated HIP runs (the fp32 atomics reorder) |d AUC| 3e-5 .. 4.7e-4 and |d logloss| 5e-5 .. 5.0e-4 — the same order of magnitude, the
HIP runs on average twice as far.  The test therefore requires
  * the STABLE statistics (rms and max over 2048 predictions, rms and worst over 80 losses) of both HIP arithmetics to stay within
    1.5 x the farthest fp32 oracle run, and
  * |d AUC| and |d logloss| within 3 x the farthest of the eight fp32 oracle runs (floor 5e-5, half of north_star's 1e-4)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
CONTROL_THREADS = (1, 2, 3, 4, 6, 8, 12, 16)

# Two geometries.  "d16": BASELINE.json configs[0]'s — the encoder runs the exact-fp32 kernels there whatever
# `arith` says, bf16x3 only reaches the head's GEMMs.  "d64_bf16x3": the geometry that SELECTS the bf16x3 encoder kernels, the ones that
# produce the headline number (embedding_dim 64, 8 heads x 10, hidden 128), small enough around them (F = 5, K = 4, depth 2, B = 128,
# DNN 64-32) for the fp64 oracle to follow 80 steps in a few minutes; the run asserts that those kernels were the ones launched.
GEOMETRIES = {
    "d16": dict(workload="mltag_like_K10_d16_B256", spec={}, encoder_bf16x3=False),
    "d64_bf16x3": dict(workload="mltag_like_K10_d16_B256",
                       spec=dict(F=5, K=4, d=64, batch=128, num_heads=8, dim_head=10, depth=2, scale_dim=2, dnn_hidden_units=[64, 32]),
                       encoder_bf16x3=True),
}


@pytest.mark.parametrize("geometry", list(GEOMETRIES))
def test_eighty_training_steps_end_as_close_to_fp64_as_the_fp32_reference_does(geometry):
    from collections import OrderedDict
    from oracle import rat_m2_oracle as orc
    from rat_amd import data as rd
    from rat_amd import models, synthetic
    from rat_amd.base_model import seed_everything
    from rat_amd.features import FeatureMap
    from rat_amd.metrics import evaluate_metrics
    assert torch.cuda.is_available()
    threads = min(16, torch.get_num_threads())                     # (the oracle's small ops thrash on a many-core host)
    torch.set_num_threads(threads)
    geo = GEOMETRIES[geometry]
    spec = dict(synthetic.WORKLOADS[geo["workload"]], **geo["spec"])
    # 40 ids per field instead of 30 000: every id recurs often enough for 80 steps to learn the rule (held-out AUC ~0.68 from 0.5)
    fm = FeatureMap.from_specs("trajectory", OrderedDict(
        ("c%02d" % i, {"source": "", "type": "categorical", "vocab_size": 40, "index": i}) for i in range(spec["F"])))
    B, K, steps = spec["batch"], spec["K"], 80
    n = B * steps + 2048
    data, idx, val, lens = rd.synthetic_split(fm, n, K, seed=4)
    batches = list(rd.RetrievalBatches(data, data, idx, val, lens, B, shuffle=False))
    train, held = batches[:steps], batches[steps:]
    yt = torch.cat([b[1][:, 0].double() for b in held]).numpy()

    def build(arith):
        seed_everything(2021)
        m = models.RAT_m2(fm, **dict(synthetic.model_kwargs(spec, gpu=0), arith=arith))
        with torch.no_grad():                  # the reference's init std of 1e-4 makes attention uniform for the first hundreds of steps
            m._flat[:m._n_feat].mul_(2000.0)
        return m

    runs = {}                                   # name -> (losses, held-out predictions)
    w0 = None
    for arith in ("auto", "f32"):
        model = build(arith)
        if w0 is None:
            w0 = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
            cfg = orc.Config(fields=orc.fields_from_specs(fm.feature_specs), embedding_dim=spec["d"], num_heads=spec["num_heads"],
                             dim_head=spec["dim_head"], depth=spec["depth"], scale_dim=spec["scale_dim"],
                             dnn_hidden_units=tuple(spec["dnn_hidden_units"]), batch_norm=spec["batch_norm"], use_wide=spec["use_wide"],
                             embedding_regularizer=model._cfg["lam_emb"], learning_rate=spec["learning_rate"])
            assert cfg.embedding_regularizer > 0 and not cfg.batch_norm
        else:
            for k, v in model.state_dict().items():
                assert torch.equal(v.detach().cpu(), w0[k]), k
        # which arithmetic the encoder kernels really ran: the bf16x3 instantiations exist for this geometry or they do not
        assert (model.arith == "bf16x3") == (geo["encoder_bf16x3"] and arith == "auto"), (geometry, arith, model.arith)
        # (RAT_m2.arith is what every encoder launch is given; "bf16x3" is only offered when the library has those instantiations
        # for this geometry: arith_modes() asks rat_attn_fwd_workspace)
        assert ("bf16x3" in model.arith_modes()) == geo["encoder_bf16x3"]
        model.train()
        losses = [float(model.train_step(b)) for b in train]
        assert any(e[1] for e in model._step_graphs.values()), "steps 3.. were hipGraph replays"
        model.check_id_errors()
        model.eval()
        with torch.no_grad():
            yp = torch.cat([model.forward(b)["y_pred"].reshape(-1).double().cpu() for b in held]).numpy()
        runs["HIP " + ("bf16x3" if arith == "auto" else "f32")] = (losses, yp)
        del model

    def oracle_run(dtype, nthreads):
        torch.set_num_threads(nthreads)
        w = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in w0.items()}
        state, losses = {}, []
        for s, b in enumerate(train):
            w, loss, *_ = orc.train_step(w, b[0].double(), b[1].double(), cfg, state, s + 1)
            losses.append(float(loss))
        with torch.no_grad():
            yp = torch.cat([orc.forward(w, b[0].double(), b[1].double(), cfg, training=False).reshape(-1).double() for b in held]).numpy()
        torch.set_num_threads(threads)
        return losses, yp

    runs["oracle fp64"] = oracle_run(torch.float64, threads)
    controls = []
    for nt in CONTROL_THREADS:                  # torch splits its reductions by thread count: each count is another fp32 rounding pattern
        controls.append("oracle fp32, %d thread%s" % (nt, "" if nt == 1 else "s"))
        runs[controls[-1]] = oracle_run(torch.float32, nt)

    truth_losses, truth_pred = runs["oracle fp64"]
    truth = evaluate_metrics(yt, truth_pred, ["AUC", "logloss"])
    assert truth["AUC"] > 0.6, "the synthetic rule must have been learnt, otherwise the comparison is vacuous"
    table = {}
    for k, (losses, yp) in runs.items():
        m = evaluate_metrics(yt, yp, ["AUC", "logloss"])
        dl = [abs(a - b) for a, b in zip(losses, truth_losses)]
        dp = yp - truth_pred
        table[k] = dict(auc=m["AUC"], logloss=m["logloss"], d_auc=abs(m["AUC"] - truth["AUC"]), d_ll=abs(m["logloss"] - truth["logloss"]),
                        first5=max(dl[:5]), worst=max(dl), rms_loss=float(np.sqrt(np.mean(np.square(dl)))),
                        pred=float(np.abs(dp).max()), rms_pred=float(np.sqrt(np.mean(dp * dp))))
        print("%-24s held-out AUC %.6f (|d| %.2e) logloss %.6f (|d| %.2e) | per-step loss vs fp64: first 5 %.1e, rms %.1e, worst %.1e | "
              "held-out y_pred vs fp64: rms %.2e, max %.2e" % (k, m["AUC"], table[k]["d_auc"], m["logloss"], table[k]["d_ll"], table[k]["first5"],
                                                               table[k]["rms_loss"], table[k]["worst"], table[k]["rms_pred"], table[k]["pred"]))
    ctrl = {q: max(table[k][q] for k in controls) for q in ("d_auc", "d_ll", "worst", "rms_loss", "rms_pred", "pred")}
    print("control (max over the fp32 oracle runs):", " ".join("%s %.2e" % kv for kv in ctrl.items()))
    for k in ("HIP bf16x3", "HIP f32"):
        t = table[k]
        assert t["first5"] < 2e-5, (k, t)                          # the first steps agree to the single-step tolerance
        # the distance between two trajectories, measured where it is a stable statistic: 2048 predictions and 80 losses
        assert t["rms_pred"] <= 1.5 * ctrl["rms_pred"], (k, t["rms_pred"], ctrl["rms_pred"])
        assert t["pred"] <= 1.5 * ctrl["pred"], (k, t["pred"], ctrl["pred"])
        assert t["rms_loss"] <= 1.5 * ctrl["rms_loss"] and t["worst"] <= 1.5 * ctrl["worst"], (k, t, ctrl)
        # ... and the two scalars north_star names.  Each is ONE number with sign cancellations inside (the fp32 oracle runs alone
        # span two orders of magnitude in |d AUC|): 3 x the farthest control
        assert t["d_auc"] <= max(3.0 * ctrl["d_auc"], 5e-5), (k, t["d_auc"], ctrl["d_auc"])
        assert t["d_ll"] <= max(3.0 * ctrl["d_ll"], 5e-5), (k, t["d_ll"], ctrl["d_ll"])
