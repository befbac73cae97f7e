"""Row-sparse / deterministic embedding-gradient path on the MI355X (-m gpu): rocPRIM's device radix sort + scan under the plan,
the hand-written segmented reduction, the row optimizer — against CPU autograd and against the dense (atomic) path."""
import numpy as np
import pytest
import torch

import sparse_cases as sc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    from rat_amd._lib import get_lib
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return get_lib()


@pytest.mark.parametrize("d", [8, 64])
@pytest.mark.parametrize("B,T", [(5, 4), (257, 11)])
def test_sorted_segmented_reduce(lib, d, B, T):
    sc.check_sorted_reduce(lib, "cuda", d, B=B, T=T)


def test_scalar_reduce_for_the_wide_tables(lib):
    sc.check_scalar_reduce(lib, "cuda")
    sc.check_scalar_reduce(lib, "cuda", B=700, T=5)


def test_merge_of_gathered_row_lists_and_row_adam(lib):
    sc.check_merge_rows(lib, "cuda")


@pytest.mark.parametrize("name", ["tiny_seq_bn", "kkbox_shape", "northstar_shape"])
def test_sorted_mode_equals_atomic_mode_and_is_reproducible(lib, name):
    if sc.gc.case_by_name(name)["embedding_dim"] % 4:
        pytest.skip("sorted / sparse modes need embedding_dim % 4 == 0")
    sc.check_model_sorted_equals_atomic(name, gpu=0)


def test_out_of_vocabulary_ids_are_treated_alike_by_every_gradient_path(lib):
    sc.check_model_sorted_equals_atomic("tiny_seq_bn", gpu=0, bad_ids=True)


@pytest.mark.parametrize("name", ["tiny_seq_bn", "northstar_shape"])
def test_sparse_training_equals_dense_training(lib, name):
    sc.check_model_sparse_training(name, gpu=0, steps=3)


def _run(workload, mode, steps, batch_rows):
    from rat_amd import models, synthetic
    from rat_amd.base_model import seed_everything
    spec = dict(synthetic.WORKLOADS[workload])
    fm = synthetic.feature_map_for(workload, spec)
    seed_everything(2021)
    model = models.RAT_m2(fm, **synthetic.model_kwargs(spec, gpu=0, embedding_regularizer=0.0), embedding_grad=mode)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.startswith("embedding_layer.") and p.shape[-1] == spec["d"]:
                p.mul_(3000.0)
    batch = tuple(t[:batch_rows] for t in synthetic.make_batch(spec, fm, seed=3))
    model.train()
    losses = [float(model.train_step(batch)) for _ in range(steps)]
    torch.cuda.synchronize()
    model.check_id_errors()
    return model, losses


def test_sparse_equals_dense_at_the_config3_rank_shape(lib):
    """BASELINE.json configs[3] per-rank shape (F = 40, B = 1024) at a tenth of its vocabulary: three steps on the same batch —
    sparse row lists + lazy row Adam against the dense atomic path + dense Adam (identical when the same rows recur)."""
    name = "synthetic_F40_V10M_K10_d64_B1024"
    dense, ld = _run(name, "atomic", 3, 1024)
    flat_d = dense._flat.detach().cpu()
    offs, order, nt = dict(dense._offsets), list(dense._order), dense._n_tab
    del dense
    torch.cuda.empty_cache()
    sparse, ls = _run(name, "sparse", 3, 1024)
    assert sparse._grad_mode == "sparse"
    flat_s = sparse._flat.detach().cpu()
    # (the dense path sums with fp32 atomics in arrival order: over repeated runs the step-3 losses differed by 1.7e-6 .. 3.4e-6)
    np.testing.assert_allclose(ls, ld, rtol=0, atol=1e-5)
    noise = sc.mc.noise_tensors(sparse)
    keep = torch.ones_like(flat_s, dtype=torch.bool)
    for n in noise:
        keep[offs[n]:offs[n] + sparse._params[n].numel()] = False
    diff = (flat_s - flat_d).abs()
    tol = 2e-6 + 2e-4 * flat_d.abs()
    bad = (diff > tol) & keep
    # Adam's first steps are sign-like (+-lr per element): an element whose gradient is rounding-level zero may step the other way;
    # bound their number and size instead of loosening the tolerance for everything
    assert float(bad.float().mean()) < 1e-5 and float(diff[keep].max()) <= 2.1e-3, (int(bad.sum()), float(diff[keep].max()))
    assert order == list(sparse._order) and nt == sparse._n_tab


def test_config3_at_its_real_vocabulary_forward_and_one_sparse_training_step():
    """BASELINE.json configs[3] per rank at its REAL size: F = 40 fields x 2.5 M rows = 100 M rows x 64 floats = 25.6 GB of feature
    tables (+ LR tables, + 51.2 GB of Adam moments) on one MI355X, B = 1024, row-sparse gradients + lazy row Adam.

    The oracle never sees 100 M rows: a batch names at most B*T ids per field, so each field's vocabulary is COMPACTED to the ids
    the batch uses (row r of the compact table = row uniq[r] of the device table, ids renumbered) — exactly the same arithmetic
    (embedding.py:158-178 reads nothing else).  Checked against it: eval-mode predictions of the full batch (2e-6), the training
    loss (2e-6), the (unique row, gradient row) lists of both table families against `orc.loss_and_grads`, every dense gradient,
    the touched rows after clip + Adam against `orc.adam_step`, and that NO other row of the 100 M changed."""
    import sys
    from oracle import rat_m2_oracle as orc
    from rat_amd import models, synthetic
    from rat_amd.base_model import seed_everything
    from rat_amd.features import FeatureMap
    name = "synthetic_F40_V100M_K10_d64_B1024"
    spec = dict(synthetic.WORKLOADS[name])
    fm = synthetic.feature_map_for(name, spec)
    seed_everything(2021)
    model = models.RAT_m2(fm, **synthetic.model_kwargs(spec, gpu=0))
    assert model._grad_mode == "sparse" and model._n_feat == 100_000_000 * 64
    d, F, B = spec["d"], spec["F"], spec["batch"]
    with torch.no_grad():
        model._flat[:model._n_feat].mul_(3000.0)          # init std 1e-4 -> 0.3: attention far from uniform
    X, y, rv, rl = synthetic.make_batch(spec, fm, seed=3)
    # ---- compact vocabulary for the oracle
    emb = "embedding_layer.embedding_layer.embedding_layer.%s.weight"
    lr = "lr_layer.embedding_layer.embedding_layer.embedding_layer.%s.weight"
    Xc = torch.empty_like(X)
    uniq, cspecs = {}, {}
    for i, (fname, fs) in enumerate(fm.feature_specs.items()):
        u, inv = torch.unique(X[:, :, i].long(), return_inverse=True)
        uniq[fname] = u
        Xc[:, :, i] = inv.to(X.dtype)
        cspecs[fname] = dict(fs, vocab_size=int(u.numel()))
    cfm = FeatureMap.from_specs(name + "_compact", cspecs)

    def snapshot():
        w = {}
        for k, v in model.state_dict().items():
            owner = next((f for f in uniq if k in (emb % f, lr % f)), None)
            w[k] = (v[uniq[owner].to(v.device)] if owner else v).detach().cpu().clone()
        return w
    w0 = snapshot()
    cfg = orc.Config(fields=orc.fields_from_specs(cfm.feature_specs), embedding_dim=d, num_heads=spec["num_heads"],
                     dim_head=spec["dim_head"], depth=spec["depth"], scale_dim=spec["scale_dim"],
                     dnn_hidden_units=tuple(spec["dnn_hidden_units"]), batch_norm=spec["batch_norm"], use_wide=spec["use_wide"],
                     embedding_regularizer=0.0, learning_rate=spec["learning_rate"])
    # ---- forward, eval mode, the whole batch
    model.eval()
    with torch.no_grad():
        mine = model.forward((X, y, rv, rl))["y_pred"].reshape(-1).double().cpu()
        ref = orc.forward(w0, Xc, y, cfg, training=False).reshape(-1).double()
    assert float(mine.std()) > 1e-3
    np.testing.assert_allclose(mine.numpy(), ref.numpy(), rtol=0, atol=2e-6)
    # ---- one training step, taken apart: backward -> row lists, then clip + row Adam
    n_rows = model._n_feat // d
    probe = torch.randint(0, n_rows, (200_000,), generator=torch.Generator().manual_seed(5)).to(model.device)
    table = model._flat[:model._n_feat].view(n_rows, d)
    probe_before = table[probe].clone()
    model.train()
    model.optimizer.zero_grad()
    loss = model.get_total_loss((X, y, rv, rl))
    loss.backward()
    torch.cuda.synchronize()
    model.check_id_errors()
    ref_loss, _, ref_grads, bn_state = orc.loss_and_grads(w0, Xc, y, cfg, training=True)
    assert abs(float(loss) - float(ref_loss)) < 2e-6, (float(loss), float(ref_loss))
    # samples with a ReLU input within rounding distance of 0 (BatchNorm on: normalised values, statistics of the whole batch) may
    # land on the other side of the ReLU on the device; their target rows are left out of the row comparisons
    sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    from test_gpu_configs import _relu_margins
    unsafe = (_relu_margins(orc, w0, Xc, y, cfg) <= 3e-6).nonzero().reshape(-1)
    print("configs[3] full vocabulary: %d of %d samples ReLU-unsafe (their target rows are not compared)" % (unsafe.numel(), B))
    assert unsafe.numel() < B // 20
    # global row id (in the flat table block) -> (field, compact row)
    offs = {f: model._offsets[emb % f] // d for f in uniq}
    skip_rows = set()
    for b in unsafe.tolist():
        for i, f in enumerate(uniq):
            skip_rows.add(offs[f] + int(X[b, 0, i]))
    (rows_f, grads_f, count_f, width_f, _tot, _base), (rows_l, grads_l, count_l, width_l, _tl, base_l) = model._sparse
    assert width_f == d and width_l == 1
    U = int(count_f.cpu()[0])
    assert U == sum(int(u.numel()) for u in uniq.values()), "every id the batch names is one unique row"
    got_rows = rows_f[:U].cpu().long()
    want_rows = torch.cat([uniq[f] + offs[f] for f in uniq])
    assert torch.equal(got_rows, want_rows), "unique rows come out sorted by (field, id)"
    want_grads = torch.cat([ref_grads[emb % f] for f in uniq]).double()
    keep = torch.tensor([r not in skip_rows for r in got_rows.tolist()])
    err = (grads_f[:U].cpu().double() - want_grads).abs()[keep].max()
    assert float(err) < 3e-4 * float(want_grads.abs().max()), (float(err), float(want_grads.abs().max()))
    UL = int(count_l.cpu()[0])
    lr_off = {f: (model._offsets[lr % f] - base_l) for f in uniq}
    want_lr_rows = torch.cat([torch.unique(X[:, 0, i].long()) + lr_off[f] for i, f in enumerate(uniq)])
    assert UL == want_lr_rows.numel() and torch.equal(rows_l[:UL].cpu().long(), want_lr_rows)
    for k, p in model.named_parameters():                          # dense part of the bucket
        if k.startswith(("query_proj", "embedding_layer.", "lr_layer.")) or k in sc.mc.noise_tensors(model):
            continue
        g, r = p.grad.detach().cpu().double(), ref_grads[k].double()
        # (a ReLU-unsafe sample — see above — may contribute to the DNN weight gradients with one hidden unit on the other side of
        # the ReLU: those tensors get head-room whenever such samples exist; every other tensor is held to 3e-4)
        # (measured: 6 such samples of 1024; one flipped unit moved dnn.dnn.0.weight by 6e-3 of its largest element when the head GEMMs
        # went from exact fp32 to bf16x3 — with 1200 ReLU inputs x 1024 samples a batch WITHOUT a near-zero pre-activation does not exist)
        # A flip at layer l changes that sample's whole contribution to the weight gradients of layers <= l (2.5e-2 of dnn.dnn.3.weight's
        # largest element was observed): with flips present the DNN tensors are only held to 5e-2 here — their tight check (3e-4, BatchNorm
        # on, a flip-free slice found by search) is tests/test_gpu_configs.py::..._with_batchnorm; every other tensor stays at 3e-4.
        tol = 5e-2 if (k.startswith("dnn.") and unsafe.numel()) else 3e-4
        assert float((g - r).abs().max()) < tol * float(r.abs().max()), (k, float((g - r).abs().max()), float(r.abs().max()))
    # ---- clip + lazy row Adam == dense Adam on the rows the step touched (step 1: zero moments everywhere else)
    model.optimizer.clip_and_step(10.0)
    torch.cuda.synchronize()
    clipped, _ = orc.clip_grad_norm(ref_grads, 10.0)
    w1 = orc.adam_step(w0, clipped, {}, cfg.learning_rate, 1)
    after = snapshot()
    noise = sc.mc.noise_tensors(model)
    n_bad = n_all = 0
    for k in w1:
        if k.startswith("query_proj") or k in noise or k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            continue
        a, r = after[k].double(), w1[k].double()
        bad = (a - r).abs() > 2e-6 + 2e-4 * r.abs()
        owner = next((f for f in uniq if k == emb % f), None)
        if owner is not None and skip_rows:
            rows_k = (uniq[owner] + offs[owner]).tolist()
            bad[torch.tensor([r_ in skip_rows for r_ in rows_k])] = False
        assert float((a - r).abs().max()) <= 2.1e-3, k              # Adam's first step is sign-like: at most 2 * lr apart
        if k.startswith("dnn.") and unsafe.numel():
            # (ReLU flips, see above: where a flipped contribution decides the SIGN of a small gradient element Adam's first step goes
            # the other way — bounded per element by the line above, counted only loosely)
            assert float(bad.double().mean()) < 2e-2, k
            continue
        n_bad += int(bad.sum())
        n_all += bad.numel()
    # (with ReLU flips present, BatchNorm's batch-mean terms carry a flipped unit's change to EVERY sample's DNN-branch gradient at the
    # 1e-3 level: table elements whose gradient is that close to zero take Adam's first step the other way — 7e-5 of the elements observed)
    assert n_bad < (3e-4 if unsafe.numel() else 1e-5) * n_all + 5, (n_bad, n_all)
    changed = (table[probe] != probe_before).any(dim=1)
    touched = torch.isin(probe, want_rows.to(probe.device))
    assert not bool((changed & ~touched).any()), "a row the batch did not name was modified"
    assert int((changed & touched).sum()) == int(touched.sum())
