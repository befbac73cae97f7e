"""Shared checks of the row-sparse / deterministic embedding-gradient path (csrc/sparse.hip, K1s) — used with the host-emulation
build on CPU (tests/test_sparse_grad.py) and the HIP build on the GPU (tests/test_gpu_sparse.py).

Kernel level: plan + segmented reduction against CPU autograd's embedding_dense_backward on ids with heavy duplication, a bag
field, padding ids and (reported elsewhere, ignored here) out-of-vocabulary ids; the merge of gathered per-rank lists.
Model level: `sorted` == `atomic` gradients (and bit-identical between two runs), `sparse` training == dense training wherever
lazy Adam and dense Adam coincide (embedding_regularizer = 0: step 1 always; later steps when the same rows are touched)."""
import numpy as np
import torch

import golden_cases as gc
import model_cases as mc
from kernel_cases import F, close, rnd
from rat_amd import ops


def check_sorted_reduce(lib, dev, d, B=5, T=4, dup_vocab=3):
    rs = np.random.RandomState(5)
    L = 5
    # tiny vocabularies -> every row is hit many times (the fixed-order segment sum is what is being tested)
    fields = [F(0, 1, dup_vocab + 4), F(1, 3, 6, padding_idx=5), F(4, 1, 9, padding_idx=8)]
    sizes = [f.vocab * d for f in fields]
    flat = torch.zeros(sum(sizes), dtype=torch.float32)
    offs = np.cumsum([0] + sizes)
    tables = [flat[offs[i]:offs[i + 1]].view(f.vocab, d) for i, f in enumerate(fields)]
    cols_hi = [fields[0].vocab, 6, 6, 6, 9]
    idx = torch.stack([torch.from_numpy(rs.randint(0, cols_hi[c], size=(B, T))) for c in range(L)], -1).int().contiguous()
    idx[0, 1, 0] = 99                                    # outside the vocabulary: clamped to the last row like the forward gather
    #                                                      and the atomic scatter (rat_check_ids reports it)
    dgrid = rnd(rs, B, T, 4, d)
    dflat = rnd(rs, B, 3 * d)
    # reference: CPU autograd of the lookups (out-of-range id clamped)
    tl = [torch.zeros(f.vocab, d, requires_grad=True) for f in fields]
    e0 = tl[0][idx[..., 0].clamp(max=fields[0].vocab - 1).long()]
    e1 = torch.nn.functional.embedding(idx[..., 1:4].long(), tl[1], padding_idx=5).sum(-2)
    e2 = torch.nn.functional.embedding(idx[..., 4].long(), tl[2], padding_idx=8)
    ref = torch.stack([e0, e1, e2], dim=2)
    ((ref * dgrid[:, :, 1:]).sum() + (ref[:, 0].reshape(B, -1) * dflat).sum()).backward()
    ref_dense = torch.cat([t.grad.reshape(-1) for t in tl])
    # device
    flat_d = flat.to(dev)
    tabs_d = [flat_d[offs[i]:offs[i + 1]].view(f.vocab, d) for i, f in enumerate(fields)]
    ftab = ops.field_table(fields, tabs_d, dev)
    c2f = ops.col2field_table(fields, L, dev)
    idx_d, dgrid_d, dflat_d = idx.to(dev), dgrid.to(dev), dflat.to(dev)
    total_rows = sum(f.vocab for f in fields)
    plan = ops.sparse_plan_ids(idx_d, ftab, c2f, 3, flat_d, d, total_rows, B, T, L, lib=lib)
    dense = torch.zeros_like(flat_d)
    cap = min(B * T * L, total_rows)
    rows = torch.full((cap,), -7, dtype=torch.int32, device=dev)
    grads = torch.zeros((cap, d), dtype=torch.float32, device=dev)
    ops.sparse_reduce_grid(plan, dgrid_d, dflat_d, c2f, B, T, L, 3, d, out_rows=rows, out_grads=grads, dense_base=dense, lib=lib)
    U = int(plan.count.cpu()[0])
    touched = (ref_dense.view(total_rows, d).abs().sum(1) > 0)
    assert U >= int(touched.sum()) and U <= cap
    # a row collects up to B*T*3 / vocab terms of size ~1 in a different (sorted) order than CPU autograd: fp32 rounding of the
    # partial sums grows with the term count
    close(dense, ref_dense, 1e-5, 1e-6 + 1e-7 * B * T, "dense sums")
    r = rows[:U].cpu().long()
    assert bool((r[1:] > r[:-1]).all()), "unique rows come out sorted"
    assert not bool(((r == 6 + fields[0].vocab - 1 + 0) & False).any())
    rebuilt = torch.zeros(total_rows, d)
    rebuilt[r] = grads[:U].cpu()
    assert torch.equal(rebuilt.reshape(-1), dense.cpu()), "row lists and the dense table hold the same bits"
    # padding rows never appear
    pad_rows = {fields[0].vocab + 5, fields[0].vocab + 6 + 8}
    assert not (set(r.tolist()) & pad_rows)
    # run-to-run reproducible
    dense2 = torch.zeros_like(flat_d)
    plan2 = ops.sparse_plan_ids(idx_d, ftab, c2f, 3, flat_d, d, total_rows, B, T, L, lib=lib)
    ops.sparse_reduce_grid(plan2, dgrid_d, dflat_d, c2f, B, T, L, 3, d, dense_base=dense2, lib=lib)
    assert torch.equal(dense, dense2)
    return rows[:U].clone(), grads[:U].clone(), dense


def check_scalar_reduce(lib, dev, B=9, T=3):
    """width-1 (LR) tables: gradient of row = sum of dlogit over the TARGET samples that name it"""
    rs = np.random.RandomState(8)
    L = 4
    fields = [F(0, 1, 5), F(1, 2, 4, padding_idx=3), F(3, 1, 6)]
    total = sum(f.vocab for f in fields)
    flat_d = torch.zeros(total, dtype=torch.float32, device=dev)
    offs = np.cumsum([0] + [f.vocab for f in fields])
    ftab = ops.field_table(fields, [flat_d[offs[i]:offs[i + 1]].view(-1, 1) for i in range(3)], dev)
    c2f = ops.col2field_table(fields, L, dev)
    idx = torch.stack([torch.from_numpy(rs.randint(0, [5, 4, 4, 6][c], size=(B, T))) for c in range(L)], -1).int().contiguous()
    dlogit = rnd(rs, B, 1)
    ref = torch.zeros(total)
    for b in range(B):
        for c in range(L):
            f = [0, 1, 1, 2][c]
            i = int(idx[b, 0, c])
            if i != (fields[f].padding_idx if fields[f].padding_idx is not None else -1):
                ref[offs[f] + i] += dlogit[b, 0]
    plan = ops.sparse_plan_ids(idx.to(dev), ftab, c2f, 3, flat_d, 1, total, B, T, L, target_only=True, lib=lib)
    dense = torch.zeros(total, dtype=torch.float32, device=dev)
    rows = torch.zeros(min(B * L, total), dtype=torch.int32, device=dev)
    vals = torch.zeros((rows.numel(), 1), dtype=torch.float32, device=dev)
    ops.sparse_reduce_scalar(plan, dlogit.to(dev), B, L, out_rows=rows, out_vals=vals, dense_base=dense, lib=lib)
    close(dense, ref, 1e-5, 1e-6)
    U = int(plan.count.cpu()[0])
    rebuilt = torch.zeros(total)
    rebuilt[rows[:U].cpu().long()] = vals[:U, 0].cpu()
    assert torch.equal(rebuilt, dense.cpu())


def check_merge_rows(lib, dev, d=8, world=3, cap=7, total_rows=11):
    """the data-parallel exchange: `world` gathered (rows, grads) lists of capacity `cap` -> union with summed duplicates"""
    rs = np.random.RandomState(9)
    counts = torch.tensor([5, 0, 7][:world], dtype=torch.int32)
    rows = torch.from_numpy(rs.randint(0, total_rows, size=(world, cap))).int()
    for r in range(world):                                   # a rank's own list has unique rows
        rows[r, :int(counts[r])] = torch.from_numpy(rs.permutation(total_rows)[:int(counts[r])]).int()
    grads = rnd(rs, world, cap, d)
    ref = torch.zeros(total_rows, d, dtype=torch.float64)
    for r in range(world):
        for i in range(int(counts[r])):
            ref[int(rows[r, i])] += grads[r, i].double()
    plan = ops.sparse_plan_rows(rows.reshape(-1).to(dev), counts.to(dev), cap, world, total_rows, lib=lib)
    ncap = min(cap * world, total_rows)
    out_rows = torch.zeros(ncap, dtype=torch.int32, device=dev)
    out_grads = torch.zeros((ncap, d), dtype=torch.float32, device=dev)
    ops.sparse_reduce_rows(plan, grads.reshape(-1, d).to(dev).contiguous(), cap, world, d, out_rows, out_grads, lib=lib)
    U = int(plan.count.cpu()[0])
    assert U == int((ref.abs().sum(1) > 0).sum())
    got = torch.zeros(total_rows, d)
    got[out_rows[:U].cpu().long()] = out_grads[:U].cpu()
    close(got, ref, 1e-6, 1e-6)
    # row optimizer: Adam on the listed rows only, clip coefficient from the device scalar
    w = rnd(rs, total_rows, d).to(dev)
    w0 = w.clone()
    m, v = torch.zeros_like(w), torch.zeros_like(w)
    nsq = torch.zeros(1, device=dev)
    ops.sumsq_rows(out_grads, plan.count, ncap, d, nsq, lib=lib)
    assert abs(float(nsq[0]) - float((ref ** 2).sum())) < 1e-4 * float((ref ** 2).sum())
    ops.adam_rows(w, m, v, out_rows, out_grads, plan.count, ncap, d, nsq, 0.5, 1e-2, 0.9, 0.999, 1e-8, 1, lib=lib)
    coef = min(1.0, 0.5 / (float(nsq[0]) ** 0.5 + 1e-6))
    g = ref.float() * coef
    want = w0.cpu() - 1e-2 * g / (g.abs() + 1e-8) * (g != 0)        # step 1 of Adam: m_hat / (sqrt(v_hat) + eps) = g / (|g| + eps)
    close(w, want, 1e-4, 1e-6)
    untouched = ref.abs().sum(1) == 0
    assert torch.equal(w.cpu()[untouched], w0.cpu()[untouched])


def _model(case_name, gpu, mode, **over):
    case = dict(gc.case_by_name(case_name))
    case.update(over)
    model = mc.build_model(case, gpu=gpu, seed=1, embedding_grad=mode)
    mc.load_weights(model, case)
    return case, model, mc.batch_of(case)


def check_model_sorted_equals_atomic(case_name, gpu, bad_ids=False):
    """bad_ids: one out-of-vocabulary and one negative id in the batch — every path clamps them the same way (the forward reads the
    clamped row, both gradient paths update it), so the modes still agree; check_id_errors() reports them."""
    grads = {}
    for mode in ("atomic", "sorted", "sorted"):
        case, model, batch = _model(case_name, gpu, mode)
        if bad_ids:
            X = batch[0].clone()
            X[0, 1, 0] = 10 ** 6
            X[1, 0, 0] = -3
            batch = (X,) + tuple(batch[1:])
        model.train()
        model.optimizer.zero_grad()
        model.get_total_loss(batch).backward()
        g = {k: p.grad.detach().clone().cpu() for k, p in model.named_parameters() if p.grad is not None}
        grads.setdefault(mode, []).append(g)
        if bad_ids:
            try:
                model.check_id_errors()
                raise AssertionError("the out-of-vocabulary ids were not reported")
            except IndexError:
                pass
    a, (s1, s2) = grads["atomic"][0], grads["sorted"]
    assert set(a) == set(s1)
    noise = mc.noise_tensors(model)
    for k in a:
        if "embedding_layer" in k:
            assert torch.equal(s1[k], s2[k]), "sorted table gradients must be bit-reproducible: " + k
        tol = 3e-5 if k in noise else 3e-6
        np.testing.assert_allclose(s1[k].numpy(), a[k].numpy(), rtol=3e-4, atol=tol, err_msg=k)


def check_model_sparse_training(case_name, gpu, steps=3):
    """embedding_regularizer = 0 and the same batch every step: lazy row Adam == dense Adam (every touched row is touched in every
    step, untouched rows have zero gradient and zero moments in both)."""
    out = {}
    for mode in ("atomic", "sparse"):
        case, model, batch = _model(case_name, gpu, mode, embedding_regularizer=0.0)
        model.train()
        losses = [float(model.train_step(batch)) for _ in range(steps)]
        out[mode] = (losses, {k: v.detach().clone().cpu() for k, v in model.state_dict().items()})
        if mode == "sparse":
            assert model._grad_mode == "sparse" and model._n_sparse == model._n_tab > 0
            assert all(p.grad is None for n, p in model.named_parameters() if n.startswith(("embedding_layer.", "lr_layer.")))
    (la, wa), (ls, ws) = out["atomic"], out["sparse"]
    np.testing.assert_allclose(ls, la, rtol=0, atol=1e-5)          # atomics: arrival-order sums in the dense path
    noise = mc.noise_tensors(model)
    for k in wa:
        if k.startswith("query_proj") or k in noise or k.endswith("num_batches_tracked"):
            continue
        atol = 3e-6 if not k.endswith("running_mean") else 3e-6 + steps * 0.1 * 1e-3 * 1.01
        np.testing.assert_allclose(ws[k].numpy(), wa[k].numpy(), rtol=3e-4, atol=atol, err_msg=k)
