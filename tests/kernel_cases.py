"""Shared helpers for the kernel parity tests (used with the host-emulation build on CPU and the HIP build on GPU)."""
import numpy as np
import torch

from oracle import rat_m2_oracle as orc


def rnd(rs, *shape, scale=1.0):
    return torch.from_numpy((scale * rs.standard_normal(shape)).astype(np.float32))


def attn_reference(x, ln_g, ln_b, w_qkv, w_out, b_out, heads, dh, mode):
    """x: [B,T,S,d] float64/32 -> y same layout, via the oracle's attention()."""
    B, T, S, d = x.shape
    w = {"p.norm.weight": ln_g, "p.norm.bias": ln_b, "p.fn.to_qkv.weight": w_qkv}
    if w_out is not None:
        w["p.fn.to_out.0.weight"] = w_out
        w["p.fn.to_out.0.bias"] = b_out
    cfg = orc.Config(fields=[], embedding_dim=d, num_heads=heads, dim_head=dh)
    if mode == "intra":
        xi = x.reshape(B * T, S, d)
        return (orc.attention(xi, w, "p.", cfg) + xi).reshape(B, T, S, d)
    xc = x.transpose(1, 2).reshape(B * S, T, d)
    return (orc.attention(xc, w, "p.", cfg) + xc).reshape(B, S, T, d).transpose(1, 2)


def ffn_reference(x, w1, b1, w2, b2):
    w = {"m.0.weight": w1, "m.0.bias": b1, "m.3.weight": w2, "m.3.bias": b2}
    return orc.feed_forward(x, w, "m.") + x


# ----------------------------------------------------------------------------- generic kernel checks
from rat_amd import ops  # noqa: E402


class F:
    def __init__(self, col, ncols, vocab, padding_idx=None):
        self.col, self.ncols, self.vocab, self.padding_idx = col, ncols, vocab, padding_idx


def close(a, b, rtol, atol, msg=""):
    np.testing.assert_allclose(a.detach().cpu().double().numpy(), b.detach().cpu().double().numpy(), rtol=rtol, atol=atol, err_msg=msg)


def check_gather(lib, dev, d, B=3, T=4):
    rs = np.random.RandomState(0)
    L = 5
    fields = [F(0, 1, 7), F(1, 3, 6, padding_idx=5), F(4, 1, 9, padding_idx=8)]
    tables = [rnd(rs, f.vocab, d) for f in fields]
    tables[1][5] = 0
    label_table = rnd(rs, 3, d)
    idx = torch.stack([torch.from_numpy(rs.randint(0, [7, 6, 6, 6, 9][c], size=(B, T))) for c in range(L)], -1).int().contiguous()
    labels = torch.from_numpy(rs.randint(0, 2, size=(B, T))).int()
    labels[:, 0] = 2
    dgrid = rnd(rs, B, T, 4, d)
    dflat = rnd(rs, B, 3 * d)
    # reference (CPU autograd)
    tl = [t.clone().requires_grad_(True) for t in tables] + [label_table.clone().requires_grad_(True)]
    e1 = torch.nn.functional.embedding(idx[..., 1:4].long(), tl[1], padding_idx=5).sum(-2)
    e2 = torch.nn.functional.embedding(idx[..., 4].long(), tl[2], padding_idx=8)
    ref = torch.stack([tl[3][labels.long()], tl[0][idx[..., 0].long()], e1, e2], dim=2)
    ((ref * dgrid).sum() + (ref[:, 0, 1:].reshape(B, -1) * dflat).sum()).backward()
    # device
    tables_d = [t.to(dev) for t in tables]
    label_d, idx_d, labels_d = label_table.to(dev), idx.to(dev), labels.to(dev)
    ftab = ops.field_table(fields, tables_d, dev)
    grid = ops.gather_fwd(idx_d, labels_d, ftab, 3, label_d, B, T, L, d, lib=lib)
    assert torch.equal(grid.cpu(), ref.detach()), "gather must be bit-exact"
    gtabs = [torch.zeros_like(t) for t in tables_d]
    dlabel = torch.zeros(3, d, device=dev)
    gftab = ops.field_table(fields, gtabs, dev)
    ops.gather_bwd(dgrid.to(dev), dflat.to(dev), idx_d, labels_d, gftab, 3, dlabel, B, T, L, d, lib=lib)
    for g, t in zip(gtabs + [dlabel], tl):        # fp32 atomics: summation order differs from the CPU's
        close(g, t.grad, 1e-4, 2e-6 * float(t.grad.abs().max()) + 1e-6)


def check_sgemm(lib, dev, ta, tb, M=70, N=37, K=29, arith="f32"):
    """arith "bf16x3": sgemm3_kernel (operands split into bf16 planes, k-major tiles through transposed block reads) — same
    tolerance as the exact-fp32 kernel; it needs lda, ldb % 4 == 0 and K >= 64, other shapes silently run exact fp32."""
    rs = np.random.RandomState(1)
    A = rnd(rs, *((K, M) if ta else (M, K)))
    Bm = rnd(rs, *((N, K) if tb else (K, N)))
    bias = rnd(rs, N)
    C = rnd(rs, M, N)
    ref = (A.t() if ta else A).double() @ (Bm.t() if tb else Bm).double() + bias.double() + 0.5 * C.double()
    Cd = C.to(dev)
    ops.sgemm(ta, tb, M, N, K, A.to(dev), A.shape[1], Bm.to(dev), Bm.shape[1], Cd, N, bias=bias.to(dev), beta=0.5, arith=arith, lib=lib)
    close(Cd, ref, 1e-5, 1e-5 * max(1.0, K ** 0.5))       # fp32 accumulation over K terms of O(1) products


def attn_weights(rs, d, heads, dh, proj):
    inner = heads * dh
    return (1 + 0.1 * rnd(rs, d), 0.1 * rnd(rs, d), rnd(rs, 3 * inner, d, scale=d ** -0.5),
            rnd(rs, d, inner, scale=inner ** -0.5) if proj else None, 0.1 * rnd(rs, d) if proj else None)


def check_attn(lib, dev, case, mode, seed=2, arith="f32"):
    """arith "bf16x3": the split-operand bf16 MFMA kernels — SAME tolerances against the float64 reference as the exact-fp32
    kernels, plus a direct comparison of the two arithmetic variants (they must agree to fp32 rounding level)."""
    B, T, S, d, heads, dh, proj = case
    rs = np.random.RandomState(seed)
    x = rnd(rs, B, T, S, d)
    ws = attn_weights(rs, d, heads, dh, proj)
    dy = rnd(rs, B, T, S, d)
    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) if w is not None else None for w in ws]
    ref = attn_reference(xr, *wr, heads, dh, mode)
    ref.backward(dy.double())
    xd, dyd = x.to(dev), dy.to(dev)
    wd = [w.to(dev) if w is not None else None for w in ws]
    params = ops.attn_params(*wd)
    smap = ops.intra_map(B, T, S) if mode == "intra" else ops.cross_map(B, T, S)
    y, o_save, lse = ops.attn_fwd(xd, params, smap, d, heads, dh, save=True, arith=arith, lib=lib)
    close(y, ref, 2e-5, 2e-5, "y")
    gs = [torch.zeros_like(w) if w is not None else None for w in wd]
    grads = ops.attn_params(*gs)
    dx, _ = ops.attn_bwd(xd, dyd, o_save, lse, params, grads, smap, d, heads, dh, arith=arith, lib=lib)
    scale = max(1.0, (B * T * S) ** 0.5 / 4)
    close(dx, xr.grad, 1e-4, 1e-4, "dx")
    for name, g, w in zip(["ln_g", "ln_b", "w_qkv", "w_out", "b_out"], gs, wr):
        if w is not None:
            close(g, w.grad, 1e-4, 1e-4 * scale, name)
    if arith != "f32":                                          # the two arithmetic variants side by side
        y0, o0, l0 = ops.attn_fwd(xd, params, smap, d, heads, dh, save=True, lib=lib)
        g0 = [torch.zeros_like(w) if w is not None else None for w in wd]
        dx0, _ = ops.attn_bwd(xd, dyd, o0, l0, params, ops.attn_params(*g0), smap, d, heads, dh, lib=lib)
        for name, a_, b_ in [("y", y, y0), ("o_save", o_save, o0), ("lse", lse, l0), ("dx", dx, dx0)] + \
                [(n, ga, gb) for n, ga, gb in zip(["ln_g", "ln_b", "w_qkv", "w_out", "b_out"], gs, g0) if ga is not None]:
            err = float((a_ - b_).abs().max()) / (float(b_.abs().max()) + 1e-30)
            assert err < 4e-6, ("bf16x3 vs exact fp32", name, err)
        # weight planes split ONCE by the caller (RatAttnParams.planes + rat_split_weights_batch): bit-identical to the per-call split
        nbytes = ops.attn_planes_bytes(d, heads, dh, lib=lib)
        if nbytes:
            planes = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            p2 = ops.attn_params(*wd, planes=planes)
            jobs = ops.attn_split_jobs(p2, d, heads, dh, planes, lib=lib)
            assert len(jobs) == 4
            arr, n = ops.split_job_array(jobs)
            ops.split_weights_batch(arr, n, xd, lib=lib)
            y2, o2, l2 = ops.attn_fwd(xd, p2, smap, d, heads, dh, save=True, arith=arith, lib=lib)
            g2 = [torch.zeros_like(w) if w is not None else None for w in wd]
            dx2, _ = ops.attn_bwd(xd, dyd, o2, l2, p2, ops.attn_params(*g2), smap, d, heads, dh, arith=arith, lib=lib)
            for name, a_, b_ in [("y", y, y2), ("o_save", o_save, o2), ("lse", lse, l2), ("dx", dx, dx2)] + \
                    [(n_, ga, gb) for n_, ga, gb in zip(["ln_g", "ln_b", "w_qkv", "w_out", "b_out"], gs, g2) if ga is not None]:
                assert torch.equal(a_, b_), ("planes split by the caller", name)


def check_attn_queries(lib, dev, case, mode, nq=1, seed=5, arith="f32"):
    """RatSeqMap.queries (include/rat_hip.h): only the outputs of positions [0, nq) of each sequence are wanted and the gradient rows
    of the other positions are zero.  Against the oracle with exactly that gradient: y at the query positions, the WHOLE dx (every
    position is a key and a value), every parameter gradient — and against the same kernels run without `queries` on the same
    zero-padded dy (the contract says skipping the other queries changes nothing)."""
    B, T, S, d, heads, dh, proj = case
    rs = np.random.RandomState(seed)
    x = rnd(rs, B, T, S, d)
    ws = attn_weights(rs, d, heads, dh, proj)
    dy = rnd(rs, B, T, S, d)
    if mode == "intra":
        dy[:, :, nq:, :] = 0.0
        qsel = lambda t: t[:, :, :nq, :]                          # noqa: E731
        full, part = ops.intra_map(B, T, S), ops.intra_map(B, T, S, queries=nq)
    else:
        dy[:, nq:, :, :] = 0.0
        qsel = lambda t: t[:, :nq, :, :]                          # noqa: E731
        full = ops.cross_map(B, T, S)
        part = ops.cross_map(B, T, S)
        part.queries = nq
    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) if w is not None else None for w in ws]
    ref = attn_reference(xr, *wr, heads, dh, mode)
    ref.backward(dy.double())
    xd, dyd = x.to(dev), dy.to(dev)
    wd = [w.to(dev) if w is not None else None for w in ws]
    params = ops.attn_params(*wd)
    names = ["ln_g", "ln_b", "w_qkv", "w_out", "b_out"]
    res = {}
    for key, smap in (("part", part), ("full", full)):
        y, o_save, lse = ops.attn_fwd(xd, params, smap, d, heads, dh, save=True, arith=arith, lib=lib)
        assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(o_save).all()) and bool(torch.isfinite(lse).all())
        gs = [torch.zeros_like(w) if w is not None else None for w in wd]
        dx, _ = ops.attn_bwd(xd, dyd, o_save, lse, params, ops.attn_params(*gs), smap, d, heads, dh, arith=arith, lib=lib)
        res[key] = (y, dx, gs)
    y, dx, gs = res["part"]
    close(qsel(y), qsel(ref), 2e-5, 2e-5, "y at the query positions")
    scale = max(1.0, (B * T * S) ** 0.5 / 4)
    close(dx, xr.grad, 1e-4, 1e-4, "dx")
    for name, g, w in zip(names, gs, wr):
        if w is not None:
            close(g, w.grad, 1e-4, 1e-4 * scale, name)
    y0, dx0, g0 = res["full"]
    assert torch.equal(qsel(y), qsel(y0))
    for name, a_, b_ in [("dx", dx, dx0)] + [(n, ga, gb) for n, ga, gb in zip(names, gs, g0) if ga is not None]:
        err = float((a_ - b_).abs().max()) / (float(b_.abs().max()) + 1e-30)
        assert err < 2e-6, ("queries vs all positions", name, err)


def check_attn_dropout(lib, dev, case, mode, p=0.25, seed=123456789, arith="f32"):
    """Dropout behind the output projection (RAT_m2.py:186-189): y = Dropout(to_out(...)) + x.  The mask is a pure function of
    (seed, element index) — the generator of rat_dropout — so the reference uses the mask that rat_dropout produces on ones."""
    B, T, S, d, heads, dh, proj = case
    assert proj
    rs = np.random.RandomState(31)
    x = rnd(rs, B, T, S, d)
    ws = attn_weights(rs, d, heads, dh, proj)
    dy = rnd(rs, B, T, S, d)
    mask = ops.dropout(torch.ones_like(x).to(dev), p, seed, lib=lib).cpu().double()
    kept = float((mask > 0).double().mean())
    assert abs(kept - (1 - p)) < 0.1 and set(np.unique(mask.numpy()).round(6)) <= {0.0, round(1 / (1 - p), 6)}
    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) for w in ws]
    ref = mask * (attn_reference(xr, *wr, heads, dh, mode) - xr) + xr
    ref.backward(dy.double())
    xd, dyd = x.to(dev), dy.to(dev)
    wd = [w.to(dev) for w in ws]
    params = ops.attn_params(*wd)
    smap = ops.intra_map(B, T, S) if mode == "intra" else ops.cross_map(B, T, S)
    y, o_save, lse = ops.attn_fwd(xd, params, smap, d, heads, dh, save=True, arith=arith, dropout=(p, seed), lib=lib)
    close(y, ref, 2e-5, 2e-5, "y")
    gs = [torch.zeros_like(w) for w in wd]
    dx, _ = ops.attn_bwd(xd, dyd, o_save, lse, params, ops.attn_params(*gs), smap, d, heads, dh, arith=arith, dropout=(p, seed), lib=lib)
    scale = max(1.0, (B * T * S) ** 0.5 / 4)
    close(dx, xr.grad, 1e-4, 1e-4, "dx")
    for name, g, w in zip(["ln_g", "ln_b", "w_qkv", "w_out", "b_out"], gs, wr):
        close(g, w.grad, 1e-4, 1e-4 * scale, name)


def check_attn_ex(lib, dev, case, mode, res_mode, out_scale, softmax_scale, seed=5, arith="f32"):
    """rat_attn_fwd_ex / rat_attn_bwd_ex: y = out_scale * attention(LN(x)) + res with res in {none, a second tensor, the output
    itself}, an explicit softmax scale; backward with the matching `add` term."""
    B, T, S, d, heads, dh, proj = case
    rs = np.random.RandomState(seed)
    x = rnd(rs, B, T, S, d)
    other = rnd(rs, B, T, S, d)
    ws = attn_weights(rs, d, heads, dh, proj)
    dy = rnd(rs, B, T, S, d)
    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) if w is not None else None for w in ws]
    w = {"p.norm.weight": wr[0], "p.norm.bias": wr[1], "p.fn.to_qkv.weight": wr[2]}
    if proj:
        w["p.fn.to_out.0.weight"], w["p.fn.to_out.0.bias"] = wr[3], wr[4]
    cfg = orc.Config(fields=[], embedding_dim=d, num_heads=heads, dim_head=dh)
    if mode == "intra":
        att = orc.attention(xr.reshape(B * T, S, d), w, "p.", cfg, scale=softmax_scale).reshape(B, T, S, d)
    else:
        att = orc.attention(xr.transpose(1, 2).reshape(B * S, T, d), w, "p.", cfg, scale=softmax_scale).reshape(B, S, T, d).transpose(1, 2)
    ref = out_scale * att + (other.double() if res_mode != "none" else 0.0)
    ref.backward(dy.double())
    xd, dyd, od = x.to(dev), dy.to(dev), other.to(dev)
    wd = [t.to(dev) if t is not None else None for t in ws]
    params = ops.attn_params(*wd)
    smap = ops.intra_map(B, T, S) if mode == "intra" else ops.cross_map(B, T, S)
    if res_mode == "acc":                                   # accumulate onto a tensor that already holds `other`
        y = od.clone()
        y, o_save, lse = ops.attn_fwd_ex(xd, y, params, smap, d, heads, dh, softmax_scale or 0.0, out_scale, save=True, out=y, arith=arith, lib=lib)
    else:
        y, o_save, lse = ops.attn_fwd_ex(xd, od if res_mode == "other" else None, params, smap, d, heads, dh, softmax_scale or 0.0,
                                         out_scale, save=True, arith=arith, lib=lib)
    close(y, ref, 2e-5, 2e-5, "y")
    gs = [torch.zeros_like(t) if t is not None else None for t in wd]
    grads = ops.attn_params(*gs)
    add = rnd(rs, B, T, S, d)
    if res_mode == "none":
        dx, _ = ops.attn_bwd_ex(xd, dyd, None, o_save, lse, params, grads, smap, d, heads, dh, softmax_scale or 0.0, out_scale, arith=arith, lib=lib)
        want = xr.grad
    else:                                                   # dx = add + ..., written in place over `add`
        buf = add.clone().to(dev)             # .to("cpu") would alias `add`, which the in-place kernel overwrites
        dx, _ = ops.attn_bwd_ex(xd, dyd, buf, o_save, lse, params, grads, smap, d, heads, dh, softmax_scale or 0.0, out_scale,
                                out=buf, arith=arith, lib=lib)
        want = xr.grad + add.double()
    scale = max(1.0, (B * T * S) ** 0.5 / 4)
    close(dx, want, 1e-4, 1e-4, "dx")
    for name, g, wt in zip(["ln_g", "ln_b", "w_qkv", "w_out", "b_out"], gs, wr):
        if wt is not None:
            close(g, wt.grad, 1e-4, 1e-4 * scale, name)


def check_attn_groups(lib, dev, case, mode, res_mode="x", dropout=0.0, seed=7):
    """rat_attn_fwd_groups: wide heads (heads = G x 8) in ONE launch that loops over the head groups inside a chunk.
    y against float64 attention over ALL heads; o_save / lse_save slice g bit-identical to what a rat_attn_fwd_ex launch on a contiguous
    copy of group g's weights saves (the per-group form the backward consumes); then the backward as G rat_attn_bwd_ex launches on those
    slices against float64 gradients."""
    B, T, S, d, heads, dh, proj = case
    assert proj and heads % 8 == 0 and heads > 8
    G, ig, I = heads // 8, 8 * dh, heads * dh
    rs = np.random.RandomState(seed)
    x = rnd(rs, B, T, S, d)
    other = rnd(rs, B, T, S, d)
    ws = attn_weights(rs, d, heads, dh, proj)
    dy = rnd(rs, B, T, S, d)
    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) for w in ws]
    att = attn_reference(xr, *wr, heads, dh, mode) - xr
    xd, dyd, od = x.to(dev), dy.to(dev), other.to(dev)
    drop = (0.0, 0)
    if dropout > 0:
        drop = (dropout, 987654321)
        mask = ops.dropout(torch.ones_like(xd), drop[0], drop[1], lib=lib).cpu().double()
        att = mask * att
    ref = att + (xr if res_mode == "x" else other.double())
    ref.backward(dy.double())
    wd = [w.to(dev) for w in ws]
    params = ops.attn_params(*wd)
    smap = ops.intra_map(B, T, S) if mode == "intra" else ops.cross_map(B, T, S)
    nbytes = ops.attn_groups_planes_bytes(d, heads, dh, lib=lib)
    assert nbytes > 0 and ops.attn_groups_planes_bytes(d, 8, dh, lib=lib) == 0
    planes = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    jobs = ops.attn_groups_split_jobs(params, d, heads, dh, planes, lib=lib)
    assert len(jobs) == 4 * G
    arr, n = ops.split_job_array(jobs)
    ops.split_weights_batch(arr, n, xd, lib=lib)
    y, o_save, lse = ops.attn_fwd_groups(xd, xd if res_mode == "x" else od, params, planes, smap, d, heads, dh, save=True, dropout=drop, lib=lib)
    close(y, ref, 2e-5, 2e-5, "y")
    y2, _, _ = ops.attn_fwd_groups(xd, xd if res_mode == "x" else od, params, planes, smap, d, heads, dh, save=False, dropout=drop, lib=lib)
    assert torch.equal(y, y2), "inference form (nothing saved)"
    # group g alone, on contiguous copies of its weight slices (what the model's grouped backward uses)
    wq = wd[2].view(3, G, ig, d).permute(1, 0, 2, 3).contiguous()
    wo = wd[3].view(d, G, ig).permute(1, 0, 2).contiguous()
    zero_b = torch.zeros_like(wd[4])
    ntok = B * T * S
    dx = None
    t_ln = torch.zeros((2, G, d), dtype=torch.float32, device=dev)
    t_w = torch.zeros((G, 3 * ig, d), dtype=torch.float32, device=dev)
    t_wo = torch.zeros((G, d, ig), dtype=torch.float32, device=dev)
    g_b, t_b = torch.zeros_like(wd[4]), torch.zeros_like(wd[4])
    for g in range(G):
        p_g = ops.attn_params(wd[0], wd[1], wq[g].view(3 * ig, d), wo[g], wd[4] if g == 0 else zero_b)
        _, o_g, l_g = ops.attn_fwd_ex(xd, None, p_g, smap, d, 8, dh, save=True, arith="bf16x3", lib=lib)
        assert torch.equal(o_g, o_save[g]) and torch.equal(l_g, lse[g]), ("saved O / lse of group", g)
        grads = ops.attn_params(t_ln[0, g], t_ln[1, g], t_w[g], t_wo[g], g_b if g == 0 else t_b)
        add = (dyd if res_mode == "x" else None) if g == 0 else dx
        dx, _ = ops.attn_bwd_ex(xd, dyd, add, o_save[g], lse[g], p_g, grads, smap, d, 8, dh, out=dx, arith="bf16x3", dropout=drop, lib=lib)
    scale = max(1.0, ntok ** 0.5 / 4)
    close(dx, xr.grad, 1e-4, 1e-4, "dx")
    close(t_w.view(G, 3, ig, d).permute(1, 0, 2, 3).reshape(3 * I, d), wr[2].grad, 1e-4, 1e-4 * scale, "w_qkv")
    close(t_wo.permute(1, 0, 2).reshape(d, I), wr[3].grad, 1e-4, 1e-4 * scale, "w_out")
    close(g_b, wr[4].grad, 1e-4, 1e-4 * scale, "b_out")
    close(t_ln[0].sum(0), wr[0].grad, 1e-4, 1e-4 * scale, "ln_g")
    close(t_ln[1].sum(0), wr[1].grad, 1e-4, 1e-4 * scale, "ln_b")


def check_attn_groups_small_d(lib, dev, case, mode, res_mode="x", dropout=0.0, seed=9):
    """rat_attn_fwd_groups / rat_attn_bwd_groups at a small embedding dimension (the shipped Tmall geometry: d = 10, 32 heads x 10):
    the whole wide-head layer in ONE launch per direction, weights addressed in place.  y, dx and every parameter gradient (delivered
    in the layer's full-width layout) against float64; the saved O / lse slices bit-identical to an 8-head launch on a contiguous copy of
    each group's weights (the same arithmetic in the same order)."""
    B, T, S, d, heads, dh, proj = case
    per = 80 // dh                                               # heads per group: 8 x 10, or (round 6, RAT_m3's geometry) 4 x 20
    assert proj and dh in (10, 20) and heads % per == 0 and heads > per and d <= 16
    G, ig, I = heads // per, per * dh, heads * dh
    assert ops.attn_groups_supported(d, heads, dh, lib=lib) == (3 if G <= 4 else 1)
    assert ops.attn_groups_supported(d, per, dh, lib=lib) == 0 and ops.attn_groups_planes_bytes(d, heads, dh, lib=lib) == 0
    rs = np.random.RandomState(seed)
    x = rnd(rs, B, T, S, d)
    other = rnd(rs, B, T, S, d)
    ws = attn_weights(rs, d, heads, dh, proj)
    dy = rnd(rs, B, T, S, d)
    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) for w in ws]
    att = attn_reference(xr, *wr, heads, dh, mode) - xr
    xd, dyd, od = x.to(dev), dy.to(dev), other.to(dev)
    drop = (0.0, 0)
    if dropout > 0:
        drop = (dropout, 192837465)
        att = ops.dropout(torch.ones_like(xd), drop[0], drop[1], lib=lib).cpu().double() * att
    ref = att + (xr if res_mode == "x" else other.double())
    ref.backward(dy.double())
    wd = [w.to(dev) for w in ws]
    params = ops.attn_params(*wd)
    smap = ops.intra_map(B, T, S) if mode == "intra" else ops.cross_map(B, T, S)
    y, o_save, lse = ops.attn_fwd_groups(xd, xd if res_mode == "x" else od, params, None, smap, d, heads, dh, save=True, dropout=drop, lib=lib)
    close(y, ref, 2e-5, 2e-5, "y")
    wq = wd[2].view(3, G, ig, d).permute(1, 0, 2, 3).contiguous()
    wo = wd[3].view(d, G, ig).permute(1, 0, 2).contiguous()
    for g in range(G):
        p_g = ops.attn_params(wd[0], wd[1], wq[g].view(3 * ig, d), wo[g], wd[4])
        _, o_g, l_g = ops.attn_fwd_ex(xd, None, p_g, smap, d, per, dh, save=True, lib=lib)
        if per == 8:                                             # the same arithmetic in the same order as the 8-head generic kernel
            assert torch.equal(o_g, o_save[g]) and torch.equal(l_g, lse[g]), ("saved O / lse of group", g)
        else:                                                    # (4 x 20: the generic <0, 20> kernel sums a head's 20 products in another order)
            close(o_save[g], o_g, 1e-5, 1e-6, "saved O of group %d" % g)
            close(lse[g], l_g, 1e-5, 1e-6, "saved lse of group %d" % g)
    if G > 4:
        return
    gs = [torch.full_like(w, 7.0) for w in wd]                   # (overwritten, not accumulated)
    if res_mode == "x":
        dx, _ = ops.attn_bwd_groups(xd, dyd, dyd, o_save, lse, params, ops.attn_params(*gs), smap, d, heads, dh, dropout=drop, lib=lib)
        want = xr.grad
    else:                                                        # dx = add + ..., in place over `add`
        add = rnd(rs, B, T, S, d)
        buf = add.clone().to(dev)
        dx, _ = ops.attn_bwd_groups(xd, dyd, buf, o_save, lse, params, ops.attn_params(*gs), smap, d, heads, dh, out=buf, dropout=drop, lib=lib)
        want = xr.grad + add.double()
    scale = max(1.0, (B * T * S) ** 0.5 / 4)
    close(dx, want, 1e-4, 1e-4, "dx")
    for name, g_, w_ in zip(["ln_g", "ln_b", "w_qkv", "w_out", "b_out"], gs, wr):
        close(g_, w_.grad, 1e-4, 1e-4 * scale, name)


def check_attn_core(lib, dev, nseq, L, heads, dh, softmax_scale=None):
    """rat_attn_core_fwd / bwd against float64 softmax attention on random projected rows."""
    rs = np.random.RandomState(17)
    I = heads * dh
    qkv = rnd(rs, nseq * L, 3 * I)
    dout = rnd(rs, nseq * L, I)
    r = qkv.double().requires_grad_(True)
    q, k, v = [t.reshape(nseq, L, heads, dh).permute(0, 2, 1, 3) for t in r.split(I, dim=-1)]
    sc = dh ** -0.5 if softmax_scale is None else softmax_scale
    p = torch.softmax((q @ k.transpose(-1, -2)) * sc, dim=-1)
    ref = (p @ v).permute(0, 2, 1, 3).reshape(nseq * L, I)
    ref.backward(dout.double())
    qd = qkv.to(dev)
    o, lse = ops.attn_core_fwd(qd, nseq, L, heads, dh, softmax_scale or 0.0, lib=lib)
    close(o, ref, 2e-5, 2e-5, "o")
    dqkv = ops.attn_core_bwd(qd, o, lse, dout.to(dev), nseq, L, heads, dh, softmax_scale or 0.0, lib=lib)
    close(dqkv, r.grad, 1e-4, 1e-4, "dqkv")


def check_attn_core_strided(lib, dev, B, T, S, heads, dh):
    """rat_attn_core_*_map with the cross-sample map (sequences of T tokens strided by S through the [B,T,S] grid)."""
    rs = np.random.RandomState(19)
    I = heads * dh
    qkv = rnd(rs, B * T * S, 3 * I)
    dout = rnd(rs, B * T * S, I)
    r = qkv.double().requires_grad_(True)
    g = r.reshape(B, T, S, 3 * I).transpose(1, 2).reshape(B * S, T, 3 * I)               # what the reference's transpose copy holds
    q, k, v = [t.reshape(B * S, T, heads, dh).permute(0, 2, 1, 3) for t in g.split(I, dim=-1)]
    p = torch.softmax((q @ k.transpose(-1, -2)) * dh ** -0.5, dim=-1)
    ref = (p @ v).permute(0, 2, 1, 3).reshape(B, S, T, I).transpose(1, 2).reshape(B * T * S, I)
    ref.backward(dout.double())
    smap = ops.cross_map(B, T, S)
    qd = qkv.to(dev)
    o, lse = ops.attn_core_fwd_map(qd, smap, heads, dh, lib=lib)
    close(o, ref, 2e-5, 2e-5, "o")
    dqkv = ops.attn_core_bwd_map(qd, o, lse, dout.to(dev), smap, heads, dh, lib=lib)
    close(dqkv, r.grad, 1e-4, 1e-4, "dqkv")


def check_ffn(lib, dev, ntok, d, hidden, arith="f32"):
    rs = np.random.RandomState(3)
    x = rnd(rs, ntok, d)
    ws = (rnd(rs, hidden, d, scale=d ** -0.5), 0.1 * rnd(rs, hidden), rnd(rs, d, hidden, scale=hidden ** -0.5), 0.1 * rnd(rs, d))
    dy = rnd(rs, ntok, d)
    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) for w in ws]
    ref = ffn_reference(xr, *wr)
    ref.backward(dy.double())
    xd, dyd = x.to(dev), dy.to(dev)
    wd = [w.to(dev) for w in ws]
    y = ops.ffn_fwd(xd, *wd, d, hidden, arith=arith, lib=lib)
    close(y, ref, 2e-5, 2e-5, "y")
    gs = [torch.zeros_like(w) for w in wd]
    dx, _ = ops.ffn_bwd(xd, dyd, *wd, gs[0], gs[1], gs[2], gs[3], d, hidden, arith=arith, lib=lib)
    scale = max(1.0, ntok ** 0.5 / 4)
    close(dx, xr.grad, 1e-4, 1e-4, "dx")
    for g, w in zip(gs, wr):
        close(g, w.grad, 1e-4, 1e-4 * scale)
    if arith != "f32":                                          # the two arithmetic variants side by side
        y0 = ops.ffn_fwd(xd, *wd, d, hidden, lib=lib)
        g0 = [torch.zeros_like(w) for w in wd]
        dx0, _ = ops.ffn_bwd(xd, dyd, *wd, g0[0], g0[1], g0[2], g0[3], d, hidden, lib=lib)
        for name, a_, b_ in [("y", y, y0), ("dx", dx, dx0)] + [("g%d" % i, ga, gb) for i, (ga, gb) in enumerate(zip(gs, g0))]:
            err = float((a_ - b_).abs().max()) / (float(b_.abs().max()) + 1e-30)
            assert err < 4e-6, ("bf16x3 vs exact fp32", name, err)
        nbytes = ops.ffn_planes_bytes(d, hidden, lib=lib)       # weight planes split once by the caller: bit-identical
        if nbytes:
            planes = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            jobs = ops.ffn_split_jobs(wd[0], wd[2], d, hidden, planes, lib=lib)
            assert len(jobs) == 3
            arr, n = ops.split_job_array(jobs)
            ops.split_weights_batch(arr, n, xd, lib=lib)
            g2 = [torch.zeros_like(w) for w in wd]
            dx2, _ = ops.ffn_bwd(xd, dyd, *wd, g2[0], g2[1], g2[2], g2[3], d, hidden, arith=arith, planes=planes, lib=lib)
            for name, a_, b_ in [("dx", dx, dx2)] + [("g%d" % i, ga, gb) for i, (ga, gb) in enumerate(zip(gs, g2))]:
                assert torch.equal(a_, b_), ("planes split by the caller", name)


def check_ffn_res(lib, dev, ntok, d, hidden, with_res, arith="f32"):
    """y = FFN(x) + res with the residual from a separate tensor (or none): rat_ffn_fwd_res / rat_ffn_bwd_res(add_dy=0)."""
    rs = np.random.RandomState(13)
    x, res = rnd(rs, ntok, d), rnd(rs, ntok, d)
    ws = (rnd(rs, hidden, d, scale=d ** -0.5), 0.1 * rnd(rs, hidden), rnd(rs, d, hidden, scale=hidden ** -0.5), 0.1 * rnd(rs, d))
    dy = rnd(rs, ntok, d)
    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) for w in ws]
    hdn = xr @ wr[0].t() + wr[1]
    ref = (0.5 * hdn * (1.0 + torch.erf(hdn / 2.0 ** 0.5))) @ wr[2].t() + wr[3]
    if with_res:
        ref = ref + res.double()
    ref.backward(dy.double())
    xd, dyd = x.to(dev), dy.to(dev)
    wd = [w.to(dev) for w in ws]
    y = ops.ffn_fwd_res(xd, res.to(dev) if with_res else None, *wd, d, hidden, arith=arith, lib=lib)
    close(y, ref, 2e-5, 2e-5, "y")
    gs = [torch.zeros_like(w) for w in wd]
    dx, _ = ops.ffn_bwd_res(xd, dyd, *wd, gs[0], gs[1], gs[2], gs[3], d, hidden, add_dy=False, arith=arith, lib=lib)
    scale = max(1.0, ntok ** 0.5 / 4)
    close(dx, xr.grad, 1e-4, 1e-4, "dx")
    for g, w in zip(gs, wr):
        close(g, w.grad, 1e-4, 1e-4 * scale)


def check_ffn_rows(lib, dev, ntok, d, hidden, period):
    """rat_ffn_bwd_res_rows (dy given as the compact rows k * period, zero elsewhere) == rat_ffn_bwd_res on the dense dy, BIT FOR BIT:
    the same kernel, zeros substituted instead of loaded"""
    rs = np.random.RandomState(31)
    x = rnd(rs, ntok, d)
    ws = (rnd(rs, hidden, d, scale=d ** -0.5), 0.1 * rnd(rs, hidden), rnd(rs, d, hidden, scale=hidden ** -0.5), 0.1 * rnd(rs, d))
    nrows = (ntok + period - 1) // period
    rows = rnd(rs, nrows, d)
    dy = torch.zeros(ntok, d)
    dy[::period] = rows
    assert ops.ffn_bwd_rows_supported(d, hidden, "bf16x3", lib)
    xd, wd = x.to(dev), [w.to(dev) for w in ws]
    g1 = [torch.zeros_like(w) for w in wd]
    dx1, _ = ops.ffn_bwd_res(xd, dy.to(dev), *wd, g1[0], g1[1], g1[2], g1[3], d, hidden, add_dy=True, arith="bf16x3", lib=lib)
    g2 = [torch.zeros_like(w) for w in wd]
    dx2, _ = ops.ffn_bwd_rows(xd, rows.to(dev), period, *wd, g2[0], g2[1], g2[2], g2[3], d, hidden, arith="bf16x3", lib=lib)
    assert torch.equal(dx1.cpu(), dx2.cpu()), "dx"
    for a, b in zip(g1, g2):
        assert torch.equal(a.cpu(), b.cpu()), "weight gradients"
    assert float(dx1.abs().sum()) > 0


def check_ffn_dropout(lib, dev, ntok, d, hidden, with_res, add_dy, p=0.3):
    """FeedForward with its two Dropout layers (rat_ffn_fwd_drop / rat_ffn_bwd_drop) against the float64 reference under the SAME masks:
    the masks are counter-based functions of (seed word, element index) — the generator of rat_dropout — so the reference takes them
    from rat_dropout on ones."""
    rs = np.random.RandomState(23)
    x, res = rnd(rs, ntok, d), rnd(rs, ntok, d)
    ws = (rnd(rs, hidden, d, scale=d ** -0.5), 0.1 * rnd(rs, hidden), rnd(rs, d, hidden, scale=hidden ** -0.5), 0.1 * rnd(rs, d))
    dy = rnd(rs, ntok, d)
    words = torch.tensor([0x1234567890ABCDE, 0x0FEDCBA987654321 & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64).to(dev)
    m1 = ops.dropout(torch.ones(ntok, hidden).to(dev), p, words[0:1], lib=lib).cpu().double()
    m2 = ops.dropout(torch.ones(ntok, d).to(dev), p, words[1:2], lib=lib).cpu().double()
    assert 0.5 < float((m1 != 0).double().mean()) < 0.9 and not torch.equal(m1[:, :d] != 0, m2 != 0)
    xr = x.double().requires_grad_(True)
    wr = [w.double().requires_grad_(True) for w in ws]
    hdn = xr @ wr[0].t() + wr[1]
    ref = ((0.5 * hdn * (1.0 + torch.erf(hdn / 2.0 ** 0.5))) * m1 @ wr[2].t() + wr[3]) * m2
    if with_res:
        ref = ref + (xr if add_dy else res.double())
    ref.backward(dy.double())
    xd, dyd = x.to(dev), dy.to(dev)
    wd = [w.to(dev) for w in ws]
    drop = (p, words[0:1], words[1:2])
    y = ops.ffn_fwd_res(xd, (xd if add_dy else res.to(dev)) if with_res else None, *wd, d, hidden, dropout=drop, lib=lib)
    close(y, ref, 2e-5, 2e-5, "y")
    gs = [torch.zeros_like(w) for w in wd]
    dx, _ = ops.ffn_bwd_res(xd, dyd, *wd, gs[0], gs[1], gs[2], gs[3], d, hidden, add_dy=bool(with_res and add_dy), dropout=drop, lib=lib)
    scale = max(1.0, ntok ** 0.5 / 4)
    close(dx, xr.grad, 1e-4, 1e-4, "dx")
    for g, w in zip(gs, wr):
        close(g, w.grad, 1e-4, 1e-4 * scale)


def check_layernorm(lib, dev, nrows, d, stride_mul, with_add):
    """rat_layernorm_fwd / bwd on strided rows against torch's float64 layer_norm."""
    rs = np.random.RandomState(14)
    stride = d * stride_mul
    xfull = rnd(rs, nrows, stride)
    gamma, beta = 1 + 0.1 * rnd(rs, d), 0.1 * rnd(rs, d)
    dy = rnd(rs, nrows, d)
    add = rnd(rs, nrows, stride) if with_add else None
    xr = xfull[:, :d].double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (d,), gr, br, 1e-5)
    ref.backward(dy.double())
    xd = xfull.to(dev)
    y = ops.layernorm_fwd(xd, stride, nrows, gamma.to(dev), beta.to(dev), d, lib=lib)
    close(y, ref, 2e-5, 2e-5, "ln y")
    sentinel = 7.5
    dx = torch.full((nrows, stride), sentinel, dtype=torch.float32, device=dev)
    dg, db = torch.zeros(d, device=dev), torch.zeros(d, device=dev)
    ops.layernorm_bwd(xd, stride, dy.to(dev), gamma.to(dev), dx, stride, dg, db, d, add=add.to(dev) if with_add else None, lib=lib)
    want = xr.grad + (add[:, :d].double() if with_add else 0.0)
    close(dx[:, :d], want, 1e-4, 1e-4, "ln dx")
    if stride_mul > 1:
        assert bool((dx[:, d:] == sentinel).all()), "rows outside the addressed tokens were touched"
    scale = max(1.0, nrows ** 0.5 / 4)
    close(dg, gr.grad, 1e-4, 1e-4 * scale, "dgamma")
    close(db, br.grad, 1e-4, 1e-4 * scale, "dbeta")


def check_bn_relu(lib, dev, M, N, use_bn):
    rs = np.random.RandomState(4)
    z = rnd(rs, M, N)
    gamma, beta = 1 + 0.1 * rnd(rs, N), 0.1 * rnd(rs, N)
    rm, rv = 0.1 * rnd(rs, N), 1 + 0.1 * rnd(rs, N).abs()
    da = rnd(rs, M, N)
    zr = z.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    if use_bn:
        yr = torch.relu(torch.nn.functional.batch_norm(zr, rm_ref, rv_ref, gr, br, training=True, momentum=0.1, eps=1e-5))
    else:
        yr = torch.relu(zr)
    yr.backward(da.double())
    zd, rmd, rvd = z.to(dev), rm.to(dev), rv.to(dev)
    gd, bd = gamma.to(dev), beta.to(dev)
    a, sm, sr = ops.bn_relu_fwd(zd, gd, bd, rmd, rvd, True, use_bn, lib=lib)
    close(a, yr, 1e-5, 1e-5, "bn fwd")
    dg, db = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
    dz = ops.bn_relu_bwd(zd, a, da.to(dev), gd, sm, sr, dg, db, use_bn, lib=lib)
    close(dz, zr.grad, 1e-4, 1e-5, "bn dz")
    if use_bn:
        close(rmd, rm_ref, 1e-5, 1e-6, "running_mean")
        close(rvd, rv_ref, 1e-5, 1e-6, "running_var")
        close(dg, gr.grad, 1e-4, 1e-4, "dgamma")
        close(db, br.grad, 1e-4, 1e-4, "dbeta")
        # eval mode uses the (updated) running stats
        a2, _, _ = ops.bn_relu_fwd(zd, gd, bd, rmd, rvd, False, True, lib=lib)
        ye = torch.relu(torch.nn.functional.batch_norm(z.double(), rm_ref, rv_ref, gamma.double(), beta.double(), training=False, eps=1e-5))
        close(a2, ye, 1e-5, 1e-5, "bn eval")
    out = torch.zeros(N, device=dev)
    ops.colsum(zd, N, out, M, N, lib=lib)
    close(out, z.double().sum(0), 1e-5, 1e-4, "colsum")


def check_bn_strip(lib, dev, M, N, use_bn, act="relu"):
    """column-strip forms (one launch per direction, Linear bias gradient included) against float64 autograd"""
    rs = np.random.RandomState(14)
    z = rnd(rs, M, N) + 3.0 * rnd(rs, N)                       # column means far from zero: the pivot form must not cancel
    gamma, beta = 1 + 0.1 * rnd(rs, N), 0.1 * rnd(rs, N)
    rm, rv = 0.1 * rnd(rs, N), 1 + 0.1 * rnd(rs, N).abs()
    da = rnd(rs, M, N)
    fn = {"relu": torch.relu, "tanh": torch.tanh, "none": lambda t: t}[act]
    zr = z.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    rm_ref, rv_ref = rm.double().clone(), rv.double().clone()
    pre = torch.nn.functional.batch_norm(zr, rm_ref, rv_ref, gr, br, training=True, momentum=0.1, eps=1e-5) if use_bn else zr
    yr = fn(pre)
    yr.backward(da.double())
    zd, rmd, rvd, gd, bd = z.to(dev), rm.to(dev), rv.to(dev), gamma.to(dev), beta.to(dev)
    code = ops.ACT[act]
    if use_bn:
        a, sm, sr = ops.bn_act_fwd_strip(zd, gd, bd, rmd, rvd, True, True, act=code, lib=lib)
    else:
        a, sm, sr = ops.bn_act_fwd_strip(zd, None, None, None, None, True, False, act=code, lib=lib)
    close(a, yr, 1e-5, 1e-5, "strip fwd")
    dg, db, dbl = torch.zeros(N, device=dev), torch.zeros(N, device=dev), torch.full((N,), 7.0, device=dev)
    dz = ops.bn_act_bwd_strip(zd, a, da.to(dev), gd if use_bn else None, sm, sr, dg if use_bn else None, db if use_bn else None, dbl,
                              use_bn, act=code, lib=lib)
    close(dz, zr.grad, 1e-4, 1e-5, "strip dz")
    close(dbl, dz.double().sum(0), 1e-5, 1e-4 * max(1.0, M ** 0.5) * 1e-1, "strip Linear bias gradient")
    if use_bn:
        close(rmd, rm_ref, 1e-5, 1e-6, "running_mean")
        close(rvd, rv_ref, 1e-5, 1e-6, "running_var")
        scale = max(1.0, M ** 0.5)
        close(dg, gr.grad, 1e-4, 1e-5 * scale, "dgamma")
        close(db, br.grad, 1e-4, 1e-5 * scale, "dbeta")
        a2, _, _ = ops.bn_act_fwd_strip(zd, gd, bd, rmd, rvd, False, True, act=code, lib=lib)
        ye = fn(torch.nn.functional.batch_norm(z.double(), rm_ref, rv_ref, gamma.double(), beta.double(), training=False, eps=1e-5))
        close(a2, ye, 1e-5, 1e-5, "strip eval")
        # and against the two-launch kernels: same per-element arithmetic, batch sums in another order
        rm2, rv2 = rm.to(dev), rv.to(dev)
        a3, sm3, sr3 = ops.bn_relu_fwd(zd, gd, bd, rm2, rv2, True, True, act=code, lib=lib)
        close(a, a3, 1e-5, 1e-5, "strip vs two-launch fwd")
        close(sm, sm3, 1e-6, 1e-6, "save_mean")
        close(sr, sr3, 1e-5, 1e-6, "save_rstd")


def check_logit(lib, dev, B, d, with_dnn=True, with_lr=True):
    rs = np.random.RandomState(5)
    T, S, L = 3, 4, 5
    fields = [F(0, 1, 7), F(1, 3, 6, padding_idx=5), F(4, 1, 9)]
    lr_tabs = [rnd(rs, f.vocab, 1, scale=0.3) for f in fields]
    lr_tabs[1][5] = 0
    grid = rnd(rs, B, T, S, d)
    idx = torch.stack([torch.from_numpy(rs.randint(0, [7, 6, 6, 6, 9][c], size=(B, T))) for c in range(L)], -1).int().contiguous()
    fc_w, fc_b = rnd(rs, 1, d, scale=d ** -0.5), 0.1 * rnd(rs, 1)
    dnn_out = rnd(rs, B, 1)
    y = torch.from_numpy(rs.randint(0, 2, size=(B,)).astype(np.float32))
    # reference
    gr = grid.double().requires_grad_(True)
    wr, br, dr = fc_w.double().requires_grad_(True), fc_b.double().requires_grad_(True), dnn_out.double().requires_grad_(True)
    lt = [t.double().requires_grad_(True) for t in lr_tabs]
    z = gr[:, 0, 0] @ wr.t() + br
    if with_dnn:
        z = z + dr
    if with_lr:
        lr = lt[0][idx[:, 0, 0].long()] + torch.nn.functional.embedding(idx[:, 0, 1:4].long(), lt[1], padding_idx=5).sum(-2) + lt[2][idx[:, 0, 4].long()]
        z = z + lr
    p = torch.sigmoid(z)
    loss = torch.nn.functional.binary_cross_entropy(p, y.double().unsqueeze(-1))
    loss.backward()
    # device
    gd, idxd, yd = grid.to(dev), idx.to(dev), y.to(dev)
    fwd, fbd, dd = fc_w.to(dev), fc_b.to(dev), dnn_out.to(dev)
    ltd = [t.to(dev) for t in lr_tabs]
    ftab = ops.field_table(fields, ltd, dev) if with_lr else None
    loss_sum = torch.zeros(1, device=dev)
    yp = ops.logit_fwd(gd, T * S * d, fwd, fbd, dd if with_dnn else None, ftab, 3, idxd, T * L, yd, loss_sum, B, d, lib=lib)
    close(yp, p, 1e-5, 1e-6, "y_pred")
    close(loss_sum, loss.reshape(1), 1e-5, 1e-6, "loss")
    dgrid = torch.zeros_like(gd)
    dfw, dfb = torch.zeros_like(fwd), torch.zeros_like(fbd)
    glt = [torch.zeros_like(t) for t in ltd]
    gftab = ops.field_table(fields, glt, dev) if with_lr else None
    dlogit = ops.logit_bwd(yp, yd, gd, T * S * d, fwd, dgrid, T * S * d, dfw, dfb, gftab, 3, idxd, T * L, 1.0, B, d, lib=lib)
    close(dgrid, gr.grad, 1e-4, 1e-7, "dcls")
    close(dfw, wr.grad, 1e-4, 1e-6, "dfc_w")
    close(dfb, br.grad, 1e-4, 1e-6, "dfc_b")
    if with_dnn:
        close(dlogit, dr.grad, 1e-4, 1e-7, "dlogit")
    if with_lr:
        for g, t in zip(glt, lt):
            close(g, t.grad, 1e-4, 1e-7, "lr grad")
    if with_dnn:
        # the DNN's one-output Linear evaluated inside the launches (rat_logit_fwd_dnn / rat_logit_bwd_dnn): K = 52 (vector path) and
        # K = 7 with an odd leading dimension (scalar path)
        for K, ld in ((52, 52), (7, 9)):
            a = rnd(rs, B, ld)
            w_o, b_o = rnd(rs, 1, K, scale=K ** -0.5), 0.1 * rnd(rs, 1)
            dn = (a[:, :K].double() @ w_o.double().t() + b_o.double())
            z2 = grid.double()[:, 0, 0] @ fc_w.double().t() + fc_b.double() + dn
            if with_lr:
                z2 = z2 + lr.detach()
            p2 = torch.sigmoid(z2)
            loss2 = torch.nn.functional.binary_cross_entropy(p2, y.double().unsqueeze(-1))
            ls2 = torch.zeros(1, device=dev)
            yp2 = ops.logit_fwd(gd, T * S * d, fwd, fbd, None, ftab, 3, idxd, T * L, yd, ls2, B, d,
                                dnn_last=(a.to(dev), ld, K, w_o.to(dev), b_o.to(dev)), lib=lib)
            close(yp2, p2, 1e-5, 1e-6, "y_pred (DNN output layer inside)")
            close(ls2, loss2.reshape(1), 1e-5, 1e-6, "loss (DNN output layer inside)")
        dgrid2, dfw2, dfb2, dob = torch.zeros_like(gd), torch.zeros_like(fwd), torch.zeros_like(fbd), torch.full((1,), 0.5, device=dev)
        dl2 = ops.logit_bwd(yp, yd, gd, T * S * d, fwd, dgrid2, T * S * d, dfw2, dfb2, None, 3, idxd, T * L, 1.0, B, d, ddnn_b=dob, lib=lib)
        close(dl2, dlogit, 1e-6, 1e-9, "dlogit (dnn variant)")
        close(dob - 0.5, dfb2.reshape(1), 1e-4, 1e-6, "ddnn_b accumulates the sum dfc_b gets")


def check_bn_strip_outer(lib, dev, M, N, use_bn):
    """last hidden layer + the one-output Linear behind it: da = dl (x) w formed inside the launch, dw returned"""
    rs = np.random.RandomState(15)
    z = rnd(rs, M, N)
    gamma, beta = (1 + 0.1 * rnd(rs, N)).to(dev), (0.1 * rnd(rs, N)).to(dev)
    rm, rv = torch.zeros(N, device=dev), torch.ones(N, device=dev)
    dl, w = rnd(rs, M, 1), rnd(rs, 1, N)
    zd, dld, wd = z.to(dev), dl.to(dev), w.to(dev)
    if use_bn:
        a, sm, sr = ops.bn_act_fwd_strip(zd, gamma, beta, rm, rv, True, True, lib=lib)
    else:
        a, sm, sr = ops.bn_act_fwd_strip(zd, None, None, None, None, True, False, lib=lib)
    da = (dl.double() @ w.double()).float().to(dev)
    dg, db, dbl = (torch.zeros(N, device=dev) for _ in range(3))
    ref = ops.bn_act_bwd_strip(zd, a, da, gamma if use_bn else None, sm, sr, dg if use_bn else None, db if use_bn else None, dbl, use_bn, lib=lib)
    dg2, db2, dbl2, dw = (torch.zeros(N, device=dev) for _ in range(4))
    dz = ops.bn_act_bwd_strip_outer(zd, a, dld, wd, dw, gamma if use_bn else None, sm, sr, dg2 if use_bn else None, db2 if use_bn else None,
                                    dbl2, use_bn, lib=lib)
    scale = max(1.0, M ** 0.5)
    close(dz, ref, 1e-5, 1e-6, "outer dz")
    close(dbl2, dbl, 1e-4, 1e-5 * scale, "outer dbias")
    close(dw, (dl.double().t() @ a.double().cpu()).reshape(-1), 1e-5, 1e-5 * scale, "dw = dl^T a")
    if use_bn:
        close(dg2, dg, 1e-4, 1e-5 * scale, "outer dgamma")
        close(db2, db, 1e-4, 1e-5 * scale, "outer dbeta")


def check_optim(lib, dev, n):
    rs = np.random.RandomState(6)
    w, g = rnd(rs, n), rnd(rs, n, scale=3.0)
    m, v = 0.1 * rnd(rs, n), 0.01 * rnd(rs, n).abs()
    lam = 0.02
    wd, gd, md, vd = (t.clone().to(dev) for t in (w, g, m, v))
    reg = torch.zeros(1, device=dev)
    ops.l2_reg(wd, gd, lam, reg, lib=lib)
    g2 = g.double() + lam * w.double()
    close(gd, g2, 1e-6, 1e-6, "reg grad")
    close(reg, (0.5 * lam * (w.double() ** 2).sum()).reshape(1), 1e-5, 1e-6, "reg value")
    nsq = torch.zeros(1, device=dev)
    ops.sumsq(gd, nsq, lib=lib)
    close(nsq, (g2 ** 2).sum().reshape(1), 1e-5, 1e-5, "sumsq")
    step, lr, b1, b2, eps, max_norm = 3, 1e-3, 0.9, 0.999, 1e-8, 10.0
    ops.clip_adam(wd, gd, md, vd, nsq, max_norm, lr, b1, b2, eps, step, lib=lib)
    coef = min(1.0, max_norm / (float(torch.sqrt((g2 ** 2).sum())) + 1e-6))
    gc_ = g2 * coef
    mr = b1 * m.double() + (1 - b1) * gc_
    vr = b2 * v.double() + (1 - b2) * gc_ * gc_
    wr = w.double() - lr / (1 - b1 ** step) * mr / (torch.sqrt(vr) / (1 - b2 ** step) ** 0.5 + eps)
    close(md, mr, 1e-5, 1e-7, "m")
    close(vd, vr, 1e-5, 1e-8, "v")
    close(wd, wr, 1e-5, 1e-6, "w")


def check_step_begin(lib, dev):
    """rat_step_begin == rat_adam_tick + cleared accumulator scalars + advanced BatchNorm counters"""
    lr = torch.tensor([3e-3], device=dev)
    s1, s2 = torch.tensor([4], dtype=torch.int32, device=dev), torch.tensor([4], dtype=torch.int32, device=dev)
    h1, h2 = torch.zeros(4, device=dev), torch.zeros(4, device=dev)
    scal = torch.tensor([1.0, 2.0, 3.0, 4.0], device=dev)
    counts = torch.tensor([7, 7, 9], dtype=torch.int64, device=dev)
    ops.adam_tick(s1, lr, 0.9, 0.999, h1, lib=lib)
    ops.step_begin(s2, lr, 0.9, 0.999, h2, scal, counts, lib=lib)
    assert int(s1) == int(s2) == 5 and torch.equal(h1.cpu(), h2.cpu())
    assert float(scal.abs().sum()) == 0.0 and counts.tolist() == [8, 8, 10]
    ops.step_begin(s2, lr, 0.9, 0.999, h2, scal, None, lib=lib)
    assert int(s2) == 6
    want = 3e-3 / (1 - 0.9 ** 6)
    assert abs(float(h2[0]) - want) < 1e-6 * want and abs(float(h2[1]) - (1 - 0.999 ** 6) ** -0.5) < 1e-4, h2


def check_deferred_reductions(lib, dev, arith="f32"):
    """three backward calls (two feed-forward layers, one attention layer) with their slab reductions recorded and run as ONE launch
    (rat_reduce_defer_begin / _end) == the same calls with their own reduction launches, bit for bit"""
    rs = np.random.RandomState(41)
    d, hidden, ntok = 64, 128, 300
    x, dy = rnd(rs, ntok, d).to(dev), rnd(rs, ntok, d).to(dev)
    ws = [w.to(dev) for w in (rnd(rs, hidden, d, scale=d ** -0.5), 0.1 * rnd(rs, hidden), rnd(rs, d, hidden, scale=hidden ** -0.5), 0.1 * rnd(rs, d))]
    B, T, S, heads, dh = 3, 4, 5, 8, 10
    xa, dya = rnd(rs, B, T, S, d).to(dev), rnd(rs, B, T, S, d).to(dev)
    aw = [t.to(dev) if t is not None else None for t in attn_weights(rs, d, heads, dh, True)]
    params = ops.attn_params(*aw)
    smap = ops.intra_map(B, T, S)
    y, o, lse = ops.attn_fwd(xa, params, smap, d, heads, dh, save=True, arith=arith, lib=lib)

    def run(defer):
        g1 = [torch.zeros_like(w) for w in ws]
        g2 = [torch.zeros_like(w) for w in ws]
        ga = [torch.zeros_like(t) for t in aw]
        with ops.deferred_reductions(x, lib, enabled=defer):
            dx1, w1 = ops.ffn_bwd_res(x, dy, *ws, g1[0], g1[1], g1[2], g1[3], d, hidden, add_dy=True, arith=arith, lib=lib)
            dx2, w2 = ops.ffn_bwd_res(x, dx1, *ws, g2[0], g2[1], g2[2], g2[3], d, hidden, add_dy=False, arith=arith, lib=lib)
            dxa, w3 = ops.attn_bwd(xa, dya, o, lse, params, ops.attn_params(*ga), smap, d, heads, dh, arith=arith, lib=lib)
            if defer:
                assert float(g1[0].abs().sum()) == 0.0 or dev == "cuda"     # (the launches are asynchronous on the GPU; on the emulator nothing ran yet)
        return [dx1, dx2, dxa] + g1 + g2 + ga

    a, b = run(False), run(True)
    for u, v in zip(a, b):
        assert torch.equal(u.cpu(), v.cpu())
    assert float(b[3].abs().sum()) > 0 and float(b[-1].abs().sum()) > 0
