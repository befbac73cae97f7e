import sys, os
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/www24-rat_amd"]
import torch, numpy as np
import test_gpu_configs as T
from oracle import rat_m2_oracle as orc
name, nslice = sys.argv[1], int(sys.argv[2])
for mode in ("atomic", "sorted"):
    spec, fm, model, batch = T._build(name, batch_norm=False)
    model._grad_mode = mode if spec["d"] % 4 == 0 else "atomic"
    sub = tuple(t[:nslice] for t in batch)
    model.train(); model.optimizer.zero_grad()
    loss = model.get_total_loss(sub); loss.backward(); torch.cuda.synchronize()
    w = {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}
    cfg = T._oracle_cfg(orc, spec, fm, batch_norm=False, embedding_regularizer=0.0)
    ref_loss, _, ref_grads, _ = orc.loss_and_grads(w, sub[0], sub[1], cfg, training=True)
    print(mode, "loss", float(loss), float(ref_loss))
    rows = []
    for k, p in model.named_parameters():
        if p.grad is None: continue
        got, ref = p.grad.detach().cpu().double(), ref_grads[k].double()
        sc = float(ref.abs().max()); d = (got - ref).abs()
        rows.append((float(d.max()) / sc, k, sc, int((d > 3e-4 * sc).sum()), d.numel()))
    for r in sorted(rows, reverse=True)[:14]: print("  %.2e %-70s scale %.2e  n_bad %d / %d" % r)
    if mode == "atomic":
        k = "embedding_layer.embedding_layer.embedding_layer.c00.weight"
        got, ref = model._params[k].grad.detach().cpu().double(), ref_grads[k].double()
        d = (got - ref).abs().max(dim=1).values
        top = torch.topk(d, 5).indices
        X = sub[0][:, :, 0].long()
        for r in top.tolist():
            occ = (X == r).nonzero()
            print("   row", r, "err", float(d[r]), "occurrences (b,t):", occ.tolist()[:8])
