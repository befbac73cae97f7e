"""Diagnostic (not product, not a test): how far apart do REPEATED runs of the same 80-step training trajectory end?
HIP runs differ run to run through the order of their fp32 atomics; the fp32 oracle through torch's thread count.
Prints held-out logloss / AUC deltas against the fp64 oracle for every run (see tests/test_gpu_trajectory.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "www24-rat_amd")):
    sys.path.insert(0, p)
from collections import OrderedDict
import numpy as np
import torch
from oracle import rat_m2_oracle as orc
from rat_amd import data as rd
from rat_amd import models, synthetic
from rat_amd.base_model import seed_everything
from rat_amd.features import FeatureMap
from rat_amd.metrics import evaluate_metrics

spec = dict(synthetic.WORKLOADS["mltag_like_K10_d16_B256"])
fm = FeatureMap.from_specs("trajectory", OrderedDict(("c%02d" % i, {"source": "", "type": "categorical", "vocab_size": 40, "index": i}) for i in range(spec["F"])))
B, K, steps = spec["batch"], spec["K"], 80
data, idx, val, lens = rd.synthetic_split(fm, B * steps + 2048, K, seed=4)
batches = list(rd.RetrievalBatches(data, data, idx, val, lens, B, shuffle=False))
train, held = batches[:steps], batches[steps:]
yt = torch.cat([b[1][:, 0].double() for b in held]).numpy()

def hip(arith, **kw):
    seed_everything(2021)
    m = models.RAT_m2(fm, **dict(synthetic.model_kwargs(spec, gpu=0), arith=arith, **kw))
    with torch.no_grad():
        m._flat[:m._n_feat].mul_(2000.0)
    w0 = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    m.train()
    for b in train:
        m.train_step(b)
    m.eval()
    with torch.no_grad():
        yp = torch.cat([m.forward(b)["y_pred"].reshape(-1).double().cpu() for b in held]).numpy()
    return w0, m, yp

w0, m, _ = hip("f32")
cfg = orc.Config(fields=orc.fields_from_specs(fm.feature_specs), embedding_dim=spec["d"], num_heads=spec["num_heads"], dim_head=spec["dim_head"],
                 depth=spec["depth"], scale_dim=spec["scale_dim"], dnn_hidden_units=tuple(spec["dnn_hidden_units"]), batch_norm=spec["batch_norm"],
                 use_wide=spec["use_wide"], embedding_regularizer=m._cfg["lam_emb"], learning_rate=spec["learning_rate"])

def oracle(dtype, nt):
    torch.set_num_threads(nt)
    w = {k: (v.to(dtype) if v.is_floating_point() else v.clone()) for k, v in w0.items()}
    state = {}
    for s, b in enumerate(train):
        w, loss, *_ = orc.train_step(w, b[0].double(), b[1].double(), cfg, state, s + 1)
    with torch.no_grad():
        return torch.cat([orc.forward(w, b[0].double(), b[1].double(), cfg, training=False).reshape(-1).double() for b in held]).numpy()

truth = oracle(torch.float64, 16)
mt = evaluate_metrics(yt, truth, ["AUC", "logloss"])
def report(name, yp):
    mm = evaluate_metrics(yt, yp, ["AUC", "logloss"])
    print("%-28s dAUC %+.2e  dlogloss %+.2e  mean(pred - fp64) %+.2e  rms %.2e" % (name, mm["AUC"] - mt["AUC"], mm["logloss"] - mt["logloss"],
          float(np.mean(yp - truth)), float(np.sqrt(np.mean((yp - truth) ** 2)))), flush=True)
for nt in (1, 2, 3, 4, 5, 6, 7, 8, 12, 16):
    report("oracle fp32 %d threads" % nt, oracle(torch.float32, nt))
for i in range(4):
    report("HIP f32 run %d" % i, hip("f32")[2])
for i in range(4):
    report("HIP bf16x3(auto) run %d" % i, hip("auto")[2])
for i in range(2):
    report("HIP f32 sorted run %d" % i, hip("f32", embedding_grad="sorted")[2])
report("HIP f32 no graph", hip("f32", use_graph=False)[2] if False else hip("f32")[2])
