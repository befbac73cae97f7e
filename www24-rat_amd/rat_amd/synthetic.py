"""Deterministic synthetic workloads shaped like BASELINE.json's configs: uniform ids per field,
Bernoulli(0.5) labels, reference init rules under seed_everything(2021).  No dataset is available offline."""
from collections import OrderedDict

import torch

from .features import FeatureMap

KKBOX_HYPER = dict(num_heads=8, dim_head=10, depth=4, scale_dim=2, dnn_hidden_units=[400, 400, 400], batch_norm=True,
                   use_wide=True, learning_rate=1e-3)

WORKLOADS = {
    # BASELINE.json configs[1]: the single-GPU configuration the metric is quoted on
    "synthetic_F20_V1M_K10_d64_B4096": dict(F=20, total_vocab=1_000_000, K=10, d=64, batch=4096, **KKBOX_HYPER),
    # BASELINE.json configs[0] shape (CPU-runnable plumbing case)
    "mltag_like_K10_d16_B256": dict(F=3, total_vocab=90_000, K=10, d=16, batch=256, num_heads=2, dim_head=10, depth=4,
                                    scale_dim=4, dnn_hidden_units=[400, 400, 400], batch_norm=False, use_wide=True,
                                    learning_rate=1e-3),
    # BASELINE.json configs[2] shape (KKBox: 13 fields, d = 64 as that config line states) and configs[4] shape (Tmall: 8 fields,
    # K = 30 retrieved, the shipped 32 heads x 10 -> heads*dim_head = 320: served by the composed attention path); 1.5 M rows each
    "kkbox_like_F13_K10_d64_B4096": dict(F=13, total_vocab=92_000, K=10, d=64, batch=4096, **KKBOX_HYPER),
    "tmall_like_F8_K30_d64_h32_B4096": dict(F=8, total_vocab=1_500_000, K=30, d=64, batch=4096, num_heads=32, dim_head=10, depth=4,
                                            scale_dim=2, dnn_hidden_units=[200, 80], batch_norm=True, use_wide=True, learning_rate=1e-3),
    # BASELINE.json configs[3] shape per GPU (F = 40, global batch 8192 over 8 GPUs = 1024 per rank) at a tenth of its vocabulary:
    # the full 100 M-row table (25.6 GB + 77 GB of gradient / Adam state per replica) fits one MI355X, but its DENSE gradient
    # all-reduce does not scale — row-sparse exchange + lazy Adam for that config are future work
    "synthetic_F40_V10M_K10_d64_B1024": dict(F=40, total_vocab=10_000_000, K=10, d=64, batch=1024, **KKBOX_HYPER),
    # BASELINE.json configs[3] per GPU at its FULL vocabulary: 100 M rows x 64 floats = 25.6 GB of feature tables (+ 51.2 GB of Adam
    # moments; no dense gradient: row-sparse lists + rat_adam_rows), global batch 8192 over 8 GPUs = 1024 per rank
    "synthetic_F40_V100M_K10_d64_B1024": dict(F=40, total_vocab=100_000_000, K=10, d=64, batch=1024, embedding_regularizer=0.0,
                                              embedding_grad="sparse", **KKBOX_HYPER),
    # the reference's own KKBox experiment (configs/RAT_m2/kkbox_x1/model_config.yaml: embedding_dim 40, K = 5): the bf16x3 kernels
    # inside their 64-wide tiles
    "kkbox_real_F13_K5_d40_B4096": dict(F=13, total_vocab=92_000, K=5, d=40, batch=4096, **KKBOX_HYPER),
    # the two other shipped experiments at their own geometry (configs/RAT_m2/movielenslatest_x1, tmall_x1_002; topK = 5 in
    # configs/datasets/*.yaml): embedding_dim 10 — generic exact-fp32 kernels (Tmall's 32 heads x 10 in groups of 8)
    "movielens_real_F3_K5_d10_B4096": dict(F=3, total_vocab=90_000, K=5, d=10, batch=4096, num_heads=2, dim_head=10, depth=4,
                                           scale_dim=4, dnn_hidden_units=[400, 400, 400], batch_norm=False, use_wide=True,
                                           learning_rate=1e-3),
    "tmall_real_F9_K5_d10_h32_B4096": dict(F=9, total_vocab=1_500_000, K=5, d=10, batch=4096, num_heads=32, dim_head=10, depth=4,
                                           scale_dim=2, dnn_hidden_units=[200, 80], batch_norm=True, use_wide=True, learning_rate=1e-3),
    # bench.py --dry-run-cpu (plumbing check of the multi-process launch over gloo + the host-emulated kernels)
    "dryrun": dict(F=3, total_vocab=90, K=2, d=16, batch=8, num_heads=2, dim_head=10, depth=1, scale_dim=2,
                   dnn_hidden_units=[16], batch_norm=True, use_wide=True, learning_rate=1e-3),
    # tiny: smoke / CI
    "tiny": dict(F=5, total_vocab=500, K=3, d=16, batch=32, num_heads=2, dim_head=10, depth=2, scale_dim=2,
                 dnn_hidden_units=[32, 16], batch_norm=True, use_wide=True, learning_rate=1e-3),
}


def feature_map_for(name, spec):
    vocab = spec["total_vocab"] // spec["F"]
    specs = OrderedDict()
    for i in range(spec["F"]):
        specs["c%02d" % i] = {"source": "", "type": "categorical", "vocab_size": vocab, "index": i}
    return FeatureMap.from_specs(name, specs)


def model_kwargs(spec, gpu, embedding_regularizer=0.0005, model_root="/tmp/rat_amd_models/"):
    embedding_regularizer = spec.get("embedding_regularizer", embedding_regularizer)
    extra = {"embedding_grad": spec["embedding_grad"]} if "embedding_grad" in spec else {}
    return dict(**extra, model_id="RAT_m2_bench", gpu=gpu, task="binary_classification", learning_rate=spec["learning_rate"],
                embedding_dim=spec["d"], dnn_hidden_units=list(spec["dnn_hidden_units"]), dnn_activations="relu",
                num_heads=spec["num_heads"], dim_head=spec["dim_head"], depth=spec["depth"], scale_dim=spec["scale_dim"],
                dropout=0.0, emb_dropout=0.0, net_dropout=0, batch_norm=spec["batch_norm"], use_wide=spec["use_wide"],
                embedding_regularizer=embedding_regularizer, net_regularizer=0, retrieval_augmented=True,
                retrieval_configs={"topK": spec["K"], "label_wise": False}, model_root=model_root,
                metrics=["AUC", "logloss"], verbose=0, optimizer="adam", loss="binary_crossentropy", monitor="AUC",
                monitor_mode="max", patience=2, every_x_epochs=1, save_best_only=True, seed=2021)


def make_batch(spec, feature_map, seed=0, batch=None, device=None, as_float64=True):
    """The DataLoader 4-tuple (data_generator.py:66-78).  With ``device`` set, ids are handed over as int32 tensors
    already resident in HBM (what bench.py times); otherwise float64 host tensors exactly like the reference's."""
    B = batch or spec["batch"]
    T = spec["K"] + 1
    g = torch.Generator().manual_seed(seed)
    cols = [torch.randint(0, s["vocab_size"], (B, T, 1), generator=g) for s in feature_map.feature_specs.values()]
    X = torch.cat(cols, dim=-1)
    y = torch.randint(0, 2, (B, T), generator=g)
    rv = torch.rand(B, T - 1, generator=g, dtype=torch.float64)
    rl = torch.full((B,), T - 1, dtype=torch.int64)
    if device is not None:
        return (X.to(torch.int32).to(device), y.to(torch.float32).to(device), rv.to(device), rl.to(device))
    if as_float64:
        return (X.to(torch.float64), y.to(torch.float64), rv, rl)
    return (X, y, rv, rl)
