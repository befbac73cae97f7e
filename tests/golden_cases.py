"""Case descriptions shared by tests/golden/make_golden.py (reference side) and the tests.

A case = a dataset *shape* (fields, columns, vocab sizes), the RAT_m2 hyper-parameters, and
seeds.  Weights and inputs are produced from ``numpy.random.RandomState`` (a frozen, versioned
stream) so that the committed ``.npz`` files only need to hold the reference's OUTPUTS.
"""
from collections import OrderedDict

import numpy as np

# ----------------------------------------------------------------------------- cases
def _cat(name, vocab, **kw):
    d = {"name": name, "type": "categorical", "vocab_size": vocab}
    d.update(kw)
    return d


def _seq(name, vocab, max_len=3):
    return {"name": name, "type": "sequence", "vocab_size": vocab, "max_len": max_len,
            "encoder": "MaskedSumPooling"}


CASES = [
    # tiny: 2 categorical + 1 sequence + 1 categorical with an explicit padding row; BN on.
    dict(name="tiny_seq_bn", batch=6, topk=3, init_seed=2021, data_seed=11, weight_seed=12, full_limit=1 << 20,
         fields=[_cat("a", 7), _cat("b", 5), _seq("c", 6), _cat("e", 9, padding_idx=8)],
         embedding_dim=8, num_heads=2, dim_head=4, depth=2, scale_dim=2, dnn_hidden_units=[16, 8],
         batch_norm=True, use_wide=True, embedding_regularizer=0.01, net_regularizer=0),
    # MovieLens-Tag shape (configs/RAT_m2/movielenslatest_x1/model_config.yaml): F=3, d=10, h=2x10, depth 4, scale 4.
    dict(name="mltag_shape", batch=8, topk=5, init_seed=2021, data_seed=21, weight_seed=22, full_limit=4096,
         fields=[_cat("user_id", 50), _cat("item_id", 40), _cat("tag_id", 30)],
         embedding_dim=10, num_heads=2, dim_head=10, depth=4, scale_dim=4, dnn_hidden_units=[32, 16, 16],
         batch_norm=False, use_wide=True, embedding_regularizer=0.03, net_regularizer=0),
    # KKBox shape (configs/datasets/kkbox_x1.yaml:64-77): 13 fields / 17 columns, two 3-id bags; BN on; 8 heads.
    dict(name="kkbox_shape", batch=8, topk=5, init_seed=2021, data_seed=31, weight_seed=32, full_limit=2048,
         fields=[_cat("msno", 23), _cat("song_id", 31), _cat("source_system_tab", 9), _cat("source_screen_name", 11),
                 _cat("source_type", 12), _cat("city", 8), _cat("gender", 4), _cat("registered_via", 6),
                 _cat("language", 10), _seq("genre_ids", 17), _seq("artist_name", 19), _cat("isrc", 13),
                 _cat("bd", 7)],
         embedding_dim=16, num_heads=8, dim_head=10, depth=2, scale_dim=2, dnn_hidden_units=[24, 24],
         batch_norm=True, use_wide=True, embedding_regularizer=0.0005, net_regularizer=0),
    # Tmall shape (configs/RAT_m2/tmall_x1_002): F=8, K large-ish, d=10, many heads; DNN [.,.] + BN.
    dict(name="tmall_shape", batch=5, topk=12, init_seed=2021, data_seed=41, weight_seed=42, full_limit=2048,
         fields=[_cat("f%d" % i, 15 + 3 * i) for i in range(8)],
         embedding_dim=10, num_heads=4, dim_head=10, depth=2, scale_dim=2, dnn_hidden_units=[20, 8],
         batch_norm=True, use_wide=True, embedding_regularizer=0.07, net_regularizer=0),
    # North-star shape (BASELINE.json configs[1]) at depth 1 and a toy batch: F=20, K=10, d=64, 8x10 heads.
    dict(name="northstar_shape", batch=3, topk=10, init_seed=2021, data_seed=51, weight_seed=52, full_limit=1024,
         fields=[_cat("c%02d" % i, 37) for i in range(20)],
         embedding_dim=64, num_heads=8, dim_head=10, depth=1, scale_dim=2, dnn_hidden_units=[16, 16],
         batch_norm=True, use_wide=True, embedding_regularizer=0.0005, net_regularizer=0),
    # no wide, no DNN, no regulariser, single head with dim_head == d -> Attention has no output projection.
    dict(name="bare_no_proj", batch=4, topk=2, init_seed=7, data_seed=61, weight_seed=62, full_limit=1 << 20,
         fields=[_cat("p", 6), _cat("q", 5)],
         embedding_dim=8, num_heads=1, dim_head=8, depth=1, scale_dim=1, dnn_hidden_units=[],
         batch_norm=False, use_wide=False, embedding_regularizer=0, net_regularizer=0),
]

# The REAL Tmall head geometry (configs/RAT_m2/tmall_x1_002/model_config.yaml: 32 heads x 10, d = 10): heads*dim_head = 320 is too
# wide for the fused attention kernel's LDS tile -> the composed path (LayerNorm -> GEMM -> attention core with strided sequences
# -> GEMM) serves both the intra and the cross phase.
CASES += [
    dict(name="tmall_real_heads", batch=4, topk=6, init_seed=2021, data_seed=151, weight_seed=152, full_limit=2048,
         fields=[_cat("f%d" % i, 11 + 2 * i) for i in range(8)],
         embedding_dim=10, num_heads=32, dim_head=10, depth=2, scale_dim=2, dnn_hidden_units=[20, 8],
         batch_norm=True, use_wide=True, embedding_regularizer=0.07, net_regularizer=0),
]

# RAT_m1 (cascaded intra / cross transformers, RAT_m1.py) — same fields / seeds machinery.
CASES += [
    dict(name="m1_tiny_seq", model="RAT_m1", batch=6, topk=3, init_seed=2021, data_seed=71, weight_seed=72, full_limit=1 << 20,
         fields=[_cat("a", 7), _cat("b", 5), _seq("c", 6), _cat("e", 9, padding_idx=8)],
         embedding_dim=8, num_heads=2, dim_head=4, depth=2, scale_dim=2, dnn_hidden_units=[16, 8],
         batch_norm=True, use_wide=True, embedding_regularizer=0.01, net_regularizer=0),
    dict(name="m1_northstar_shape", model="RAT_m1", batch=3, topk=10, init_seed=2021, data_seed=81, weight_seed=82, full_limit=1024,
         fields=[_cat("c%02d" % i, 37) for i in range(20)],
         embedding_dim=64, num_heads=8, dim_head=10, depth=1, scale_dim=2, dnn_hidden_units=[16, 16],
         batch_norm=True, use_wide=True, embedding_regularizer=0.0005, net_regularizer=0),
    dict(name="m1_bare", model="RAT_m1", batch=4, topk=2, init_seed=7, data_seed=91, weight_seed=92, full_limit=1 << 20,
         fields=[_cat("p", 6), _cat("q", 5)],
         embedding_dim=16, num_heads=1, dim_head=16, depth=1, scale_dim=2, dnn_hidden_units=[],
         batch_norm=False, use_wide=False, embedding_regularizer=0, net_regularizer=0),
]

# RAT_m3 (parallel intra / cross attention sharing W_q, heads/2 heads of width 2*dim_head, mean fusion — RAT_m3.py)
CASES += [
    dict(name="m3_tiny_seq", model="RAT_m3", batch=6, topk=3, init_seed=2021, data_seed=101, weight_seed=102, full_limit=1 << 20,
         fields=[_cat("a", 7), _cat("b", 5), _seq("c", 6), _cat("e", 9, padding_idx=8)],
         embedding_dim=8, num_heads=2, dim_head=4, depth=2, scale_dim=2, dnn_hidden_units=[16, 8],
         batch_norm=True, use_wide=True, embedding_regularizer=0.01, net_regularizer=0),
    dict(name="m3_northstar_shape", model="RAT_m3", batch=3, topk=10, init_seed=2021, data_seed=111, weight_seed=112, full_limit=1024,
         fields=[_cat("c%02d" % i, 37) for i in range(20)],
         embedding_dim=64, num_heads=8, dim_head=10, depth=1, scale_dim=2, dnn_hidden_units=[16, 16],
         batch_norm=True, use_wide=True, embedding_regularizer=0.0005, net_regularizer=0),
    dict(name="m3_mltag_shape", model="RAT_m3", batch=8, topk=5, init_seed=2021, data_seed=121, weight_seed=122, full_limit=4096,
         fields=[_cat("user_id", 50), _cat("item_id", 40), _cat("tag_id", 30)],
         embedding_dim=10, num_heads=2, dim_head=10, depth=2, scale_dim=4, dnn_hidden_units=[32, 16],
         batch_norm=False, use_wide=True, embedding_regularizer=0.03, net_regularizer=0),
]

# Round 6: RAT_m3 outside the one geometry its fused kernel serves.  (a) the head geometry of the README's RAT_PA-on-Tmall run
# (configs/RAT_m2/tmall_x1_002/model_config.yaml:23 — 32 heads x 10 at d = 10 -> RAT_m3 runs 16 heads of width 20, inner 320): head
# groups; (b) dim_head 16 -> heads of width 32, wider than any fused instantiation: the composed path.
CASES += [
    dict(name="m3_tmall_real_heads", model="RAT_m3", batch=4, topk=6, init_seed=2021, data_seed=161, weight_seed=162, full_limit=2048,
         fields=[_cat("f%d" % i, 11 + 2 * i) for i in range(8)],
         embedding_dim=10, num_heads=32, dim_head=10, depth=2, scale_dim=2, dnn_hidden_units=[20, 8],
         batch_norm=True, use_wide=True, embedding_regularizer=0.07, net_regularizer=0),
    dict(name="m3_wide_dim_head", model="RAT_m3", batch=5, topk=4, init_seed=2021, data_seed=171, weight_seed=172, full_limit=2048,
         fields=[_cat("a", 7), _cat("b", 5), _seq("c", 6), _cat("e", 9, padding_idx=8)],
         embedding_dim=16, num_heads=4, dim_head=16, depth=2, scale_dim=2, dnn_hidden_units=[16, 8],
         batch_norm=False, use_wide=True, embedding_regularizer=0.01, net_regularizer=0),
]

# RAT_m0 (one Transformer over the joint (t n) sequence of a sample — RAT_m0.py)
CASES += [
    dict(name="m0_tiny_seq", model="RAT_m0", batch=6, topk=3, init_seed=2021, data_seed=131, weight_seed=132, full_limit=1 << 20,
         fields=[_cat("a", 7), _cat("b", 5), _seq("c", 6), _cat("e", 9, padding_idx=8)],
         embedding_dim=8, num_heads=2, dim_head=4, depth=2, scale_dim=2, dnn_hidden_units=[16, 8],
         batch_norm=True, use_wide=True, embedding_regularizer=0.01, net_regularizer=0),
    dict(name="m0_northstar_shape", model="RAT_m0", batch=3, topk=10, init_seed=2021, data_seed=141, weight_seed=142, full_limit=1024,
         fields=[_cat("c%02d" % i, 37) for i in range(20)],
         embedding_dim=64, num_heads=8, dim_head=10, depth=1, scale_dim=2, dnn_hidden_units=[16, 16],
         batch_norm=True, use_wide=True, embedding_regularizer=0.0005, net_regularizer=0),
]

# The three shipped experiments (exps/RAT_m2/*/*.log "Total number of parameters").
# Round 4: the constructor options the shipped configs never vary — hidden activations other than ReLU (per-layer list, with and
# without BatchNorm, a layer WITHOUT an activation module: the Sequential's indices shift), the other optimizers get_optimizer can
# build (torch defaults), task = "regression" with F.mse_loss (no output activation).
_OPT_FIELDS = [_cat("a", 7), _cat("b", 5), _seq("c", 6), _cat("e", 9, padding_idx=8)]
CASES += [
    dict(name="act_tanh_sigmoid_leaky_bn", batch=6, topk=3, init_seed=2021, data_seed=211, weight_seed=212, full_limit=1 << 20,
         fields=_OPT_FIELDS, embedding_dim=8, num_heads=2, dim_head=4, depth=1, scale_dim=2, dnn_hidden_units=[16, 8, 8],
         dnn_activations=["Tanh", "sigmoid", "LeakyReLU"], batch_norm=True, use_wide=True, embedding_regularizer=0.01, net_regularizer=0),
    dict(name="act_elu_none_relu", batch=6, topk=3, init_seed=2021, data_seed=221, weight_seed=222, full_limit=1 << 20,
         fields=_OPT_FIELDS, embedding_dim=8, num_heads=2, dim_head=4, depth=1, scale_dim=2, dnn_hidden_units=[16, 8, 8],
         dnn_activations=["ELU", None, "relu"], batch_norm=False, use_wide=True, embedding_regularizer=0.01, net_regularizer=0),
    dict(name="opt_sgd", batch=6, topk=3, init_seed=2021, data_seed=231, weight_seed=232, full_limit=1 << 20, optimizer="SGD",
         learning_rate=0.05, fields=_OPT_FIELDS, embedding_dim=8, num_heads=2, dim_head=4, depth=1, scale_dim=2, dnn_hidden_units=[16, 8],
         batch_norm=True, use_wide=True, embedding_regularizer=0.01, net_regularizer=0),
    dict(name="opt_adagrad", batch=6, topk=3, init_seed=2021, data_seed=241, weight_seed=242, full_limit=1 << 20, optimizer="Adagrad",
         learning_rate=0.01, fields=_OPT_FIELDS, embedding_dim=8, num_heads=2, dim_head=4, depth=1, scale_dim=2, dnn_hidden_units=[16, 8],
         batch_norm=True, use_wide=True, embedding_regularizer=0.01, net_regularizer=0),
    dict(name="opt_rmsprop", batch=6, topk=3, init_seed=2021, data_seed=251, weight_seed=252, full_limit=1 << 20, optimizer="RMSprop",
         learning_rate=0.001, fields=_OPT_FIELDS, embedding_dim=8, num_heads=2, dim_head=4, depth=1, scale_dim=2, dnn_hidden_units=[16, 8],
         batch_norm=False, use_wide=True, embedding_regularizer=0.01, net_regularizer=0),
    dict(name="regression_mse", batch=6, topk=3, init_seed=2021, data_seed=261, weight_seed=262, full_limit=1 << 20, task="regression",
         loss="mse_loss", fields=_OPT_FIELDS, embedding_dim=8, num_heads=2, dim_head=4, depth=1, scale_dim=2, dnn_hidden_units=[16, 8],
         batch_norm=True, use_wide=True, embedding_regularizer=0.01, net_regularizer=0),
]

KNOWN_COUNT_CASES = [
    dict(name="count_mltag", expected_params=1337241, topk=5,
         fields=[_cat("user_id", 90239 - 2000 - 1000), _cat("item_id", 2000), _cat("tag_id", 1000)],
         embedding_dim=10, num_heads=2, dim_head=10, depth=4, scale_dim=4, dnn_hidden_units=[400, 400, 400],
         batch_norm=False, use_wide=True, embedding_regularizer=0.03, net_regularizer=0),
    dict(name="count_kkbox", expected_params=4714649, topk=5,
         fields=[_cat("msno", 92247 - 12 * 100)] + [_cat("k%d" % i, 100) for i in range(8)] +
                [_seq("genre_ids", 100), _seq("artist_name", 100), _cat("isrc", 100), _cat("bd", 100)],
         embedding_dim=40, num_heads=8, dim_head=10, depth=4, scale_dim=2, dnn_hidden_units=[400, 400, 400],
         batch_norm=True, use_wide=True, embedding_regularizer=0.0005, net_regularizer=0),
    dict(name="count_tmall", expected_params=16970282, topk=5,
         fields=[_cat("t0", 1529680 - 7 * 1000)] + [_cat("t%d" % i, 1000) for i in range(1, 8)],
         embedding_dim=10, num_heads=32, dim_head=10, depth=4, scale_dim=2, dnn_hidden_units=[200, 80],
         batch_norm=True, use_wide=True, embedding_regularizer=0.07, net_regularizer=0),
]


def case_by_name(name):
    for c in CASES + KNOWN_COUNT_CASES:
        if c["name"] == name:
            return c
    raise KeyError(name)


# ----------------------------------------------------------------------------- feature map / kwargs
def feature_specs(case):
    """feature_map.json-style ``feature_specs`` with column indices assigned like
    FeatureMap.set_feature_index (fuxictr/features.py:46-57)."""
    specs = OrderedDict()
    col = 0
    for f in case["fields"]:
        spec = {"source": "", "type": f["type"], "vocab_size": f["vocab_size"]}
        if f["type"] == "sequence":
            spec["index"] = list(range(col, col + f["max_len"]))
            spec["max_len"] = f["max_len"]
            spec["encoder"] = f["encoder"]
            spec["padding_idx"] = f["vocab_size"] - 1
            col += f["max_len"]
        else:
            spec["index"] = col
            if "padding_idx" in f:
                spec["padding_idx"] = f["padding_idx"]
            col += 1
        specs[f["name"]] = spec
    return specs


def input_length(case):
    return sum(f.get("max_len", 1) for f in case["fields"])


def model_kwargs(case):
    """The flattened params dict run_expid.py would splat into the constructor."""
    return dict(model_id=case.get("model", "RAT_m2") + "_" + case["name"], gpu=-1, task=case.get("task", "binary_classification"),
                learning_rate=case.get("learning_rate", 1e-3),
                embedding_dim=case["embedding_dim"], dnn_hidden_units=list(case["dnn_hidden_units"]),
                dnn_activations=case.get("dnn_activations", "relu"), num_heads=case["num_heads"], dim_head=case["dim_head"],
                depth=case["depth"], scale_dim=case["scale_dim"], dropout=0.0, emb_dropout=0.0, net_dropout=0,
                batch_norm=case["batch_norm"], use_wide=case["use_wide"],
                embedding_regularizer=case["embedding_regularizer"], net_regularizer=case["net_regularizer"],
                retrieval_augmented=True, retrieval_configs={"topK": case["topk"], "label_wise": False},
                model_root="./_golden_models/", metrics=["AUC", "logloss"], verbose=0, optimizer=case.get("optimizer", "adam"),
                loss=case.get("loss", "binary_crossentropy"), monitor="AUC", monitor_mode="max", patience=2, every_x_epochs=1,
                save_best_only=True, layer_norm=True, use_scale=True, use_residual=True, pool="cls", seed=2021)


# ----------------------------------------------------------------------------- deterministic data
def make_inputs(case):
    """The 4-tuple a DataLoader batch carries (fuxictr/pytorch/data_generator.py:66-78):
    X [B,1+K,L] float64 ids, y [B,1+K] float64 {0,1}, retrieved_values [B,K] f64, retrieved_lens [B] i64."""
    rs = np.random.RandomState(case["data_seed"])
    b, t = case["batch"], case["topk"] + 1
    cols = []
    for f in case["fields"]:
        v = f["vocab_size"]
        if f["type"] == "sequence":
            ids = rs.randint(0, v - 1, size=(b, t, f["max_len"]))
            lens = rs.randint(0, f["max_len"] + 1, size=(b, t, 1))          # 0..max_len real ids, rest padding
            pad = np.arange(f["max_len"])[None, None, :] >= lens
            ids[pad] = v - 1
            cols.append(ids)
        else:
            ids = rs.randint(0, v, size=(b, t, 1))                          # may hit an explicit padding row
            cols.append(ids)
    X = np.concatenate(cols, axis=-1).astype(np.float64)
    y = rs.randint(0, 2, size=(b, t)).astype(np.float64)
    y[0, 0], y[1, 0] = 1.0, 0.0                                             # both classes present
    rv = rs.rand(b, t - 1)
    rl = np.full((b,), t - 1, dtype=np.int64)
    return X, y, rv, rl


def make_weights(case, shapes):
    """Non-degenerate weights for every state_dict entry (shapes: name -> tuple)."""
    rs = np.random.RandomState(case["weight_seed"])
    pad_rows = {}
    for f in case["fields"]:
        if f["type"] == "sequence":
            pad_rows[f["name"]] = f["vocab_size"] - 1
        elif "padding_idx" in f:
            pad_rows[f["name"]] = f["padding_idx"]
    out = OrderedDict()
    for name, shp in shapes.items():
        if name.endswith("num_batches_tracked"):
            out[name] = np.array(0, dtype=np.int64)
        elif name.endswith("running_mean"):
            out[name] = (0.1 * rs.standard_normal(shp)).astype(np.float32)
        elif name.endswith("running_var"):
            out[name] = (1.0 + 0.2 * np.abs(rs.standard_normal(shp))).astype(np.float32)
        elif "embedding_layer" in name and len(shp) == 2:
            scale = 1.0 if name.startswith("label_embedding") else (0.3 if shp[1] > 1 else 0.2)
            wt = (scale * rs.standard_normal(shp)).astype(np.float32)
            fname = name.split(".")[-2]
            if fname in pad_rows:
                wt[pad_rows[fname]] = 0.0
            out[name] = wt
        elif name.endswith("norm.weight") or (len(shp) == 1 and name.startswith("dnn.") and name.endswith("weight")):
            out[name] = (1.0 + 0.1 * rs.standard_normal(shp)).astype(np.float32)
        elif len(shp) == 1:
            out[name] = (0.1 * rs.standard_normal(shp)).astype(np.float32)
        else:
            out[name] = (rs.standard_normal(shp) / np.sqrt(shp[1])).astype(np.float32)
    return out


# ----------------------------------------------------------------------------- compact summaries
_PROJ_SEED = 977


def _summary_vectors(n):
    rs = np.random.RandomState(_PROJ_SEED + (n % 1000003))
    proj = rs.standard_normal((4, n))
    idx = rs.randint(0, n, size=min(256, n))
    return proj, idx


def put_summary(store, key, arr, full_limit):
    """Store ``arr`` in full if small, else (l2, sum, 4 random projections, 256 sampled entries)."""
    a = np.array(arr, copy=True)          # never alias a live parameter
    if a.size <= full_limit:
        store[key] = a
        return
    flat = a.reshape(-1).astype(np.float64)
    proj, idx = _summary_vectors(flat.size)
    store[key + "#summary"] = np.concatenate([[np.sqrt((flat ** 2).sum()), flat.sum()], proj @ flat, flat[idx]])


def check_summary(store, key, arr, rtol, atol):
    """Compare ``arr`` with what put_summary stored under ``key``.  Returns max abs error seen."""
    a = np.asarray(arr)
    if key in store:
        ref = store[key]
        assert ref.shape == a.shape, (key, ref.shape, a.shape)
        np.testing.assert_allclose(a.astype(np.float64), ref.astype(np.float64), rtol=rtol, atol=atol, err_msg=key)
        return float(np.max(np.abs(a.astype(np.float64) - ref.astype(np.float64)))) if a.size else 0.0
    ref = store[key + "#summary"]
    flat = a.reshape(-1).astype(np.float64)
    proj, idx = _summary_vectors(flat.size)
    got = np.concatenate([[np.sqrt((flat ** 2).sum()), flat.sum()], proj @ flat, flat[idx]])
    scale = max(ref[0], 1e-30)
    # l2 / sum / projections are O(l2)-sized quantities: compare relative to the tensor's norm
    np.testing.assert_allclose(got[:6], ref[:6], rtol=rtol, atol=atol + rtol * scale * 4, err_msg=key + " (aggregates)")
    np.testing.assert_allclose(got[6:], ref[6:], rtol=rtol, atol=atol, err_msg=key + " (samples)")
    return float(np.max(np.abs(got[6:] - ref[6:])))


def summary_scale(store, key):
    """largest |element| the fixture holds for `key` (the whole tensor, or the sampled elements of a summary)"""
    ref = store[key] if key in store else store[key + "#summary"][6:]
    return float(np.max(np.abs(np.asarray(ref, dtype=np.float64)))) if np.size(ref) else 0.0


def has(store, key):
    return key in store or (key + "#summary") in store
