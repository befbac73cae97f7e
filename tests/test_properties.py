"""Size-independent properties of the hot path, checked on the oracle (fast) and on the product
through the emulated kernels (tiny): the target's prediction is invariant to the ORDER of its retrieved samples
(cross-sample attention has no positional term, RAT_m2.py:128-133 `del position_embedding`), padding rows and the dead
query_proj never receive gradient, and a K=0 grid (no retrieved samples) is legal."""
import os
import sys

import numpy as np
import pytest
import torch

import golden_cases as gc
import model_cases as mc
from test_oracle_golden import load_case
from oracle import rat_m2_oracle as orc

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "emu"))


@pytest.mark.parametrize("name", ["tiny_seq_bn", "mltag_shape", "tmall_shape"])
def test_oracle_prediction_invariant_to_retrieved_order(name):
    case, cfg, gold, w, X, y = load_case(name)
    perm = torch.from_numpy(np.random.RandomState(0).permutation(X.shape[1] - 1)) + 1
    Xp = torch.cat([X[:, :1], X[:, perm]], dim=1)
    yp = torch.cat([y[:, :1], y[:, perm]], dim=1)
    with torch.no_grad():
        a = orc.forward(w, X, y, cfg, training=False)
        b = orc.forward(w, Xp, yp, cfg, training=False)
    np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=0, atol=2e-6)


def test_oracle_k0_grid():
    case, cfg, gold, w, X, y = load_case("tiny_seq_bn")
    with torch.no_grad():
        out = orc.forward(w, X[:, :1], y[:, :1], cfg, training=False)
    assert out.shape == (X.shape[0], 1) and torch.isfinite(out).all()


@pytest.fixture(scope="module")
def emu_lib():
    import build_emu
    import rat_amd._lib as L
    old = L._default
    L._default = L.RatLib(build_emu.build())
    yield L._default
    L._default = old


def test_product_invariance_padding_and_dead_params(emu_lib):
    case = gc.case_by_name("tiny_seq_bn")
    model = mc.build_model(case, gpu=-1, seed=1)
    mc.load_weights(model, case)
    X, y, rv, rl = mc.batch_of(case)
    perm = torch.from_numpy(np.random.RandomState(1).permutation(X.shape[1] - 1)) + 1
    model.eval()
    with torch.no_grad():
        a = model.forward((X, y, rv, rl))["y_pred"]
        b = model.forward((torch.cat([X[:, :1], X[:, perm]], 1), torch.cat([y[:, :1], y[:, perm]], 1), rv, rl))["y_pred"]
        k0 = model.forward((X[:, :1], y[:, :1], rv[:, :0], rl))["y_pred"]          # no retrieved samples at all
    np.testing.assert_allclose(a.numpy(), b.numpy(), rtol=0, atol=2e-6)
    assert torch.isfinite(k0).all()
    model.train()
    model.get_total_loss((X, y, rv, rl)).backward()
    params = dict(model.named_parameters())
    assert params["query_proj.weight"].grad is None and params["query_proj.bias"].grad is None
    # padding rows: sequence field "c" pads with id 5, categorical "e" declares padding_idx=8 (regulariser adds lambda*0 = 0)
    for prefix in ("embedding_layer.embedding_layer.embedding_layer.", "lr_layer.embedding_layer.embedding_layer.embedding_layer."):
        assert float(params[prefix + "c.weight"].grad[5].abs().max()) == 0.0
        assert float(params[prefix + "e.weight"].grad[8].abs().max()) == 0.0
