"""GPU parity of the product model against the reference's golden vectors — run on the MI355X box with -m gpu."""
import pytest
import torch

import golden_cases as gc
import model_cases as mc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def hip_lib():
    from rat_amd._lib import get_lib
    assert torch.cuda.is_available(), "GPU tests need a GPU"
    return get_lib()


@pytest.mark.parametrize("name", [c["name"] for c in gc.CASES])
def test_init_matches_reference_bit_for_bit(name):
    mc.check_init(name, gpu=0)


@pytest.mark.parametrize("name", [c["name"] for c in gc.CASES])
def test_eval_forward(name):
    mc.check_eval(name, gpu=0)


@pytest.mark.parametrize("name", [c["name"] for c in gc.CASES])
def test_two_training_steps(name):
    mc.check_training(name, gpu=0)


@pytest.mark.parametrize("name", [c["name"] for c in gc.CASES])
def test_train_step_api_matches_the_reference_run(name):
    """train_step(): fused iteration; steps 1-2 eager against the golden run, steps 3-5 are hipGraph REPLAYS (captured at the third
    call of the batch shape) against the literal zero_grad / backward / clip / step sequence on a twin model"""
    model = mc.check_train_step_api(name, gpu=0, steps=5)
    graphs = [e[1] for e in model._step_graphs.values() if e[1]]
    assert len(graphs) == 1 and model.optimizer._step == 5


@pytest.mark.parametrize("name", ["tiny_seq_bn", "northstar_shape"])
@pytest.mark.parametrize("mode", ["sorted", "sparse"])
def test_train_step_graph_with_sorted_and_sparse_table_gradients(name, mode):
    """the rocPRIM sort / scan of the plan and the row optimizer inside the captured step"""
    kw = dict(embedding_grad=mode)
    if mode == "sparse":
        kw["embedding_regularizer"] = 0.0
    case = gc.case_by_name(name)
    a = mc.build_model(case, gpu=0, seed=1, **kw)
    b = mc.build_model(case, gpu=0, seed=1, **kw)
    mc.load_weights(a, case), mc.load_weights(b, case)
    b.use_graph = False
    batch = mc.batch_of(case)
    a.train(), b.train()
    for step in range(5):
        la, lb = float(a.train_step(batch)), float(b.train_step(batch))
        assert abs(la - lb) < 1e-6, (step, la, lb)
    assert any(e[1] for e in a._step_graphs.values())
    # same kernels in the same order; not bit-identical run to run (the loss / fc gradients are fp32 atomic sums), and Adam turns
    # rounding-level gradient differences into +-lr steps: bound the outliers like check_train_step_api does
    noise = mc.noise_tensors(a)              # biases in front of BatchNorm: true gradient 0, Adam steps on rounding noise
    for (k, va), vb in zip(a.state_dict().items(), b.state_dict().values()):
        if k.endswith("num_batches_tracked"):
            assert int(va) == int(vb) == 5, k
            continue
        if k in noise or k.endswith("running_mean"):
            continue
        x, y = va.detach().cpu().double(), vb.detach().cpu().double()
        bad = (x - y).abs() > 1.5e-5 + 3e-4 * y.abs()
        assert float(bad.double().mean()) < 1e-3 and float((x - y).abs().max()) <= 1.05e-2, (k, float((x - y).abs().max()))


@pytest.mark.parametrize("mode", ["sorted", "sparse"])
def test_train_step_graphs_of_two_batch_shapes_keep_their_plans(mode):
    """a captured step has the sort plan's workspace / count pointers baked in.  Shape A is captured, then an
    epoch's tail batch (shape B) builds its own plan — which used to REPLACE A's, so that every later replay of A sorted and reduced
    through a freed block.  Run A x4 (capture at 3), B x4 (capture), garbage allocations in between, A x2 again: the graph model must
    follow an eager twin, and A's plan object must still be the one the first capture saw."""
    kw = dict(embedding_grad=mode)
    if mode == "sparse":
        kw["embedding_regularizer"] = 0.0
    case = gc.case_by_name("northstar_shape")
    a = mc.build_model(case, gpu=0, seed=1, **kw)
    b = mc.build_model(case, gpu=0, seed=1, **kw)
    mc.load_weights(a, case), mc.load_weights(b, case)
    b.use_graph = False
    full = mc.batch_of(case)
    nb = full[0].shape[0]
    tail = tuple(t[: max(1, nb // 2)].contiguous() for t in full)
    a.train(), b.train()
    plans_a = None
    schedule = [full] * 4 + [tail] * 4 + [full] * 2 + [tail] + [full]
    for step, batch in enumerate(schedule):
        la, lb = float(a.train_step(batch)), float(b.train_step(batch))
        assert abs(la - lb) < 2e-6, (step, la, lb)
        if step == 3:
            plans_a = {k: id(v) for k, v in a._ws.items() if isinstance(k, tuple) and k[0] == "plan"}
        if step in (7, 9):      # whatever the allocator handed back is overwritten before the next replay
            junk = [torch.full((1 << 20,), float("nan"), device="cuda") for _ in range(64)]
            del junk
    assert sum(1 for e in a._step_graphs.values() if e[1]) == 2
    for k, v in plans_a.items():
        assert id(a._ws[k]) == v, "the plan of the first captured shape was replaced: %r" % (k,)
    noise = mc.noise_tensors(a)
    for (k, va), vb in zip(a.state_dict().items(), b.state_dict().values()):
        if k.endswith("num_batches_tracked") or k in noise or k.endswith("running_mean"):
            continue
        x, y = va.detach().cpu().double(), vb.detach().cpu().double()
        assert bool(torch.isfinite(x).all()), k
        bad = (x - y).abs() > 3e-5 + 3e-4 * y.abs()
        assert float(bad.double().mean()) < 1e-3 and float((x - y).abs().max()) <= 2.5e-2, (k, float((x - y).abs().max()))


@pytest.mark.parametrize("name", ["kkbox_shape", "tmall_real_heads"])
def test_train_step_graph_replays_with_dropout(name):
    """the shipped KKBox / Tmall configs have emb_dropout 0.1 (Tmall also net_dropout 0.08) and used to stay eager
    because the seeds were drawn on the host.  The generator state now lives on the device (rat_dropout_seeds), so the captured step
    draws new masks on every replay: a graph model and an eager twin with the same base seed must agree step by step, and the
    loss on the SAME batch must move from step to step by more than Adam alone would explain identically in both."""
    case = gc.case_by_name(name)
    kw = dict(emb_dropout=0.1, net_dropout=0.08, dropout=0.05)
    models = []
    for use_graph in (True, False):
        torch.manual_seed(4321)                      # the base seed is drawn from torch's generator at the first training forward
        m = mc.build_model(case, gpu=0, seed=1, **kw)
        mc.load_weights(m, case)
        m.use_graph = use_graph
        m.train()
        models.append(m)
    a, b = models
    batch = mc.batch_of(case)
    losses = []
    for step in range(6):
        torch.manual_seed(99 + step)
        la = float(a.train_step(batch))
        torch.manual_seed(99 + step)
        lb = float(b.train_step(batch))
        assert abs(la - lb) < 2e-6 * max(1.0, abs(lb)), (step, la, lb)
        losses.append(la)
    assert any(e[1] for e in a._step_graphs.values()), "the step with dropout was not captured"
    assert int(a._drop_counter) == 6 and int(b._drop_counter) == 6
    # the masks really change between replays: with a frozen mask the eval-mode prediction drift alone would be monotone and tiny
    a.eval(), b.eval()
    with torch.no_grad():
        ya, yb = a.forward(batch)["y_pred"], b.forward(batch)["y_pred"]
    assert float((ya - yb).abs().max()) < 2e-3      # (six Adam steps on rounding-level gradient differences; the per-step losses above are the check)
    words = a._drop_words.clone()
    a.train()
    a.train_step(batch)
    assert not torch.equal(words, a._drop_words)


def test_a_changed_clip_norm_takes_a_new_capture():
    """max_gradient_norm (and the other baked launch arguments) are part of the step-graph key"""
    case = gc.case_by_name("tiny_seq_bn")
    a = mc.build_model(case, gpu=0, seed=1)
    mc.load_weights(a, case)
    batch = mc.batch_of(case)
    a.train()
    for _ in range(3):
        a.train_step(batch)
    assert sum(1 for e in a._step_graphs.values() if e[1]) == 1
    a._max_gradient_norm = 1e-3
    a.graph_shapes = 4
    for _ in range(3):
        a.train_step(batch)
    assert sum(1 for e in a._step_graphs.values() if e[1]) == 2
    b = mc.build_model(case, gpu=0, seed=1)
    mc.load_weights(b, case)
    b.use_graph = False
    b.train()
    for i in range(6):
        b._max_gradient_norm = 10.0 if i < 3 else 1e-3
        b.train_step(batch)
    noise = mc.noise_tensors(a)
    for (k, va), vb in zip(a.state_dict().items(), b.state_dict().values()):
        if k.endswith("num_batches_tracked") or k in noise or k.endswith("running_mean"):
            continue
        assert float((va - vb).abs().max()) <= 1.3e-2, k


def test_train_step_graph_in_segments_with_eager_closures_between():
    """what a captured step looks like under data parallelism: collectives are not captured, they run eagerly BETWEEN graph segments
    (StepGraph.between_segments).  One GPU has no collectives, so the model's test knob inserts two no-op ones: 3 segments + 2
    closures must reproduce the single-graph result."""
    case = gc.case_by_name("northstar_shape")
    a = mc.build_model(case, gpu=0, seed=1)
    b = mc.build_model(case, gpu=0, seed=1)
    mc.load_weights(a, case), mc.load_weights(b, case)
    a._graph_test_splits = True
    batch = mc.batch_of(case)
    a.train(), b.train()
    for step in range(5):
        la, lb = float(a.train_step(batch)), float(b.train_step(batch))
        assert abs(la - lb) < 1e-6, (step, la, lb)
    ga = [e[1] for e in a._step_graphs.values() if e[1]][0]
    gb = [e[1] for e in b._step_graphs.values() if e[1]][0]
    assert sum(isinstance(i, torch.cuda.CUDAGraph) for i in ga.items) == 3 and len(ga.items) == 5 and len(gb.items) == 1
    noise = mc.noise_tensors(a)
    for (k, va), vb in zip(a.state_dict().items(), b.state_dict().values()):
        if k.endswith("num_batches_tracked") or k in noise or k.endswith("running_mean"):
            continue
        x, y = va.detach().cpu().double(), vb.detach().cpu().double()
        bad = (x - y).abs() > 1.5e-5 + 3e-4 * y.abs()
        assert float(bad.double().mean()) < 1e-3 and float((x - y).abs().max()) <= 1.05e-2, k


@pytest.mark.parametrize("name", ["tiny_seq_bn", "mltag_shape", "kkbox_shape", "northstar_shape", "bare_no_proj", "tmall_real_heads", "tmall_shape",
                                  "m3_tiny_seq", "m3_northstar_shape", "m3_mltag_shape"])
def test_dead_token_pruning_changes_nothing(name):
    mc.check_pruning_equivalence(name, gpu=0)


@pytest.mark.parametrize("name", mc.CHECKPOINT_CASES)
def test_reference_written_checkpoint_loads_and_round_trips(name, tmp_path):
    """`.model` files written by the reference's own RAT_m2 / m0 / m1 / m3 classes (tests/golden/make_golden_checkpoints.py)"""
    mc.check_checkpoint(name, gpu=0, tmpdir=tmp_path)


@pytest.mark.parametrize("name", ["tiny_seq_bn", "northstar_shape", "kkbox_shape"])
def test_composed_attention_path(name, monkeypatch):
    """fused-kernel threshold lowered: every attention layer runs LayerNorm -> GEMM -> attention core (strided sequences in the
    cross phase) -> GEMM; same golden vectors."""
    from rat_amd import models
    monkeypatch.setattr(models.RAT_m2, "FUSED_MAX_L", 3)
    mc.check_eval(name, gpu=0)
    mc.check_training(name, gpu=0)


@pytest.mark.parametrize("heads,dropout", [(32, 0.0), (16, 0.2)], ids=["h32", "h16_dropout"])
def test_wide_heads_forward_group_loop_equals_the_per_group_launches(heads, dropout):
    """BASELINE configs[4]'s head geometry (32 x 10 at d = 64): rat_attn_fwd_groups inside the model against the per-group launches"""
    mc.check_wide_heads_group_loop(gpu=0, batch=40, topk=6, nfields=7, heads=heads, depth=2, dropout=dropout)


@pytest.mark.parametrize("variant", ["RAT_m1", "RAT_m0", "RAT_m3"])
def test_variants_with_the_shipped_tmall_heads(variant):
    """32 heads x 10 at d = 10 in the other model variants: RAT_m1's transformers run the wide-head layers through the same one-launch
    forward / backward as RAT_m2 (equal to the per-group launches), RAT_m0's joint sequences take the composed path whatever the head
    count, RAT_m3's shared-query attention (16 heads of width 20) runs as four head groups of the fused kernel — against the golden
    vectors of the real RAT_m3 class at this geometry (m3_tmall_real_heads); heads of width 32 take the composed path (m3_wide_dim_head)"""
    import golden_cases as gc
    case = dict(gc.case_by_name("tmall_real_heads"), name="tm_" + variant, model=variant, batch_norm=False)
    if variant == "RAT_m3":
        from rat_amd import ops
        model = mc.build_model(gc.case_by_name("m3_tmall_real_heads"), gpu=0, seed=1)
        assert model._m3_mode(ops.intra_map(4, 7, 9)) == ("grouped", 4) and model._m3_mode(ops.cross_map(4, 7, 9)) == ("grouped", 4)
        mc.check_training("m3_tmall_real_heads", gpu=0)
        model = mc.build_model(gc.case_by_name("m3_wide_dim_head"), gpu=0, seed=1)
        assert model._m3_mode(ops.intra_map(5, 5, 5))[0] == "composed"
        mc.check_training("m3_wide_dim_head", gpu=0)
        # (and, below: the one-launch form of the head groups against the per-group launches, like the other variants)
    out = {}
    for loop in (True, False):
        model = mc.build_model(case, gpu=0, seed=1)
        mc.load_weights(model, case)
        model.group_loop = loop
        model.train()
        model.optimizer.zero_grad()
        loss = model.get_total_loss(tuple(t.to(model.device) for t in mc.batch_of(case)))
        loss.backward()
        out[loop] = (float(loss.detach()), {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None})
    (l1, g1), (l0, g0) = out[True], out[False]
    assert abs(l1 - l0) < 2e-6 and len(g1) >= 20
    for k in g1:
        assert float((g1[k] - g0[k]).abs().max()) / (float(g0[k].abs().max()) + 1e-30) < 1e-4, k


def test_parallel_variant_with_wide_heads_at_the_north_star_embedding_dimension():
    """RAT_m3 with 16 x 10 heads at d = 64 (8 heads of width 20, inner 160: too wide for one launch): two head groups of 4 x 20, each on the
    bf16x3 kernels' 4-head instantiation (round 6) — loss, predictions and every gradient against the oracle"""
    from rat_amd import ops
    model, worst = mc.check_variant_against_oracle(0, "RAT_m3", num_heads=16, batch=5, topk=4, depth=2)
    assert model._m3_mode(ops.intra_map(5, 5, 21)) == ("grouped", 4) and model._m3_arith(4) == "bf16x3"
    assert worst < 2e-5, worst


def test_grouped_heads_mode_is_selected_for_wide_heads():
    """32 heads x 10: four launches of the fused kernel on 8 heads each (model._attn_mode)"""
    import golden_cases as gc
    from rat_amd import ops
    model = mc.build_model(gc.case_by_name("tmall_real_heads"), gpu=0, seed=1)
    assert model._attn_mode(ops.intra_map(4, 7, 9)) == ("grouped", 8)
    assert model._attn_mode(ops.cross_map(4, 7, 9)) == ("grouped", 8)


def test_evaluate_generator_on_device_tuples_keeps_every_batchs_labels_under_the_eval_graph():
    """base_model.py:232-247 keeps each batch's y_true until the end of the pass.  Once a shape's inference graph is captured,
    rat_batch_prepare writes the labels into the graph's static inputs: forward() must hand out a copy, not that view
    (ADVICE r5, high).  Seven same-shape CUDA 4-tuples with different labels: graph on == graph off, batch by batch."""
    import golden_cases as gc
    case = gc.case_by_name("tiny_seq_bn")
    X, y, rv, rl = mc.batch_of(case)
    rng = torch.Generator().manual_seed(5)
    batches = []
    for i in range(7):
        perm = torch.randperm(X.shape[0], generator=rng)
        yi = y[perm].clone()
        yi[:, 0] = (torch.rand(X.shape[0], generator=rng) < 0.5).to(y.dtype)
        batches.append(tuple(t.cuda() for t in (X[perm], yi, rv[perm], rl[perm])))
    out = {}
    for graph in (True, False):
        model = mc.build_model(case, gpu=0, seed=1)
        mc.load_weights(model, case)
        model.eval_graph = graph
        model.eval()
        with torch.no_grad():
            per_batch = [model.forward(b) for b in batches]
        torch.cuda.synchronize()
        assert bool(model.__dict__.get("_eval_graphs")) == graph
        out[graph] = ([o["y_true"].cpu() for o in per_batch], [o["y_pred"].cpu() for o in per_batch], model.evaluate_generator(batches))
    for i, b in enumerate(batches):
        assert torch.equal(out[True][0][i].reshape(-1), b[1][:, 0].float().cpu()), i
        assert torch.equal(out[True][0][i], out[False][0][i]), i
        assert float((out[True][1][i] - out[False][1][i]).abs().max()) < 1e-6
    assert len({tuple(t.reshape(-1).tolist()) for t in out[True][0]}) > 1       # the labels really differ from batch to batch
    for k, v in out[False][2].items():
        assert abs(out[True][2][k] - v) < 1e-6, (k, out[True][2], out[False][2])
