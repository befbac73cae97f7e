"""RCCL on the MI355X box (-m gpu).  The box has ONE GPU and RCCL refuses two ranks on a device, so the N > 1 path cannot be run with
its real backend here — except with a process group of ONE rank: `dp_single_rank = True` (rat_amd/base_model.py) sends that rank
through the data-parallel code path anyway, and every collective of a training step is then issued to RCCL on real hardware:
the asynchronous all-reduce of the dense-net gradients started inside backward, SyncBN's all-gathers of the statistics, the
all-gathers of the (row ids, gradient rows, count) lists with the merge behind them or the dense all-reduce of the table block, the
label-table all-reduce, the int32 all-reduce of the id-error counters, the float64 broadcast and the barrier of the epoch end — eagerly,
and between the hipGraph segments of a captured step (`graph_under_dp`).  What this proves is that the calls are valid RCCL calls on
device tensors in the stream order the step needs; that the sums over ranks are right is what the 2-rank gloo tests prove
(tests/test_dp_gloo.py, tests/test_gpu_dp.py).

Checked: five steps end at the weights of a plain single process (2e-4, the tolerance of the other DP tests), the per-step losses agree,
and the graph variant really replayed segments with closures between them."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

WORKER = r'''
import os, sys
ROOT, out, case_name, mode, steps = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4], int(sys.argv[5])
sys.path[:0] = [ROOT, os.path.join(ROOT, "www24-rat_amd"), os.path.join(ROOT, "tests")]
import torch
import torch.distributed as dist
import golden_cases as gc
import model_cases as mc
rccl = mode != "plain"
if rccl:
    torch.cuda.set_device(0)
    try:
        dist.init_process_group("nccl", rank=0, world_size=1)
        probe = torch.ones(4, device="cuda:0")
        dist.all_reduce(probe)                    # the communicator comes up with the first collective
        torch.cuda.synchronize()
    except Exception as exc:                      # no usable RCCL in this environment: nothing this test could say (exit code 77 = skip)
        print("RCCL_UNAVAILABLE: %s: %s" % (type(exc).__name__, exc))
        sys.exit(77)
case = dict(gc.case_by_name(case_name))
model = mc.build_model(case, gpu=0, seed=1)
mc.load_weights(model, case)
batch = mc.batch_of(case)
if rccl:
    model.dp_single_rank = True
    model.row_list_exchange = "lists" in mode
    model.graph_under_dp = "graph" in mode
    assert model._dp() and model._world_size() == 1
model.train()
losses = [float(model.train_step(batch)) for _ in range(steps)]
torch.cuda.synchronize()
agreed = None
if rccl:
    model.check_id_errors()                       # int32 all-reduce of the counters
    agreed = model._agreed_value(0.625)           # float64 broadcast
    model.checkpoint = out + ".model"
    model._save_checkpoint()                      # rank 0 writes, barrier
graphs = [e[1] for e in getattr(model, "_step_graphs", {}).values() if e[1]]
segs = sum(isinstance(i, torch.cuda.CUDAGraph) for i in graphs[0].items) if graphs else 0
torch.save({"flat": model._flat.detach().cpu(), "losses": losses, "segments": segs, "closures": (len(graphs[0].items) - segs) if graphs else 0,
            "noise": sorted(mc.noise_tensors(model)), "offsets": dict(model._offsets), "agreed": agreed,
            "sizes": {k: v.numel() for k, v in model._params.items()}, "backend": dist.get_backend() if rccl else None}, out)
if rccl:
    dist.barrier()
    dist.destroy_process_group()
'''


def _run(out, case, mode, steps, port):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    if mode != "plain":
        env.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run([sys.executable, "-c", WORKER, ROOT, str(out), case, mode, str(steps)], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.STDOUT, text=True, timeout=900)
    if r.returncode == 77 and "RCCL_UNAVAILABLE" in r.stdout:
        pytest.skip("RCCL could not bring up a one-rank communicator here: " + r.stdout.strip().splitlines()[-1][:300])
    assert r.returncode == 0, r.stdout[-3000:]
    return torch.load(str(out))


@pytest.mark.parametrize("mode", ["rccl_lists_graph", "rccl_dense_graph", "rccl_lists_eager", "rccl_dense_eager"])
def test_every_collective_of_a_step_runs_on_rccl_with_one_rank(tmp_path, mode):
    assert torch.cuda.is_available()
    case, steps = "kkbox_shape", 5          # BatchNorm on (SyncBN), wide part, two 3-id bag fields, d = 16
    port = 33500 + (os.getpid() % 2000)
    one = _run(tmp_path / "plain.pt", case, "plain", steps, port)
    got = _run(tmp_path / "rccl.pt", case, mode, steps, port)
    assert got["backend"] == "nccl" and got["agreed"] == 0.625
    assert os.path.exists(str(tmp_path / "rccl.pt.model"))
    assert one["segments"] == 1 and one["closures"] == 0
    if "graph" in mode:
        assert got["segments"] >= 8 and got["closures"] == got["segments"] - 1, (got["segments"], got["closures"])
    else:
        assert got["segments"] == 0
    keep = torch.ones_like(one["flat"], dtype=torch.bool)
    for name in one["noise"]:                # biases in front of BatchNorm: true gradient 0, Adam steps on rounding noise
        keep[one["offsets"][name]:one["offsets"][name] + one["sizes"][name]] = False
    a, b = got["flat"][keep].double(), one["flat"][keep].double()
    bad = (a - b).abs() > 3e-6 + 3e-4 * b.abs()
    assert float(bad.double().mean()) < 2e-3 and float((a - b).abs().max()) <= 1.05e-2, (float(bad.double().mean()), float((a - b).abs().max()))
    for s in range(steps):
        assert abs(got["losses"][s] - one["losses"][s]) < 2e-5, (s, got["losses"][s], one["losses"][s])


def _bench(args, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_PORT"] = str(35500 + (os.getpid() % 2000))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                       timeout=timeout)
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    return r, lines


def test_bench_rehearsal_measures_eager_first_then_graphs_and_reports_the_exchange():
    """bench.py's N > 1 path on one RCCL rank: the eager regions are measured first, then the segmented-graph regions; the line carries
    both attempts, the world size, the exchange form with what it put on the links and the per-phase HIP-event times"""
    import json
    r, lines = _bench(["--dp-rehearsal", "--graph-dp", "--workload", "tiny", "--steps", "3", "--warmup", "2"])      # (--graph-dp: opt-in since round 6)
    if "RCCL" in r.stderr and r.returncode != 0 and "init_process_group" in r.stderr:
        pytest.skip("RCCL could not bring up a one-rank communicator here")
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    att = out["step_mode"]["attempts"]
    assert att["graph_captured"] is True and att["reported"] in ("graph", "eager") and att["eager_ms_per_step"] > 0 and att["graph_ms_per_step"] > 0
    assert out["world"] == 1 and out["first_barrier_s"] is not None
    comm = out["communication"]["weak"]
    assert comm["backend"] == "nccl" and comm["exchange_form"] in ("owner_lists", "dense_allreduce", "gathered_lists")
    assert {"exchange", "optimizer"} <= set(comm["phases_ms_per_step"])
    if comm["exchange_form"] == "owner_lists":
        assert comm["owner_lists"]["collectives_per_step"] == 3 and "exchange_counts" in comm["phases_ms_per_step"]
    assert "strong_scaling" in out


def test_bench_rehearsal_survives_a_stalled_graph_attempt():
    """the watchdog around the graph attempt: with a timeout it cannot meet, every rank exits 0 and rank 0 prints the EAGER line with a note"""
    import json
    r, lines = _bench(["--dp-rehearsal", "--graph-dp", "--workload", "tiny", "--steps", "3", "--warmup", "2", "--graph-attempt-timeout", "0.001"])
    if r.returncode != 0 and "init_process_group" in r.stderr:
        pytest.skip("RCCL could not bring up a one-rank communicator here")
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert "abandoned" in out["step_mode"]["graph_attempt"] and out["step_mode"]["graph"] is False and out["value"] > 0


def test_bench_rehearsal_default_is_the_eager_step():
    """round 6: without --graph-dp the N > 1 line is the eager fused step only (no graph attempt, nothing to abandon)"""
    import json
    r, lines = _bench(["--dp-rehearsal", "--workload", "tiny", "--steps", "3", "--warmup", "2"])
    if r.returncode != 0 and "init_process_group" in r.stderr:
        pytest.skip("RCCL could not bring up a one-rank communicator here")
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["step_mode"]["graph"] is False and "attempts" not in out["step_mode"] and out["value"] > 0 and "strong_scaling" in out


def test_bench_rehearsal_keeps_the_headline_when_the_strong_region_fails():
    """the weak-scaling line is ready before the strong-scaling region starts: an exception there (injected) must not cost the job's
    `value` — rc 0, one line, no `strong_scaling`, the error named"""
    import json
    r, lines = _bench(["--dp-rehearsal", "--workload", "tiny", "--steps", "3", "--warmup", "2", "--fault", "0:strong", "--no-eager-first"])
    if r.returncode != 0 and "init_process_group" in r.stderr:
        pytest.skip("RCCL could not bring up a one-rank communicator here")
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert "strong_scaling" not in out and "injected" in out["strong_scaling_error"] and out["value"] > 0 and out["scaling"] == "weak"
