"""bench.py's contract around the measurement: `python bench.py --gpus N` (the form the driver uses) must start its N worker
processes itself, run the weak- AND the strong-scaling region and print ONE JSON line.  There is no GPU here, so the workers
run `--dry-run-cpu`: gloo process group, the host-emulation build of the kernel sources, a toy workload — the numbers mean
nothing, the plumbing (self-launch, rank environment, SyncBN collectives, batch rotation, JSON schema) is what is checked."""
import json
import os
import subprocess
import pytest
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_gpus2_self_launch_dry_run():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 1 and out["scaling"] == "weak" and out["dry_run_cpu"] is True
    assert out["config"]["global_batch"] == 2 * out["config"]["batch_per_gpu"] and out["config"]["sync_batch_norm"] is True
    st = out["strong_scaling"]
    assert st["global_batch"] == out["config"]["batch_per_gpu"] and st["batch_per_gpu"] * 2 == st["global_batch"]
    for key in ("metric", "value", "unit", "ms_per_step", "higher_is_better", "vs_baseline", "dtype", "arith", "data", "roofline",
                "targets", "kernels"):
        assert key in out, key
    assert out["metric"] == json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    # (no rat_gather_bwd under data parallelism: the table gradients are produced as row lists for the exchange)
    assert {"rat_gather_fwd", "cross_attention"} <= set(out["targets"])
    assert out["step_mode"]["graph"] is False             # hipGraph capture needs a GPU
    assert "cpu_baseline" not in out                      # N = 1 only (and never in a dry run)
    # what the N > 1 line says about its communication: world as torch.distributed reports it, and per region the
    # exchange form with the bytes it put on the links and the per-phase times of the step
    assert out["world"] == 2 and out["first_barrier_s"] is not None
    for region in ("weak", "strong"):
        comm = out["communication"][region]
        assert comm["world"] == 2 and comm["backend"] == "gloo" and comm["exchange_form"] in ("owner_lists", "gathered_lists", "dense_allreduce")
        assert {"exchange", "optimizer", "sync_bn"} <= set(comm["phases_ms_per_step"])
        assert comm["dense_net_bytes_sent_per_rank"] > 0 and comm["table_bytes_sent_per_rank"] is not None
        if comm["exchange_form"] == "owner_lists":
            assert comm["owner_lists"]["collectives_per_step"] == 3 and comm["owner_lists"]["pairs_sent"] > 0


def _clean_env():
    return {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}


@pytest.mark.parametrize("where", ["init", "barrier", "step"])
def test_bench_launcher_ends_the_job_when_one_rank_dies(where, tmp_path):
    """a rank that dies (before the rendezvous, after the first barrier, inside the timed region) must not leave
    rank 0 waiting in a collective: the launcher sees the exit code, stops the other rank and returns non-zero — within a minute."""
    import time
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--steps", "2", "--warmup", "1",
                        "--fault", "1:" + where, "--log-dir", str(tmp_path), "--init-timeout", "600"],
                       capture_output=True, text=True, timeout=600, env=_clean_env())
    took = time.monotonic() - t0
    assert r.returncode == 17, (r.returncode, r.stderr[-2000:])
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], "no result line from a job that lost a rank"
    assert "rank 1 (exit code 17)" in r.stderr and "---- rank 0" in r.stderr
    assert os.path.exists(os.path.join(str(tmp_path), "rank1.err"))
    # `init` / `barrier` die before any kernel is built or run: the launcher's reaction time is what is measured
    if where != "step":
        assert took < 60.0, took


def test_a_rank_that_fails_in_the_strong_region_does_not_cost_the_weak_scaling_line(tmp_path):
    """rank 1 raises inside the strong-scaling region and leaves; rank 0 sits in a collective whose own timeout is
    --init-timeout.  The region guard must fire BEFORE that (bench.py clamps --region-timeout to 0.8 x --init-timeout; here 600 -> 32 s):
    every rank exits 0 and rank 0's line comes out with the weak-scaling value and `strong_scaling_error` instead of `strong_scaling`."""
    import time
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--steps", "1", "--warmup", "1",
                        "--fault", "1:strong", "--log-dir", str(tmp_path), "--init-timeout", "40", "--region-timeout", "600"],
                       capture_output=True, text=True, timeout=900, env=_clean_env())
    took = time.monotonic() - t0
    assert r.returncode == 0, (r.returncode, r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-2000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    assert "strong_scaling" not in out and "strong_scaling_error" in out, sorted(out)
    assert took < 400, took


def test_a_rank_that_dies_in_the_strong_region_leaves_the_weak_scaling_line_behind(tmp_path):
    """a rank that DIES (exit code 17, like a segfault / OOM kill) inside the strong-scaling region: the launcher stops the others and returns
    17 — and rank 0, whose weak-scaling measurement was complete, writes that line with a `terminated` note when it is stopped (a watcher
    thread takes SIGTERM with sigwait, so it is served even while the main thread sits in a collective); the launcher relays it"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--steps", "1", "--warmup", "1",
                        "--fault", "1:strongexit", "--log-dir", str(tmp_path), "--init-timeout", "120"],
                       capture_output=True, text=True, timeout=900, env=_clean_env())
    assert r.returncode == 17, (r.returncode, r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout, r.stderr[-2000:])
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0 and "strong_scaling" not in out
    assert "strong_scaling_error" in out          # (+ `terminated` when the launcher's SIGTERM got there before the dead peer surfaced in a collective)


def test_bench_launcher_deadline(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry-run-cpu", "--steps", "1", "--warmup", "1",
                        "--launch-timeout", "1", "--log-dir", str(tmp_path)],
                       capture_output=True, text=True, timeout=300, env=_clean_env())
    assert r.returncode == 124, (r.returncode, r.stderr[-1500:])
    assert "launch timeout" in r.stderr


@pytest.mark.skipif(os.environ.get("RAT_CPU_FULL") != "1", reason="two minutes of 8 emulated ranks; the 2-rank launch above and "
                    "tests/test_dp_gloo.py::test_eight_rank_step_equals_full_batch_step cover the same code; RAT_CPU_FULL=1 runs it")
def test_bench_gpus8_self_launch_dry_run():
    """the scaling job's largest invocation, `python bench.py --gpus 8`, end to end on CPU (8 gloo ranks, host-emulated kernels): weak region
    on the dense all-reduce, strong region (one sample per rank) on the owner exchange with eight owners, one JSON line"""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run-cpu", "--steps", "1", "--warmup", "1"],
                       capture_output=True, text=True, timeout=1500, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["world"] == 8 and out["config"]["global_batch"] == 8 * out["config"]["batch_per_gpu"]
    assert out["strong_scaling"]["batch_per_gpu"] * 8 == out["strong_scaling"]["global_batch"]
    comm = out["communication"]
    assert comm["weak"]["world"] == 8 and comm["strong"]["exchange_form"] in ("owner_lists", "dense_allreduce", "gathered_lists")
    if comm["strong"]["exchange_form"] == "owner_lists":
        assert comm["strong"]["owner_lists"]["collectives_per_step"] == 3
