"""Data parallelism on the MI355X box (-m gpu): TWO ranks share the one GPU of the box (RCCL refuses two ranks on a device, so the
process group is gloo and the collectives stage through the host — `RAT_m2._staged`), every kernel is the HIP build, and from the
third step on each rank REPLAYS its captured training step: graph segments with the SyncBN exchanges, the dense-net all-reduce and the
table-gradient exchange running eagerly between them (rat_amd/graph.py).  This is the part of the N > 1 path that a 1-GPU box can
prove on real hardware; RCCL itself only runs in the driver's scaling job.

Checked: the two replicas stay bit-identical, and after five steps they hold what ONE process holds after five steps on the full
batch (2e-4, the tolerance of tests/test_dp_gloo.py) — once with the table gradients as all-gathered row lists, once as the dense
all-reduce."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

WORKER = r'''
import os, sys
ROOT, out, case_name, row_lists, steps, graph = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4] == "1", int(sys.argv[5]), sys.argv[6] == "1"
sys.path[:0] = [ROOT, os.path.join(ROOT, "www24-rat_amd"), os.path.join(ROOT, "tests")]
import torch
import torch.distributed as dist
import golden_cases as gc
import model_cases as mc
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
case = dict(gc.case_by_name(case_name))
model = mc.build_model(case, gpu=0, seed=1)
mc.load_weights(model, case)
batch = mc.batch_of(case)
if world > 1:
    per = batch[0].shape[0] // world
    batch = tuple(t[rank * per:(rank + 1) * per] for t in batch)
    model.row_list_exchange = row_lists
    model.graph_under_dp = graph
model.train()
losses = [float(model.train_step(batch)) for _ in range(steps)]
torch.cuda.synchronize()
graphs = [e[1] for e in getattr(model, "_step_graphs", {}).values() if e[1]]
segs = sum(isinstance(i, torch.cuda.CUDAGraph) for i in graphs[0].items) if graphs else 0
torch.save({"flat": model._flat.detach().cpu(), "losses": losses, "segments": segs, "closures": (len(graphs[0].items) - segs) if graphs else 0,
            "noise": sorted(mc.noise_tensors(model)), "offsets": dict(model._offsets),
            "sizes": {k: v.numel() for k, v in model._params.items()}}, out)
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
'''


def _spawn(out, case, row_lists, steps, env_extra, graph=True):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.Popen([sys.executable, "-c", WORKER, ROOT, str(out), case, "1" if row_lists else "0", str(steps), "1" if graph else "0"], env=env,
                            stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


@pytest.mark.parametrize("row_lists,graph", [(True, True), (False, True), (True, False)], ids=["row_lists", "dense_tables", "eager_default"])
def test_two_ranks_on_one_gpu_replay_graph_segments_and_match_the_full_batch(tmp_path, row_lists, graph):
    """`eager_default`: graph_under_dp off — what a data-parallel run does unless asked otherwise (the fused step with eager launches)"""
    assert torch.cuda.is_available()
    case, steps = "kkbox_shape", 5          # batch 8 -> 4 per rank; BatchNorm on (SyncBN), wide part, two 3-id bag fields, d = 16
    port = 32500 + (os.getpid() % 2000)
    procs = [_spawn(tmp_path / "single.pt", case, row_lists, steps, {})]
    procs += [_spawn(tmp_path / ("rank%d.pt" % r), case, row_lists, steps,
                     dict(RANK=str(r), LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)), graph) for r in (0, 1)]
    outs = [p.communicate(timeout=900)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    one = torch.load(str(tmp_path / "single.pt"))
    r0, r1 = torch.load(str(tmp_path / "rank0.pt")), torch.load(str(tmp_path / "rank1.pt"))
    assert torch.equal(r0["flat"], r1["flat"]), "replicas diverged"
    # one process: the whole step is ONE graph; a rank: segments with the collectives between them
    assert one["segments"] == 1 and one["closures"] == 0
    if graph:
        assert r0["segments"] == r1["segments"] >= 8 and r0["closures"] == r0["segments"] - 1, (r0["segments"], r0["closures"])
    else:
        assert r0["segments"] == r1["segments"] == 0
    keep = torch.ones_like(one["flat"], dtype=torch.bool)
    for name in one["noise"]:                # biases in front of BatchNorm: true gradient 0, Adam steps on rounding noise
        keep[one["offsets"][name]:one["offsets"][name] + one["sizes"][name]] = False
    a, b = r0["flat"][keep].double(), one["flat"][keep].double()
    bad = (a - b).abs() > 3e-6 + 3e-4 * b.abs()
    assert float(bad.double().mean()) < 2e-3 and float((a - b).abs().max()) <= 1.05e-2, (float(bad.double().mean()), float((a - b).abs().max()))
    for s in range(steps):                   # each rank reports (local BCE + reg) / world
        assert abs(r0["losses"][s] + r1["losses"][s] - one["losses"][s]) < 2e-5, s
