"""The reference's one entry point under data parallelism (run_expid.py): launched the way a torch.distributed
launcher would (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment), 2 ranks over gloo on CPU with the kernels through the
host emulation.  `batch_size` is the GLOBAL batch: each rank trains on its half of every batch, BatchNorm statistics, the dense-net
gradients and the table row lists are exchanged, rank 0 alone writes the checkpoint and the result line — and the checkpoint must be
the one a single process writes for the same config and seed (up to fp32 summation order)."""
import os
import subprocess
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

RUNNER = r'''
import os, sys
ROOT = %r
sys.path[:0] = [ROOT, os.path.join(ROOT, "www24-rat_amd"), os.path.join(ROOT, "tests", "emu")]
import build_emu
import rat_amd._lib as L
L._default = L.RatLib(build_emu.build())
import run_expid
run_expid.main(["--config", os.path.join(ROOT, "tests", "fixtures_cfg", "RAT_m2", "demo_dp"), "--expid", "RAT_m2_demo", "--gpu", "-1",
                "--synthetic", "32", "--epochs", "1"])
''' % ROOT


def _run(workdir, env_extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    return subprocess.Popen([sys.executable, "-c", RUNNER], cwd=str(workdir), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT,
                            text=True)


def test_sharded_batch_sources_cover_each_global_batch_once():
    sys.path.insert(0, os.path.join(ROOT, "www24-rat_amd"))
    from rat_amd import data as rd
    rs = np.random.RandomState(0)
    n, L, K = 37, 3, 2
    data = np.concatenate([rs.randint(0, 9, size=(n, L)), rs.randint(0, 2, size=(n, 1))], axis=1).astype(np.float64)
    idx, val, lens = rs.randint(0, n, size=(n, K)), rs.rand(n, K), np.full(n, K)
    full = list(rd.RetrievalBatches(data, data, idx, val, lens, 8, shuffle=True, seed=5))
    parts = [list(rd.RetrievalBatches(data, data, idx, val, lens, 8, shuffle=True, seed=5, shard=(r, 2))) for r in (0, 1)]
    assert len(parts[0]) == len(parts[1]) == len(full)
    for b, (X, y, v, ln) in enumerate(full):
        per = X.shape[0] // 2
        for r in (0, 1):
            Xr, yr, vr, lr = parts[r][b]
            assert torch.equal(Xr, X[r * per:(r + 1) * per]) and torch.equal(yr, y[r * per:(r + 1) * per])
            assert Xr.shape[0] == per                      # (the tail batch of 5 rows: 2 + 2, one sample dropped)


def test_run_expid_two_ranks_equals_one_process(tmp_path):
    sys.path.insert(0, os.path.join(HERE, "emu"))
    import build_emu
    build_emu.build()                                       # once, before the workers race for it
    single, dp = tmp_path / "single", tmp_path / "dp"
    single.mkdir(), dp.mkdir()
    port = 31500 + (os.getpid() % 2000)
    procs = [_run(single, {})]
    procs += [_run(dp, dict(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port)))
              for r in (0, 1)]
    outs = [p.communicate(timeout=1500)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    rel = os.path.join("_demo_models", "demo_x1_retrieval")
    a = torch.load(str(single / rel / "RAT_m2_demo.model"), map_location="cpu")
    b = torch.load(str(dp / rel / "RAT_m2_demo.model"), map_location="cpu")
    assert list(a) == list(b)
    for k in a:
        if k.endswith("num_batches_tracked"):
            assert int(a[k]) == int(b[k]), k
            continue
        if k.startswith("dnn.dnn.0.bias") or k.endswith("running_mean"):
            continue        # bias in front of BatchNorm: true gradient 0, Adam turns rounding noise into +-lr steps (and the batch mean follows)
        np.testing.assert_allclose(b[k].numpy(), a[k].numpy(), rtol=2e-4, atol=3e-6, err_msg=k)
    # rank 0 alone wrote the result line; its metrics are the single process's (same weights up to rounding, same validation set)
    lines = open(str(dp / rel / "RAT_m2_demo.csv")).read().strip().splitlines()
    assert len(lines) == 1 and "[val] AUC" in lines[0]
    assert os.path.exists(str(dp / rel / "RAT_m2_demo.rank1.log"))
    one = open(str(single / rel / "RAT_m2_demo.csv")).read().strip().splitlines()[0]
    val = lambda line: float(line.split("[val] AUC: ")[1].split(" ")[0])          # noqa: E731
    assert abs(val(lines[0]) - val(one)) < 1e-3
