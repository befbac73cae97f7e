#!/usr/bin/env python3
"""bench.py — training samples/sec of the RAT_m2 hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full training iteration of BaseModel.train_one_epoch on one synthetic batch that is already
resident in HBM: zero_grad -> forward -> BCE(+L2) -> backward -> [all-reduce] -> clip_grad_norm(10) -> Adam.
Workload at every N: BASELINE.json configs[1] (F=20 fields, 1M-row vocab, K=10 retrieved, d=64, batch 4096 PER GPU,
weak scaling), KKBox hyper-parameters for what BASELINE.json leaves open.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "www24-rat_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2, dense, exact fp32
PEAK_HBM_GBS = 8000.0             # HBM3E spec
# HBM bytes per launch from the PMC passes (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE), collected with rocprofv3 in
# separate runs (bench.py cannot host the profiler) and committed next to the kernel stats; valid for the north-star shapes only
TRAFFIC_FILE = os.path.join(ROOT, "profiles", "round1", "r1_traffic_pmc.json")


def pmc_traffic(kernel, workload):
    if workload != "synthetic_F20_V1M_K10_d64_B4096" or not os.path.exists(TRAFFIC_FILE):
        return None
    try:
        with open(TRAFFIC_FILE) as f:
            return json.load(f)["kernels"][kernel]["hbm_bytes_per_launch"]
    except (KeyError, ValueError):
        return None


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="synthetic_F20_V1M_K10_d64_B4096")
    ap.add_argument("--model", default="RAT_m2", choices=["RAT_m2", "RAT_m1", "RAT_m3", "RAT_m0"],
                    help="RAT_m2 (default) is the BASELINE.json metric; RAT_m1 times the cascaded variant (SURVEY §8f rank 2) on the same workload")
    ap.add_argument("--step-times", action="store_true", help="print the host-side issue time of every timed step to stderr")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-batch", type=int, default=256)
    ap.add_argument("--time-all-kernels", action="store_true",
                    help="HIP-event timing around EVERY C-ABI launch (adds ~2 events x 60 launches of host work per step); "
                         "default: only the encoder kernels (attention / FFN forward and backward), which hold >90 %% of the step")
    return ap.parse_args()


class KernelTimer:
    """HIP-event timing of every C-ABI launch on torch's current stream (the stream the kernels are launched on)."""

    HEAVY = ("rat_attn_fwd", "rat_attn_bwd", "rat_attn_fwd_ex", "rat_attn_bwd_ex", "rat_ffn_fwd", "rat_ffn_bwd", "rat_ffn_fwd_res",
             "rat_ffn_bwd_res", "rat_attn_core_fwd", "rat_attn_core_bwd", "rat_attn_core_fwd_map", "rat_attn_core_bwd_map")
    ATTN_ARGS = {"rat_attn_fwd": (5, 7), "rat_attn_bwd": (9, 11), "rat_attn_fwd_ex": (6, 8), "rat_attn_bwd_ex": (10, 12)}   # (map, heads)

    def __init__(self, lib, everything=False):
        self.lib, self.inner, self.records, self.enabled, self.everything = lib, lib.call, [], False, everything
        self.calls_seen, self.pool = 0, []
        lib.call = self._call

    def prepare(self, steps, warmup_steps):
        """Create (and once record, which is what actually creates the HIP event) every event the timed region will need, from
        the launch count seen during warm-up: hipEventCreate inside the timed loop would be the benchmark timing itself."""
        need = 2 * (self.calls_seen // max(warmup_steps, 1) + 8) * steps
        self.pool = [torch.cuda.Event(enable_timing=True) for _ in range(need)]
        for ev in self.pool:
            ev.record()
        torch.cuda.synchronize()

    def _event(self):
        return self.pool.pop() if self.pool else torch.cuda.Event(enable_timing=True)

    def _call(self, name, *args):
        timed = self.everything or name in self.HEAVY
        if not self.enabled or not timed:
            self.calls_seen += 1 if timed else 0
            return self.inner(name, *args)
        s, e = self._event(), self._event()
        s.record()
        self.inner(name, *args)
        e.record()
        self.records.append((name, self._tag(name, args), s, e))

    @staticmethod
    def _tag(name, args):
        if name in KernelTimer.ATTN_ARGS:                                # "L<seq len>h<heads of this launch>" (grouped mode: heads / 4)
            mi, hi = KernelTimer.ATTN_ARGS[name]
            return "L%dh%d" % (args[mi]._obj.L, int(args[hi]))
        if name in ("rat_attn_core_fwd", "rat_attn_core_bwd"):           # RAT_m0: joint sequences of T*S tokens
            return "L%d" % int(args[4 if name == "rat_attn_core_fwd" else 6])
        if name in ("rat_attn_core_fwd_map", "rat_attn_core_bwd_map"):
            return "L%d" % args[3 if name == "rat_attn_core_fwd_map" else 5]._obj.L
        if name in ("rat_ffn_fwd_res", "rat_ffn_bwd_res"):               # RAT_m1 runs the block MLP at two token counts
            return "n%d" % int(args[7 if name == "rat_ffn_fwd_res" else 13])
        return ""

    def summary(self, steps):
        out = {}
        for name, tag, s, e in self.records:
            d = out.setdefault((name, tag), [0, 0.0])
            d[0] += 1
            d[1] += s.elapsed_time(e)
        return {k: dict(launches_per_step=v[0] / steps, avg_ms=v[1] / v[0], ms_per_step=v[1] / steps) for k, v in out.items()}


def algorithmic_work(spec, model="RAT_m2"):
    """-> f(kernel name, tag) = (bound, FLOPs or bytes per LAUNCH) or None (padded MFMA lanes and recompute do not
    count; backward = 2x forward).  Tags come from KernelTimer._tag."""
    B, F, K, d = spec["batch"], spec["F"], spec["K"], spec["d"]
    T, S = K + 1, F + 1
    dh, H = spec["dim_head"], d * spec["scale_dim"]
    tok = B * T * S

    def tokens_of(L):                          # RAT_m1's cross transformer sees one token per sample; everything else the grid
        return B * T if (model == "RAT_m1" and L == T) else tok

    def work(name, tag):
        bwd = 2 if "_bwd" in name else 1
        if name in KernelTimer.ATTN_ARGS:                                # fused kernel: projections of `heads` heads + core
            L, h = [int(v) for v in tag[1:].split("h")]
            inner = h * dh
            return "mfma", bwd * tokens_of(L) * (8 * d * inner + 4 * inner * L)
        if name.startswith("rat_attn_core"):                             # core only (fp32 VALU, priced against the fp32 peak)
            L = int(tag[1:])
            return "mfma", bwd * tokens_of(L) * 4 * spec["num_heads"] * dh * L
        if name in ("rat_ffn_fwd", "rat_ffn_bwd"):
            return "mfma", bwd * tok * 4 * d * H
        if name in ("rat_ffn_fwd_res", "rat_ffn_bwd_res"):
            return "mfma", bwd * int(tag[1:]) * 4 * d * H
        if name == "rat_gather_fwd":
            return "hbm", B * (T * F * d * 4 + T * S * d * 4 + T * F * 4)
        if name == "rat_gather_bwd":
            return "hbm", B * (T * S * d * 4 + 2 * T * F * d * 4 + T * F * 4)
        return None
    return work


def cpu_baseline(spec, fm, batch_size, seed, model="RAT_m2"):
    """The oracle (a port: oracle/rat_m2_oracle.py, pinned to the reference's golden vectors) timed on this box's host
    cores on a bounded sample of the same workload: full training steps at a reduced batch."""
    from oracle import rat_m2_oracle as orc
    from rat_amd import synthetic
    # many-core hosts thrash on these small ops with one thread per core; 32 threads is where the oracle peaks
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    cfg = orc.Config(fields=orc.fields_from_specs(fm.feature_specs), embedding_dim=spec["d"], num_heads=spec["num_heads"],
                     dim_head=spec["dim_head"], depth=spec["depth"], scale_dim=spec["scale_dim"],
                     dnn_hidden_units=tuple(spec["dnn_hidden_units"]), batch_norm=spec["batch_norm"], use_wide=spec["use_wide"],
                     embedding_regularizer=0.0005, learning_rate=spec["learning_rate"],
                     variant={"RAT_m2": "m2", "RAT_m1": "m1", "RAT_m3": "m3", "RAT_m0": "m0"}[model])
    g = torch.Generator().manual_seed(seed)
    w = {}
    for name, shp in orc.parameter_shapes(cfg).items():         # reference-like init scales
        if len(shp) == 2 and "embedding_layer" in name:
            w[name] = torch.randn(shp, generator=g) * (1.0 if name.startswith("label") else 1e-4)
        elif len(shp) == 2:
            w[name] = torch.randn(shp, generator=g) * (2.0 / (shp[0] + shp[1])) ** 0.5
        elif name.endswith("norm.weight") or (name.startswith("dnn.") and name.endswith("weight")):
            w[name] = torch.ones(shp)
        else:
            w[name] = torch.zeros(shp)
    layers, _ = orc.dnn_layout(cfg)
    for _, bn in layers:
        if bn is not None:
            n = w["dnn.dnn.%d.weight" % bn].shape[0]
            w["dnn.dnn.%d.running_mean" % bn] = torch.zeros(n)
            w["dnn.dnn.%d.running_var" % bn] = torch.ones(n)
            w["dnn.dnn.%d.num_batches_tracked" % bn] = torch.zeros((), dtype=torch.int64)
    X, y, _, _ = synthetic.make_batch(spec, fm, seed=seed, batch=batch_size)
    state = {}
    w, *_ = orc.train_step(w, X, y, cfg, state, 1)                    # warm-up
    t0 = time.perf_counter()
    nsteps = 2
    for s in range(nsteps):
        w, *_ = orc.train_step(w, X, y, cfg, state, 2 + s)
    dt = time.perf_counter() - t0
    return dict(value=batch_size * nsteps / dt, unit="samples/s", cores=torch.get_num_threads(), kind="port",
                sample="%d full training steps (fwd+loss+bwd+clip+Adam) of the CPU oracle at batch %d of the same workload "
                       "(F=%d, K=%d, d=%d, %d-row vocab), %d torch threads" % (nsteps, batch_size, spec["F"], spec["K"], spec["d"],
                                                                                spec["total_vocab"], torch.get_num_threads()))


def main():
    args = parse()
    import torch.distributed as dist
    from rat_amd import synthetic
    from rat_amd.base_model import seed_everything
    from rat_amd import models

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)

    spec = synthetic.WORKLOADS[args.workload]
    fm = synthetic.feature_map_for(args.workload, spec)
    seed_everything(2021)
    model = getattr(models, args.model)(fm, **synthetic.model_kwargs(spec, gpu=local_rank))
    batch = synthetic.make_batch(spec, fm, seed=1000 + rank, device=model.device)
    model.train()
    timer = KernelTimer(model._lib, everything=args.time_all_kernels)

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # W untimed warm-up steps.  The one-off host work (creating the timing events, gc.freeze() — what fit_generator does before
    # its first batch: no full-heap GC walks mid-loop) happens BEFORE the last of them, so that the timed region starts on a busy
    # device instead of one that idled (and clocked down) through ~100 ms of host-only set-up.
    for _ in range(max(args.warmup - 1, 0)):
        model.train_step(batch)
    sync()
    timer.prepare(args.steps, max(args.warmup - 1, 1))
    model.freeze_host_heap()
    if args.warmup > 0:
        model.train_step(batch)
    sync()
    timer.enabled = True
    t0 = time.perf_counter()
    stamps = []
    for _ in range(args.steps):
        if args.step_times:
            stamps.append(time.perf_counter())
        model.train_step(batch)
    sync()
    if args.step_times:                # diagnostic: host-side issue time of every step (stderr), e.g. to spot interpreter stalls
        stamps.append(time.perf_counter())
        print("step issue times (ms): " + " ".join("%.1f" % ((b - a) * 1e3) for a, b in zip(stamps, stamps[1:])), file=sys.stderr)
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=model.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])

    if rank == 0:
        B = spec["batch"]
        ksum = timer.summary(args.steps)
        work = algorithmic_work(spec, args.model)
        kernels = []
        for key, st in sorted(ksum.items(), key=lambda kv: -kv[1]["ms_per_step"]):
            row = dict(kernel=key[0] + (":" + key[1] if key[1] else ""), launches_per_step=round(st["launches_per_step"], 2),
                       avg_ms=round(st["avg_ms"], 4), ms_per_step=round(st["ms_per_step"], 4))
            wk = work(*key)
            if wk:
                bound, amount = wk
                if bound == "mfma":
                    row.update(bound="mfma", achieved=round(amount / (st["avg_ms"] * 1e-3) / 1e12, 3), unit="TFLOP/s")
                else:
                    row.update(bound="hbm", achieved=round(amount / (st["avg_ms"] * 1e-3) / 1e9, 1), unit="GB/s")
            kernels.append(row)
        # dominant kernel = the C-ABI entry point with the most time per step (all its launches, both attention phases pooled)
        pooled = {}
        for key, st in ksum.items():
            p = pooled.setdefault(key[0], dict(ms=0.0, n=0.0, amount=0.0, bound=None))
            p["ms"] += st["ms_per_step"]
            p["n"] += st["launches_per_step"]
            wk = work(*key)
            if wk:
                p["bound"] = wk[0]
                p["amount"] += wk[1] * st["launches_per_step"]
        dom_name, dom = max(((k, v) for k, v in pooled.items() if v["bound"]), key=lambda kv: kv[1]["ms"])
        avg_s = dom["ms"] / dom["n"] * 1e-3
        per_launch = dom["amount"] / dom["n"]
        if dom["bound"] == "mfma":
            achieved, peak, unit = per_launch / avg_s / 1e12, PEAK_F32_MFMA_TFLOPS, "TFLOP/s"
        else:
            achieved, peak, unit = per_launch / avg_s / 1e9, PEAK_HBM_GBS, "GB/s"
        roofline = dict(kernel=dom_name, bound=dom["bound"], achieved=round(achieved, 3), peak=peak, unit=unit,
                        frac=round(achieved / peak, 4), traffic=pmc_traffic(dom_name, args.workload) if args.model == "RAT_m2" else None,
                        traffic_unit="bytes/launch (rocprofv3 PMC: 2 x FETCH_SIZE + WRITE_SIZE, profiles/round1/r1_traffic_pmc.json)",
                        algorithmic=round(per_launch, 1), algorithmic_unit="FLOP/launch" if dom["bound"] == "mfma" else "bytes/launch",
                        avg_launch_ms=round(avg_s * 1e3, 4), launches_per_step=round(dom["n"], 2))
        result = dict(metric="training samples/sec at B=4096, K=10 retrieved, d=64; 1/2/4/8 MI355X", value=round(B * world * args.steps / elapsed, 1),
                      unit="samples/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                      ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True, scaling="weak", vs_baseline=None,
                      dtype="f32", data="synthetic",
                      config=dict(workload=args.workload, fields=spec["F"], vocab_rows=spec["total_vocab"], retrieved=spec["K"],
                                  embedding_dim=spec["d"], batch_per_gpu=B, global_batch=B * world, heads=spec["num_heads"],
                                  dim_head=spec["dim_head"], depth=spec["depth"], scale_dim=spec["scale_dim"],
                                  dnn=spec["dnn_hidden_units"], step="fwd+bwd+clip+adam",
                                  parallelism="dp%d" % world),
                      roofline=roofline, kernels=kernels)
        if args.model != "RAT_m2":
            result["config"]["variant"] = args.model
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(spec, fm, args.cpu_batch, seed=1000, model=args.model)
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
