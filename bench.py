#!/usr/bin/env python3
"""bench.py — training samples/sec of the RAT_m2 hot path on MI355X (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--no-cpu-baseline]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full training iteration of BaseModel.train_one_epoch on one synthetic batch that is already
resident in HBM: zero_grad -> forward -> BCE(+L2) -> backward -> [all-reduce] -> clip_grad_norm(10) -> Adam.
The timed loop rotates over NBATCH distinct batches (different ids every step).

`--gpus N` without a torch.distributed launcher around it starts the N worker processes itself (children are spawned before
this process touches the GPU; nothing is exec'ed) — one process per GPU, RCCL process group over 127.0.0.1.

Workload: BASELINE.json configs[1] (F=20 fields, 1M-row vocab, K=10 retrieved, d=64), KKBox hyper-parameters for what
BASELINE.json leaves open.  Partitioning at N > 1 (pure data parallelism over the batch):
  * `value` / `ms_per_step`  — WEAK scaling: every rank trains on its own B = 4096 batch (global batch 4096 N);
  * `strong_scaling`         — the SAME line also carries the strong-scaling measurement: ONE global batch of 4096 samples,
                               rank r takes rows [r B/N, (r+1) B/N) — §8e's partitioning — timed right after the weak run.
BatchNorm statistics are exchanged (SyncBN) so that N ranks compute what one device would on the global batch.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "www24-rat_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2, dense, exact fp32
PEAK_BF16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: dense bf16 MFMA (the sparsity figures are never used)
PEAK_HBM_GBS = 8000.0             # HBM3E spec
ROCPROF_NOTE = "profiles/round6/r6_bench_kernel_stats.csv (in the step), r6_gather_V100M_kernel_stats.csv (25.6 GB table)"
NBATCH = 4                        # distinct batches rotated through the timed loop
# HBM bytes per launch from the PMC passes (FETCH_SIZE x2 per the gfx950 correction + WRITE_SIZE), collected with rocprofv3 in
# separate runs (bench.py cannot host the profiler) and committed next to the kernel stats; valid for the north-star shapes only
TRAFFIC_FILES = [os.path.join(ROOT, "profiles", "round6", "r6_traffic_pmc.json"),
                 os.path.join(ROOT, "profiles", "round5", "r5_traffic_pmc.json"),
                 os.path.join(ROOT, "profiles", "round4", "r4_traffic_pmc.json"),
                 os.path.join(ROOT, "profiles", "round3", "r3_traffic_pmc.json"),
                 os.path.join(ROOT, "profiles", "round2", "r2_traffic_pmc.json"),
                 os.path.join(ROOT, "profiles", "round1", "r1_traffic_pmc.json")]


def pmc_traffic(kernel, workload):
    if workload != "synthetic_F20_V1M_K10_d64_B4096":
        return None, None
    for path in TRAFFIC_FILES:
        try:
            with open(path) as f:
                return json.load(f)["kernels"][kernel]["hbm_bytes_per_launch"], os.path.relpath(path, ROOT)
        except (OSError, KeyError, ValueError):
            continue
    return None, None


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="synthetic_F20_V1M_K10_d64_B4096")
    ap.add_argument("--model", default="RAT_m2", choices=["RAT_m2", "RAT_m1", "RAT_m3", "RAT_m0"],
                    help="RAT_m2 (default) is the BASELINE.json metric; the others time the variants (SURVEY §8f rank 2) on the same workload")
    ap.add_argument("--scaling", default="both", choices=["both", "weak", "strong"],
                    help="N > 1 only.  both (default): `value` = weak scaling (batch per GPU fixed) and a `strong_scaling` object "
                         "(one global batch split by rank) in the same line; weak / strong: only that measurement, as `value`")
    ap.add_argument("--arith", default=None, choices=["f32", "bf16x3"],
                    help="arithmetic of the encoder GEMMs (default: the library's default; both are timed when available)")
    ap.add_argument("--embedding-grad", default=None, choices=["atomic", "sorted", "sparse"],
                    help="how table gradients are produced (default: the workload's / the model's `auto` rule)")
    ap.add_argument("--step-times", action="store_true", help="print the host-side issue time of every timed step to stderr")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="time the eager fused step instead of hipGraph replays")
    ap.add_argument("--prune", action="store_true",
                    help="time the product's default step (dead-token pruning of the last block ON) as the headline instead of the "
                         "reference-work step; without it the pruned step is an extra object of the line")
    ap.add_argument("--dp-rehearsal", action="store_true",
                    help="with one rank: run the N > 1 code path anyway (RCCL process group of one rank, the model's data-parallel "
                         "path with every collective issued, barriers, weak and strong regions) — the 1-GPU rehearsal of the scaling job")
    ap.add_argument("--graph-dp", action="store_true",
                    help="N > 1: ALSO time hipGraph segments with the collectives between them (after the eager measurement, under a watchdog) "
                         "and report the faster form; default since round 6: the eager fused step only")
    ap.add_argument("--no-eager-first", action="store_true",
                    help="N > 1 with graphs: skip the eager measurement that is taken first as the fall-back of a stalled graph attempt")
    ap.add_argument("--graph-attempt-timeout", type=float, default=240.0,
                    help="N > 1 with graphs: seconds the segmented-graph regions may take before every rank gives up and rank 0 prints "
                         "the eager result")
    ap.add_argument("--region-timeout", type=float, default=600.0,
                    help="N > 1: seconds the strong-scaling region may take once the weak region's line is ready; after that every rank "
                         "exits 0 and rank 0 prints the line without `strong_scaling`")
    ap.add_argument("--no-graph-dp", action="store_true", help="(the default since round 6; kept so that older command lines still parse)")
    ap.add_argument("--no-group-loop", action="store_true",
                    help="diagnostic (wide heads: heads = G x 8 at d = 64): G forward launches per layer instead of the one that loops over "
                         "the head groups (rat_attn_fwd_groups)")
    ap.add_argument("--batch", type=int, default=0, help="diagnostic: override the workload's batch size (the line then names it in config)")
    ap.add_argument("--inference", action="store_true", help="add the `inference` object (eval forward, eager and hipGraph) even with --no-extras")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the side measurements of the default N = 1 line (exact-fp32 run, per-rank B/8 shape, 100 M-row gather)")
    ap.add_argument("--cpu-batch", type=int, default=0, help="batch of the CPU baseline (default 0 = the workload's own batch)")
    ap.add_argument("--cpu-steps", type=int, default=3, help="timed steps of the CPU baseline (after 1 warm-up step)")
    ap.add_argument("--time-all-kernels", action="store_true",
                    help="HIP-event timing around EVERY C-ABI launch (adds ~2 events x 60 launches of host work per step); "
                         "default: the encoder kernels (attention / FFN forward and backward, >90 %% of the step) and the "
                         "embedding gather / scatter")
    ap.add_argument("--log-dir", default=None, help="--gpus N self-launch: directory for every rank's stdout / stderr (default: a new temp dir)")
    ap.add_argument("--launch-timeout", type=float, default=3000.0, help="--gpus N self-launch: seconds after which the whole job is ended")
    ap.add_argument("--init-timeout", type=float, default=300.0,
                    help="N > 1: bound (s) on the process-group rendezvous and on every collective's completion (RCCL watchdog / gloo)")
    ap.add_argument("--fault", default=None,
                    help="test hook RANK:WHERE — that rank exits with code 17 at `init` (before the rendezvous), `barrier` (after the "
                         "first barrier) or `step` (inside the timed region): the launcher must notice and end the job; `strong`: that rank "
                         "raises inside the strong-scaling region: the line must still come out, without `strong_scaling`; `strongexit`: that "
                         "rank DIES there: rank 0 writes its last complete measurement when the launcher stops it, the job returns 17")
    ap.add_argument("--dry-run-cpu", action="store_true",
                    help="plumbing check without a GPU: gloo process group, the host-emulation build of the kernels (tests/emu), "
                         "workload `dryrun`; the printed numbers mean nothing")
    args = ap.parse_args(argv)
    # The guards of the N > 1 regions must fire BEFORE the process group's own collective timeout (= --init-timeout: RCCL's watchdog
    # aborts a rank that sits in a collective that long, SIGABRT, and the launcher then discards rank 0's line): a rank that raised
    # has left, the others wait in a collective, and only their own timer can still get the finished weak-scaling line out.
    limit = 0.8 * args.init_timeout
    args.region_timeout = min(args.region_timeout, limit)
    args.graph_attempt_timeout = min(args.graph_attempt_timeout, limit)
    return args


# ------------------------------------------------------------------------------------------------- self-launch
def self_launch(args):
    """`python bench.py --gpus N` on its own: start N workers (this file, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the
    environment) as CHILD processes — before anything in this process has touched the GPU; nothing is exec'ed — and WATCH them:
      * every rank's stderr goes to its own file (`--log-dir`, default a fresh directory under the system's temp dir); rank 0's
        stdout (the one JSON line) is relayed at the end, the other ranks' stdout goes to their log too;
      * all children are polled: the first non-zero exit ends the others (by PID: terminate, then kill) and this process exits
        non-zero at once instead of leaving rank 0 inside a collective until RCCL's own timeout;
      * `--launch-timeout` seconds bound the whole job the same way.
    Exit code: 0 only when every rank returned 0; otherwise the first failing rank's code (or 124 on the deadline), after the tail
    of every rank's log has been copied to stderr."""
    import tempfile
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    log_dir = args.log_dir or tempfile.mkdtemp(prefix="rat_bench_")
    os.makedirs(log_dir, exist_ok=True)
    procs, logs, outs = [], [], []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        err = open(os.path.join(log_dir, "rank%d.err" % r), "w")
        out = open(os.path.join(log_dir, "rank%d.out" % r), "w+")
        logs.append(err)
        outs.append(out)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env, stdout=out, stderr=err))
    print("bench.py: %d ranks started, logs in %s" % (args.gpus, log_dir), file=sys.stderr)
    deadline = time.monotonic() + args.launch_timeout
    failed = None                                   # (rank, return code) of the first rank seen failing
    while True:
        codes = [p.poll() for p in procs]
        bad = [(r, c) for r, c in enumerate(codes) if c not in (None, 0)]
        if bad:
            failed = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            failed = (-1, 124)
            break
        time.sleep(0.2)
    if failed is not None:
        for p in procs:                             # exactly the PIDs started above
            if p.poll() is None:
                p.terminate()
        t_end = time.monotonic() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, t_end - time.monotonic()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    for f in logs:
        f.close()
    outs[0].seek(0)
    text = outs[0].read()
    for f in outs:
        f.close()
    if failed is None:
        sys.stdout.write(text)
        sys.stdout.flush()
        return 0
    # a rank died AFTER rank 0 had a complete measurement (e.g. inside the strong-scaling region, once the weak one was done): rank 0's
    # termination handler wrote that line with a `terminated` note — relay it; the exit code still says that the job lost a rank
    # (or its strong-region guard did, when the dead peer surfaced as an exception in rank 0's collective first)
    salvaged = [ln for ln in text.splitlines() if ln.startswith("{") and ('"terminated"' in ln or '"strong_scaling_error"' in ln)]
    if salvaged:
        sys.stdout.write(salvaged[-1] + "\n")
        sys.stdout.flush()
    who = "the %d s launch timeout" % args.launch_timeout if failed[0] < 0 else "rank %d (exit code %d)" % failed
    print("bench.py: %s ended the job; the other ranks were stopped.  Log tails:" % who, file=sys.stderr)
    for r in range(args.gpus):
        try:
            with open(os.path.join(log_dir, "rank%d.err" % r)) as f:
                tail = f.read()[-1500:]
        except OSError:
            tail = "(no log)"
        print("---- rank %d (rc %s) ----\n%s" % (r, procs[r].returncode, tail), file=sys.stderr)
    return abs(failed[1]) or 1


# ------------------------------------------------------------------------------------------------- timing
class _HostEvent:
    """stand-in for torch.cuda.Event in --dry-run-cpu"""

    def record(self):
        self.t = time.perf_counter()

    def elapsed_time(self, other):
        return (other.t - self.t) * 1e3


class KernelTimer:
    """HIP-event timing of C-ABI launches on torch's current stream (the stream the kernels are launched on)."""

    HEAVY = ("rat_attn_fwd", "rat_attn_bwd", "rat_attn_fwd_ex", "rat_attn_bwd_ex", "rat_attn_fwd_groups", "rat_attn_bwd_groups", "rat_ffn_fwd", "rat_ffn_bwd", "rat_ffn_fwd_res",
             "rat_ffn_bwd_res", "rat_ffn_bwd_res_rows", "rat_attn_core_fwd", "rat_attn_core_bwd", "rat_attn_core_fwd_map", "rat_attn_core_bwd_map",
             "rat_gather_fwd", "rat_gather_bwd", "rat_gather_bwd_sorted")
    ATTN_ARGS = {"rat_attn_fwd": (5, 7), "rat_attn_bwd": (9, 11), "rat_attn_fwd_ex": (6, 8), "rat_attn_bwd_ex": (10, 12),
                 "rat_attn_fwd_groups": (8, 10), "rat_attn_bwd_groups": (11, 13)}                         # (map, heads)

    def __init__(self, lib, everything=False, host_events=False):
        self.lib, self.inner, self.records, self.enabled, self.everything = lib, lib.call, [], False, everything
        self.calls_seen, self.pool, self.host_events = 0, [], host_events
        lib.call = self._call

    def _new(self):
        import torch
        return _HostEvent() if self.host_events else torch.cuda.Event(enable_timing=True)

    def prepare(self, steps, warmup_steps):
        """Create (and once record, which is what actually creates the HIP event) every event the timed region will need, from
        the launch count seen during warm-up: hipEventCreate inside the timed loop would be the benchmark timing itself."""
        import torch
        need = 2 * (self.calls_seen // max(warmup_steps, 1) + 8) * steps
        self.pool = [self._new() for _ in range(need)]
        for ev in self.pool:
            ev.record()
        if not self.host_events:
            torch.cuda.synchronize()

    def _event(self):
        return self.pool.pop() if self.pool else self._new()

    def _call(self, name, *args):
        timed = self.everything or name in self.HEAVY
        if not self.enabled or not timed:
            self.calls_seen += 1 if timed else 0
            return self.inner(name, *args)
        s, e = self._event(), self._event()
        s.record()
        self.inner(name, *args)
        e.record()
        if name == "rat_ffn_bwd_res_rows":                               # the last block's backward (head gradient as compact rows): the same
            name, args = "rat_ffn_bwd_res", tuple(args[:2]) + tuple(args[3:])   # kernel, same algorithmic work: counted with its siblings
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
            return "n%d" % int(args[7 if name == "rat_ffn_fwd_res" else 14])
        return ""

    def summary(self, steps):
        out = {}
        for name, tag, s, e in self.records:
            d = out.setdefault((name, tag), [0, 0.0])
            d[0] += 1
            d[1] += s.elapsed_time(e)
        return {k: dict(launches_per_step=v[0] / steps, avg_ms=v[1] / v[0], ms_per_step=v[1] / steps) for k, v in out.items()}

    def reset(self):
        self.records = []


class PhaseTimer:
    """HIP-event pairs (on torch's current stream — the stream the launches and the collectives' waits are issued on) around the
    communication phases and the optimizer of a data-parallel step: the gradient exchange (`_exchange_gradients`: table lists or the
    dense all-reduce + the wait for the dense-net all-reduce started inside backward), the start-of-step count exchange of the owner
    form, SyncBN's forward / backward launches with their collectives, and the two-sweep optimizer.  Active only in the eager
    instrumented pass (a graph replay cannot host events)."""

    def __init__(self, model, host_events=False):
        from rat_amd import ops
        self.enabled, self.records, self.host_events = False, [], host_events
        self._wrap(model, "_exchange_gradients", "exchange")
        if hasattr(model, "_owner_prepare"):
            self._wrap(model, "_owner_prepare", "exchange_counts")
        self._wrap(model.optimizer, "fused_step", "optimizer")
        self._wrap(ops, "bn_relu_fwd_sync", "sync_bn")
        self._wrap(ops, "bn_relu_bwd_sync", "sync_bn")

    def _wrap(self, owner, name, tag):
        inner = getattr(owner, name)

        def timed(*a, **k):
            if not self.enabled:
                return inner(*a, **k)
            import torch
            s, e = (_HostEvent(), _HostEvent()) if self.host_events else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            s.record()
            try:
                return inner(*a, **k)
            finally:
                e.record()
                self.records.append((tag, s, e))
        setattr(owner, name, timed)

    def summary(self, steps):
        out = {}
        for tag, s, e in self.records:
            out[tag] = out.get(tag, 0.0) + s.elapsed_time(e)
        return {k: round(v / steps, 4) for k, v in out.items()}

    def reset(self):
        self.records = []


def algorithmic_work(spec, batch, model="RAT_m2"):
    """-> f(kernel name, tag) = (bound, FLOPs or bytes per LAUNCH) or None (padded MFMA lanes and recompute do not
    count; backward = 2x forward).  Tags come from KernelTimer._tag."""
    B, F, K, d = batch, spec["F"], spec["K"], spec["d"]
    T, S = K + 1, F + 1
    dh, H = spec["dim_head"], d * spec["scale_dim"]
    tok = B * T * S

    def tokens_of(L):                          # RAT_m1's cross transformer sees one token per sample; everything else the grid
        return B * T if (model == "RAT_m1" and L == T) else tok

    def work(name, tag):
        bwd = 2 if "_bwd" in name else 1
        if name in KernelTimer.ATTN_ARGS:                                # fused kernel: projections of `heads` heads + core
            L, h = [int(v) for v in tag[1:].split("h")]
            # RAT_m3 launches heads / 2 heads of width 2 * dim_head (RAT_m3.py:181): the tag's h counts THOSE heads
            inner = h * (2 * dh if model == "RAT_m3" else dh)
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
        if name in ("rat_gather_bwd", "rat_gather_bwd_sorted"):
            return "hbm", B * (T * S * d * 4 + 2 * T * F * d * 4 + T * F * 4)
        return None
    return work


def host_cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _cpu_oracle_run(workload, model, batch_size, seed, threads, timed_steps):
    """1 warm-up + `timed_steps` full training steps of the CPU oracle with `threads` torch threads -> (samples/s, s/step)"""
    import torch
    from oracle import rat_m2_oracle as orc
    from rat_amd import synthetic
    spec = synthetic.WORKLOADS[workload]
    fm = synthetic.feature_map_for(workload, spec)
    torch.set_num_threads(threads)
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
    for s in range(timed_steps):
        w, *_ = orc.train_step(w, X, y, cfg, state, 2 + s)
    dt = time.perf_counter() - t0
    return batch_size * timed_steps / dt, dt / timed_steps


def cpu_baseline(workload, spec, batch_size, seed, model="RAT_m2", timed_steps=3, all_cores_budget_s=45.0):
    """SURVEY.md §8d "CPU baseline beside it": the oracle (a port: oracle/rat_m2_oracle.py, pinned to the reference's golden vectors)
    on this box's host cores, SAME workload and batch as the GPU run, 1 warm-up + `timed_steps` (>= 3) timed full training steps
    (zero_grad -> loss(+reg) -> backward -> clip_grad_norm_(10) -> Adam).
      * 32 torch threads: run to completion, in this process (~2 min on the MI355X box's EPYC host at B = 4096);
      * torch.set_num_threads(os.cpu_count()) — the survey's literal recipe: on the 256-thread host it is ~12x SLOWER (measured:
        357 s per step against 31 s, one thread per hardware thread thrashes on the path's thousands of tiny per-head operations), so
        it runs afterwards in a CHILD process (CPU only) under a time budget; when the budget ends first, the child is ended by its PID and the
        entry says so instead of holding the driver's bench for 25 minutes.
    `value` = the fastest completed run; every attempt is listed in `runs`."""
    ncpu = os.cpu_count() or 1
    runs = []
    fast_threads = min(ncpu, 32)
    v, sps = _cpu_oracle_run(workload, model, batch_size, seed, fast_threads, timed_steps)
    runs.append(dict(threads=fast_threads, value=round(v, 1), s_per_step=round(sps, 3), steps="1 warm-up + %d timed" % timed_steps))
    if ncpu > fast_threads:
        cmd = [sys.executable, os.path.abspath(__file__), "--cpu-oracle-child", "%s,%s,%d,%d,%d,%d" % (workload, model, batch_size, seed, ncpu, timed_steps)]
        t0 = time.perf_counter()
        child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        try:
            out, _ = child.communicate(timeout=all_cores_budget_s)
            rec = json.loads(out.strip().splitlines()[-1])
            runs.append(dict(threads=ncpu, value=round(rec["value"], 1), s_per_step=round(rec["s_per_step"], 3),
                             steps="1 warm-up + %d timed" % timed_steps))
        except (subprocess.TimeoutExpired, ValueError, IndexError, KeyError):
            child.kill()
            child.wait()
            runs.append(dict(threads=ncpu, value=None, s_per_step=None, ended_after_s=round(time.perf_counter() - t0, 1),
                             note="torch.set_num_threads(os.cpu_count()) did not finish 1 + %d steps within the budget and was ended; "
                                  "measured once to completion on this host type: 357 s per step at B = 4096 (11.5 samples/s), "
                                  "profiles/round3" % timed_steps))
    done = [r for r in runs if r["value"]]
    best = max(done, key=lambda r: r["value"])
    import torch
    return dict(value=best["value"], unit="samples/s", cores=best["threads"], threads=best["threads"], runs=runs,
                host_cores=ncpu, cpu_model=host_cpu_model(), torch=torch.__version__, kind="port", sample_batch=batch_size,
                timed_steps=timed_steps, warmup_steps=1,
                sample="1 warm-up + %d timed full training steps (zero_grad, fwd, loss+reg, bwd, clip_grad_norm_(10), Adam) of the CPU "
                       "oracle at batch %d of the same workload (F=%d, K=%d, d=%d, %d-row vocab) on a %d-core host; value = the fastest "
                       "thread count of `runs`" % (timed_steps, batch_size, spec["F"], spec["K"], spec["d"], spec["total_vocab"], ncpu))


def big_table_gather(lib, device, rows_per_field=2_500_000, F=40, d=64, K=10, B=1024, reps=12):
    """rat_gather_fwd at BASELINE.json configs[3]'s per-rank shape on its REAL table: F = 40 fields x 2.5 M rows x 64 floats = 25.6 GB
    in HBM (uniform random ids: no cache can hold it), B = 1024, T = 11 -> 450 560 row reads of 256 B per launch.  The table is
    allocated and filled on the device for this measurement only and freed afterwards."""
    import torch
    from rat_amd import ops
    from types import SimpleNamespace
    T, S = K + 1, F + 1
    table = torch.empty((F * rows_per_field, d), dtype=torch.float32, device=device)
    table.normal_(0.0, 1e-2)
    fields = [SimpleNamespace(col=i, ncols=1, vocab=rows_per_field, padding_idx=None) for i in range(F)]
    tabs = [table[i * rows_per_field:(i + 1) * rows_per_field] for i in range(F)]
    ftab = ops.field_table(fields, tabs, device)
    label_tab = torch.randn(3, d, device=device)
    g = torch.Generator().manual_seed(11)
    idxs = [torch.randint(0, rows_per_field, (B, T, F), generator=g).to(torch.int32).to(device) for _ in range(4)]
    labels = torch.randint(0, 2, (B, T), generator=g).to(torch.int32).to(device)
    for i in range(3):
        ops.gather_fwd(idxs[i % 4], labels, ftab, F, label_tab, B, T, F, d, lib=lib)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s_, e_ in evs:
        s_.record(), e_.record()
    torch.cuda.synchronize()
    for i, (s_, e_) in enumerate(evs):
        s_.record()
        ops.gather_fwd(idxs[i % 4], labels, ftab, F, label_tab, B, T, F, d, lib=lib)
        e_.record()
    torch.cuda.synchronize()
    ms = sum(s_.elapsed_time(e_) for s_, e_ in evs) / reps    # the HIP-event pair as it stands: nothing subtracted
    nbytes = B * (T * F * d * 4 + T * S * d * 4 + T * F * 4)
    del table, tabs
    torch.cuda.empty_cache()
    return dict(bound="hbm", what="rat_gather_fwd alone at configs[3]'s per-rank shape (F=40, B=1024, K=10, d=64) on a 25.6 GB table "
                "(100 M rows), uniform ids, %d launches over 4 id sets" % reps, avg_launch_ms=round(ms, 4),
                timing="HIP-event pair around the launch, nothing subtracted; rocprofv3 durations of the same kernel: " + ROCPROF_NOTE,
                algorithmic_bytes=nbytes, achieved_GBps=round(nbytes / (ms * 1e-3) / 1e9, 1),
                frac_of_8TBps=round(nbytes / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4))


# ------------------------------------------------------------------------------------------------- the worker
def worker(args):
    # The last COMPLETE measurement of a data-parallel run (rank 0): when a peer dies later (the launcher / torchrun then sends SIGTERM to
    # the survivors) a watcher thread writes it out with a `terminated` note before the process ends.  SIGTERM is blocked HERE, before
    # torch (and with it every helper thread, which inherits the mask) exists, and taken with sigwait by the watcher — so that it is
    # served even while the main thread sits inside a collective or a device synchronisation.
    last_complete = {"line": None, "why": None, "emit": None}
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and int(os.environ.get("RANK", "0")) == 0:
        import signal
        import threading
        signal.pthread_sigmask(signal.SIG_BLOCK, {signal.SIGTERM})

        def _on_sigterm():
            signal.sigwait({signal.SIGTERM})
            try:
                if last_complete["line"] is not None and last_complete["emit"] is not None:
                    d = dict(last_complete["line"])
                    d["terminated"] = ("SIGTERM %s: a peer rank failed or the job was stopped; this is the last complete measurement"
                                       % last_complete["why"])
                    last_complete["emit"](json.dumps(d))
            finally:
                os._exit(143)
        threading.Thread(target=_on_sigterm, daemon=True).start()
    import torch
    import torch.distributed as dist
    from rat_amd import models, synthetic
    from rat_amd.base_model import seed_everything

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dry = args.dry_run_cpu
    if dry:
        sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))
        import build_emu
        import rat_amd._lib as L
        L._default = L.RatLib(build_emu.build())
        args.workload = "dryrun"
    else:
        assert torch.cuda.is_available(), "bench.py needs a GPU (the product path has no CPU fallback)"
        torch.cuda.set_device(local_rank)
    # --dp-rehearsal: ONE rank goes through everything the N > 1 run goes through — an RCCL process group, the data-parallel code path
    # of the model (dp_single_rank: every collective of the step is issued), the barriers and the max-over-ranks reduction of this
    # file, the weak AND the strong region.  It is the only way to put the N > 1 bench on the one GPU of a gpurun box.
    dp = world > 1 or args.dp_rehearsal
    first_barrier_s = None
    result_fd = None

    def emit(line):
        if result_fd is None:
            print(line)
            sys.stdout.flush()
        else:
            os.write(result_fd, (line + "\n").encode())
    last_complete["emit"] = emit

    def fault(where):
        if args.fault and args.fault == "%d:%s" % (rank, where):
            print("bench.py: --fault %s: rank %d exits" % (args.fault, rank), file=sys.stderr)
            sys.stderr.flush()
            os._exit(17)
    if dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        # RCCL prints a version banner on STDOUT when a communicator comes up (with eager initialisation: whenever its helper threads
        # get to it); stdout must carry the one JSON line only.  So for the whole life of a data-parallel worker file descriptor 1
        # points at stderr, and the result line is written to a private duplicate of the original stdout (`emit`).
        sys.stdout.flush()
        result_fd = os.dup(1)
        os.dup2(2, 1)
        import datetime
        fault("init")
        # bounded: a rank that never arrives fails the rendezvous of the others after --init-timeout instead of the 10 / 30 min
        # defaults; the same bound is what RCCL's watchdog / gloo apply to every later collective (a dead peer mid-run)
        dist.init_process_group("gloo" if dry else "nccl", rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=args.init_timeout),
                                **({} if dry else {"device_id": torch.device("cuda", local_rank)}))
        t_b = time.perf_counter()
        dist.barrier()
        if not dry:
            torch.cuda.synchronize()
        first_barrier_s = time.perf_counter() - t_b
        fault("barrier")

    spec = synthetic.WORKLOADS[args.workload]
    if args.batch:
        spec = dict(spec, batch=args.batch)
    fm = synthetic.feature_map_for(args.workload, spec)
    seed_everything(2021)
    gpu = -1 if dry else local_rank
    kwargs = synthetic.model_kwargs(spec, gpu=gpu)
    if args.embedding_grad is not None:
        kwargs["embedding_grad"] = args.embedding_grad
        if args.embedding_grad == "sparse":
            kwargs["embedding_regularizer"] = 0.0          # lazy row updates cannot carry the dense lambda*W term (declared)
    model = getattr(models, args.model)(fm, **kwargs)
    if args.arith is not None:
        model.set_arith(args.arith)
    if args.dp_rehearsal and world == 1:
        model.dp_single_rank = True
    B = spec["batch"]
    dev = model.device if not dry else None
    model.train()
    timer = KernelTimer(model._lib, everything=args.time_all_kernels, host_events=dry)
    phases = PhaseTimer(model, host_events=dry) if dp else None

    def sync():
        if dp:
            dist.barrier()
        if not dry:
            torch.cuda.synchronize()

    def make(seed, lo=None, hi=None):
        b = synthetic.make_batch(spec, fm, seed=seed, device=dev, as_float64=False)
        return b if lo is None else tuple(t[lo:hi].contiguous() for t in b)

    # The headline `value` is measured with the reference's amount of work: every token of every block.  The product's default
    # also prunes the last block's dead tokens (RAT_m2.prune_dead_tokens: identical predictions and gradients) — that step is timed
    # separately and reported beside it as `dead_token_pruning`; --prune makes it the measured step instead.
    can_prune = hasattr(model, "prune_dead_tokens") and args.model in ("RAT_m2", "RAT_m3")
    if can_prune:
        model.prune_dead_tokens = bool(args.prune)
    if args.no_group_loop:
        model.group_loop = False
    # Under data parallelism the segmented-graph form (graph segments with the collectives between them) is OPT-IN since round 6
    # (`--graph-dp`): measured with one RCCL rank it buys 0.4 % at B = 4096 (19.79 against 19.86 ms) and 4 % at the per-rank shape — and no
    # run with more than one RCCL rank exists yet, so the default N > 1 line is the eager fused step, the form with the fewest moving parts.
    graph_under_dp = bool(args.graph_dp) and not args.no_graph_dp
    graph_mode = bool(getattr(model, "use_graph", False)) and not dry and not args.no_graph and (not dp or graph_under_dp)
    model.use_graph = graph_mode
    model.graph_under_dp = graph_under_dp
    model.graph_shapes = 8                       # weak / strong / per-rank shapes and both arithmetics each get their own graph
    region_info = {}

    def timed_region(batches, steps, warmup, label, kernel_pass=True):
        """W untimed warm-up steps, then K steps between barrier + synchronize.  The one-off host work (creating the timing
        events, gc.freeze() — what fit_generator does before its first batch) happens BEFORE the last warm-up step, so that the
        timed region starts on a busy device instead of one that idled (and clocked down) through ~100 ms of host-only set-up.

        train_step() replays a captured hipGraph from its (graph_warmup + 1)-th call of a batch shape on.  The capture has to be over
        before the LAST warm-up step; when W is too small for that, extra untimed steps are run first (`graph_prepare_steps`).  A
        replay cannot host per-launch HIP events, so in graph mode the per-kernel numbers come from `kernel_pass`: the SAME K steps
        once more, eagerly (same kernels, same arguments, same stream), with HIP events around every timed launch — after the
        timed region, never inside it."""
        timer.enabled = False
        timer.reset()
        nb = len(batches)
        prep = 0
        if graph_mode:
            prep = max(0, model.graph_warmup + 1 - max(warmup - 1, 0))
            for i in range(prep):
                model.train_step(batches[i % nb])
        for i in range(max(warmup - 1, 0)):
            model.train_step(batches[i % nb])
        sync()
        if not graph_mode:
            timer.prepare(steps, max(warmup - 1, 1))
        model.freeze_host_heap()
        if warmup > 0:
            model.train_step(batches[(warmup - 1) % nb])
        sync()
        timer.enabled = not graph_mode
        if phases is not None:
            phases.reset()
            phases.enabled = not graph_mode
        t0 = time.perf_counter()
        stamps = []
        for i in range(steps):
            if args.step_times:
                stamps.append(time.perf_counter())
            model.train_step(batches[i % nb])
            if i == 0:
                fault("step")
                if label == "strong" and args.fault == "%d:strong" % rank:
                    raise RuntimeError("--fault %s: injected into the strong-scaling region" % args.fault)
                if label == "strong" and args.fault == "%d:strongexit" % rank:        # (a rank that DIES there: segfault, OOM kill)
                    os._exit(17)
        sync()
        elapsed = time.perf_counter() - t0
        timer.enabled = False
        if phases is not None:
            phases.enabled = False
        if args.step_times:                # diagnostic: host-side issue time of every step (stderr), e.g. to spot interpreter stalls
            stamps.append(time.perf_counter())
            print("%s step issue times (ms): %s" % (label, " ".join("%.1f" % ((b - a) * 1e3) for a, b in zip(stamps, stamps[1:]))),
                  file=sys.stderr)
        if dp:
            t = torch.tensor([elapsed], dtype=torch.float64, device=model.device if not dry else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t[0])
        info = dict(graph=graph_mode, graph_prepare_steps=prep)
        if graph_mode and kernel_pass:
            model.use_graph = False
            timer.calls_seen = 0
            model.train_step(batches[0])
            sync()
            timer.prepare(steps, 1)
            model.train_step(batches[1 % nb])
            sync()
            timer.enabled = True
            if phases is not None:
                phases.enabled = True
            t1 = time.perf_counter()
            for i in range(steps):
                model.train_step(batches[i % nb])
            sync()
            info["eager_instrumented_ms_per_step"] = round((time.perf_counter() - t1) / steps * 1e3, 3)
            timer.enabled = False
            if phases is not None:
                phases.enabled = False
            model.use_graph = True
        model.check_id_errors()
        if phases is not None and (kernel_pass or not graph_mode):
            # HIP-event times per step of the communication phases and the optimizer (eager instrumented pass of the same steps),
            # the exchange form the model chose for this region's batch shape and what it put on the links
            ex = dict(model.__dict__.get("_exchange_info") or {})
            own = model.__dict__.get("_owner_stats") if str(ex.get("form", "")).startswith("owner") else None
            info["communication"] = dict(world=dist.get_world_size(), backend=dist.get_backend(), phases_ms_per_step=phases.summary(steps),
                                         exchange_form=ex.get("form"), table_bytes_sent_per_rank=ex.get("table_bytes"),
                                         dense_net_bytes_sent_per_rank=ex.get("dense_net_bytes"),
                                         owner_lists=dict(pairs_sent=own["sent"], pairs_received=own["received"], capacity=list(own["capacity"]),
                                                          collectives_per_step=own["collectives"]) if own else None,
                                         note="bytes: all-reduce priced as 2 (N-1)/N of its size; owner form: chunks for the other owners + the "
                                              "reduced list to every peer; phases: HIP events on the launch stream, `exchange` includes the wait "
                                              "for the dense-net all-reduce started inside backward")
        region_info[label] = info
        return elapsed, (timer.summary(steps) if (kernel_pass or not graph_mode) else {})

    def assemble(weak, strong, region_info, graph_mode):
        primary, per_rank_batch, scaling = (weak, B, "weak") if weak is not None else (strong, B // world, "strong")
        elapsed, ksum = primary
        work = algorithmic_work(spec, per_rank_batch, args.model)
        arith = getattr(model, "arith", "f32")
        kernels = []
        for key, st in sorted(ksum.items(), key=lambda kv: -kv[1]["ms_per_step"]):
            row = dict(kernel=key[0] + (":" + key[1] if key[1] else ""), launches_per_step=round(st["launches_per_step"], 2),
                       avg_ms=round(st["avg_ms"], 4), ms_per_step=round(st["ms_per_step"], 4))
            wk = work(*key)
            if wk:
                bound, amount = wk
                if bound == "mfma":
                    tf = amount / (st["avg_ms"] * 1e-3) / 1e12
                    row.update(bound="mfma", achieved=round(tf, 3), unit="TFLOP/s", frac=round(tf / PEAK_F32_MFMA_TFLOPS, 4))
                else:
                    gbs = amount / (st["avg_ms"] * 1e-3) / 1e9
                    row.update(bound="hbm", achieved=round(gbs, 1), unit="GB/s", frac=round(gbs / PEAK_HBM_GBS, 4))
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
        traffic, traffic_src = pmc_traffic(dom_name, args.workload) if (args.model == "RAT_m2" and per_rank_batch == B) else (None, None)
        roofline = dict(kernel=dom_name, bound=dom["bound"], achieved=round(achieved, 3), peak=peak, unit=unit,
                        frac=round(achieved / peak, 4), traffic=traffic,
                        traffic_unit="bytes/launch (rocprofv3 PMC: 2 x FETCH_SIZE + WRITE_SIZE, %s)" % traffic_src,
                        algorithmic=round(per_launch, 1), algorithmic_unit="FLOP/launch" if dom["bound"] == "mfma" else "bytes/launch",
                        avg_launch_ms=round(avg_s * 1e3, 4), launches_per_step=round(dom["n"], 2))
        if dom["bound"] == "mfma" and arith == "bf16x3":
            # 3-way bf16 split, 6 of the 9 cross products, fp32 accumulate: `peak` stays the exact-fp32 MFMA peak (continuity
            # with round 1); `peak_effective` = dense bf16 MFMA peak / 6 products per fp32-equivalent FLOP
            eff = PEAK_BF16_MFMA_TFLOPS / 6.0
            roofline.update(peak_effective=round(eff, 1), frac_effective=round(achieved / eff, 4),
                            peak_effective_note="bf16 MFMA dense peak 2500 TFLOP/s / 6 bf16 products per fp32 product")
        # north_star's two named targets, from the same HIP-event records
        targets = {}
        for nm in ("rat_gather_fwd", "rat_gather_bwd", "rat_gather_bwd_sorted"):
            p = pooled.get(nm)
            if p and p["n"] > 0 and p["bound"] == "hbm":
                ms = p["ms"] / p["n"]                         # the HIP-event pair as it stands (no overhead subtraction)
                gbs = p["amount"] / p["n"] / (ms * 1e-3) / 1e9
                targets[nm] = dict(bound="hbm", avg_launch_ms=round(ms, 4),
                                   timing="HIP-event pair around the launch inside the training step, nothing subtracted; rocprofv3 "
                                          "durations of the same kernel: " + ROCPROF_NOTE,
                                   algorithmic_bytes=round(p["amount"] / p["n"]), achieved_GBps=round(gbs, 1),
                                   frac_of_8TBps=round(gbs / PEAK_HBM_GBS, 4))
        if gather_big is not None:
            targets["rat_gather_fwd_V100M"] = gather_big
        T = spec["K"] + 1
        cross_ms = cross_fl = 0.0
        for key, st in ksum.items():
            if key[0] in KernelTimer.ATTN_ARGS and key[1].startswith("L%dh" % T) and T != spec["F"] + 1:
                cross_ms += st["ms_per_step"]
                cross_fl += work(*key)[1] * st["launches_per_step"]
        if cross_ms > 0:
            tf = cross_fl / (cross_ms * 1e-3) / 1e12
            targets["cross_attention"] = dict(bound="mfma", what="all L=T=%d fused-attention launches of a step, fwd + bwd" % T,
                                              ms_per_step=round(cross_ms, 4), algorithmic_flop_per_step=round(cross_fl),
                                              achieved_TFLOPs=round(tf, 2), frac_of_f32_mfma_peak=round(tf / PEAK_F32_MFMA_TFLOPS, 4))
        gbatch = per_rank_batch * world
        result = dict(metric="training samples/sec at B=4096, K=10 retrieved, d=64; 1/2/4/8 MI355X",
                      value=round(gbatch * args.steps / elapsed, 1),
                      unit="samples/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                      ms_per_step=round(elapsed / args.steps * 1e3, 3), higher_is_better=True, scaling=scaling, vs_baseline=None,
                      dtype="f32", arith={"f32": "exact fp32 MFMA (v_mfma_f32_16x16x4_f32), fp32 accumulate",
                                          "bf16x3": "bf16x3-split MFMA, fp32 accumulate"}.get(arith, arith),
                      data="synthetic (%d distinct batches rotated)" % NBATCH,
                      config=dict(workload=args.workload, fields=spec["F"], vocab_rows=spec["total_vocab"], retrieved=spec["K"],
                                  embedding_dim=spec["d"], batch_per_gpu=per_rank_batch, global_batch=gbatch, heads=spec["num_heads"],
                                  dim_head=spec["dim_head"], depth=spec["depth"], scale_dim=spec["scale_dim"],
                                  dnn=spec["dnn_hidden_units"], step="fwd+bwd+clip+adam", embedding_grad=model._grad_mode,
                                  embedding_regularizer=model._cfg["lam_emb"], sync_batch_norm=bool(dp and spec["batch_norm"]),
                                  parallelism="dp%d" % world),
                      roofline=roofline, targets=targets, kernels=kernels)
        if dp:
            result["world"] = dist.get_world_size()
            result["first_barrier_s"] = round(first_barrier_s, 3) if first_barrier_s is not None else None
            result["communication"] = {k: v.get("communication") for k, v in region_info.items() if v.get("communication")}
        if strong is not None and weak is not None:
            el, _ = strong
            result["strong_scaling"] = dict(value=round(B * args.steps / el, 1), unit="samples/s", ms_per_step=round(el / args.steps * 1e3, 3),
                                            global_batch=B, batch_per_gpu=B // world,
                                            partitioning="one global batch of %d samples, rank r trains on rows [r*%d, (r+1)*%d)" % (B, B // world, B // world))
        result["step_mode"] = dict(region_info.get(scaling, {}),
                                   what="train_step(): fused iteration (forward, BCE, backward, [exchange], two-sweep clip+Adam with the "
                                        "regulariser folded in)" + (", replayed as a captured hipGraph; `kernels` / `roofline` / `targets` "
                                        "are HIP-event timings of the same K steps run eagerly right after the timed region" if graph_mode
                                        else ", eager launches"))
        result["config"]["dead_token_pruning"] = bool(can_prune and model.prune_dead_tokens)
        if spec["num_heads"] > 8 and hasattr(model, "_groups_supported") and model._groups_supported():
            both = bool(model._groups_supported() & 2)
            result["config"]["wide_heads"] = ("one %s launch per layer, head groups looped inside" % ("forward and one backward" if both else "forward")
                                              if getattr(model, "group_loop", False) else "one launch per head group and direction")
        if args.dp_rehearsal:
            result["dp_rehearsal"] = "one rank through the N > 1 code path (RCCL group of one rank, every collective of the step issued)"
        if pruned is not None:
            pruned["speedup"] = round(pruned["value"] / result["value"], 4)
            result["dead_token_pruning"] = pruned
        if per_rank is not None:
            per_rank["ratio_to_headline"] = round(per_rank["value"] / result["value"], 4)
            result["per_rank_shape"] = per_rank
        if inference is not None:
            result["inference"] = inference
        if alt is not None:
            result["exact_f32" if alt["arith"] == "f32" else "alt_arith"] = alt
        if args.model != "RAT_m2":
            result["config"]["variant"] = args.model
        if dry:
            result["dry_run_cpu"] = True
        return result

    want_weak = not dp or args.scaling in ("both", "weak")
    want_strong = dp and args.scaling in ("both", "strong")

    def run_regions(mode):
        """weak region, then strong region.  The strong region is the one that runs the row-list (owner) exchange — uneven all-to-alls
        whose sizes come from a host read-back — and the weak region's number is the job's `value`: once the weak region is through,
        its line is kept ready and the strong region runs under a watchdog and a try / except.  If it stalls (a rank died, a collective
        hangs) or raises, every rank leaves with exit code 0 and rank 0 prints the line WITHOUT `strong_scaling` but with
        `strong_scaling_error` — the headline of an 8-GPU run is not lost to its secondary measurement."""
        import threading
        weak = strong = None
        if want_weak:                                  # every rank its own batches of B samples
            batches = [make(1000 + 16 * rank + i) for i in range(NBATCH)]
            weak = timed_region(batches, args.steps, args.warmup, "weak")
            del batches
        if want_strong:                                # one global batch of B samples, rank r takes rows [r B/N, (r+1) B/N)
            per = B // world
            batches = [make(2000 + i, rank * per, (rank + 1) * per) for i in range(NBATCH)]
            guard = None
            if weak is not None and (not dry or ":strong" in (args.fault or "")):      # (dry runs: only for the injected faults' tests)
                keep = assemble(weak, None, dict(region_info), mode) if rank == 0 else None
                if rank == 0 and last_complete["line"] is None:
                    last_complete.update(line=dict(keep, strong_scaling_error="the job ended inside the strong-scaling region"),
                                         why="inside the strong-scaling region")

                def give_up(why):
                    if rank == 0:
                        keep["strong_scaling_error"] = why
                        emit(json.dumps(keep))
                    os._exit(0)
                guard = threading.Timer(args.region_timeout, give_up, args=("the strong-scaling region did not finish within %.0f s" % args.region_timeout,))
                guard.daemon = True
                guard.start()
            try:
                strong = timed_region(batches, args.steps, max(args.warmup, 2) if weak is None else 2, "strong")
            except Exception as exc:                   # (this rank's error; the other ranks' watchdogs end them)
                if guard is None:
                    raise
                guard.cancel()
                give_up("%s: %s" % (type(exc).__name__, exc))
            finally:
                if guard is not None:
                    guard.cancel()
            del batches
        return weak, strong

    alt = per_rank = pruned = inference = gather_big = None
    attempts = None
    if dp and graph_mode and not args.no_eager_first:
        # First contact with N > 1 RCCL ranks must be SURVIVABLE: the eager step (the form that has met RCCL) is measured first and
        # its result line is ready before the segmented-graph form is tried; a watchdog ends a graph attempt that stalls — every rank
        # exits 0 and rank 0 prints the eager line (with a note) instead of leaving the job in a collective until some outer timeout.
        import threading
        graph_mode = False
        model.use_graph = False
        weak_e, strong_e = run_regions(False)
        info_e = dict(region_info)
        fallback = assemble(weak_e, strong_e, info_e, False) if rank == 0 else None
        if rank == 0:
            last_complete.update(line=fallback, why="during the segmented-graph attempt (these are the eager numbers)")

        def bail():
            if rank == 0:
                fallback["step_mode"]["graph_attempt"] = "the segmented-graph form did not finish within %.0f s and was abandoned; these are the eager numbers" % args.graph_attempt_timeout
                emit(json.dumps(fallback))
            os._exit(0)
        dog = threading.Timer(args.graph_attempt_timeout, bail)
        dog.daemon = True
        sync()
        dog.start()
        graph_mode = True
        model.use_graph = True
        region_info.clear()
        weak, strong = run_regions(True)
        dog.cancel()
        captured = any(e[1] for e in model.__dict__.get("_step_graphs", {}).values())
        first_e, first_g = (weak_e or strong_e)[0], (weak or strong)[0]
        attempts = dict(eager_ms_per_step=round(first_e / args.steps * 1e3, 3), graph_ms_per_step=round(first_g / args.steps * 1e3, 3),
                        graph_captured=bool(captured), reported="graph" if first_g <= first_e else "eager")
        if first_g > first_e:                          # the eager form was the faster one on this machine: report it
            weak, strong, graph_mode = weak_e, strong_e, False
            region_info.clear()
            region_info.update(info_e)
    else:
        weak, strong = run_regions(graph_mode)

    # exact-fp32 arithmetic timed beside the default one in the SAME invocation; N = 1 only
    alt = None
    if not dp and args.arith is None and not args.no_extras and hasattr(model, "arith_modes") and len(model.arith_modes()) > 1:
        default_arith = model.arith
        other = [m for m in model.arith_modes() if m != default_arith][0]
        model.set_arith(other)
        batches = [make(1000 + i) for i in range(NBATCH)]
        el, ks = timed_region(batches, args.steps, 2, other)
        alt = dict(arith=other, value=round(B * args.steps / el, 1), ms_per_step=round(el / args.steps * 1e3, 3),
                   kernels={k[0] + (":" + k[1] if k[1] else ""): round(v["avg_ms"], 4) for k, v in ks.items()})
        model.set_arith(default_arith)
        del batches

    extras = not dp and not args.no_extras and not dry
    if alt is not None and args.no_extras:
        alt = None
    # §8e's strong-scaling partitioning on ONE GPU: the per-rank shape of an 8-GPU run of the global batch (B/8 samples per step)
    per_rank = None
    if extras and B % 8 == 0 and B // 8 >= 8:
        pb = B // 8
        batches = [make(3000 + i, 0, pb) for i in range(NBATCH)]
        el, _ = timed_region(batches, args.steps * 4, 3, "per_rank_shape", kernel_pass=False)
        per_rank = dict(batch=pb, value=round(pb * args.steps * 4 / el, 1), unit="samples/s", ms_per_step=round(el / (args.steps * 4) * 1e3, 3),
                        steps=args.steps * 4, what="the same training step at the batch ONE rank gets when 8 GPUs split the global batch of "
                        "%d (SURVEY 8e partitioning), measured on this single GPU: value / the headline value = what the fixed per-step "
                        "costs leave of linear strong scaling before any communication" % B)
        del batches
    pruned = None
    if extras and can_prune and not args.prune:
        model.prune_dead_tokens = True
        batches = [make(1000 + i) for i in range(NBATCH)]
        el, _ = timed_region(batches, args.steps, 3, "dead_token_pruning", kernel_pass=False)
        pruned = dict(value=round(B * args.steps / el, 1), unit="samples/s", ms_per_step=round(el / args.steps * 1e3, 3),
                      what="the product's default step: in the LAST encoder block only the class token's dependencies are computed "
                           "(cross-sample attention over B instead of B*S sequences, the block MLP on one token per sample); predictions "
                           "and gradients are those of the full computation (tests: check_pruning_equivalence, the golden cases, the "
                           "full-size oracle comparisons all run with it on)")
        model.prune_dead_tokens = False
        del batches
    # the embedding gather on a table that cannot sit in any cache: BASELINE.json configs[3]'s 100 M rows x 64 floats = 25.6 GB
    gather_big = None
    if extras and args.workload == "synthetic_F20_V1M_K10_d64_B4096" and args.model == "RAT_m2":
        gather_big = big_table_gather(model._lib, model.device)

    # Inference (the reference's logs quote inference throughput: BASELINE.md §1, base_model.py:232-247): the eval
    # forward of evaluate_generator / predict_generator — model.eval(), no saved activations — on resident batches, eagerly and as a
    # hipGraph replay (graph.EvalGraph), at the workload's batch and at 256 samples (where the ~25 launches, not the GPU, set the pace)
    inference = None
    if not dp and not dry and (extras or args.inference):
        def forward_flops_per_sample():
            F_, K_, d_ = spec["F"], spec["K"], spec["d"]
            T_, S_ = K_ + 1, F_ + 1
            I_, H_ = spec["num_heads"] * spec["dim_head"], d_ * spec["scale_dim"]
            if args.model != "RAT_m2":
                return None
            enc = spec["depth"] * T_ * S_ * (16 * d_ * I_ + 4 * d_ * H_ + 4 * I_ * (S_ + T_))
            widths = [F_ * d_] + list(spec["dnn_hidden_units"])
            head = 2 * (sum(a * b for a, b in zip(widths, widths[1:])) + widths[-1]) + 2 * d_
            return enc + head

        def eval_region(batches, steps, graph):
            model.eval()
            model.eval_graph, model.eval_graph_max_batch = graph, 1 << 30
            nb = len(batches)
            with torch.no_grad():
                for i in range(model.graph_warmup + 3):
                    model.forward(batches[i % nb])
                sync()
                t0 = time.perf_counter()
                for i in range(steps):
                    model.forward(batches[i % nb])
                sync()
            return (time.perf_counter() - t0) / steps

        keep = (model.eval_graph, model.eval_graph_max_batch, model.training)
        model.__dict__.pop("_eval_graphs", None)
        fl = forward_flops_per_sample()
        inference = dict(what="eval forward (model.eval(), torch.no_grad(): evaluate_generator / predict_generator's per-batch call, "
                              "base_model.py:232-273) on resident batches, %d distinct batches rotated; every token computed unless "
                              "dead_token_pruning says otherwise" % NBATCH, dead_token_pruning=bool(can_prune and model.prune_dead_tokens),
                         algorithmic_flop_per_sample=fl, shapes=[])
        for nb_, st_ in ((B, args.steps * 2), (256, args.steps * 8)):
            if nb_ > B or (nb_ == 256 and B == 256 and inference["shapes"]):
                continue
            batches = [make(5000 + i, 0, nb_) for i in range(NBATCH)]
            row = dict(batch=nb_, steps=st_)
            for graph in (False, True):
                sec = eval_region(batches, st_, graph)
                tag = "graph" if graph else "eager"
                row[tag + "_ms_per_batch"] = round(sec * 1e3, 4)
                row[tag + "_samples_per_s"] = round(nb_ / sec, 1)
            best = min(row["eager_ms_per_batch"], row["graph_ms_per_batch"])
            row["value"] = round(nb_ / (best * 1e-3), 1)
            if fl:
                tf = fl * nb_ / (best * 1e-3) / 1e12
                row["roofline"] = dict(bound="mfma", achieved=round(tf, 2), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s", frac=round(tf / PEAK_F32_MFMA_TFLOPS, 4))
            # per-kernel HIP events of the same forwards, eagerly
            model.eval_graph = False
            timer.reset()
            timer.calls_seen = 0
            with torch.no_grad():
                model.forward(batches[0])
                sync()
                timer.prepare(st_, 1)
                timer.enabled = True
                for i in range(st_):
                    model.forward(batches[i % NBATCH])
                sync()
                timer.enabled = False
            ks = timer.summary(st_)
            row["kernels"] = {k[0] + (":" + k[1] if k[1] else ""): round(v["ms_per_step"], 4) for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["ms_per_step"])}
            row["kernels_ms_per_batch"] = round(sum(v["ms_per_step"] for v in ks.values()), 4)
            inference["shapes"].append(row)
            del batches
        inference["value"], inference["unit"] = inference["shapes"][0]["value"], "samples/s"
        if can_prune and not model.prune_dead_tokens:          # the product's default forward (identical predictions): beside, not instead
            model.prune_dead_tokens = True
            model.__dict__.pop("_eval_graphs", None)
            batches = [make(5000 + i) for i in range(NBATCH)]
            sec = min(eval_region(batches, args.steps * 2, g_) for g_ in (False, True))
            inference["with_dead_token_pruning"] = dict(batch=B, ms_per_batch=round(sec * 1e3, 4), value=round(B / sec, 1), unit="samples/s")
            model.prune_dead_tokens = False
            model.__dict__.pop("_eval_graphs", None)
            del batches
        inference["reference_logs"] = "BASELINE.md §1: the reference's published logs give 107k / 37.6k / 22.9k inference samples/s on its own hardware and geometries (MovieLens / KKBox / Tmall), not on this workload"
        model.eval_graph, model.eval_graph_max_batch = keep[0], keep[1]
        model.train(keep[2])
        timer.reset()

    if rank == 0:
        result = assemble(weak, strong, region_info, graph_mode)
        if attempts is not None:
            result["step_mode"]["attempts"] = attempts
        if not dp and not args.no_cpu_baseline and not dry:
            result["cpu_baseline"] = cpu_baseline(args.workload, spec, args.cpu_batch or B, seed=1000, model=args.model,
                                                  timed_steps=args.cpu_steps)
        emit(json.dumps(result))
    if dp:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-oracle-child":          # cpu_baseline's all-cores attempt (CPU only, no GPU touched)
        wl, model, batch, seed, threads, steps = sys.argv[2].split(",")
        v, sps = _cpu_oracle_run(wl, model, int(batch), int(seed), int(threads), int(steps))
        print(json.dumps(dict(value=v, s_per_step=sps)))
        return 0
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args)
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE=%s" % (args.gpus, os.environ.get("WORLD_SIZE")))
    return worker(args)


if __name__ == "__main__":
    sys.exit(main())
