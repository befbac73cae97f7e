"""StepGraph — one fused training iteration (RAT_m2._fused_iteration: ~100 kernel launches through the C ABI) captured into
hipGraphs and replayed, so that a step costs the host a handful of calls instead of one ctypes launch per kernel.  This is the
MI355X-native answer to the per-op Python dispatch of the reference's loop (base_model.py:213-230): at the strong-scaling rank
shape (B = 512) and at BASELINE.json configs[0] (B = 256) the eager step is bound by launch issue, not by the GPU.

How it works
  * every launch goes through `ops._stream()` = torch's CURRENT stream, so inside `torch.cuda.graph(...)` the library's kernels
    are recorded into the capturing stream like torch's own; the library never synchronises and allocates nothing.
  * all tensors a step touches are static: parameters / moments / gradient bucket / workspaces exist before capture (the model runs
    `graph_warmup` eager steps of the same batch shape first), activations are allocated during capture from the graph's private
    pool, the batch is copied into three static input tensors before each replay, and the optimizer's clock (step count, lr, bias
    corrections) lives on the device (rat_adam_tick).
  * communication is NOT captured: `model._collective(fn)` closes the current capture segment, keeps the closure and opens the
    next segment; a replay alternates graph launches and those closures (RCCL calls on the same static buffers).  With one GPU
    there is exactly one segment.
Capturing does not execute anything, so after recording the first real execution is the first replay; the gradient bucket is
zeroed once by hand (every later step leaves it zero, rat_clip_adam_fused).
"""
import os

import torch


class StepGraph:
    def __init__(self, model, batch):
        self.model = model
        self.static = tuple(torch.empty_like(t) for t in batch)
        self.items = []                 # torch.cuda.CUDAGraph | callable, in execution order
        self._pool = torch.cuda.graph_pool_handle()
        self._stream = torch.cuda.Stream(device=batch[0].device)
        self._ctx = None
        for s, t in zip(self.static, batch):
            s.copy_(t)
        self._record()

    # ---- recording ------------------------------------------------------------------------------------------------------
    def _begin(self):
        g = torch.cuda.CUDAGraph()
        # thread_local: RCCL's watchdog / proxy threads keep making HIP calls of their own while this thread records
        self._ctx = torch.cuda.graph(g, pool=self._pool, stream=self._stream, capture_error_mode="thread_local")
        self._ctx.__enter__()
        self._open = g

    def _end(self):
        self._ctx.__exit__(None, None, None)
        self.items.append(self._open)
        self._ctx = self._open = None

    def between_segments(self, fn):
        """called by model._collective while recording: run `fn` eagerly now (keeps the ranks in step; its data are whatever the
        buffers hold — nothing has executed yet) and again at this position of every replay"""
        self._end()
        fn()
        self.items.append(fn)
        self._begin()

    def _record(self):
        model = self.model
        model.optimizer.prepare_step()
        torch.cuda.synchronize()
        model._tape = self
        try:
            with torch.no_grad():
                self._begin()
                try:
                    loss = model._fused_iteration(self.static, count=False)
                    self.loss = loss.reshape(1)
                finally:
                    self._end()
        finally:
            model._tape = None
        if os.environ.get("RAT_GRAPH_DEBUG"):
            print("StepGraph: %d graph segment(s), %d eager closure(s) for batch %s" %
                  (sum(isinstance(i, torch.cuda.CUDAGraph) for i in self.items),
                   sum(not isinstance(i, torch.cuda.CUDAGraph) for i in self.items), tuple(self.static[0].shape)))
        # nothing ran: the bucket may hold what the recorded collectives reduced into it; every replay expects (and leaves) zeros
        if model._gbuf is not None:
            model._gbuf[0].zero_()
            model._gbuf_clean = True
        model._sparse = None
        model._table_lists = None
        model._pending_reduce = None

    # ---- replay ---------------------------------------------------------------------------------------------------------
    def run(self, batch):
        model = self.model
        opt = model.optimizer
        opt.prepare_step()
        if not model._gbuf_clean:       # something else used the bucket since the last fused step: the graph expects zeros
            model._gbuf[0].zero_()
        for s, t in zip(self.static, batch):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t, non_blocking=True)
        for item in self.items:
            if isinstance(item, torch.cuda.CUDAGraph):
                item.replay()
            else:
                item()
        opt._step += 1                  # host mirror of the device clock the graph just advanced
        opt._clock_step = opt._step
        model._gbuf_clean = True
        return self.loss.clone()[0]


class EvalGraph:
    """The inference forward (BaseModel.evaluate_generator / predict_generator's `self.forward(batch)`, base_model.py:232-273) of one
    batch shape as ONE hipGraph: ~25 launches replayed with a single call.  It pays where the forward is shorter than the host needs to
    issue it (BASELINE.json configs[0]: B = 256, d = 16) and changes nothing where the GPU is the limit (bench.py's `inference` object
    reports both).  Single device only; the weights are read at replay time (the bf16x3 planes are re-split inside the graph), so a
    graph stays valid across optimizer steps; `load_state_dict` copies in place and keeps it valid too."""

    def __init__(self, model, batch):
        self.model = model
        self.static = tuple(torch.empty_like(t) for t in batch)
        for s, t in zip(self.static, batch):
            s.copy_(t)
        self._stream = torch.cuda.Stream(device=batch[0].device)
        self.graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        with torch.no_grad(), torch.cuda.graph(self.graph, stream=self._stream, capture_error_mode="thread_local"):
            y_pred, _loss, _reg, _saved = model._run_forward(self.static, save=False, with_reg=False)
        self.y_pred = y_pred

    def run(self, batch):
        for s, t in zip(self.static, batch):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t, non_blocking=True)
        self.graph.replay()
        return self.y_pred.clone()
