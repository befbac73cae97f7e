"""FusedClipAdam — the optimizer half of BaseModel.train_one_epoch (base_model.py:224-225) on the HIP path:
``clip_grad_norm_(params, max_norm)`` + ``torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8)`` (torch_utils.py:41-49)
as three kernel launches over the model's FLAT parameter / gradient buffers (rat_sumsq, rat_clip_adam).

It is a ``torch.optim.Optimizer`` so the reference's ``lr_decay`` (param_groups[...]["lr"]) and ``zero_grad`` keep
working.  Parameters without a gradient (the dead ``query_proj``) live outside the flat buffer and are never touched.
"""
import torch

from . import ops


KINDS = {"Adam": 0, "SGD": 1, "Adagrad": 2, "RMSprop": 3}          # RAT_OPT_* (include/rat_hip.h)


class FusedClipAdam(torch.optim.Optimizer):
    """kind: "Adam" (every shipped config) or one of the other optimizers torch_utils.get_optimizer can name — "SGD", "Adagrad",
    "RMSprop" — built like the reference builds them, `getattr(torch.optim, name)(params, lr=lr)`: torch's default hyper-parameters."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, kind="Adam"):
        params = [p for p in model.parameters() if p.requires_grad]
        if kind not in KINDS:
            raise NotImplementedError("optimizer=%r: the HIP path implements %s" % (kind, ", ".join(KINDS)))
        defaults = {"Adam": dict(lr=lr, betas=betas, eps=eps), "SGD": dict(lr=lr), "Adagrad": dict(lr=lr, eps=1e-10),
                    "RMSprop": dict(lr=lr, alpha=0.99, eps=1e-8)}[kind]
        super().__init__(params, defaults)
        self.kind, self._kind = kind, KINDS[kind]
        self._model = model
        self._step = 0
        self._m = None
        self._v = None
        self._norm_sq = None

    def _buffers(self):
        flat = self._model._flat
        if self._norm_sq is None or self._norm_sq.device != flat.device or (self._v is not None and self._v.numel() != flat.numel()):
            self._m = torch.zeros_like(flat) if self._kind == 0 else None          # first moment: Adam only
            self._v = torch.zeros_like(flat) if self._kind != 1 else None          # second moment / accumulator: all but SGD
            self._norm_sq = torch.zeros(1, dtype=torch.float32, device=flat.device)
        return self._m, self._v

    @torch.no_grad()
    def clip_and_step(self, max_norm=None):
        model = self._model
        grad = model._gather_flat_grad()                 # the dense part of the bucket (everything, unless the tables are sparse)
        sparse = getattr(model, "_sparse", None) or []
        if grad is None and not sparse:
            return None
        m, v = self._buffers()
        group = self.param_groups[0]
        lr, (b1, b2), eps = group["lr"], group.get("betas", (0.0, 0.0)), group.get("eps", 0.0)
        if self._kind != 0 and sparse:
            raise NotImplementedError("row-sparse table gradients are wired for Adam only")
        self._step += 1
        self._clock_step = None
        ns = getattr(model, "_n_sparse", 0)
        norm_sq = None
        if max_norm is not None:                         # ONE global norm over dense gradients and sparse row lists alike
            self._norm_sq.zero_()
            if grad is not None:
                ops.sumsq(grad, self._norm_sq, lib=model._lib)
            for rows, g, count, width, _total, _base in sparse:
                ops.sumsq_rows(g, count, rows.numel(), width, self._norm_sq, lib=model._lib)
            norm_sq = self._norm_sq
        self._last_norm_sq = norm_sq                     # (None: this step did not clip)
        self._last_norm_value = None
        if grad is not None and self._kind != 0:
            ops.clip_opt(model._flat[ns:], grad, v[ns:] if v is not None else None, norm_sq, max_norm or 0.0, lr, self._kind,
                         group.get("alpha", 0.0), eps, lib=model._lib)
        elif grad is not None:
            ops.clip_adam(model._flat[ns:], grad, m[ns:], v[ns:], norm_sq, max_norm or 0.0, lr, b1, b2, eps, self._step, lib=model._lib)
        for rows, g, count, width, _total, base in sparse:   # lazy Adam on the touched table rows (rat_adam_rows)
            ops.adam_rows(model._flat[base:], m[base:], v[base:], rows, g, count, rows.numel(), width, norm_sq, max_norm or 0.0,
                          lr, b1, b2, eps, self._step, lib=model._lib)
        model._sparse = None                             # consumed: a second step without a new backward must not re-apply them
        return norm_sq

    def zero_grad(self, set_to_none=True):
        super().zero_grad(set_to_none=set_to_none)
        if hasattr(self._model, "_sparse"):
            self._model._sparse = None

    # ---- the two-sweep form used by BaseModel.train_step (ABI v4) -------------------------------------------------------------
    def _clock(self):
        """device-side step counter, learning rate and the derived Adam scalars (rat_adam_tick): what lets a captured step be
        replayed without new kernel arguments.  Host mirrors: self._step, param_groups[0]["lr"]."""
        flat = self._model._flat
        if getattr(self, "_step_dev", None) is None or self._step_dev.device != flat.device:
            self._step_dev = torch.zeros(1, dtype=torch.int32, device=flat.device)
            self._lr_dev = torch.zeros(1, dtype=torch.float32, device=flat.device)
            self._hyper = torch.zeros(4, dtype=torch.float32, device=flat.device)
            self._scal = torch.zeros(4, dtype=torch.float32, device=flat.device)     # [BCE sum | clip norm^2 | regulariser value | -]
            self._reg_value = self._scal[2:3]
            self._clock_step, self._clock_lr = None, None
        return self._step_dev, self._lr_dev, self._hyper

    def step_scalars(self):
        """the 4-float tensor behind the fused step's accumulators; the caller zeroes it once per step and passes zeroed=True"""
        self._clock()
        self._buffers()
        return self._scal

    def begin_step(self, counters=None, count=True):
        """first launch of a fused iteration (rat_step_begin): clock tick, accumulator scalars cleared, `counters` (int64, the BatchNorm
        layers' num_batches_tracked; None = none) advanced — fused_step(..., zeroed=True, ticked=True) follows at the end of the step"""
        step_dev, lr_dev, hyper = self._clock()
        self._buffers()
        group = self.param_groups[0]
        b1, b2 = group.get("betas", (0.9, 0.999))
        ops.step_begin(step_dev, lr_dev, b1, b2, hyper, self._scal, counters, lib=self._model._lib)
        if count:
            self._step += 1
            self._clock_step = self._step
        return self._scal

    def prepare_step(self):
        """host -> device synchronisation of the clock; cheap, and a no-op unless the learning rate changed (lr_decay) or the step
        count was set from outside (load_state_dict).  Never called while a graph is being captured."""
        step_dev, lr_dev, _ = self._clock()
        self._buffers()
        lr = float(self.param_groups[0]["lr"])
        if self._clock_lr != lr:
            lr_dev.fill_(lr)
            self._clock_lr = lr
        if self._clock_step != self._step:
            step_dev.fill_(self._step)
            self._clock_step = self._step

    @torch.no_grad()
    def fused_step(self, grad, max_norm, count=True, zeroed=False, ticked=False):
        """clip_grad_norm_(max_norm) + Adam + zero_grad over the flat buffers with the regulariser folded in (rat_sumsq_reg,
        rat_clip_adam_fused): `grad` is the flat gradient WITHOUT the lambda*W terms; afterwards it holds zeros.  Returns the
        regulariser's value (device scalar, (lambda/2)||W||^2 over the tensors the reference regularises).  Call prepare_step()
        first.  count=False: the launches are being recorded, not executed (graph capture) — the host step mirror stays."""
        model = self._model
        c = model._cfg
        sparse = getattr(model, "_sparse", None) or []
        m, v = self._buffers()
        step_dev, lr_dev, hyper = self._clock()
        group = self.param_groups[0]
        (b1, b2), eps = group.get("betas", (0.9, 0.999)), group.get("eps", 0.0)
        ns = getattr(model, "_n_sparse", 0)
        n_split = model._n_emb - ns
        lib = model._lib
        if self._kind != 0 and sparse:
            raise NotImplementedError("row-sparse table gradients are wired for Adam only")
        if not ticked:                                               # ticked: begin_step() advanced the clock at the start of this iteration
            ops.adam_tick(step_dev, lr_dev, b1, b2, hyper, lib=lib)
            if count:
                self._step += 1
                self._clock_step = self._step
        acc = self._scal[1:2] if zeroed else self._norm_sq           # zeroed: step_scalars() was cleared at the start of this step
        if not zeroed:
            self._norm_sq.zero_()
            self._reg_value.zero_()
        ops.sumsq_reg(grad, model._flat[ns:], n_split, c["lam_emb"], c["lam_net"], acc, reg_out=self._reg_value, lib=lib)
        for rows, g, cnt, width, _total, _base in sparse:
            ops.sumsq_rows(g, cnt, rows.numel(), width, acc, lib=lib)
        self._last_norm_sq = acc                                     # (a view of the step's scalars: begin_step() of the NEXT step clears it)
        self._last_norm_value = None
        norm_sq = acc if max_norm is not None else None
        if self._kind != 0:
            ops.clip_opt_fused(model._flat[ns:], grad, v[ns:] if v is not None else None, n_split, c["lam_emb"], c["lam_net"], norm_sq,
                               max_norm or 0.0, hyper, self._kind, group.get("alpha", 0.0), eps, zero_g=True, lib=lib)
        else:
            ops.clip_adam_fused(model._flat[ns:], grad, m[ns:], v[ns:], n_split, c["lam_emb"], c["lam_net"], norm_sq, max_norm or 0.0,
                                hyper, b1, b2, eps, zero_g=True, lib=lib)
        for rows, g, cnt, width, _total, base in sparse:
            ops.adam_rows_dev(model._flat[base:], m[base:], v[base:], rows, g, cnt, rows.numel(), width, norm_sq, max_norm or 0.0,
                              hyper, b1, b2, eps, lib=lib)
        model._sparse = None
        return self._reg_value

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self.clip_and_step(None)
        return loss

    def last_grad_norm(self):
        """global gradient norm of the LAST optimizer step (either form), or None if that step did not compute one.  Read it between
        steps: the fused step keeps the value in the scalars the next begin_step() clears."""
        t = getattr(self, "_last_norm_sq", None)
        return float(torch.sqrt(t)[0]) if t is not None else None

    def untick(self, counters=None):
        """take back begin_step()'s tick after an iteration that failed before its update (RAT_m2._fused_iteration)"""
        self._step -= 1
        self._clock_step = None                          # the device clock is re-synchronised by the next prepare_step()
        if counters is not None:
            counters.sub_(1)

    def state_dict(self):
        sd = super().state_dict()
        sd["rat_step"] = self._step
        sd["rat_m"] = self._m
        sd["rat_v"] = self._v
        drop = getattr(self._model, "dropout_state", None)
        sd["rat_dropout"] = drop() if drop is not None else None          # (base seed, counter) of the device-side mask generator
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        self._step = state_dict.pop("rat_step", 0)
        self._clock_step = None                          # the device clock is re-synchronised by the next prepare_step()
        m, v = state_dict.pop("rat_m", None), state_dict.pop("rat_v", None)
        drop = state_dict.pop("rat_dropout", None)
        if drop and hasattr(self._model, "load_dropout_state"):
            self._model.load_dropout_state(drop)
        super().load_state_dict(state_dict)
        mm, vv = self._buffers()
        if m is not None and mm is not None:
            mm.copy_(m)
        if v is not None and vv is not None:
            vv.copy_(v)
