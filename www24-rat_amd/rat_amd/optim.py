"""FusedClipAdam — the optimizer half of BaseModel.train_one_epoch (base_model.py:224-225) on the HIP path:
``clip_grad_norm_(params, max_norm)`` + ``torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8)`` (torch_utils.py:41-49)
as three kernel launches over the model's FLAT parameter / gradient buffers (rat_sumsq, rat_clip_adam).

It is a ``torch.optim.Optimizer`` so the reference's ``lr_decay`` (param_groups[...]["lr"]) and ``zero_grad`` keep
working.  Parameters without a gradient (the dead ``query_proj``) live outside the flat buffer and are never touched.
"""
import torch

from . import ops


class FusedClipAdam(torch.optim.Optimizer):
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        params = [p for p in model.parameters() if p.requires_grad]
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))
        self._model = model
        self._step = 0
        self._m = None
        self._v = None
        self._norm_sq = None

    def _buffers(self):
        flat = self._model._flat
        if self._m is None or self._m.numel() != flat.numel() or self._m.device != flat.device:
            self._m = torch.zeros_like(flat)
            self._v = torch.zeros_like(flat)
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
        lr, (b1, b2), eps = group["lr"], group["betas"], group["eps"]
        self._step += 1
        ns = getattr(model, "_n_sparse", 0)
        norm_sq = None
        if max_norm is not None:                         # ONE global norm over dense gradients and sparse row lists alike
            self._norm_sq.zero_()
            if grad is not None:
                ops.sumsq(grad, self._norm_sq, lib=model._lib)
            for rows, g, count, width, _total, _base in sparse:
                ops.sumsq_rows(g, count, rows.numel(), width, self._norm_sq, lib=model._lib)
            norm_sq = self._norm_sq
        if grad is not None:
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

    def step(self, closure=None):
        loss = closure() if closure is not None else None
        self.clip_and_step(None)
        return loss

    def last_grad_norm(self):
        return float(torch.sqrt(self._norm_sq)[0]) if self._norm_sq is not None else None

    def state_dict(self):
        sd = super().state_dict()
        sd["rat_step"] = self._step
        sd["rat_m"] = self._m
        sd["rat_v"] = self._v
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        self._step = state_dict.pop("rat_step", 0)
        m, v = state_dict.pop("rat_m", None), state_dict.pop("rat_v", None)
        super().load_state_dict(state_dict)
        if m is not None:
            mm, vv = self._buffers()
            mm.copy_(m)
            vv.copy_(v)
