"""RAT_m2 — the reference's default model (fuxictr/pytorch/models/RAT_m2.py:28-152) as a FuxiCTR model plugin
whose forward/backward is the hand-written HIP path of librat_hip.so.

What is kept from the reference (so that it drops in):
  * the constructor signature and every kwarg the shipped configs pass (RAT_m2.py:29-56, base_model.py:32-46);
  * the module tree, hence the ``state_dict`` keys/shapes (checkpoints load both ways) and, because modules are
    created and re-initialised in the same order, the same initial weights under the same seed;
  * ``forward(inputs) -> {"y_true": [B,1], "y_pred": [B,1]}`` on the DataLoader's 4-tuple
    (fuxictr/pytorch/data_generator.py:66-78) and the BaseModel training surface.
What is different: the torch modules below only HOLD parameters — none of their ``forward`` methods is ever called.
All parameters with a gradient live in ONE flat fp32 buffer (tables first), the whole forward+backward is a single
autograd node, and its backward fills one flat gradient buffer that the fused clip+Adam consumes.
"""
import ctypes
from collections import OrderedDict

import torch
from torch import nn

from . import ops
from ._lib import get_lib
from .base_model import BaseModel, parse_regularizer
from .data import DeviceBatch
from .dp import DataParallelExchange
from .features import field_infos


# ------------------------------------------------------------------------------------ parameter containers
class _EmbeddingDict(nn.Module):
    """EmbeddingDictLayer's parameters (fuxictr/pytorch/layers/embedding.py:46-95): one nn.Embedding per field."""

    def __init__(self, feature_map, width):
        super().__init__()
        self.embedding_layer = nn.ModuleDict()
        for name, spec in feature_map.feature_specs.items():
            if spec.get("embedding_dim", width) != width and width != 1:
                raise NotImplementedError("per-field embedding_dim is outside the RAT_m2 hot path")
            if spec["type"] == "categorical":
                pad = spec.get("padding_idx", None)
            elif spec["type"] == "sequence":
                pad = spec["vocab_size"] - 1
            else:
                raise NotImplementedError("feature type %r is outside the RAT_m2 hot path" % spec["type"])
            self.embedding_layer[name] = nn.Embedding(spec["vocab_size"], width, padding_idx=pad)


class _EmbeddingLayer(nn.Module):
    def __init__(self, feature_map, width):
        super().__init__()
        self.embedding_layer = _EmbeddingDict(feature_map, width)


class _LRLayer(nn.Module):
    """LR_Layer with use_bias=False (shallow.py:23-34, RAT_m2.py:85-86)."""

    def __init__(self, feature_map):
        super().__init__()
        self.bias = None
        self.embedding_layer = _EmbeddingLayer(feature_map, 1)


class _Attention(nn.Module):
    def __init__(self, dim, heads, dim_head, dropout):
        super().__init__()
        inner = heads * dim_head
        self.heads, self.dim_head = heads, dim_head
        self.to_qkv = nn.Linear(dim, inner * 3, bias=False)
        project_out = not (heads == 1 and dim_head == dim)
        self.to_out = nn.Sequential(nn.Linear(inner, dim), nn.Dropout(dropout)) if project_out else nn.Identity()


class _PreNorm(nn.Module):
    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn


class _FeedForward(nn.Module):
    def __init__(self, dim, hidden):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden), nn.GELU(), nn.Dropout(0.0), nn.Linear(hidden, dim), nn.Dropout(0.0))


class _Block(nn.Module):
    def __init__(self, dim, heads, dim_head, dropout, hidden):
        super().__init__()
        self.cross_attention = _PreNorm(dim, _Attention(dim, heads, dim_head, dropout))
        self.intra_attention = _PreNorm(dim, _Attention(dim, heads, dim_head, dropout))
        self.mlp = _FeedForward(dim, hidden)


class _Encoder(nn.Module):
    def __init__(self, dim, heads, dim_head, dropout, depth, hidden):
        super().__init__()
        self.encoder = nn.ModuleList([_Block(dim, heads, dim_head, dropout, hidden) for _ in range(depth)])


def _activation_module(name):
    """torch_utils.get_activation (torch_utils.py:83-94): "relu" / "sigmoid" / "tanh" in any case, otherwise the torch.nn class of that
    name; a module (or None) passes through"""
    if isinstance(name, str):
        low = name.lower()
        if low == "relu":
            return nn.ReLU()
        if low == "sigmoid":
            return nn.Sigmoid()
        if low == "tanh":
            return nn.Tanh()
        return getattr(nn, name)()
    return name


def _activation_code(mod):
    """the RAT_ACT_* code (include/rat_hip.h) of a hidden-layer activation module, or NotImplementedError"""
    if isinstance(mod, nn.ReLU):
        return 0
    if isinstance(mod, nn.Identity):
        return 1
    if isinstance(mod, nn.Sigmoid):
        return 2
    if isinstance(mod, nn.Tanh):
        return 3
    if isinstance(mod, nn.LeakyReLU) and mod.negative_slope == 0.01:
        return 4
    if isinstance(mod, nn.ELU) and mod.alpha == 1.0:
        return 5
    raise NotImplementedError("dnn_activations: %r has no kernel on the HIP path (have ReLU, Sigmoid, Tanh, LeakyReLU(0.01), ELU(1.0), "
                              "Identity / None)" % (mod,))


class _MLP(nn.Module):
    """MLP_Layer's module list (deep.py:108-139): [Linear, (BatchNorm1d), (activation), (Dropout)]* + Linear(.,1) — the same modules
    in the same order, hence the same state_dict keys whatever the activations are."""

    def __init__(self, input_dim, hidden_units, activation, dropout, batch_norm):
        super().__init__()
        rates = dropout if isinstance(dropout, list) else [dropout] * len(hidden_units)
        acts = activation if isinstance(activation, list) else [activation] * len(hidden_units)
        acts = [_activation_module(a) for a in acts]
        for a in acts:
            if a:
                _activation_code(a)                          # refuse what the kernels do not have at construction time
        mods = []
        widths = [input_dim] + list(hidden_units)
        for j in range(len(widths) - 1):
            mods.append(nn.Linear(widths[j], widths[j + 1], bias=True))
            if batch_norm:
                mods.append(nn.BatchNorm1d(widths[j + 1]))
            if acts[j]:
                mods.append(acts[j])
            if rates[j] > 0:
                mods.append(nn.Dropout(p=rates[j]))
        mods.append(nn.Linear(widths[-1], 1, bias=True))
        self.dnn = nn.Sequential(*mods)


# ------------------------------------------------------------------------------------ the autograd node
class _RATFunction(torch.autograd.Function):
    """forward = K1 -> depth x (intra attn, cross attn, FFN) -> head -> sigmoid/BCE (+ L2 value);
    backward = the same chain reversed, writing every parameter gradient into one flat buffer."""

    @staticmethod
    def forward(ctx, model, batch, with_reg, *params):
        model._drop_private = True         # this forward's backward may run after ANOTHER forward: private copy of the seed words
        try:
            y_pred, loss, reg, saved = model._run_forward(batch, save=True, with_reg=with_reg)
        finally:
            model._drop_private = False
        ctx.model, ctx.saved_state, ctx.with_reg = model, saved, with_reg
        ctx.mark_non_differentiable(y_pred)
        return y_pred, loss, reg

    @staticmethod
    def backward(ctx, _g_pred, g_loss, g_reg):
        # the incoming gradients stay DEVICE scalars (the kernels multiply them in): no host read-back, so the host can enqueue
        # the whole backward while the forward is still running
        grads = ctx.model._run_backward(ctx.saved_state, g_loss, g_reg if ctx.with_reg else None)
        ctx.saved_state = None
        return (None, None, None) + tuple(grads)


class RAT_m2(DataParallelExchange, BaseModel):
    def __init__(self, feature_map, model_id="RAT_m2", gpu=-1, task="binary_classification", learning_rate=1e-3,
                 embedding_dim=10, dnn_hidden_units=[64, 64, 64], dnn_activations="ReLU", attention_layers=2,
                 num_heads=1, attention_dim=8, net_dropout=0, batch_norm=False, layer_norm=False, use_scale=False,
                 use_wide=False, use_residual=True, embedding_regularizer=None, net_regularizer=None, depth=4, heads=4,
                 pool="cls", dim_head=10, dropout=0., emb_dropout=0., scale_dim=4, **kwargs):
        super().__init__(feature_map, model_id=model_id, gpu=gpu, embedding_regularizer=embedding_regularizer,
                         net_regularizer=net_regularizer, **kwargs)
        d, nf = embedding_dim, feature_map.num_fields
        self._cfg = dict(d=d, heads=num_heads, dh=dim_head, depth=depth, hidden=d * scale_dim, nf=nf,
                         batch_norm=bool(batch_norm), use_wide=bool(use_wide), emb_dropout=float(emb_dropout),
                         attn_dropout=float(dropout or 0.0),
                         net_dropout=net_dropout, lam_emb=parse_regularizer(embedding_regularizer),
                         lam_net=parse_regularizer(net_regularizer))
        self._fields = field_infos(feature_map)
        # --- module tree in the reference's registration order (RAT_m2.py:63-98) -------------------------
        self.embedding_layer = _EmbeddingLayer(feature_map, d)
        self.label_embedding_layer = nn.Embedding(num_embeddings=3, embedding_dim=d)
        self.query_proj = nn.Linear(d * nf, d * nf)           # dead in the reference too (RAT_m2.py:66-67), kept for state_dict
        self.query_dropout = nn.Dropout(net_dropout) if (not isinstance(net_dropout, list) and net_dropout > 0) else None
        kwargs["retrieval_configs"]["topK"]                    # the reference requires the key (RAT_m2.py:73)
        self._make_encoder(d, num_heads, dim_head, dropout, depth, d * scale_dim)
        self.dropout = nn.Dropout(emb_dropout)
        self.lr_layer = _LRLayer(feature_map) if use_wide else None
        self.dnn = _MLP(d * nf, dnn_hidden_units, dnn_activations, net_dropout, batch_norm) if dnn_hidden_units else None
        self.fc = nn.Linear(d, 1)
        self.output_activation = self.get_output_activation(task)
        self._flat = None
        self._lib = None
        self._last_gflat = None
        self._gbuf = None              # persistent flat gradient buffer + the gradient field tables that point into it
        self._gbuf_clean = False       # True: the buffer is known to hold zeros (left so by the two-sweep optimizer)
        self._tape = None              # a StepGraph that is recording this model's training step (graph.py)
        self._sync_bn = bool(kwargs.get("sync_batch_norm", True))     # under data parallelism: BatchNorm over the GLOBAL batch
        if "graph_under_dp" in kwargs:
            self.graph_under_dp = bool(kwargs["graph_under_dp"])
        # how the embedding-table gradients are produced:
        #   "atomic" fp32 atomics into dense tables (fastest, not run-to-run reproducible); "sorted" the same dense tables from a
        #   stable sort + segmented reduction (bit-reproducible); "sparse" (unique rows, gradient rows) lists + lazy row Adam, no
        #   dense table gradient at all (BASELINE configs[3]); "auto" = sparse above 8 GB of tables when embedding_regularizer
        #   is 0, else atomic
        self._arith_request = str(kwargs.get("arith", "auto"))
        self._embedding_grad = str(kwargs.get("embedding_grad", "auto"))
        if self._embedding_grad not in ("auto", "atomic", "sorted", "sparse"):
            raise ValueError("embedding_grad=%r" % self._embedding_grad)
        self._sparse = None
        self._sparse_is_global = False   # the row lists in _sparse were already merged over the ranks
        self._table_lists = None         # dense modes under data parallelism: this backward's table gradients as row lists
        self._pending_reduce = None      # (work handle, gradient buffer) of the dense-net all-reduce started inside backward
        self._owner_state = None         # owner exchange: this step's plans + count matrix (_owner_prepare)
        self._validate_ids = bool(kwargs.get("validate_ids", True))
        self._id_errors = None
        self._ws = {}
        self._fused_cache = {}
        self.compile(kwargs["optimizer"], loss=kwargs["loss"], lr=learning_rate)
        self.reset_parameters()
        self.model_to_device()

    # ------------------------------------------------------------------------------ arithmetic of the encoder GEMMs
    # "f32": v_mfma_f32_16x16x4_f32, exact fp32 (bit-identical to a k-ordered fmaf chain).  "bf16x3": operands split exactly into
    # three bf16 chunks, six cross products on v_mfma_f32_16x16x32_bf16 with fp32 accumulation — fp32-class accuracy (every parity
    # gate of tests/ holds at unchanged tolerances) at ~2.4x the MFMA rate; compiled for embedding_dim 64 with 8 heads x 10, other
    # shapes run exact fp32 whatever is selected.  kwarg / attribute `arith`: "auto" (= bf16x3 where compiled), "f32", "bf16x3".
    arith = "f32"
    gemm_arith = "f32"

    def arith_modes(self):
        """arithmetic variants the loaded library offers for THIS model's encoder shapes ("f32" is always present)"""
        c = self._cfg
        modes = ["f32"]
        ok = lambda h: self._lib.size("rat_attn_fwd_workspace", c["d"], h, c["dh"]) > 0            # noqa: E731
        if self._lib is not None and (ok(c["heads"]) or (c["heads"] % 8 == 0 and ok(8))):       # (wide heads run in groups of 8)
            modes.append("bf16x3")
        return modes

    def set_arith(self, mode):
        # the plain GEMMs (DNN head, composed attention) have a bf16x3 kernel for every shape: they follow the REQUEST, the fused
        # encoder kernels what is compiled for their geometry
        self.gemm_arith = "f32" if mode == "f32" else "bf16x3"
        if mode == "auto":
            mode = self.arith_modes()[-1]
        if mode not in self.arith_modes():
            raise ValueError("arith=%r is not available for this shape (have %s)" % (mode, self.arith_modes()))
        self.arith = mode

    # ------------------------------------------------------------------------------ encoder (overridden by the variants)
    def _make_encoder(self, d, num_heads, dim_head, dropout, depth, hidden):
        torch.randn(1, 1, d)                                   # `space_token` (RAT_m2.py:74): unregistered, but it advances the RNG
        self.encoder = _Encoder(d, num_heads, dim_head, dropout, depth, hidden)

    def _attn_descriptor(self, p):
        """(names, RatAttnParams) of the PreNorm(Attention) whose parameters live under prefix p."""
        has_out = (p + "fn.to_out.0.weight") in self._params
        names = [p + "norm.weight", p + "norm.bias", p + "fn.to_qkv.weight",
                 p + "fn.to_out.0.weight" if has_out else None, p + "fn.to_out.0.bias" if has_out else None]
        return names, ops.attn_params(*[self._p(n) if n else None for n in names])

    # ---- one PreNorm(Attention)(x) + x layer: the fused kernel when it serves the dimensions, otherwise composed from K2c
    #      LayerNorm -> rat_sgemm (to_qkv) -> K2d attention core (seq-map addressing) -> rat_sgemm (to_out + bias + residual)
    FUSED_MAX_L = 64            # tests lower this to send short sequences through the composed path as well

    def _attn_mode(self, smap):
        """-> ("fused", heads) | ("grouped", heads per group) | ("composed", None).

        fused: the whole layer is one rat_attn_fwd / rat_attn_bwd launch.  grouped: heads*dim_head is too wide for the fused
        kernel's LDS tile (the shipped Tmall config: 32 heads x 10) but the sequences are short: the heads are independent
        given LayerNorm(x), so the layer runs as heads/g launches of the fused kernel on g heads each — every launch gets its
        rows of W_q / W_k / W_v and its columns of W_out, the first one carries the bias and the residual, the others
        accumulate onto its output (rat_attn_fwd_ex, res = y); backward chains the `add` term the same way.  composed: long
        sequences (RAT_m0) — LayerNorm, GEMMs and the K2d core as separate launches."""
        key = (int(smap.L), self.FUSED_MAX_L)
        hit = self._fused_cache.get(key)
        if hit is None:
            c, L = self._cfg, int(smap.L)
            ok = lambda h: L <= self.FUSED_MAX_L and ops.attn_fused_supported(c["d"], h, c["dh"], L, lib=self._lib)   # noqa: E731
            if ok(c["heads"]):
                hit = ("fused", c["heads"])
            else:
                per = next((h for h in (8, 4, 2, 1) if c["heads"] % h == 0 and h < c["heads"] and ok(h)), None)
                hit = ("grouped", per) if per else ("composed", None)
            self._fused_cache[key] = hit
        return hit

    def _groups_supported(self):
        """rat_attn_groups_supported for this model's head geometry (bit 0: one-launch forward, bit 1: one-launch backward)"""
        v = self.__dict__.get("_groups_sup")
        if v is None:
            c = self._cfg
            v = self._groups_sup = ops.attn_groups_supported(c["d"], c["heads"], c["dh"], lib=self._lib)
        return v

    def _attn_is_fused(self, smap):
        return self._attn_mode(smap)[0] == "fused"

    def _group_weights(self, names, per, gplanes=None):
        """contiguous per-group copies of the projection weights: rows of Q | K | V and columns of to_out.  gplanes: the layer's per-group
        bf16x3 plane sets (rat_attn_groups_split_jobs, refreshed once per step): slice g rides along as RatAttnParams.planes, so that the
        launch on group g does not split its weights again (three small launches per call)"""
        c = self._cfg
        d, dh, groups = c["d"], c["dh"], c["heads"] // per
        ig = per * dh
        ln_g, ln_b, w_qkv, w_out, b_out = [self._p(n) if n else None for n in names]
        # one permuting copy per tensor (not one per group): [groups][Q|K|V][ig][d] and [groups][d][ig], group g a contiguous slice
        wq = w_qkv.view(3, groups, ig, d).permute(1, 0, 2, 3).contiguous()
        wo = w_out.view(d, groups, ig).permute(1, 0, 2).contiguous()
        zero_bias = torch.zeros_like(b_out)
        out = []
        for g in range(groups):
            w_g, wo_g = wq[g].view(3 * ig, d), wo[g]
            pl = None
            if gplanes is not None and self.arith == "bf16x3":
                nb = gplanes.numel() // groups
                pl = gplanes[g * nb:(g + 1) * nb]
            out.append((w_g, wo_g, ops.attn_params(ln_g, ln_b, w_g, wo_g, b_out if g == 0 else zero_bias, planes=pl), zero_bias))
        return out

    def _attn_layer_forward(self, desc, x, smap, save, out=None):
        """desc = (names, RatAttnParams) from _attn_descriptor; returns (y, whatever the backward needs)."""
        c, lib = self._cfg, self._lib
        d, heads, dh = c["d"], c["heads"], c["dh"]
        mode, per = self._attn_mode(smap)
        # Dropout behind the output projection (RAT_m2.py:186-189), training only; the seed is one of this step's device words
        # (_dropout_begin) and is kept for the backward, which re-derives the mask
        drop = (c["attn_dropout"], self._dropout_word()) if (self.training and c["attn_dropout"] > 0) else (0.0, 0)
        if mode == "fused":
            y, o, l = ops.attn_fwd(x, desc[1], smap, d, heads, dh, save=save, out=out, arith=self.arith, dropout=drop, lib=lib)
            return y, (o, l, drop)
        if mode == "grouped":
            if desc[0][3] is None:
                raise NotImplementedError("grouped attention needs an output projection")
            gplanes = getattr(desc[1], "gplanes", None)
            sup = self._groups_supported()
            if per == 8 and self.group_loop and ((gplanes is not None and self.arith == "bf16x3") if d == 64 else bool(sup & 1)):
                # ONE launch: LayerNorm / x / residual once per chunk, the head groups looped over inside it (rat_attn_fwd_groups);
                # o, l are group-major, slice g is what the backward's launch on group g reads.  d = 64: bf16x3 group planes; small d
                # (the shipped Tmall geometry): exact fp32, weights in place, and the backward is one launch too
                y, o, l = ops.attn_fwd_groups(x, x, desc[1], gplanes if d == 64 else None, smap, d, heads, dh, save=save, out=out,
                                              dropout=drop, lib=lib)
                return y, (("loop", o, l, drop) if save else None)
            y, kept = None, []
            for g, (w_g, wo_g, params_g, zb) in enumerate(self._group_weights(desc[0], per)):
                # Dropout(sum of the groups' partial projections + bias) = sum of the equally masked partials: same seed in every launch
                y, o, l = ops.attn_fwd_ex(x, x if g == 0 else y, params_g, smap, d, per, dh, 0.0, 1.0, save=save, out=y,
                                          arith=self.arith, dropout=drop, lib=lib)
                kept.append((w_g, wo_g, params_g, zb, o, l, drop))
            return y, (kept if save else None)
        inner, ntok = heads * dh, x.numel() // d
        ln_g, ln_b, w_qkv, w_out, b_out = [self._p(n) if n else None for n in desc[0]]
        if w_out is None:
            raise NotImplementedError("attention without an output projection (heads == 1 and dim_head == embedding_dim) is only "
                                      "implemented in the fused kernel (sequences up to 64 tokens)")
        xn = ops.layernorm_fwd(x, d, ntok, ln_g, ln_b, d, lib=lib)
        qkv = torch.empty((ntok, 3 * inner), dtype=torch.float32, device=x.device)
        ops.sgemm(0, 1, ntok, 3 * inner, d, xn, d, w_qkv, d, qkv, 3 * inner, arith=self.gemm_arith, lib=lib)                     # to_qkv (no bias)
        o, lse = ops.attn_core_fwd_map(qkv, smap, heads, dh, save=True, lib=lib)
        if drop[0] > 0:                                              # y = x + Dropout(to_out(o)) (RAT_m0.py / RAT_m2.py Attention.to_out)
            t = torch.empty((ntok, d), dtype=torch.float32, device=x.device)
            ops.sgemm(0, 1, ntok, d, inner, o, inner, w_out, inner, t, d, bias=b_out, arith=self.gemm_arith, lib=lib)
            ops.dropout(t, drop[0], drop[1], out=t, lib=lib)
            y = torch.add(x.reshape(ntok, d), t).view_as(x)
        else:
            y = x.clone()                                                                                 # the residual, accumulated by beta = 1
            ops.sgemm(0, 1, ntok, d, inner, o, inner, w_out, inner, y, d, bias=b_out, beta=1.0, arith=self.gemm_arith, lib=lib)  # to_out + x
        return y, ((qkv, o, lse, drop) if save else None)

    def _attn_layer_backward(self, desc, x_in, dy, att, smap, G, out=None, ws_key="attn"):
        c, lib = self._cfg, self._lib
        d, heads, dh = c["d"], c["heads"], c["dh"]
        names = desc[0]
        mode, per = self._attn_mode(smap)
        if mode == "fused":
            ws = self._workspace(ws_key, lib.size("rat_attn_bwd_workspace", d, heads, dh))
            grads = ops.attn_params(*[G(n) if n else None for n in names])
            dx, _ = ops.attn_bwd(x_in, dy, att[0], att[1], desc[1], grads, smap, d, heads, dh, workspace=ws, arith=self.arith,
                                 dropout=att[2], out=out, lib=lib)
            return dx
        assert out is None or mode == "grouped", "a caller-provided gradient grid is wired for the fused and the grouped kernels only"
        if mode == "grouped" and att[0] == "loop" and (self._groups_supported() & 2):
            # small embedding dimension: the whole wide-head backward in one launch, gradients straight into the full-width tensors
            _, o_all, l_all, drop = att
            ws = self._workspace("attn_groups", lib.size("rat_attn_bwd_groups_workspace", d, heads, dh))
            grads = ops.attn_params(*[G(n) for n in names])
            dx, _ = ops.attn_bwd_groups(x_in, dy, dy, o_all, l_all, desc[1], grads, smap, d, heads, dh, workspace=ws, out=out,
                                        dropout=drop, lib=lib)
            return dx
        if mode == "grouped":
            groups, ig = heads // per, per * dh
            ws = self._workspace("attn", lib.size("rat_attn_bwd_workspace", d, per, dh))
            g_lng, g_lnb, g_wqkv, g_wout, g_bout = [G(n) for n in names]
            gq, go = g_wqkv.view(3, groups, ig, d), g_wout.view(d, groups, ig)
            # every group writes its gradients into its own slice of group-major buffers; ONE permuting copy per weight tensor and one
            # sum per LayerNorm vector move them into the gradient bucket afterwards (not two copies and two adds per group)
            t_ln = torch.empty((2, groups, g_lng.numel()), dtype=torch.float32, device=dy.device)
            t_b = torch.empty_like(g_bout)
            t_w = torch.empty((groups, 3 * ig, d), dtype=torch.float32, device=dy.device)
            t_wo = torch.empty((groups, d, ig), dtype=torch.float32, device=dy.device)
            dx = out                                     # (a caller's grid: must NOT be dy itself — every group reads dy)
            if att[0] == "loop":                         # the forward ran as one launch: per-group weight copies are made here
                _, o_all, l_all, drop = att
                att = [(w_g, wo_g, params_g, zb, o_all[g], l_all[g], drop)
                       for g, (w_g, wo_g, params_g, zb) in enumerate(self._group_weights(names, per, getattr(desc[1], "gplanes", None)))]
            for g, (w_g, wo_g, params_g, zb, o, l, drop) in enumerate(att):
                first = g == 0
                grads_g = ops.attn_params(t_ln[0, g], t_ln[1, g], t_w[g], t_wo[g], g_bout if first else t_b)
                dx, _ = ops.attn_bwd_ex(x_in, dy, dy if first else dx, o, l, params_g, grads_g, smap, d, per, dh, 0.0, 1.0,
                                        workspace=ws, out=dx, arith=self.arith, dropout=drop, lib=lib)   # dx = dy + sum over groups, in place
            gq.copy_(t_w.view(groups, 3, ig, d).permute(1, 0, 2, 3))
            go.copy_(t_wo.permute(1, 0, 2))
            torch.sum(t_ln[0], 0, out=g_lng)                                               # LayerNorm sees every group's gradient
            torch.sum(t_ln[1], 0, out=g_lnb)
            return dx
        inner, ntok = heads * dh, x_in.numel() // d
        ln_g, ln_b, w_qkv, w_out, b_out = [self._p(n) if n else None for n in names]
        qkv, o, lse, drop = att
        dev = dy.device
        xn = ops.layernorm_fwd(x_in, d, ntok, ln_g, ln_b, d, lib=lib)                                     # recomputed, not stored
        do = torch.empty((ntok, inner), dtype=torch.float32, device=dev)
        dyp = dy if drop[0] == 0 else ops.dropout(dy.reshape(ntok, d), drop[0], drop[1], lib=lib)         # through the projection's Dropout
        ops.sgemm(0, 0, ntok, inner, d, dyp, d, w_out, inner, do, inner, arith=self.gemm_arith, lib=lib)                         # dO = dy W_out
        ops.sgemm(1, 0, d, inner, ntok, dyp, d, o, inner, G(names[3]), inner, arith=self.gemm_arith, lib=lib)                    # dW_out = dy^T O
        ops.colsum(dyp, d, G(names[4]), ntok, d, lib=lib)
        dqkv = ops.attn_core_bwd_map(qkv, o, lse, do, smap, heads, dh, lib=lib)
        dxn = torch.empty((ntok, d), dtype=torch.float32, device=dev)
        ops.sgemm(0, 0, ntok, d, 3 * inner, dqkv, 3 * inner, w_qkv, d, dxn, d, arith=self.gemm_arith, lib=lib)                   # d(norm(x)) = dQKV W_qkv
        ops.sgemm(1, 0, 3 * inner, d, ntok, dqkv, 3 * inner, xn, d, G(names[2]), d, arith=self.gemm_arith, lib=lib)              # dW_qkv = dQKV^T norm(x)
        return ops.layernorm_bwd(x_in, d, dxn, ln_g, dxn, d, G(names[0]), G(names[1]), d, add=dy, lib=lib).view_as(x_in)

    def _build_encoder_descriptors(self):
        self._blocks = []
        for i in range(self._cfg["depth"]):
            blk = {}
            for which in ("intra", "cross"):
                blk[which] = self._attn_descriptor("encoder.encoder.%d.%s_attention." % (i, which))
            p = "encoder.encoder.%d.mlp.net." % i
            blk["ffn"] = [p + "0.weight", p + "0.bias", p + "3.weight", p + "3.bias"]
            blk["ffn_planes"] = None
            self._blocks.append(blk)
        self._build_weight_planes()

    def _build_weight_planes(self):
        """bf16x3 fragment planes of every encoder weight matrix, refreshed by ONE launch per forward (`_refresh_weight_planes`)
        instead of 2 - 3 split launches inside every attention / FFN call (52 per north-star step).  One uint8 buffer, a 16-byte
        aligned slot per layer; the job list holds raw pointers into the flat parameter buffer and into that buffer, both of
        which live as long as these descriptors."""
        c, lib = self._cfg, self._lib
        d, heads, dh, H = c["d"], c["heads"], c["dh"], c["hidden"]
        self._split_jobs = (None, 0)
        a_bytes = ops.attn_planes_bytes(d, heads, dh, lib=lib)
        g_bytes = ops.attn_groups_planes_bytes(d, heads, dh, lib=lib)      # wide heads: per-group planes of the one-launch forward
        f_bytes = ops.ffn_planes_bytes(d, H, lib=lib)
        slots = []
        for blk in self._blocks:
            for which in ("intra", "cross"):
                if (a_bytes or g_bytes) and blk[which][0][3] is not None:
                    slots.append((blk, which, a_bytes or g_bytes))
            if f_bytes:
                slots.append((blk, "ffn", f_bytes))
        if not slots:
            return
        total = sum((b + 15) // 16 * 16 for _, _, b in slots)
        self._planes = torch.empty(total, dtype=torch.uint8, device=self.device)
        jobs, off = [], 0
        for blk, which, nbytes in slots:
            view = self._planes[off:off + nbytes]
            off += (nbytes + 15) // 16 * 16
            if which == "ffn":
                w1, _, w2, _ = [self._p(n) for n in blk["ffn"]]
                got = ops.ffn_split_jobs(w1, w2, d, H, view, lib=lib)
                if got:
                    blk["ffn_planes"] = view
            elif a_bytes:
                got = ops.attn_split_jobs(blk[which][1], d, heads, dh, view, lib=lib)
                if got:
                    blk[which][1].planes = view.data_ptr()
            else:
                got = ops.attn_groups_split_jobs(blk[which][1], d, heads, dh, view, lib=lib)
                if got:
                    blk[which][1].gplanes = view
            jobs += got
        self._split_jobs = ops.split_job_array(jobs)

    def _refresh_weight_planes(self):
        """the weights may have changed since the last forward (optimizer step, load_state_dict): split them again — one launch"""
        arr, n = getattr(self, "_split_jobs", (None, 0))
        if n and self.arith == "bf16x3":
            ops.split_weights_batch(arr, n, self._flat, lib=self._lib)

    def _encoder_forward(self, x, x0, dims, save, saved):
        """depth x (intra attention, cross attention, FFN), each with its residual (RAT_m2.py:219-236).
        Returns (tensor holding the class token rows, row stride in floats)."""
        c, lib = self._cfg, self._lib
        B, T, L, S = dims
        d, H, heads, dh = c["d"], c["hidden"], c["heads"], c["dh"]
        imap, cmap = ops.intra_map(B, T, S), ops.cross_map(B, T, S)
        self._refresh_weight_planes()
        last = len(self._blocks) - 1
        prune = self.prune_dead_tokens and self._attn_mode(cmap)[0] in ("fused", "grouped")
        for bi, blk in enumerate(self._blocks):
            inplace = (not save) and (bi > 0 or x is not x0)   # eval: x0 must survive (DNN input), later grids are reused
            w1, b1, w2, b2 = [self._p(n) for n in blk["ffn"]]
            if bi == last and prune:
                # Only token (t = 0, s = 0) of the encoder's output is ever read (RAT_m2.py:138-140: x[:, 0][:, 0]).  In the LAST block
                # that token depends on the cross-sample sequence of token position 0 alone (B sequences instead of B S) — of which
                # only position 0 is a query —, on the feed-forward of ONE token per sample, and on the intra-sample layer's output
                # at token position 0 of every sample (all S positions are its keys and values, ONE is a query: RatSeqMap.queries).
                # Every other output of these three layers is dead — computed by the reference, read by nobody, with a gradient of
                # exactly zero.  They are not computed: same y_pred, same gradients.
                xa, a1 = self._attn_layer_forward(blk["intra"], x, ops.intra_map(B, T, S, queries=1), save, out=x if inplace else None)
                cm0 = ops.cross_map_label_token(B, T, S, queries=1)
                xb, a2 = self._attn_layer_forward(blk["cross"], xa, cm0, save, out=xa if not save else None)
                xcls_in = xb[:, 0, 0, :].contiguous()                                  # [B, d]
                xc = ops.ffn_fwd(xcls_in, w1, b1, w2, b2, d, H, arith=self.arith, lib=lib)
                if save:
                    saved["blocks"].append((x, a1, xa, a2, xcls_in))
                    saved["pruned"] = True
                return xc, d
            xa, a1 = self._attn_layer_forward(blk["intra"], x, imap, save, out=x if inplace else None)
            xb, a2 = self._attn_layer_forward(blk["cross"], xa, cmap, save, out=xa if not save else None)
            xc = ops.ffn_fwd(xb, w1, b1, w2, b2, d, H, out=xb if not save else None, arith=self.arith, lib=lib)
            if save:
                saved["blocks"].append((x, a1, xa, a2, xb))
            x = xc
        return x, T * S * d

    def _encoder_backward(self, saved, dx, G, dy_period=0):
        """dx: gradient of the tensor _encoder_forward returned -> gradient of the [B,T,S,d] grid.
        dy_period > 0: dx holds only the rows k * dy_period of that gradient ([B, d]: the class tokens'), every other row is zero."""
        c, lib = self._cfg, self._lib
        B, T, L, S = saved["dims"]
        d, H, heads, dh = c["d"], c["hidden"], c["heads"], c["dh"]
        imap, cmap = ops.intra_map(B, T, S), ops.cross_map(B, T, S)
        pruned = bool(saved.get("pruned"))
        # Every backward kernel ends with a reduction of its per-work-group gradient slabs.  With one slab workspace PER LAYER the
        # 3 x depth reductions run as one launch behind the last kernel (ops.deferred_reductions); wide heads (grouped launches) read
        # their weight gradients back between the groups and keep the immediate form.
        defer = (self.defer_slab_reductions and self._attn_mode(imap)[0] == "fused" and self._attn_mode(cmap)[0] == "fused"
                 and (not pruned or self._attn_is_fused(ops.cross_map_label_token(B, T, S, queries=1))))
        n_ffn = lib.size("rat_ffn_bwd_workspace", d, H)
        with ops.deferred_reductions(dx, lib, enabled=defer):
            for bi, (blk, (x_in, a1, xa, a2, xb)) in enumerate(zip(reversed(self._blocks), reversed(saved["blocks"]))):
                w1, b1, w2, b2 = [self._p(n) for n in blk["ffn"]]
                gw = [G(n) for n in blk["ffn"]]
                ws_ffn = self._workspace(("ffn", bi) if defer else "ffn", n_ffn)
                kc, ki = (("attn", bi, "cross"), ("attn", bi, "intra")) if defer else ("attn", "attn")
                if bi == 0 and pruned:                       # the last block (see _encoder_forward): dx is [B, d], the class tokens' gradient
                    dcls, _ = ops.ffn_bwd(xb, dx, w1, b1, w2, b2, gw[0], gw[1], gw[2], gw[3], d, H, workspace=ws_ffn, arith=self.arith,
                                          planes=blk["ffn_planes"], lib=lib)
                    dgrid = torch.zeros((B, T, S, d), dtype=torch.float32, device=dx.device)
                    dgrid[:, 0, 0, :] = dcls
                    # cross-sample attention backward over the B sequences of token position 0, IN PLACE: their rows of dgrid become dx,
                    # every other row stays zero — exactly the gradient the intra-sample layer below would have been handed
                    # (both layers with ONE query position per sequence — RatSeqMap.queries —: the gradient rows of the others are zero)
                    cm0 = ops.cross_map_label_token(B, T, S, queries=1)
                    # (wide heads run in groups and every group reads dy again: there the result goes to a second zero grid)
                    dxg = dgrid if self._attn_is_fused(cm0) else torch.zeros_like(dgrid)
                    dx = self._attn_layer_backward(blk["cross"], xa, dgrid, a2, cm0, G, out=dxg, ws_key=kc)
                    dx = self._attn_layer_backward(blk["intra"], x_in, dx, a1, ops.intra_map(B, T, S, queries=1), G, ws_key=ki)
                    continue
                if bi == 0 and dy_period:
                    dx, _ = ops.ffn_bwd_rows(xb, dx, dy_period, w1, b1, w2, b2, gw[0], gw[1], gw[2], gw[3], d, H, workspace=ws_ffn,
                                             arith=self.arith, planes=blk["ffn_planes"], lib=lib)
                else:
                    dx, _ = ops.ffn_bwd(xb, dx, w1, b1, w2, b2, gw[0], gw[1], gw[2], gw[3], d, H, workspace=ws_ffn, arith=self.arith,
                                        planes=blk["ffn_planes"], lib=lib)
                dx = self._attn_layer_backward(blk["cross"], xa, dx, a2, cmap, G, ws_key=kc)
                dx = self._attn_layer_backward(blk["intra"], x_in, dx, a1, imap, G, ws_key=ki)
        return dx

    # ------------------------------------------------------------------------------ flat parameter buffer
    def _trainable(self):
        """(name, param) of every tensor that receives a gradient, "embedding_layer" tensors first."""
        named = [(n, p) for n, p in self.named_parameters() if p.requires_grad and not n.startswith("query_proj")]
        emb = [(n, p) for n, p in named if "embedding_layer" in n]
        rest = [(n, p) for n, p in named if "embedding_layer" not in n]
        # inside the "embedding_layer" block: feature tables, then the LR ("wide") tables, then the label table — the two table
        # families are what the row-sparse gradient path covers, so they form one contiguous prefix of the buffer
        rank = lambda n: 0 if n.startswith("embedding_layer.") else (1 if n.startswith("lr_layer.") else 2)   # noqa: E731
        emb.sort(key=lambda np_: rank(np_[0]))
        return emb, rest

    def _after_device_move(self):
        """Re-home every trainable tensor into one flat buffer (16-byte aligned slots) and cache kernel descriptors."""
        emb, rest = self._trainable()
        offsets, off = OrderedDict(), 0
        for n, p in emb:
            offsets[n] = off
            off += (p.numel() + 3) // 4 * 4
        self._n_emb = off
        for n, p in rest:
            offsets[n] = off
            off += (p.numel() + 3) // 4 * 4
        flat = torch.zeros(off, dtype=torch.float32, device=self.device)
        with torch.no_grad():
            for n, p in emb + rest:
                view = flat[offsets[n]:offsets[n] + p.numel()].view_as(p)
                view.copy_(p.data)
                p.data = view
        self._flat, self._offsets, self._order = flat, offsets, [n for n, _ in emb + rest]
        self._params = OrderedDict(emb + rest)
        self._lib = get_lib() if self._lib is None else self._lib
        # table families at the head of the buffer: [0, _n_feat) feature tables (rows of d floats), [_n_feat, _n_tab) LR tables
        self._n_feat = sum((p.numel() + 3) // 4 * 4 for n, p in emb if n.startswith("embedding_layer."))
        self._n_tab = self._n_feat + sum((p.numel() + 3) // 4 * 4 for n, p in emb if n.startswith("lr_layer."))
        mode = self._embedding_grad
        d = self._cfg["d"]
        row_aligned = d % 4 == 0 and all(self._offsets[n] % d == 0 for n, _ in emb if n.startswith("embedding_layer."))
        if mode == "auto":
            big = self._n_feat * 4 > (8 << 30)
            mode = "sparse" if (big and self._cfg["lam_emb"] == 0 and row_aligned) else "atomic"
        if mode in ("sorted", "sparse") and not row_aligned:
            raise NotImplementedError("embedding_grad=%r needs embedding_dim %% 4 == 0 (table rows on 16-byte boundaries)" % mode)
        if mode == "sparse" and self._cfg["lam_emb"] != 0:
            raise NotImplementedError("embedding_grad='sparse' touches only the rows of the batch: the reference's dense L2 term "
                                      "(lambda*W on EVERY row, base_model.py:79-94) cannot be carried — use embedding_regularizer=0")
        if mode == "sparse" and getattr(self.optimizer, "kind", "Adam") != "Adam":
            if self._embedding_grad == "sparse":
                raise NotImplementedError("embedding_grad='sparse' (row lists + lazy row updates) is wired for Adam only; optimizer=%r "
                                          "needs dense table gradients ('atomic' / 'sorted')" % self.optimizer.kind)
            mode = "atomic"                     # `auto` on a large table with another optimizer: dense gradients, said here, not at step 1
        self._grad_mode = mode
        self._lists_possible = row_aligned      # (sort + segmented reduction needs table rows on 16-byte boundaries)
        self._n_sparse = self._n_tab if mode == "sparse" else 0
        self._gbuf = None
        self.__dict__.pop("_step_graphs", None)        # captured steps point into the old buffers
        self.__dict__.pop("_eval_graphs", None)
        self._build_descriptors()
        self.set_arith(self._arith_request)

    def _p(self, name):
        return self._params[name].data

    def _build_descriptors(self):
        c, dev = self._cfg, self.device
        emb_prefix = "embedding_layer.embedding_layer.embedding_layer."
        self._tables = [self._p(emb_prefix + f.name + ".weight") for f in self._fields]
        self._ftab = ops.field_table(self._fields, self._tables, dev)
        if c["use_wide"]:
            lr_prefix = "lr_layer.embedding_layer.embedding_layer.embedding_layer."
            self._lr_tables = [self._p(lr_prefix + f.name + ".weight") for f in self._fields]
            self._lr_ftab = ops.field_table(self._fields, self._lr_tables, dev)
        else:
            self._lr_tables, self._lr_ftab = None, None
        self._col2field = None                                  # built on first use (needs the batch's column count)
        self._build_encoder_descriptors()
        # DNN head layout: [(linear_idx, bn_idx or None, dropout_p)], out linear idx
        self._dnn_layers, self._dnn_acts, self._dnn_out = [], [], None
        if self.dnn is not None:
            mods = list(self.dnn.dnn)
            j, last = 0, len(mods) - 1
            while j < last:
                lin, bn, pdrop, act = j, None, 0.0, 1            # (act 1 = RAT_ACT_NONE: no activation module behind this layer)
                j += 1
                if j < last and isinstance(mods[j], nn.BatchNorm1d):
                    bn = j
                    j += 1
                if j < last and not isinstance(mods[j], (nn.Linear, nn.Dropout)):
                    act = _activation_code(mods[j])
                    j += 1
                if j < last and isinstance(mods[j], nn.Dropout):
                    pdrop = mods[j].p
                    j += 1
                self._dnn_layers.append((lin, bn, pdrop))
                self._dnn_acts.append(act)
            self._dnn_out = last
        # BatchNorm's num_batches_tracked (deep.py:128-132 -> nn.BatchNorm1d): one int64 tensor, every layer's buffer a 0-dim view of
        # it, so that a training forward advances all of them with ONE launch (state_dict keys / values are unchanged)
        bns = [self.dnn.dnn[bn] for _, bn, _ in self._dnn_layers if bn is not None] if self.dnn is not None else []
        self._bn_counts = None
        if bns:
            counts = torch.stack([m.num_batches_tracked.detach().to(dev).reshape(()) for m in bns]).contiguous()
            for i, m in enumerate(bns):
                m._buffers["num_batches_tracked"] = counts[i]
            self._bn_counts = counts

    def _grad_buffer(self):
        """-> (zeroed flat gradient buffer, its embedding / LR gradient field tables).  ONE persistent buffer is reused (so the
        RatField tables that point into it are built and uploaded once) whenever the previous step's gradients were released
        (optimizer.zero_grad()); if some p.grad still aliases it — gradient accumulation over several backward calls — a fresh
        buffer is used so that autograd's `p.grad += new` never adds a tensor to itself."""
        dev = self._flat.device
        emb_prefix = "embedding_layer.embedding_layer.embedding_layer."
        lr_prefix = "lr_layer.embedding_layer.embedding_layer.embedding_layer."
        self._settle_pending_reduce()        # RCCL may still be reducing the buffer that is about to be zeroed / replaced

        def tables(g):
            view = lambda n: self._gflat_view(g, n)           # noqa: E731
            gft = ops.field_table(self._fields, [view(emb_prefix + f.name + ".weight") for f in self._fields], dev)
            lft = ops.field_table(self._fields, [view(lr_prefix + f.name + ".weight") for f in self._fields], dev) \
                if self._cfg["use_wide"] else None
            return gft, lft
        held = any(p.grad is not None for p in self._params.values())
        size = self._flat.numel() - self._n_sparse
        if self._gbuf is not None and not held and self._gbuf[0].numel() == size and self._gbuf[0].device == dev:
            g, gft, lft = self._gbuf
            if not self._gbuf_clean:           # (the two-sweep optimizer leaves g = 0 behind: rat_clip_adam_fused, zero_g)
                g.zero_()
            self._gbuf_clean = False
            return g, gft, lft
        g = torch.zeros(size, dtype=torch.float32, device=dev)
        gft, lft = tables(g) if self._n_sparse == 0 else (None, None)
        if not held:
            self._gbuf = (g, gft, lft)
            self._gbuf_clean = False
        return g, gft, lft

    def _settle_pending_reduce(self):
        """A backward that was NOT followed by _exchange_gradients (a custom loop, a second backward) leaves the dense-net
        all-reduce it started in flight: wait for it before its buffer is zeroed, replaced or accumulated into."""
        pending, self._pending_reduce = self._pending_reduce, None
        if pending is not None:
            pending[0][0].wait()

    def _gflat_view(self, gflat, name):
        """view of parameter `name`'s gradient inside the flat gradient buffer (which, in sparse mode, starts BEHIND the tables)"""
        p = self._params[name]
        o = self._offsets[name] - self._n_sparse
        assert o >= 0, "the tables have no dense gradient in sparse mode"
        return gflat[o:o + p.numel()].view_as(p)

    def _dense_names(self):
        return [n for n in self._order if self._offsets[n] >= self._n_sparse]

    def _gather_flat_grad(self):
        """The flat gradient the last backward produced, or one assembled from p.grad if autograd copied them."""
        g = self._last_gflat
        names = self._dense_names()
        if g is not None and all(self._params[n].grad is not None and
                                 self._params[n].grad.data_ptr() == g.data_ptr() + 4 * (self._offsets[n] - self._n_sparse) for n in names):
            return g
        if all(self._params[n].grad is None for n in names):
            return None
        g = torch.zeros(self._flat.numel() - self._n_sparse, dtype=torch.float32, device=self._flat.device)
        for n in names:
            if self._params[n].grad is not None:
                self._gflat_view(g, n).copy_(self._params[n].grad)
        return g

    def check_id_errors(self):
        """nn.Embedding raises IndexError on an out-of-vocabulary id (embedding.py:158-178); the kernels clamp for memory safety
        and rat_check_ids counts the offenders on the device.  This reads the counters (ONE host synchronisation — called where
        the training loop synchronises anyway: end of an epoch / of an evaluation) and raises like the reference would."""
        world = self._world_size()
        if self._id_errors is None:
            if not self._dp():
                return
            self._id_errors = torch.zeros(2, dtype=torch.int32, device=self.device)
        counts = self._id_errors
        if self._dp():           # every rank raises together (a rank that raised alone would leave the others in the next collective)
            counts = self._all_reduce_sum(counts.clone())
        bad_ids, bad_labels = [int(v) for v in counts.tolist()]
        if bad_ids or bad_labels:
            self._id_errors.zero_()
            raise IndexError("index out of range in self: %d feature id(s) outside their embedding table and %d label id(s) "
                             "outside {0, 1} were fed to the model since the last check (feature_map / data mismatch?)"
                             % (bad_ids, bad_labels))

    # ------------------------------------------------------------------------------ batch plumbing
    def _prepare_batch(self, inputs, out=None):
        """inputs_to_device (base_model.py:125-133) + the slicing of RAT_m2.forward lines 110-116: ids become int32
        once, on the host side of the boundary; the target row's label token is id 2.  A 4-tuple that already sits on the device
        takes ONE launch (rat_batch_prepare), written straight into `out` (a captured step's static inputs) when given."""
        if isinstance(inputs, DeviceBatch):            # assembled on the device by rat_batch_assemble (data.py)
            return inputs.idx, inputs.label_ids, inputs.y_true
        X, y = inputs[0], inputs[1]
        if len(inputs) >= 4:
            assert inputs[3].ndim == 1, "RIM does not support label-wise retrieval-enhanced training"
        assert X.ndim == 3, "retrieval augmented mode requires input_shape like [Bx(1+K)xF]"
        if X.is_cuda and y.is_cuda and X.device == self.device and y.dtype in (torch.float32, torch.float64) \
                and X.dtype in (torch.int32, torch.int64, torch.float32, torch.float64):
            return ops.batch_prepare(X, y, out=out, lib=self._lib)
        idx = X.to(torch.int32) if X.dtype != torch.int32 else X
        labels = y.to(torch.int32).clone()
        labels[:, 0] = 2
        y_true = y[:, 0].to(torch.float32)
        dev = self.device
        return (idx.contiguous().to(dev, non_blocking=True), labels.contiguous().to(dev, non_blocking=True),
                y_true.contiguous().to(dev, non_blocking=True))

    # ------------------------------------------------------------------------------ public forward
    def forward(self, inputs):
        """RAT_m2.forward (RAT_m2.py:104-152)."""
        ready = None
        if not self.training and not isinstance(inputs, DeviceBatch) and inputs[0].is_cuda and inputs[0].ndim == 3:
            entry = self.__dict__.get("_eval_graphs", {}).get(self._eval_graph_key(tuple(inputs[0].shape)))
            ready = entry[1] if entry is not None and entry[1] else None
        batch = self._prepare_batch(inputs, out=ready.static if ready is not None else None)
        self.batch_size = batch[0].shape[0]
        if torch.is_grad_enabled() and self.training:
            y_pred, _, _ = _RATFunction.apply(self, batch, False, *[self._params[n] for n in self._order])
        else:
            graph = self._eval_graph_for(batch) if not self.training else None
            if graph is not None:
                y_pred = graph.run(batch)
            else:
                y_pred, _, _, _ = self._run_forward(batch, save=False, with_reg=False)
        # `ready` means batch[2] IS the captured graph's static input: the next same-shape batch overwrites it, and evaluate_generator
        # keeps every batch's y_true until the end of the pass (base_model.py:232-251) - hand out a copy, never the static view
        y_true = batch[2].clone() if ready is not None else batch[2]
        return {"y_true": y_true.unsqueeze(-1), "y_pred": y_pred}

    # The inference forward as a hipGraph (graph.EvalGraph): used for batches of at most `eval_graph_max_batch` samples — where the
    # forward is short enough for its ~25 launches to be the limit (measured: bench.py `inference`) — after `graph_warmup` eager
    # forwards of a shape; at most `graph_shapes` shapes (the full batch and an evaluation set's tail batch).
    eval_graph = True
    eval_graph_max_batch = 4096

    def _eval_graph_key(self, shape):
        return (tuple(shape), self.arith, self.gemm_arith, bool(self.prune_dead_tokens), bool(self._validate_ids), bool(self._head_strips))

    def _eval_graph_for(self, batch):
        if not (self.eval_graph and self.use_graph and batch[0].is_cuda and not self._dp() and batch[0].shape[0] <= self.eval_graph_max_batch):
            return None
        graphs = self.__dict__.setdefault("_eval_graphs", {})
        key = self._eval_graph_key(tuple(batch[0].shape))
        entry = graphs.get(key)
        if entry is None:
            if len(graphs) >= self.graph_shapes:
                return None
            entry = graphs[key] = [0, None]
        if entry[1] is None:
            entry[0] += 1
            if entry[0] <= self.graph_warmup:
                return None
            from .graph import EvalGraph
            try:
                entry[1] = EvalGraph(self, batch)
            except Exception as exc:
                import logging
                logging.warning("hipGraph capture of the inference forward failed (%s: %s); continuing with eager launches",
                                type(exc).__name__, exc)
                entry[1] = False
        return entry[1] or None

    # ------------------------------------------------------------------------------ the fused training iteration
    def _fused_train_step(self, inputs):
        """BaseModel.train_one_epoch's iteration (base_model.py:220-226) as ONE pass over the flat buffers, no autograd node:
        forward -> BCE -> backward (gradients WITHOUT the regulariser term) -> [exchange] -> rat_sumsq_reg (clip norm of g + lambda W
        and the regulariser's value in one sweep) -> rat_clip_adam_fused (Adam on g + lambda W, leaves g = 0: the next zero_grad).
        Same arithmetic per element as the reference's sequence; p.grad is not populated (it would be None after zero_grad anyway).
        On a GPU the iteration is captured into a hipGraph after `graph_warmup` eager steps of the same batch shape and replayed
        from then on (graph.StepGraph; `use_graph = False` keeps it eager).  Dropout does not: its generator state lives on the
        device (_dropout_begin), so every replay draws new masks."""
        if not self.training:
            raise RuntimeError("train_step() on a model in eval mode")
        # a batch shape that already has its graph: the conversion writes into the graph's static inputs (no copies in front of the replay)
        ready = None
        if not isinstance(inputs, DeviceBatch) and inputs[0].is_cuda and inputs[0].ndim == 3:
            entry = self.__dict__.get("_step_graphs", {}).get(self._step_graph_key(tuple(inputs[0].shape)))
            ready = entry[1] if entry is not None and entry[1] else None
        batch = self._prepare_batch(inputs, out=ready.static if ready is not None else None)
        self.batch_size = batch[0].shape[0]
        if any(p.grad is not None for p in self._params.values()):       # optimizer.zero_grad() of the reference's iteration
            self.optimizer.zero_grad()
            self._gbuf_clean = False
        self._settle_pending_reduce()
        graph = self._step_graph_for(batch)
        if graph is not None:
            return graph.run(batch)
        with torch.no_grad():
            self.optimizer.prepare_step()
            return self._fused_iteration(batch, count=True)

    def _fused_iteration(self, batch, count):
        """the launches of one fused iteration on `batch` = (idx, label ids, y_true) device tensors -> total loss (device scalar)"""
        world = self._world_size()
        inv = self._inv_world()
        # the step's accumulator scalars — BCE sum, clip norm^2, regulariser value — share one 4-float tensor: ONE fill per step
        # (rat_step_begin: that fill, the optimizer's clock tick and the BatchNorm layers' num_batches_tracked in one launch)
        counts = self._bn_counts if self.training else None
        scal = self.optimizer.begin_step(counts, count=count)
        self._step_loss = scal[0:1]
        self._bn_counted = counts is not None
        lists = self._dp() and self._row_lists_travel_lighter(batch[0].shape, world)
        self._owner_state = None
        if self._dp() and self._use_owner_exchange() and self._lists_possible and (lists or self._grad_mode == "sparse"):
            B, T, L = batch[0].shape
            self._owner_prepare(batch[0], (B, T, L, self._cfg["nf"] + 1))
        try:
            try:
                _y_pred, loss, _reg, saved = self._run_forward(batch, save=True, with_reg=False)
            finally:
                self.__dict__.pop("_bn_counted", None)
                self.__dict__.pop("_step_loss", None)
            if self._graph_test_splits:
                self._collective(lambda: None)             # (test knob: a segment boundary where SyncBN / the exchange would put one)
            self._run_backward(saved, inv, None, table_lists=lists)
            if self._graph_test_splits:
                self._collective(lambda: None)
            g = self._last_gflat
            self._exchange_gradients(g)
        except BaseException:
            # rat_step_begin already advanced the optimizer's clock and the BatchNorm layers' num_batches_tracked, and no update will
            # follow: take the tick back (host mirror now, device clock at the next prepare_step) so that a caller who catches the
            # error — a bad batch skipped, an out-of-memory retried at a smaller size — continues with an unshifted bias correction
            self._owner_state = None
            if count and self._tape is None:
                self.optimizer.untick(counts)
            raise
        reg = self.optimizer.fused_step(g, self._max_gradient_norm, count=count, zeroed=True, ticked=True)
        self._gbuf_clean = self._gbuf is not None and g is self._gbuf[0]
        total = loss + reg[0]
        return total if world == 1 else total * inv[0]

    def _row_lists_travel_lighter(self, idx_shape, world):
        """Table gradients under data parallelism in the dense modes: all-gather of (row ids, gradient rows) lists at capacity (every
        rank receives world x min(pairs, rows) x (4 d + 4) bytes, then sorts and merges them) or one dense all-reduce of the table block
        (a ring moves ~2 x its size per rank)?  Lists win at the strong-scaling rank shape (north star, 8 ranks of 512 samples: 8 x 28.8 MB
        against 2 x 257 MB) and lose when every rank brings a full batch (weak scaling at B = 4096: 8 x 230 MB): pick by bytes, with
        a margin for the merge."""
        if self.row_list_exchange is not None:
            return bool(self.row_list_exchange)
        B, T, L = [int(v) for v in idx_shape]
        d = self._cfg["d"]
        rows_feat, rows_lr = self._n_feat // d, self._n_tab - self._n_feat
        lists = world * (min(B * T * L, rows_feat) * (4 * d + 4) + min(B * L, rows_lr) * 8)
        dense = 2 * 4 * self._n_tab
        return lists < 0.6 * dense

    def _inv_world(self):
        world = self._world_size()
        t = getattr(self, "_inv_world_t", None)
        if t is None or t.device != self._flat.device or self._inv_world_n != world:
            t = torch.full((1,), 1.0 / world, dtype=torch.float32, device=self._flat.device)
            self._inv_world_t, self._inv_world_n = t, world
        return t

    # Dead-token pruning of the last encoder block (see _encoder_forward): on by default — identical predictions and gradients; of the
    # twelve encoder layers (depth 4) two shrink to 1/S and 1/(T S) of their size and one keeps a single query per sequence.  bench.py
    # reports the step with it OFF as its headline value (the reference's amount of work) and the step with it ON beside it.
    prune_dead_tokens = True
    row_list_exchange = None    # None: decide by traffic (_row_lists_travel_lighter); True / False: force (tests, experiments)
    _graph_test_splits = False  # tests: cut the captured step into segments the way collectives do under data parallelism
    use_graph = True           # capture the fused iteration into a hipGraph (CUDA devices only)
    # Under data parallelism the captured step is a chain of graph segments with the collectives between them as eager closures
    # (graph.py): proven on one GPU with two gloo ranks (tests/test_gpu_dp.py) and with one RCCL rank (tests/test_gpu_rccl.py,
    # bench.py --dp-rehearsal).  OFF by default on the model: no run with more than one RCCL rank exists yet, and fit_generator has no
    # watchdog that could take a stalled segmented-graph collective back to the eager step.  bench.py opts in — it measures
    # the eager form first and runs the graph form under a timer; `graph_under_dp=True` in the model's kwargs opts a training run in.
    # A capture that fails leaves that batch shape on the eager step (_step_graph_for).
    graph_under_dp = False
    group_loop = True          # wide heads (heads = G x 8 at d = 64): the forward of a layer as ONE launch that loops over the head groups
                               # (rat_attn_fwd_groups) instead of G launches; False: the per-group launches (A/B, tests)
    graph_warmup = 2           # eager fused steps of a batch shape before it is captured
    graph_shapes = 2           # at most this many batch shapes get a graph (the full batch and an epoch's tail batch)

    def _step_graph_key(self, shape):
        c = self._cfg
        group = self.optimizer.param_groups[0]
        return (tuple(shape), self.arith, self.gemm_arith, self._world_size(), self._dp(), bool(self.prune_dead_tokens),
                # kernel arguments and control flow the recorded launches carry: a change of any of them takes a new capture
                self._max_gradient_norm, self.optimizer.kind, tuple(group.get("betas", ())), group.get("eps"), group.get("alpha"),
                c["lam_emb"], c["lam_net"], self._grad_mode,
                self.row_list_exchange, bool(self._graph_test_splits), bool(self._validate_ids), bool(self._head_strips),
                bool(self.defer_slab_reductions), bool(self.owner_exchange))

    def _step_graph_for(self, batch):
        if not (self.use_graph and batch[0].is_cuda):
            return None
        if self._dp() and not self.graph_under_dp:
            return None
        graphs = self.__dict__.setdefault("_step_graphs", {})
        key = self._step_graph_key(tuple(batch[0].shape))
        entry = graphs.get(key)
        if entry is None:
            if len(graphs) >= self.graph_shapes:
                return None
            entry = graphs[key] = [0, None]
        if entry[1] is None:
            entry[0] += 1
            if entry[0] <= self.graph_warmup:
                return None                                # eager: fills every lazily built cache (workspaces, plans, tables)
            from .graph import StepGraph
            try:
                entry[1] = StepGraph(self, batch)
            except Exception as exc:                       # capture refused (driver / runtime / collective inside a segment ...):
                import logging                             # this shape stays on the eager fused step
                logging.warning("hipGraph capture of the training step failed (%s: %s); continuing with eager launches",
                                type(exc).__name__, exc)
                self._tape = None
                self._sparse = self._table_lists = self._pending_reduce = None
                if self._gbuf is not None:
                    self._gbuf[0].zero_()
                    self._gbuf_clean = True
                entry[1] = False
        return entry[1] or None

    def _loss_terms(self, inputs, with_reg):
        batch = self._prepare_batch(inputs)
        self.batch_size = batch[0].shape[0]
        if torch.is_grad_enabled():
            y_pred, loss, reg = _RATFunction.apply(self, batch, with_reg, *[self._params[n] for n in self._order])
        else:
            y_pred, loss, reg, _ = self._run_forward(batch, save=False, with_reg=with_reg)
        return y_pred, loss, reg

    def _regularization_value(self):
        c = self._cfg
        reg = torch.zeros(1, dtype=torch.float32, device=self.device)
        if c["lam_emb"] > 0 and self._n_emb > 0:
            ops.sumsq(self._flat[:self._n_emb], reg, lib=self._lib)
            reg = reg * (0.5 * c["lam_emb"])
        if c["lam_net"] > 0:
            rest = torch.zeros(1, dtype=torch.float32, device=self.device)
            ops.sumsq(self._flat[self._n_emb:], rest, lib=self._lib)
            reg = reg + rest * (0.5 * c["lam_net"])
        return reg[0]

    # ------------------------------------------------------------------------------ the HIP pipeline
    def _run_forward(self, batch, save, with_reg):
        c, lib = self._cfg, self._lib
        idx, labels, y_true = batch
        B, T, L = idx.shape
        d, F, H, heads, dh = c["d"], c["nf"], c["hidden"], c["heads"], c["dh"]
        S = F + 1
        training = self.training
        if self._validate_ids:
            if self._id_errors is None:
                self._id_errors = torch.zeros(2, dtype=torch.int32, device=idx.device)
            ops.check_ids(idx, labels, self._ftab, F, self._id_errors, B, T, L, lib=lib)
        x0 = ops.gather_fwd(idx, labels, self._ftab, F, self._p("label_embedding_layer.weight"), B, T, L, d, lib=lib)
        saved = {"batch": batch, "dims": (B, T, L, S), "blocks": [], "dnn": []}
        # dropout: the masks are counter-based functions of per-layer seed WORDS that live on the device and are refreshed once per
        # training forward (rat_dropout_seeds) — no host draw per step, so a captured step replays with new masks; backward re-derives
        drop = training and (c["emb_dropout"] > 0 or any(p > 0 for _, _, p in self._dnn_layers))
        if training and (drop or c["attn_dropout"] > 0):
            self._dropout_begin()
        seeds = [self._dropout_word() for _ in range(1 + len(self._dnn_layers))] if drop else None
        saved["seeds"] = seeds
        # ---- DNN branch on the target sample's raw field embeddings (RAT_m2.py:145-146; deep.py:126-141)
        dnn_out = dnn_last = None
        if training and self._bn_counts is not None and not self.__dict__.get("_bn_counted", False):
            self._bn_counts.add_(1)                                         # every BatchNorm layer's num_batches_tracked (shared storage)
        if self.dnn is not None:
            mods = self.dnn.dnn
            a_prev, lda, K = x0[:, 0, 1:, :], T * S * d, F * d
            for li, (lin, bn, pdrop) in enumerate(self._dnn_layers):
                W, bvec = mods[lin].weight.data, mods[lin].bias.data
                N = W.shape[0]
                z = torch.empty((B, N), dtype=torch.float32, device=x0.device)
                ops.sgemm(0, 1, B, N, K, a_prev, lda, W, K, z, N, bias=bvec, arith=self.gemm_arith, lib=lib)
                if bn is not None:
                    m = mods[bn]
                    if training and self._sync_bn and self._dp():                     # SyncBN: statistics of the GLOBAL batch
                        a, sm, sr, gstats = ops.bn_relu_fwd_sync(z, m.weight.data, m.bias.data, m.running_mean, m.running_var,
                                                                 self._all_gather_flat, eps=m.eps, momentum=m.momentum,
                                                                 act=self._dnn_acts[li], lib=lib)
                        sm = (sm, gstats)
                    else:
                        fwd = ops.bn_act_fwd_strip if self._head_strips and B <= self._strip_fwd_rows and ops.bn_strip_ok(B, N, lib) else ops.bn_relu_fwd
                        a, sm, sr = fwd(z, m.weight.data, m.bias.data, m.running_mean, m.running_var, training, True,
                                        eps=m.eps, momentum=m.momentum, act=self._dnn_acts[li], lib=lib)
                else:
                    fwd = ops.bn_act_fwd_strip if self._head_strips and B <= self._strip_fwd_rows and ops.bn_strip_ok(B, N, lib) else ops.bn_relu_fwd
                    a, sm, sr = fwd(z, None, None, None, None, training, False, act=self._dnn_acts[li], lib=lib)
                a_act = a
                if drop and pdrop > 0:                                      # net_dropout (deep.py:133-134)
                    a = ops.dropout(a_act, pdrop, seeds[1 + li], lib=lib)
                if save:
                    saved["dnn"].append((a_prev, lda, K, z, a_act, sm, sr))
                a_prev, lda, K = a, N, N
            W, bvec = mods[self._dnn_out].weight.data, mods[self._dnn_out].bias.data
            if self._head_strips:                                           # the one-output Linear is evaluated inside rat_logit_fwd_dnn
                dnn_last = (a_prev, lda, K, W, bvec)
            else:
                dnn_out = torch.empty((B, 1), dtype=torch.float32, device=x0.device)
                ops.sgemm(0, 1, B, 1, K, a_prev, lda, W, K, dnn_out, 1, bias=bvec, arith=self.gemm_arith, lib=lib)
            if save:
                saved["dnn_last"] = (a_prev, lda, K)
        # ---- encoder on the token grid
        x = x0
        if drop and c["emb_dropout"] > 0:                                  # self.dropout(x) (RAT_m2.py:135); X_emb for the DNN stays un-dropped
            x = ops.dropout(x0, c["emb_dropout"], seeds[0], lib=lib)
        x, cls_stride = self._encoder_forward(x, x0, (B, T, L, S), save, saved)
        # ---- logit = fc(cls) + dnn + wide ; sigmoid ; BCE
        loss = self.__dict__.pop("_step_loss", None)                      # the fused iteration's zeroed scalar (one fill per step)
        if loss is None:
            loss = torch.zeros(1, dtype=torch.float32, device=x0.device)
        y_pred = ops.logit_fwd(x, cls_stride, self.fc.weight.data, self.fc.bias.data, dnn_out, self._lr_ftab, F, idx, T * L,
                               y_true, loss, B, d, head=self._head, dnn_last=dnn_last, lib=lib)
        reg = self._regularization_value() if with_reg else None
        if save:
            saved["x_final"], saved["cls_stride"], saved["y_pred"] = x, cls_stride, y_pred
        return y_pred, loss[0], reg, saved

    _DROP_WORDS = 64

    def _dropout_begin(self):
        """Start of a training forward with a positive dropout rate: advance the device-side generator state (base seed drawn ONCE
        from torch's CPU generator — so seed_everything governs it — plus a device counter) and hand out this step's seed words.
        Under autograd (_RATFunction: another forward may run before this one's backward) the step works on a private copy of the
        words; the fused / captured iteration reads the shared words directly."""
        if self.__dict__.get("_drop_words") is None or self._drop_words.device != self._flat.device:
            self._dropout_state_init()
        ops.dropout_seeds(self._drop_words, self._drop_base, self._drop_counter, lib=self._lib)
        self._drop_step_words = self._drop_words.clone() if self.__dict__.get("_drop_private") else self._drop_words
        self._drop_cursor = 0

    def _dropout_state_init(self, base=None, counter=0):
        """(base seed, step counter) of the device-side dropout generator.  The base is drawn ONCE from torch's CPU generator
        (seed_everything governs it) and — under data parallelism, where every rank draws the same number — mixed with the rank, so that
        the ranks mask their different shards with different masks, like the reference's per-process generators would.  The UNMIXED
        draw is what a checkpoint stores (only rank 0 writes one): every rank re-applies its own mix on load, so a resumed run draws
        the masks the uninterrupted run would have drawn."""
        if base is None:
            base = int(torch.randint(0, 2 ** 62, (1,)))
        self._drop_base_unmixed = int(base)
        if self._dp():
            import torch.distributed as dist
            base = (base ^ (0x9E3779B97F4A7C15 * (dist.get_rank() + 1))) & (2 ** 62 - 1)
        dev = self._flat.device
        self._drop_base = int(base)
        self._drop_words = torch.zeros(self._DROP_WORDS, dtype=torch.int64, device=dev)
        self._drop_counter = torch.full((1,), int(counter), dtype=torch.int64, device=dev)

    def dropout_state(self):
        """-> {"base", "counter"} or None when no training forward with dropout has run: what a checkpoint needs to continue the mask
        sequence (`base` is the draw before the rank is mixed in).  The reference's `.model` file is the bare state_dict and must stay
        loadable by it, so the state travels with the optimizer's state_dict (`rat_dropout`, next to the moments and the step count —
        the resume state the reference does not have)"""
        if self.__dict__.get("_drop_words") is None:
            return None
        return {"base": int(self._drop_base_unmixed), "counter": int(self._drop_counter.cpu()[0])}

    def load_dropout_state(self, state):
        if state:
            self._dropout_state_init(int(state["base"]), int(state["counter"]))

    def _dropout_word(self):
        i = self._drop_cursor
        if i >= self._DROP_WORDS:
            raise RuntimeError("more than %d dropout layers" % self._DROP_WORDS)
        self._drop_cursor = i + 1
        return self._drop_step_words[i:i + 1]

    # column-strip BatchNorm / activation launches (rat_bn_act_*_strip: one launch per layer and direction, the Linear's bias gradient and —
    # last hidden layer — the one-output Linear's gradients included).  A strip work-group walks ALL rows of its 8 columns, so its time grows
    # with the batch while the row-split two-launch forms spread a tall matrix over the chip: measured on MI355X (rocprofv3, N = 400) the
    # forward strip wins up to 2048 rows (B = 512: 7.9 against 10.7 us; B = 4096: 25.4 against 17.6), the backward strip — which also
    # replaces the two column-sum launches — up to 4096 (B = 4096: 29.2 against 39.8 us).  False = always the older forms.
    _head_strips = True
    defer_slab_reductions = True     # RAT_m2's encoder backward: one slab-reduction launch per step instead of one per layer (see _encoder_backward)
    _strip_fwd_rows = 2048
    _strip_bwd_rows = 4096

    def _workspace(self, key, nbytes):
        ws = self._ws.get(key)
        if ws is None or ws.numel() * 4 < nbytes:
            if ws is not None:                                   # a captured step may have this buffer's address baked in
                self._ws.setdefault("retired", []).append(ws)
            ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=self.device)
            self._ws[key] = ws
        return ws

    def _run_backward(self, saved, g_loss, g_reg, table_lists=False):
        """g_loss / g_reg: DEVICE scalars (autograd's incoming gradients of the loss / regulariser outputs; g_reg None = no
        regulariser term) — multiplied in by rat_logit_bwd / rat_l2_reg, never read on the host.
        table_lists (dense modes only): leave the table block of the gradient buffer untouched (zero) and keep this backward's
        table gradients as (unique rows, gradient rows, count) lists in `_table_lists` — what _exchange_gradients ships under data
        parallelism instead of the dense table block."""
        c, lib = self._cfg, self._lib
        idx, labels, y_true = saved["batch"]
        B, T, L, S = saved["dims"]
        d, F, H, heads, dh = c["d"], c["nf"], c["hidden"], c["heads"], c["dh"]
        dev = self._flat.device
        gflat, gftab, lr_gftab = self._grad_buffer()
        mode = self._grad_mode
        as_lists = bool(table_lists) and mode != "sparse" and self._lists_possible
        if as_lists:
            mode = "lists"
        if mode != "atomic":
            lr_gftab = None                                     # the LR rows come from the sorted reduction below, not from atomics
            if self._col2field is None or self._col2field.numel() != L:
                self._col2field = ops.col2field_table(self._fields, L, dev)
        G = lambda name: self._gflat_view(gflat, name)          # noqa: E731
        x_final, y_pred = saved["x_final"], saved["y_pred"]
        g_loss = g_loss.reshape(1).to(torch.float32).contiguous()
        # ---- head
        cs = saved["cls_stride"]
        # the head's gradient lives on ONE token row per sample.  Where the last block's feed-forward backward can take it as compact
        # rows (rat_ffn_bwd_res_rows) the [B][T][S][d] zero grid that used to carry it is never filled nor read.
        dy_period = 0
        if (not saved.get("pruned") and cs == T * S * d and len(saved["blocks"]) > 0 and self._head_strips
                and type(self)._encoder_backward is RAT_m2._encoder_backward and ops.ffn_bwd_rows_supported(d, H, self.arith, lib)):
            dy_period = T * S
            dx = torch.empty((B, d), dtype=torch.float32, device=dev)
        else:
            dx = torch.zeros_like(x_final)
        dcs = d if dy_period else cs
        # the DNN's one-output Linear (deep.py:135-137) rides along with its neighbours when the last hidden layer runs as a column
        # strip and has no Dropout behind it: bias gradient from rat_logit_bwd_dnn, weight gradient and the outer product da = dlogit W
        # inside the last hidden layer's strip launch (rat_bn_act_bwd_strip_outer)
        seeds = saved["seeds"]
        fold = False
        if self.dnn is not None and self._head_strips and saved["dnn"]:
            (_, bn_l, pdrop_l), (_, _, _, z_l, _, sm_l, _) = self._dnn_layers[-1], saved["dnn"][-1]
            fold = (not (seeds is not None and pdrop_l > 0) and not (bn_l is not None and isinstance(sm_l, tuple))
                    and B <= self._strip_bwd_rows and ops.bn_strip_ok(B, z_l.shape[1], lib))
        pre_out = "dnn.dnn.%d." % self._dnn_out if self.dnn is not None else None
        dlogit = ops.logit_bwd(y_pred, y_true, x_final, cs, self.fc.weight.data, dx, dcs, G("fc.weight"),
                               G("fc.bias"), lr_gftab, F, idx, T * L, 1.0, B, d, gscale_dev=g_loss, head=self._head,
                               ddnn_b=G(pre_out + "bias") if fold else None, lib=lib)
        dflat = None
        if self.dnn is not None:
            mods = self.dnn.dnn
            pre = pre_out
            a_prev, lda, K = saved["dnn_last"]
            W = mods[self._dnn_out].weight.data
            da = None
            if not fold:
                ops.sgemm(1, 0, 1, K, B, dlogit, 1, a_prev, lda, G(pre + "weight"), K, arith=self.gemm_arith, lib=lib)       # dW = dlogit^T a
                ops.colsum(dlogit, 1, G(pre + "bias"), B, 1, lib=lib)
                da = torch.empty((B, K), dtype=torch.float32, device=dev)
                ops.sgemm(0, 0, B, K, 1, dlogit, 1, W, K, da, K, arith=self.gemm_arith, lib=lib)                             # da = dlogit W
            for li, ((lin, bn, pdrop), (a_in, lda_in, K_in, z, a, sm, sr)) in reversed(list(enumerate(zip(self._dnn_layers, saved["dnn"])))):
                N = z.shape[1]
                if seeds is not None and pdrop > 0:
                    da = ops.dropout(da, pdrop, seeds[1 + li], out=da, lib=lib)
                pre = "dnn.dnn.%d." % lin
                strip = (self._head_strips and B <= self._strip_bwd_rows and not (bn is not None and isinstance(sm, tuple))
                         and ops.bn_strip_ok(B, N, lib))
                if da is None:                              # fold: the last hidden layer
                    m = mods[bn] if bn is not None else None
                    dz = ops.bn_act_bwd_strip_outer(z, a, dlogit, W, G(pre_out + "weight"), m.weight.data if m is not None else None,
                                                    sm, sr, G("dnn.dnn.%d.weight" % bn) if m is not None else None,
                                                    G("dnn.dnn.%d.bias" % bn) if m is not None else None, G(pre + "bias"),
                                                    m is not None, act=self._dnn_acts[li], lib=lib)
                elif bn is not None and isinstance(sm, tuple):                       # SyncBN (see _run_forward)
                    m = mods[bn]
                    dz = ops.bn_relu_bwd_sync(z, a, da, m.weight.data, sm[0], sr, G("dnn.dnn.%d.weight" % bn),
                                              G("dnn.dnn.%d.bias" % bn), self._all_reduce_sum, sm[1], act=self._dnn_acts[li], lib=lib)
                elif strip:                                 # one launch: [BatchNorm backward +] activation backward + the Linear's bias gradient
                    m = mods[bn] if bn is not None else None
                    dz = ops.bn_act_bwd_strip(z, a, da, m.weight.data if m is not None else None, sm, sr,
                                              G("dnn.dnn.%d.weight" % bn) if m is not None else None,
                                              G("dnn.dnn.%d.bias" % bn) if m is not None else None, G(pre + "bias"), m is not None,
                                              act=self._dnn_acts[li], lib=lib)
                elif bn is not None:
                    m = mods[bn]
                    dz = ops.bn_relu_bwd(z, a, da, m.weight.data, sm, sr, G("dnn.dnn.%d.weight" % bn), G("dnn.dnn.%d.bias" % bn),
                                         True, act=self._dnn_acts[li], lib=lib)
                else:
                    dz = ops.bn_relu_bwd(z, a, da, None, None, None, None, None, False, act=self._dnn_acts[li], lib=lib)
                ops.sgemm(1, 0, N, K_in, B, dz, N, a_in, lda_in, G(pre + "weight"), K_in, arith=self.gemm_arith, lib=lib)  # dW = dz^T a_in
                if not strip:
                    ops.colsum(dz, N, G(pre + "bias"), B, N, lib=lib)
                da = torch.empty((B, K_in), dtype=torch.float32, device=dev)
                ops.sgemm(0, 0, B, K_in, N, dz, N, mods[lin].weight.data, K_in, da, K_in, arith=self.gemm_arith, lib=lib)   # da_in = dz W
            dflat = da                                                                             # [B, F*d]
        # ---- encoder, reversed
        dx = self._encoder_backward(saved, dx, G, dy_period=dy_period) if dy_period else self._encoder_backward(saved, dx, G)
        # every dense-net gradient (encoder, DNN, fc) is final now: add its regulariser term and, under data parallelism, start
        # its all-reduce — it travels over xGMI while the embedding-table gradients below are still being produced
        n_dense0 = self._n_emb - self._n_sparse
        if g_reg is not None and c["lam_net"] > 0:
            g_reg_dev = g_reg.reshape(1).to(torch.float32).contiguous()
            ops.l2_reg(self._flat[self._n_emb:], gflat[n_dense0:], c["lam_net"], None, lam_scale_dev=g_reg_dev, lib=lib)
        if self._dp() and n_dense0 < gflat.numel() and self._gbuf is not None and gflat is self._gbuf[0]:
            # (a fresh buffer means some p.grad was still held — gradient accumulation: autograd will ADD this buffer into p.grad,
            # so nothing is reduced early; _exchange_gradients falls back to one all-reduce of the accumulated gradients)
            import torch.distributed as dist
            handle, part = [None], gflat[n_dense0:]

            if self._staged(part):
                class _Done:
                    def wait(self):
                        return True

                def start():
                    h = part.detach().cpu()
                    dist.all_reduce(h, op=dist.ReduceOp.SUM)
                    part.copy_(h)
                    handle[0] = _Done()
            else:
                def start():
                    handle[0] = dist.all_reduce(part, op=dist.ReduceOp.SUM, async_op=True)
            self._collective(start)
            self._pending_reduce = (handle, gflat)
        if saved["seeds"] is not None and c["emb_dropout"] > 0:
            dx = ops.dropout(dx, c["emb_dropout"], saved["seeds"][0], out=dx, lib=lib)
        # ---- embedding tables
        if mode == "atomic":
            ops.gather_bwd(dx, dflat, idx, labels, gftab, F, G("label_embedding_layer.weight"), B, T, L, d, lib=lib)
            self._sparse = None
        else:
            self._table_gradients_sorted(dx, dflat, dlogit, idx, labels, gflat, (B, T, L, S), mode)
            ops.label_grad(dx, labels, G("label_embedding_layer.weight"), B * T, S, d, lib=lib)
            if as_lists:
                self._table_lists, self._sparse = self._sparse, None
        # ---- L2 regulariser gradient (base_model.py:79-94): lambda * W on the "embedding_layer" tensors
        if g_reg is not None:
            g_reg = g_reg.reshape(1).to(torch.float32).contiguous()
            if c["lam_emb"] > 0 and self._n_emb > 0:
                ops.l2_reg(self._flat[:self._n_emb], gflat[:self._n_emb], c["lam_emb"], None, lam_scale_dev=g_reg, lib=lib)
        self._last_gflat = gflat
        return [self._gflat_view(gflat, n) if self._offsets[n] >= self._n_sparse else None for n in self._order]

    def _table_gradients_sorted(self, dx, dflat, dlogit, idx, labels, gflat, dims, mode):
        """K1s: stable sort of the batch's (sample, id column) pairs by table row + segmented reduction in batch order.
        mode "sorted": sums land in the dense gradient tables (bit-reproducible replacement of the fp32 atomics);
        mode "sparse": (unique rows, gradient rows, count) lists per table family, consumed by rat_adam_rows."""
        c, lib = self._cfg, self._lib
        B, T, L, S = dims
        d, F = c["d"], c["nf"]
        dev = dx.device
        rows_feat = self._n_feat // d
        # (under the owner exchange the plans were built at the start of the step: _owner_prepare)
        st = self.__dict__.get("_owner_state")
        plan, plan_lr = st["plans"] if st is not None else self._build_plans(idx, dims)
        sparse = []
        if mode == "sorted":
            ops.sparse_reduce_grid(plan, dx, dflat, self._col2field, B, T, L, F, d, dense_base=gflat, lib=lib)
        else:
            cap = min(B * T * L, rows_feat)
            rows = torch.empty(cap, dtype=torch.int32, device=dev)
            grads = torch.empty((cap, d), dtype=torch.float32, device=dev)
            ops.sparse_reduce_grid(plan, dx, dflat, self._col2field, B, T, L, F, d, out_rows=rows, out_grads=grads, lib=lib)
            sparse.append((rows, grads, plan.count.clone(), d, rows_feat, 0))
        if c["use_wide"]:                                       # LR tables: width-1 rows, gradient = dlogit of the TARGET sample
            rows_lr = self._n_tab - self._n_feat
            if mode == "sorted":
                ops.sparse_reduce_scalar(plan_lr, dlogit, B, L, dense_base=gflat[self._n_feat:], lib=lib)
            else:
                cap = min(B * L, rows_lr)
                rows = torch.empty(cap, dtype=torch.int32, device=dev)
                vals = torch.empty((cap, 1), dtype=torch.float32, device=dev)
                ops.sparse_reduce_scalar(plan_lr, dlogit, B, L, out_rows=rows, out_vals=vals, lib=lib)
                sparse.append((rows, vals, plan_lr.count.clone(), 1, rows_lr, self._n_feat))
        self._sparse = sparse if mode in ("sparse", "lists") else None
        self._sparse_is_global = False
