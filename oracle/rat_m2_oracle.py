"""CPU oracle for the RAT_m2 hot path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch restatement (plain torch tensor math on the CPU, no
``nn.Module`` forward calls, no einops, no fuxictr import) of the arithmetic the
reference performs on the path BASELINE.json's north_star names.  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import it; the product package (``www24-rat_amd/rat_amd``) never does.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real
reference from /root/reference (build container only), runs forward / loss /
backward / clip / Adam on seeded inputs and commits inputs + expected outputs
as ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks every function
here against those vectors.

Every function cites the reference lines it restates (paths relative to
/root/reference).

All functions take the model's weights as a flat ``dict`` keyed by the
reference's ``state_dict`` names and are dtype-generic (float32 for parity with
the reference, float64 for an "exact" comparator of the HIP kernels).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import torch

Tensor = torch.Tensor

EMB = "embedding_layer.embedding_layer.embedding_layer."
LR = "lr_layer.embedding_layer.embedding_layer.embedding_layer."


# --------------------------------------------------------------------------- config
@dataclass
class Field:
    """One entry of FeatureMap.feature_specs (fuxictr/features.py:46-57)."""
    name: str
    kind: str                 # "categorical" | "sequence"
    vocab_size: int
    index: object             # int, or list of column ids for a sequence field
    padding_idx: Optional[int] = None

    @property
    def columns(self) -> List[int]:
        return list(self.index) if isinstance(self.index, (list, tuple)) else [int(self.index)]


@dataclass
class Config:
    """Constructor kwargs that change arithmetic (fuxictr/pytorch/models/RAT_m2.py:29-56)."""
    fields: List[Field]
    embedding_dim: int = 10
    num_heads: int = 1
    dim_head: int = 10
    depth: int = 4
    scale_dim: int = 4
    dnn_hidden_units: Sequence[int] = (64, 64, 64)
    batch_norm: bool = False
    use_wide: bool = False
    embedding_regularizer: float = 0.0
    net_regularizer: float = 0.0
    learning_rate: float = 1e-3
    max_gradient_norm: float = 10.0
    bn_eps: float = 1e-5
    bn_momentum: float = 0.1
    ln_eps: float = 1e-5
    dnn_activations: object = "relu"    # str or per-layer list (MLP_Layer hidden_activations, deep.py:108-141; torch_utils.get_activation)
    optimizer: str = "adam"             # torch_utils.get_optimizer (torch_utils.py:41-49): "adam" or a torch.optim class name
    task: str = "binary_classification"   # "regression": no output activation (base_model.py:286-292), loss F.mse_loss
    variant: str = "m2"       # "m2": cross/intra encoder blocks (RAT_m2.py); "m1": cascaded transformers (RAT_m1.py);
                              # "m0": ONE transformer over all T*S tokens of a sample (RAT_m0.py);
                              # "m3": parallel intra/cross attention with a shared query projection, mean fusion (RAT_m3.py)

    @property
    def num_fields(self) -> int:
        return len(self.fields)

    @property
    def inner_dim(self) -> int:
        return self.num_heads * self.dim_head

    @property
    def hidden_dim(self) -> int:
        return self.embedding_dim * self.scale_dim


def fields_from_specs(feature_specs) -> List[Field]:
    """Turn a feature_map.json style ``feature_specs`` mapping into Field records.

    Sequence fields pad with ``vocab_size - 1`` (fuxictr/pytorch/layers/embedding.py:90-93);
    categorical fields may carry an explicit ``padding_idx`` (embedding.py:78-81).
    """
    out = []
    for name, spec in feature_specs.items():
        kind = spec["type"]
        if kind == "sequence":
            pad = spec["vocab_size"] - 1
        elif kind == "categorical":
            pad = spec.get("padding_idx", None)
        else:
            raise NotImplementedError("feature type %r is outside the RAT_m2 hot path" % kind)
        out.append(Field(name, kind, int(spec["vocab_size"]), spec["index"], pad))
    return out


# --------------------------------------------------------------------------- parameter inventory
def parameter_shapes(cfg: Config) -> "OrderedDict[str, Tuple[int, ...]]":
    """Shapes of every trainable tensor, in the reference's registration order.

    RAT_m2.__init__ (RAT_m2.py:63-98): embedding tables, label table, the dead
    ``query_proj``, ``depth`` encoder blocks (cross_attention registered before
    intra_attention, RAT_m2.py:210-217), LR tables, DNN, fc.
    """
    d, inner, hid = cfg.embedding_dim, cfg.inner_dim, cfg.hidden_dim
    nf = cfg.num_fields
    shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    for f in cfg.fields:
        shapes[EMB + f.name + ".weight"] = (f.vocab_size, d)
    shapes["label_embedding_layer.weight"] = (3, d)
    shapes["query_proj.weight"] = (d * nf, d * nf)
    shapes["query_proj.bias"] = (d * nf,)
    def attn_shapes(p):
        shapes[p + "norm.weight"] = (d,)
        shapes[p + "norm.bias"] = (d,)
        shapes[p + "fn.to_qkv.weight"] = (3 * inner, d)
        if not (cfg.num_heads == 1 and cfg.dim_head == d):       # project_out, RAT_m2.py:180 / RAT_m1.py:166
            shapes[p + "fn.to_out.0.weight"] = (d, inner)
            shapes[p + "fn.to_out.0.bias"] = (d,)

    def mlp_shapes(p):
        shapes[p + "0.weight"] = (hid, d)
        shapes[p + "0.bias"] = (hid,)
        shapes[p + "3.weight"] = (d, hid)
        shapes[p + "3.bias"] = (d,)

    if cfg.variant in ("m1", "m0"):
        # RAT_m1.__init__ (RAT_m1.py:71-72): intra_transformer, cross_transformer; RAT_m0.__init__ (RAT_m0.py:70): encoder.
        # Transformer registers `layers` before `norm` (RAT_m1.py:194-203): layers.i.0 = PreNorm(Attention),
        # layers.i.1 = PreNorm(FeedForward)
        for t in (("intra_transformer.", "cross_transformer.") if cfg.variant == "m1" else ("encoder.",)):
            for i in range(cfg.depth):
                attn_shapes(t + "layers.%d.0." % i)
                shapes[t + "layers.%d.1.norm.weight" % i] = (d,)
                shapes[t + "layers.%d.1.norm.bias" % i] = (d,)
                mlp_shapes(t + "layers.%d.1.fn.net." % i)
            shapes[t + "norm.weight"] = (d,)
            shapes[t + "norm.bias"] = (d,)
    elif cfg.variant == "m3":
        # CrossIntraEncoderBlock of RAT_m3 (RAT_m3.py:196-213): five bias-free projections owned by the block (W_q is shared
        # by both attentions), then intra_attention BEFORE cross_attention (each: norm + to_out), then the MLP
        for i in range(cfg.depth):
            p = "encoder.encoder.%d." % i
            for name in ("W_q", "W_k_s", "W_v_s", "W_k_t", "W_v_t"):
                shapes[p + name + ".weight"] = (inner, d)
            for which in ("intra_attention.", "cross_attention."):
                shapes[p + which + "norm.weight"] = (d,)
                shapes[p + which + "norm.bias"] = (d,)
                if not (cfg.num_heads == 1 and cfg.dim_head == d):
                    shapes[p + which + "fn.to_out.0.weight"] = (d, inner)
                    shapes[p + which + "fn.to_out.0.bias"] = (d,)
            mlp_shapes(p + "mlp.net.")
    else:
        for i in range(cfg.depth):
            for which in ("cross_attention", "intra_attention"):
                attn_shapes("encoder.encoder.%d.%s." % (i, which))
            mlp_shapes("encoder.encoder.%d.mlp.net." % i)
    if cfg.use_wide:
        for f in cfg.fields:
            shapes[LR + f.name + ".weight"] = (f.vocab_size, 1)
    if cfg.dnn_hidden_units:
        widths = [d * nf] + list(cfg.dnn_hidden_units)
        pos = 0
        acts = activation_names(cfg)
        for j in range(len(widths) - 1):
            shapes["dnn.dnn.%d.weight" % pos] = (widths[j + 1], widths[j])
            shapes["dnn.dnn.%d.bias" % pos] = (widths[j + 1],)
            pos += 1
            if cfg.batch_norm:
                shapes["dnn.dnn.%d.weight" % pos] = (widths[j + 1],)
                shapes["dnn.dnn.%d.bias" % pos] = (widths[j + 1],)
                pos += 1
            if acts[j] is not None:
                pos += 1                               # the activation module (deep.py:121-123: only when the entry is truthy)
        shapes["dnn.dnn.%d.weight" % pos] = (1, widths[-1])
        shapes["dnn.dnn.%d.bias" % pos] = (1,)
    shapes["fc.weight"] = (1, d)
    shapes["fc.bias"] = (1,)
    return shapes


def count_parameters(cfg: Config) -> int:
    """BaseModel.count_parameters (base_model.py:294-301): every requires_grad tensor."""
    total = 0
    for shp in parameter_shapes(cfg).values():
        n = 1
        for s in shp:
            n *= s
        total += n
    return total


def activation_names(cfg: Config) -> List[Optional[str]]:
    """per hidden layer: None (no activation module) or the lower-cased activation name"""
    acts = cfg.dnn_activations
    acts = list(acts) if isinstance(acts, (list, tuple)) else [acts] * len(cfg.dnn_hidden_units)
    return [(a.lower() if isinstance(a, str) and a else None) for a in acts]


def apply_activation(z: Tensor, name: Optional[str]) -> Tensor:
    """torch_utils.get_activation (torch_utils.py:83-94) -> the nn module's function (default hyper-parameters)"""
    if name is None or name == "identity":
        return z
    if name == "relu":
        return torch.relu(z)
    if name == "sigmoid":
        return torch.sigmoid(z)
    if name == "tanh":
        return torch.tanh(z)
    if name == "leakyrelu":
        return torch.nn.functional.leaky_relu(z, 0.01)
    if name == "elu":
        return torch.nn.functional.elu(z, 1.0)
    raise NotImplementedError(name)


def dnn_layout(cfg: Config):
    """Indices inside ``dnn.dnn`` of (linear, bn or None) per hidden layer, plus the output linear (MLP_Layer appends
    Linear, [BatchNorm1d], [activation module], [Dropout: p = 0 here] per layer, deep.py:117-125)."""
    layers = []
    pos = 0
    for act in activation_names(cfg):
        lin = pos
        pos += 1
        bn = None
        if cfg.batch_norm:
            bn = pos
            pos += 1
        if act is not None:
            pos += 1
        layers.append((lin, bn))
    return layers, pos


# --------------------------------------------------------------------------- building blocks
def embed_fields(X_long: Tensor, cfg: Config, w: Dict[str, Tensor], prefix: str = EMB) -> Tensor:
    """EmbeddingDictLayer.forward + dict2tensor (embedding.py:158-178,138-156).

    X_long: [..., L] integer column ids.  Returns [..., F, width].  Sequence
    fields are looked up per position and summed (MaskedSumPooling,
    sequence.py:32-38); the padding row of their table is all-zero by
    construction so the sum ignores padding.
    """
    per_field = []
    for f in cfg.fields:
        table = w[prefix + f.name + ".weight"]
        cols = f.columns
        ids = X_long[..., cols[0]] if f.kind == "categorical" else X_long[..., cols]
        rows = table[ids]
        if f.padding_idx is not None:
            # nn.Embedding(padding_idx=...) never sends gradient to the padding row (embedding.py:78-93)
            rows = torch.where((ids == f.padding_idx).unsqueeze(-1), rows.detach(), rows)
        per_field.append(rows if f.kind == "categorical" else rows.sum(dim=-2))
    return torch.stack(per_field, dim=-2)


def layer_norm(x: Tensor, gamma: Tensor, beta: Tensor, eps: float) -> Tensor:
    """nn.LayerNorm(d) inside PreNorm (RAT_m2.py:155-161): biased variance over the last dim."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * gamma + beta


def gelu_erf(x: Tensor) -> Tensor:
    """nn.GELU() default = exact erf form (RAT_m2.py:168)."""
    return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))


def attention(x: Tensor, w: Dict[str, Tensor], prefix: str, cfg: Config, w_qkv: Optional[Tensor] = None,
              heads: Optional[int] = None, dim_head: Optional[int] = None, scale: Optional[float] = None) -> Tensor:
    """PreNorm(Attention) (RAT_m2.py:155-161,176-202) on x: [N, L, d] -> [N, L, d] (no residual).
    The optional arguments serve RAT_m3 (RAT_m3.py:164-189): separate W_q / W_k / W_v Linear layers (passed stacked as
    ``w_qkv``), heads/2 heads of twice the width, and the softmax scale of the CONFIGURED dim_head."""
    h, dh = heads or cfg.num_heads, dim_head or cfg.dim_head
    scale = dh ** -0.5 if scale is None else scale
    n, l, _ = x.shape
    xn = layer_norm(x, w[prefix + "norm.weight"], w[prefix + "norm.bias"], cfg.ln_eps)
    qkv = xn @ (w[prefix + "fn.to_qkv.weight"] if w_qkv is None else w_qkv).t()   # [N, L, 3*h*dh]
    q, k, v = qkv.split(h * dh, dim=-1)
    q = q.reshape(n, l, h, dh).permute(0, 2, 1, 3)                     # [N, h, L, dh]
    k = k.reshape(n, l, h, dh).permute(0, 2, 1, 3)
    v = v.reshape(n, l, h, dh).permute(0, 2, 1, 3)
    scores = (q @ k.transpose(-1, -2)) * scale
    scores = scores - scores.max(dim=-1, keepdim=True).values
    p = torch.exp(scores)
    p = p / p.sum(dim=-1, keepdim=True)
    o = (p @ v).permute(0, 2, 1, 3).reshape(n, l, h * dh)
    if (prefix + "fn.to_out.0.weight") in w:
        o = o @ w[prefix + "fn.to_out.0.weight"].t() + w[prefix + "fn.to_out.0.bias"]
    return o


def feed_forward(x: Tensor, w: Dict[str, Tensor], prefix: str) -> Tensor:
    """FeedForward (RAT_m2.py:163-174): Linear -> GELU -> Linear (both dropouts are p=0)."""
    hdn = gelu_erf(x @ w[prefix + "0.weight"].t() + w[prefix + "0.bias"])
    return hdn @ w[prefix + "3.weight"].t() + w[prefix + "3.bias"]


def encoder_block(x: Tensor, w: Dict[str, Tensor], i: int, cfg: Config) -> Tensor:
    """CrossIntraEncoderBlock.forward (RAT_m2.py:219-236) on the [B, T, S, d] grid."""
    b, t, s, d = x.shape
    p = "encoder.encoder.%d." % i
    xi = x.reshape(b * t, s, d)
    xi = attention(xi, w, p + "intra_attention.", cfg) + xi            # over the S field tokens
    xc = xi.reshape(b, t, s, d).transpose(1, 2).reshape(b * s, t, d)
    xc = attention(xc, w, p + "cross_attention.", cfg) + xc            # over the T samples
    xc = feed_forward(xc, w, p + "mlp.net.") + xc                      # no norm before the MLP
    return xc.reshape(b, s, t, d).transpose(1, 2)


def m3_state_aliases(cfg: Config) -> "OrderedDict[str, str]":
    """state_dict-only names of RAT_m3: the shared projection modules are also registered inside both Attention modules
    (RAT_m3.py:168-170,204-209), so ``state_dict()`` lists them again under those paths.  alias -> owning parameter, in
    state_dict order (``load_state_dict`` copies in that order, so the LAST alias of a tensor decides its loaded value)."""
    out: "OrderedDict[str, str]" = OrderedDict()
    for i in range(cfg.depth):
        p = "encoder.encoder.%d." % i
        for which, (k, v) in (("intra_attention.", ("W_k_s", "W_v_s")), ("cross_attention.", ("W_k_t", "W_v_t"))):
            out[p + which + "fn.W_q.weight"] = p + "W_q.weight"
            out[p + which + "fn.W_k.weight"] = p + k + ".weight"
            out[p + which + "fn.W_v.weight"] = p + v + ".weight"
    return out


def encoder_block_m3(x: Tensor, w: Dict[str, Tensor], i: int, cfg: Config) -> Tensor:
    """CrossIntraEncoderBlock.forward of RAT_m3 (RAT_m3.py:215-243): intra and cross attention both read the block INPUT
    (no residual of their own), share W_q, use heads/2 heads of width 2*dim_head with the softmax scale dim_head^-0.5
    (RAT_m3.py:172-189); their mean goes through the MLP, whose residual is the block input."""
    b, t, s, d = x.shape
    p = "encoder.encoder.%d." % i
    h = int(cfg.num_heads / 2)
    dh = cfg.inner_dim // h
    scale = cfg.dim_head ** -0.5
    wq = w[p + "W_q.weight"]
    w_s = torch.cat([wq, w[p + "W_k_s.weight"], w[p + "W_v_s.weight"]], dim=0)
    w_t = torch.cat([wq, w[p + "W_k_t.weight"], w[p + "W_v_t.weight"]], dim=0)
    out_s = attention(x.reshape(b * t, s, d), w, p + "intra_attention.", cfg, w_qkv=w_s, heads=h, dim_head=dh, scale=scale)
    out_s = out_s.reshape(b, t, s, d)
    xc = x.transpose(1, 2).reshape(b * s, t, d)
    out_t = attention(xc, w, p + "cross_attention.", cfg, w_qkv=w_t, heads=h, dim_head=dh, scale=scale)
    out_t = out_t.reshape(b, s, t, d).transpose(1, 2)
    out = torch.cat((out_s.unsqueeze(1), out_t.unsqueeze(1)), dim=1).mean(dim=1)
    return feed_forward(out, w, p + "mlp.net.") + x


def transformer(x: Tensor, w: Dict[str, Tensor], prefix: str, cfg: Config) -> Tensor:
    """RAT_m1's Transformer.forward (RAT_m1.py:205-209) on x: [N, L, d]: depth x (PreNorm attention + residual,
    PreNorm feed-forward + residual), then the final LayerNorm."""
    for i in range(cfg.depth):
        p = prefix + "layers.%d." % i
        x = attention(x, w, p + "0.", cfg) + x
        xn = layer_norm(x, w[p + "1.norm.weight"], w[p + "1.norm.bias"], cfg.ln_eps)
        x = feed_forward(xn, w, p + "1.fn.net.") + x
    return layer_norm(x, w[prefix + "norm.weight"], w[prefix + "norm.bias"], cfg.ln_eps)


def dnn_head(flat: Tensor, w: Dict[str, Tensor], cfg: Config, training: bool,
             bn_state: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """MLP_Layer.forward (deep.py:108-141) with ReLU hidden activations.

    BatchNorm1d: training uses biased batch variance for normalisation and
    updates running stats with the unbiased one (momentum 0.1); eval uses the
    running stats.  ``bn_state`` (if given) receives the updated buffers.
    """
    layers, out_pos = dnn_layout(cfg)
    acts = activation_names(cfg)
    z = flat
    for (lin, bn), act in zip(layers, acts):
        z = z @ w["dnn.dnn.%d.weight" % lin].t() + w["dnn.dnn.%d.bias" % lin]
        if bn is not None:
            g, bta = w["dnn.dnn.%d.weight" % bn], w["dnn.dnn.%d.bias" % bn]
            if training:
                mu = z.mean(dim=0)
                var = ((z - mu) ** 2).mean(dim=0)
                if bn_state is not None:
                    n = z.shape[0]
                    rm = w["dnn.dnn.%d.running_mean" % bn]
                    rv = w["dnn.dnn.%d.running_var" % bn]
                    bn_state["dnn.dnn.%d.running_mean" % bn] = \
                        ((1 - cfg.bn_momentum) * rm + cfg.bn_momentum * mu).detach()
                    bn_state["dnn.dnn.%d.running_var" % bn] = \
                        ((1 - cfg.bn_momentum) * rv + cfg.bn_momentum * var * n / max(n - 1, 1)).detach()
                    bn_state["dnn.dnn.%d.num_batches_tracked" % bn] = \
                        w["dnn.dnn.%d.num_batches_tracked" % bn] + 1
            else:
                mu = w["dnn.dnn.%d.running_mean" % bn]
                var = w["dnn.dnn.%d.running_var" % bn]
            z = (z - mu) / torch.sqrt(var + cfg.bn_eps) * g + bta
        z = apply_activation(z, act)
    return z @ w["dnn.dnn.%d.weight" % out_pos].t() + w["dnn.dnn.%d.bias" % out_pos]


# --------------------------------------------------------------------------- forward
def build_grid(X: Tensor, y: Tensor, w: Dict[str, Tensor], cfg: Config) -> Tuple[Tensor, Tensor]:
    """Token grid assembly (RAT_m2.py:113-126).

    X: [B, T, L] integer-valued, y: [B, T].  Row t=0 is the target.  Returns
    (grid [B, T, S, d] with token order [label, f0..], target field embeddings [B, F, d]).
    The target's label token is id 2; retrieved samples use their own label {0,1}.
    """
    Xl = X.long()
    label_ids = y.long().clone()
    label_ids[:, 0] = 2
    fields = embed_fields(Xl, cfg, w)                                   # [B, T, F, d]
    labels = w["label_embedding_layer.weight"][label_ids].unsqueeze(2)  # [B, T, 1, d]
    return torch.cat([labels, fields], dim=2), fields[:, 0]


def forward(w: Dict[str, Tensor], X: Tensor, y: Tensor, cfg: Config, training: bool = False,
            bn_state: Optional[Dict[str, Tensor]] = None, return_logit: bool = False):
    """RAT_m2.forward (RAT_m2.py:104-152) with dropout p=0.  Returns y_pred [B, 1]."""
    grid, target_fields = build_grid(X, y, w, cfg)
    if cfg.variant == "m0":
        # RAT_m0.forward (RAT_m0.py:121-127): one joint sequence of T*S tokens per sample; [:, :, 0][:, 0] = token (t=0, n=0)
        b, t, s, d = grid.shape
        cls = transformer(grid.reshape(b, t * s, d), w, "encoder.", cfg)[:, 0]
    elif cfg.variant == "m1":
        # RAT_m1.forward (RAT_m1.py:121-130): every sample's S tokens through the intra transformer, its label token
        # [:, 0] becomes that sample's vector; the T sample vectors go through the cross transformer; target = [:, 0]
        b, t, s, d = grid.shape
        xi = transformer(grid.reshape(b * t, s, d), w, "intra_transformer.", cfg)
        xc = transformer(xi[:, 0].reshape(b, t, d), w, "cross_transformer.", cfg)
        cls = xc[:, 0]
    else:
        x = grid
        for i in range(cfg.depth):
            x = encoder_block_m3(x, w, i, cfg) if cfg.variant == "m3" else encoder_block(x, w, i, cfg)
        cls = x[:, 0, 0]                                                # target sample, label token
    logit = cls @ w["fc.weight"].t() + w["fc.bias"]
    if cfg.dnn_hidden_units:
        logit = logit + dnn_head(target_fields.flatten(1), w, cfg, training, bn_state)
    if cfg.use_wide:
        # LR_Layer (shallow.py:36-45): width-1 tables, sum over fields; the mean over
        # dim 1 is over a singleton because X is passed as [B, 1, F] (RAT_m2.py:119,148).
        lr = embed_fields(X[:, :1].long(), cfg, w, prefix=LR).sum(dim=-2).mean(dim=1)
        logit = logit + lr
    if return_logit or cfg.task == "regression":            # get_output_activation: None for regression (base_model.py:286-292)
        return logit
    return torch.sigmoid(logit)


def bce_mean(y_pred: Tensor, y_true: Tensor) -> Tensor:
    """F.binary_cross_entropy(reduction='mean') (torch_utils.py:51-63); log clamped at -100 like torch."""
    lp = torch.clamp(torch.log(y_pred), min=-100.0)
    l1p = torch.clamp(torch.log(1.0 - y_pred), min=-100.0)
    return -(y_true * lp + (1.0 - y_true) * l1p).mean()


def regularization(w: Dict[str, Tensor], cfg: Config, trainable: Sequence[str]) -> Tensor:
    """BaseModel.add_regularization (base_model.py:79-94) with the float -> l2 rule of
    get_regularizer (torch_utils.py:65-81): (lambda/2)*||W||_2^2 for every trainable tensor whose
    name contains "embedding_layer" (feature tables, LR tables AND the label table), net lambda for the rest."""
    total = torch.zeros((), dtype=next(iter(w.values())).dtype)
    for name in trainable:
        lam = cfg.embedding_regularizer if "embedding_layer" in name else cfg.net_regularizer
        if lam:
            total = total + (lam / 2.0) * (w[name] ** 2).sum()
    return total


def total_loss(w: Dict[str, Tensor], X: Tensor, y: Tensor, cfg: Config, training: bool = True,
               bn_state: Optional[Dict[str, Tensor]] = None) -> Tuple[Tensor, Tensor]:
    """BaseModel.get_total_loss (base_model.py:97-99).  Returns (loss, y_pred)."""
    y_pred = forward(w, X, y, cfg, training=training, bn_state=bn_state)
    y_true = y[:, :1].to(y_pred.dtype)
    names = list(parameter_shapes(cfg).keys())
    data_loss = ((y_pred - y_true) ** 2).mean() if cfg.task == "regression" else bce_mean(y_pred, y_true)     # F.mse_loss / BCE
    return data_loss + regularization(w, cfg, names), y_pred


# --------------------------------------------------------------------------- backward / optimizer
def loss_and_grads(w: Dict[str, Tensor], X: Tensor, y: Tensor, cfg: Config, training: bool = True):
    """loss.backward() (base_model.py:223).  ``query_proj`` never enters the graph -> grad None.

    Returns (loss, y_pred, grads dict, bn_state dict)."""
    names = list(parameter_shapes(cfg).keys())
    leaves = {k: (v.detach().clone().requires_grad_(True) if k in names else v.detach()) for k, v in w.items()}
    bn_state: Dict[str, Tensor] = {}
    loss, y_pred = total_loss(leaves, X, y, cfg, training=training, bn_state=bn_state)
    used = [k for k in names if not k.startswith("query_proj")]
    gl = torch.autograd.grad(loss, [leaves[k] for k in used], allow_unused=True)
    grads = {k: g for k, g in zip(used, gl) if g is not None}
    return loss.detach(), y_pred.detach(), grads, bn_state


def clip_grad_norm(grads: Dict[str, Tensor], max_norm: float) -> Tuple[Dict[str, Tensor], Tensor]:
    """nn.utils.clip_grad_norm_ (base_model.py:224): scale = min(1, max_norm / (||g||_2 + 1e-6))."""
    total = torch.sqrt(sum((g.double() ** 2).sum() for g in grads.values())).to(next(iter(grads.values())).dtype)
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    return {k: g * coef for k, g in grads.items()}, total


def adam_step(w: Dict[str, Tensor], grads: Dict[str, Tensor], state: Dict[str, Dict[str, Tensor]],
              lr: float, step: int, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-8):
    """torch.optim.Adam defaults (torch_utils.py:41-49): no weight decay, no amsgrad.

    ``state[name] = {"m": ..., "v": ...}`` is created on first use; ``step`` is 1-based.
    Tensors without a gradient (``query_proj``) are left untouched, like torch does."""
    new_w = dict(w)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    for name, g in grads.items():
        st = state.setdefault(name, {"m": torch.zeros_like(g), "v": torch.zeros_like(g)})
        st["m"] = beta1 * st["m"] + (1 - beta1) * g
        st["v"] = beta2 * st["v"] + (1 - beta2) * g * g
        denom = torch.sqrt(st["v"]) / math.sqrt(bc2) + eps
        new_w[name] = w[name] - (lr / bc1) * st["m"] / denom
    return new_w


def other_optimizer_step(w: Dict[str, Tensor], grads: Dict[str, Tensor], state: Dict[str, Dict[str, Tensor]], lr: float, kind: str):
    """torch.optim.SGD / Adagrad / RMSprop as get_optimizer builds them — getattr(torch.optim, name)(params, lr=lr), torch_utils.py:41-49:
    torch's defaults (SGD without momentum; Adagrad lr_decay 0, eps 1e-10; RMSprop alpha 0.99, eps 1e-8, no momentum, not centered)."""
    new_w = dict(w)
    for name, g in grads.items():
        if kind == "SGD":
            new_w[name] = w[name] - lr * g
            continue
        st = state.setdefault(name, {"s": torch.zeros_like(g)})
        if kind == "Adagrad":
            st["s"] = st["s"] + g * g
            new_w[name] = w[name] - lr * g / (torch.sqrt(st["s"]) + 1e-10)
        elif kind == "RMSprop":
            st["s"] = 0.99 * st["s"] + (1 - 0.99) * g * g
            new_w[name] = w[name] - lr * g / (torch.sqrt(st["s"]) + 1e-8)
        else:
            raise NotImplementedError(kind)
    return new_w


def train_step(w: Dict[str, Tensor], X: Tensor, y: Tensor, cfg: Config, state: Dict, step: int):
    """One iteration of BaseModel.train_one_epoch (base_model.py:220-226).

    Returns (new weights incl. updated BN buffers, loss, y_pred, raw grads, grad norm)."""
    loss, y_pred, grads, bn_state = loss_and_grads(w, X, y, cfg, training=True)
    clipped, gnorm = clip_grad_norm(grads, cfg.max_gradient_norm)
    if cfg.optimizer.lower() == "adam":
        new_w = adam_step(w, clipped, state, cfg.learning_rate, step)
    else:
        new_w = other_optimizer_step(w, clipped, state, cfg.learning_rate, cfg.optimizer)
    new_w.update(bn_state)
    return new_w, loss, y_pred, grads, gnorm


# --------------------------------------------------------------------------- metrics
def auc(y_true, y_pred) -> float:
    """roc_auc_score (fuxictr/metrics.py:28) via the rank statistic with average ranks for ties."""
    import numpy as np
    y_true = np.asarray(y_true, dtype=np.float64).reshape(-1)
    y_pred = np.asarray(y_pred, dtype=np.float64).reshape(-1)
    order = np.argsort(y_pred, kind="mergesort")
    ranks = np.empty(len(y_pred), dtype=np.float64)
    sp = y_pred[order]
    i = 0
    while i < len(sp):
        j = i
        while j + 1 < len(sp) and sp[j + 1] == sp[i]:
            j += 1
        ranks[order[i:j + 1]] = 0.5 * (i + j) + 1.0
        i = j + 1
    npos = y_true.sum()
    nneg = len(y_true) - npos
    return float((ranks[y_true == 1].sum() - npos * (npos + 1) / 2.0) / (npos * nneg))


def logloss(y_true, y_pred, eps: float = 1e-7) -> float:
    """log_loss(y_true, y_pred, eps=1e-7) of the sklearn the reference pinned (metrics.py:26)."""
    import numpy as np
    y_true = np.asarray(y_true, dtype=np.float64).reshape(-1)
    p = np.clip(np.asarray(y_pred, dtype=np.float64).reshape(-1), eps, 1 - eps)
    return float(-(y_true * np.log(p) + (1 - y_true) * np.log(1 - p)).mean())
