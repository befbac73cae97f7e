"""The reference's other three model plugins behind the same kernels: RAT_m1 (cascaded intra / cross
transformers, RAT_m1.py), RAT_m3 (parallel intra / cross attention with a shared query projection and mean fusion, RAT_m3.py) and RAT_m0
(one transformer over the joint sequence of a sample, RAT_m0.py).  Split out of model.py in round 6 (no behaviour change): each class
overrides RAT_m2's encoder hooks — `_make_encoder` (the module tree that fixes the state_dict keys and the init order),
`_build_encoder_descriptors`, `_encoder_forward`, `_encoder_backward` — and inherits everything else (tables, head, loss, optimizer,
data parallelism, graphs)."""
import torch
from torch import nn

from . import ops
from .model import RAT_m2, _Attention, _FeedForward, _PreNorm  # noqa: F401


class _AttentionM3(nn.Module):
    """RAT_m3's Attention (RAT_m3.py:164-178): the three projections are modules OWNED BY THE BLOCK and only referenced here
    (so state_dict lists them again under fn.W_q / fn.W_k / fn.W_v); the output projection is its own."""

    def __init__(self, W_q, W_k, W_v, dim, inner_dim, heads, dim_head, dropout):
        super().__init__()
        self.inner_dim = inner_dim
        project_out = not (heads == 1 and dim_head == dim)
        self.W_q, self.W_k, self.W_v = W_q, W_k, W_v
        self.heads, self.scale = heads, dim_head ** -0.5
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim), nn.Dropout(dropout)) if project_out else nn.Identity()


class _BlockM3(nn.Module):
    """CrossIntraEncoderBlock of RAT_m3 (RAT_m3.py:191-213): registration order W_q, W_k_s, W_v_s, W_k_t, W_v_t,
    intra_attention, cross_attention, mlp."""

    def __init__(self, dim, heads, dim_head, dropout, hidden):
        super().__init__()
        self.inner_dim = heads * dim_head
        self.W_q = nn.Linear(dim, self.inner_dim, bias=False)
        self.W_k_s = nn.Linear(dim, self.inner_dim, bias=False)
        self.W_v_s = nn.Linear(dim, self.inner_dim, bias=False)
        self.W_k_t = nn.Linear(dim, self.inner_dim, bias=False)
        self.W_v_t = nn.Linear(dim, self.inner_dim, bias=False)
        self.intra_attention = _PreNorm(dim, _AttentionM3(self.W_q, self.W_k_s, self.W_v_s, dim, self.inner_dim, heads, dim_head, dropout))
        self.cross_attention = _PreNorm(dim, _AttentionM3(self.W_q, self.W_k_t, self.W_v_t, dim, self.inner_dim, heads, dim_head, dropout))
        self.mlp = _FeedForward(dim, hidden)


class _EncoderM3(nn.Module):
    def __init__(self, dim, heads, dim_head, dropout, depth, hidden):
        super().__init__()
        self.encoder = nn.ModuleList([_BlockM3(dim, heads, dim_head, dropout, hidden) for _ in range(depth)])


class _Transformer(nn.Module):
    """RAT_m1's Transformer (RAT_m1.py:194-203): `layers` is registered before `norm`; layers[i] = [PreNorm(Attention),
    PreNorm(FeedForward)]."""

    def __init__(self, dim, depth, heads, dim_head, hidden, dropout):
        super().__init__()
        self.layers = nn.ModuleList([])
        self.norm = nn.LayerNorm(dim)
        for _ in range(depth):
            self.layers.append(nn.ModuleList([_PreNorm(dim, _Attention(dim, heads, dim_head, dropout)),
                                              _PreNorm(dim, _FeedForward(dim, hidden))]))


class RAT_m1(RAT_m2):
    """RAT_m1 (fuxictr/pytorch/models/RAT_m1.py:24-130): the cascaded variant.  Every sample's S = F+1 tokens go through
    an intra `Transformer` (depth x [PreNorm attention + residual, PreNorm feed-forward + residual], final LayerNorm);
    the label token of each of the T samples is kept, the [B, T, d] result goes through a second, cross `Transformer`,
    and the target's row feeds `fc`.  Same constructor, batch layout, head, loss, optimizer and C-ABI as RAT_m2; the
    encoder is K2a (attention) + K2c (LayerNorm) + K2b with a separate residual (rat_ffn_fwd_res)."""

    def __init__(self, feature_map, model_id="RAT_m1", **kwargs):
        super().__init__(feature_map, model_id=model_id, **kwargs)

    def _make_encoder(self, d, num_heads, dim_head, dropout, depth, hidden):
        self.intra_transformer = _Transformer(d, depth, num_heads, dim_head, hidden, dropout)     # RAT_m1.py:71
        self.cross_transformer = _Transformer(d, depth, num_heads, dim_head, hidden, dropout)     # RAT_m1.py:72

    def _build_encoder_descriptors(self):
        self._stacks = {}
        for t in ("intra_transformer", "cross_transformer"):
            layers = []
            for i in range(self._cfg["depth"]):
                p = "%s.layers.%d." % (t, i)
                layers.append(dict(attn=self._attn_descriptor(p + "0."), ln=[p + "1.norm.weight", p + "1.norm.bias"],
                                   ffn=[p + "1.fn.net.0.weight", p + "1.fn.net.0.bias", p + "1.fn.net.3.weight", p + "1.fn.net.3.bias"]))
            self._stacks[t] = (layers, [t + ".norm.weight", t + ".norm.bias"])

    def _stack_forward(self, which, x, smap, ntok, cls_stride, ncls, save, saved):
        """One Transformer (RAT_m1.py:205-209) on `ntok` tokens; the final LayerNorm is applied to token 0 of every
        sequence only (the sole rows read afterwards, RAT_m1.py:125,128) -> [ncls, d]."""
        c, lib = self._cfg, self._lib
        d, H, heads, dh = c["d"], c["hidden"], c["heads"], c["dh"]
        layers, norm = self._stacks[which]
        rec = []
        for lay in layers:
            xa, att = self._attn_layer_forward(lay["attn"], x, smap, save)                            # attn(norm(x)) + x
            xn = ops.layernorm_fwd(xa, d, ntok, self._p(lay["ln"][0]), self._p(lay["ln"][1]), d, lib=lib)
            w1, b1, w2, b2 = [self._p(n) for n in lay["ffn"]]
            # FeedForward(dim, mlp_dim, dropout) carries two Dropout layers of the same rate (RAT_m1.py:151-161,202; RAT_m0.py:150-160,201)
            fdrop = (c["attn_dropout"], self._dropout_word(), self._dropout_word()) if (self.training and c["attn_dropout"] > 0) else None
            xb = ops.ffn_fwd_res(xn, xa, w1, b1, w2, b2, d, H, arith=self.arith, dropout=fdrop, lib=lib)   # ff(norm(x)) + x
            if save:
                rec.append((x, att, xa, xn, fdrop))
            x = xb
        out = ops.layernorm_fwd(x, cls_stride, ncls, self._p(norm[0]), self._p(norm[1]), d, lib=lib)
        if save:
            saved[which] = (rec, x)
        return out

    def _stack_backward(self, which, saved, dcls, smap, cls_stride, shape, G):
        c, lib = self._cfg, self._lib
        d, H, heads, dh = c["d"], c["hidden"], c["heads"], c["dh"]
        layers, norm = self._stacks[which]
        rec, x_last = saved[which]
        ws_ffn = self._workspace("ffn", lib.size("rat_ffn_bwd_workspace", d, H))
        dx = torch.zeros(shape, dtype=torch.float32, device=dcls.device)          # only the class-token rows get a gradient
        ops.layernorm_bwd(x_last, cls_stride, dcls, self._p(norm[0]), dx, cls_stride, G(norm[0]), G(norm[1]), d, lib=lib)
        for lay, (x_in, att, xa, xn, fdrop) in zip(reversed(layers), reversed(rec)):
            w1, b1, w2, b2 = [self._p(n) for n in lay["ffn"]]
            gw = [G(n) for n in lay["ffn"]]
            dxn, _ = ops.ffn_bwd_res(xn, dx, w1, b1, w2, b2, gw[0], gw[1], gw[2], gw[3], d, H, add_dy=False, workspace=ws_ffn,
                                     arith=self.arith, dropout=fdrop, lib=lib)
            dxa = ops.layernorm_bwd(xa, d, dxn, self._p(lay["ln"][0]), dxn, d, G(lay["ln"][0]), G(lay["ln"][1]), d, add=dx, lib=lib)
            dx = self._attn_layer_backward(lay["attn"], x_in, dxa, att, smap, G)
        return dx

    def _encoder_forward(self, x, x0, dims, save, saved):
        B, T, L, S = dims
        d = self._cfg["d"]
        xi = self._stack_forward("intra_transformer", x, ops.intra_map(B, T, S), B * T * S, S * d, B * T, save, saved)
        xc = self._stack_forward("cross_transformer", xi, ops.intra_map(B, 1, T), B * T, T * d, B, save, saved)
        return xc, d

    def _encoder_backward(self, saved, dx, G):
        B, T, L, S = saved["dims"]
        d = self._cfg["d"]
        dxi = self._stack_backward("cross_transformer", saved, dx, ops.intra_map(B, 1, T), T * d, (B, T, d), G)
        return self._stack_backward("intra_transformer", saved, dxi, ops.intra_map(B, T, S), S * d, (B, T, S, d), G)


class RAT_m3(RAT_m2):
    """RAT_m3 (fuxictr/pytorch/models/RAT_m3.py:27-243): intra and cross attention run IN PARALLEL on the block input,
    share the query projection, use heads/2 heads of width 2*dim_head (softmax scale still dim_head^-0.5), carry no
    residual of their own; their mean goes through the MLP, whose residual is the block input.  On the HIP path:
        out  = 0.5 * intra(x)                 rat_attn_fwd_ex(res = NULL,  out_scale = 0.5)
        out += 0.5 * cross(x)                 rat_attn_fwd_ex(res = out,   out_scale = 0.5)   (same memory, strided sequences)
        x'   = FFN(out) + x                   rat_ffn_fwd_res(x = out, res = x)
    The stacked [W_q; W_k; W_v] operand of each attention is assembled per step from the block's five projections."""

    def __init__(self, feature_map, model_id="RAT_m3", **kwargs):
        super().__init__(feature_map, model_id=model_id, **kwargs)

    def _make_encoder(self, d, num_heads, dim_head, dropout, depth, hidden):
        if num_heads < 2:
            raise ValueError("RAT_m3 splits the projections into num_heads/2 heads (RAT_m3.py:181): num_heads must be >= 2")
        self.encoder = _EncoderM3(d, num_heads, dim_head, dropout, depth, hidden)

    def _build_encoder_descriptors(self):
        c = self._cfg
        inner = c["heads"] * c["dh"]
        self._m3_heads = int(c["heads"] / 2)
        self._m3_dh = inner // self._m3_heads
        self._m3_scale = float(c["dh"]) ** -0.5
        self._blocks = []
        for i in range(c["depth"]):
            p = "encoder.encoder.%d." % i
            blk = {"proj": [p + n + ".weight" for n in ("W_q", "W_k_s", "W_v_s", "W_k_t", "W_v_t")],
                   "ffn": [p + "mlp.net.%s" % n for n in ("0.weight", "0.bias", "3.weight", "3.bias")]}
            for which in ("intra", "cross"):
                q = p + which + "_attention."
                has_out = (q + "fn.to_out.0.weight") in self._params
                blk[which] = [q + "norm.weight", q + "norm.bias", q + "fn.to_out.0.weight" if has_out else None,
                              q + "fn.to_out.0.bias" if has_out else None]
            # stacked projection operands, refreshed from the parameters at every forward
            blk["w_s"] = torch.empty((3 * inner, c["d"]), dtype=torch.float32, device=self.device)
            blk["w_t"] = torch.empty((3 * inner, c["d"]), dtype=torch.float32, device=self.device)
            self._blocks.append(blk)

    def _m3_params(self, blk, which, w_qkv, source):
        ln_g, ln_b, w_out, b_out = [source(n) if n else None for n in blk[which]]
        return ops.attn_params(ln_g, ln_b, w_qkv, w_out, b_out)

    # ---- one of the block's two attentions, y = 0.5 * Dropout(to_out(softmax(Q K^T dim_head^-0.5) V)) (+ res), at ANY head geometry
    #      (RAT_m3.py:164-189 takes any num_heads x dim_head; the README's RAT_PA-on-Tmall run is 32 x 10 at d = 10).  The same three
    #      forms as RAT_m2._attn_mode, on heads/2 heads of width 2 * dim_head:
    #        fused     one rat_attn_fwd_ex / rat_attn_bwd_ex launch;
    #        grouped   heads * dim_head too wide for the fused kernels' LDS tile: the heads only meet in to_out, so the layer is G
    #                  launches on `per` heads each — group g gets its rows of W_q | W_k | W_v and its columns of to_out, the first one
    #                  the bias, every later one accumulates onto the output of the one before (res = y; backward: add = dx);
    #        composed  heads wider than any fused instantiation (2 * dim_head > 20) or sequences above 64 tokens: LayerNorm ->
    #                  rat_sgemm -> rat_attn_core_*_map -> rat_sgemm.
    def _m3_arith(self, heads=None):
        """the arithmetic of a fused attention launch on `heads` (default: all heads / 2) heads of width 2 * dim_head: the model's, where the
        library has a bf16x3 instantiation for that geometry (round 6: 4 x 20 at embedding_dim 64 — the north-star config, and the head
        groups of a wider layer at that embedding_dim), else exact fp32"""
        heads = self._m3_heads if heads is None else heads
        cache = self.__dict__.setdefault("_m3_b3", {})
        v = cache.get(heads)
        if v is None:
            v = cache[heads] = self._lib.size("rat_attn_fwd_workspace", self._cfg["d"], heads, self._m3_dh) > 0
        return self.arith if v else "f32"

    def _m3_groups_supported(self):
        """rat_attn_groups_supported for heads / 2 heads of width 2 * dim_head (bit 0: one-launch forward, bit 1: one-launch backward)"""
        v = self.__dict__.get("_m3_groups_sup")
        if v is None:
            v = self._m3_groups_sup = ops.attn_groups_supported(self._cfg["d"], self._m3_heads, self._m3_dh, lib=self._lib)
        return v if self.group_loop else 0                       # (`group_loop = False`: the per-group launches, A/B and tests)

    def _m3_mode(self, smap):
        key = ("m3", int(smap.L), self.FUSED_MAX_L)
        hit = self._fused_cache.get(key)
        if hit is None:
            c, L, h, dh = self._cfg, int(smap.L), self._m3_heads, self._m3_dh
            ok = lambda n: L <= self.FUSED_MAX_L and ops.attn_fused_supported(c["d"], n, dh, L, lib=self._lib)   # noqa: E731
            if ok(h):
                hit = ("fused", h)
            else:
                per = next((n for n in (8, 4, 2, 1) if h % n == 0 and n < h and ok(n)), None)
                hit = ("grouped", per) if per else ("composed", None)
            self._fused_cache[key] = hit
        return hit

    def _m3_group_params(self, blk, which, w_stack, per):
        """per-group (RatAttnParams, w_qkv_g, w_out_g): contiguous copies of group g's rows of each of the Q | K | V blocks and of its
        columns of to_out; the bias rides with group 0"""
        d, dh, groups = self._cfg["d"], self._m3_dh, self._m3_heads // per
        ig = per * dh
        ln_g, ln_b, w_out, b_out = [self._p(n) if n else None for n in blk[which]]
        if w_out is None:
            raise NotImplementedError("grouped attention needs an output projection")
        wq = w_stack.view(3, groups, ig, d).permute(1, 0, 2, 3).contiguous()
        wo = w_out.view(d, groups, ig).permute(1, 0, 2).contiguous()
        zero_bias = torch.zeros_like(b_out)
        return [(ops.attn_params(ln_g, ln_b, wq[g].view(3 * ig, d), wo[g], b_out if g == 0 else zero_bias), wq[g], wo[g], zero_bias)
                for g in range(groups)]

    def _m3_attn_forward(self, blk, which, w_stack, x, res, smap, save, drop, out=None):
        """-> (y, what the backward needs); y = 0.5 * attention(LayerNorm(x)) + res (res None: no addend; res may be `out` itself)"""
        c, lib = self._cfg, self._lib
        d, h, dh, sc = c["d"], self._m3_heads, self._m3_dh, self._m3_scale
        mode, per = self._m3_mode(smap)
        if mode == "fused":
            y, o, l = ops.attn_fwd_ex(x, res, self._m3_params(blk, which, w_stack, self._p), smap, d, h, dh, sc, 0.5, save=save, out=out,
                                      arith=self._m3_arith(), dropout=drop, lib=lib)
            return y, (o, l)
        if mode == "grouped" and (self._m3_groups_supported() & (3 if save else 1)) == (3 if save else 1) and blk[which][2] is not None:
            # small embedding dimension (the Tmall geometry): the whole layer in ONE launch that loops over the head groups inside a chunk
            # (rat_attn_fwd_groups, exact fp32, weights addressed in place) — and the backward in one launch too (rat_attn_bwd_groups)
            y, o, l = ops.attn_fwd_groups(x, res, self._m3_params(blk, which, w_stack, self._p), None, smap, d, h, dh, sc, 0.5, save=save,
                                          out=out, dropout=drop, lib=lib)
            return y, (("loop", o, l) if save else None)
        if mode == "grouped":
            y, kept = out, []
            for g, (params_g, w_g, wo_g, zb) in enumerate(self._m3_group_params(blk, which, w_stack, per)):
                # Dropout(sum of the groups' partial projections + bias) = the sum of the equally masked partials: same seed everywhere
                y, o, l = ops.attn_fwd_ex(x, res if g == 0 else y, params_g, smap, d, per, dh, sc, 0.5, save=save, out=y,
                                          arith=self._m3_arith(per), dropout=drop, lib=lib)
                kept.append((params_g, w_g, wo_g, zb, o, l))
            return y, (kept if save else None)
        inner, ntok = h * dh, x.numel() // d
        ln_g, ln_b, w_out, b_out = [self._p(n) if n else None for n in blk[which]]
        if w_out is None:
            raise NotImplementedError("attention without an output projection is only implemented in the fused kernel")
        xn = ops.layernorm_fwd(x, d, ntok, ln_g, ln_b, d, lib=lib)
        qkv = torch.empty((ntok, 3 * inner), dtype=torch.float32, device=x.device)
        ops.sgemm(0, 1, ntok, 3 * inner, d, xn, d, w_stack, d, qkv, 3 * inner, arith=self.gemm_arith, lib=lib)
        o, lse = ops.attn_core_fwd_map(qkv, smap, h, dh, softmax_scale=sc, save=True, lib=lib)
        t = torch.empty((ntok, d), dtype=torch.float32, device=x.device)
        ops.sgemm(0, 1, ntok, d, inner, o, inner, w_out, inner, t, d, bias=b_out, arith=self.gemm_arith, lib=lib)
        if drop[0] > 0:
            ops.dropout(t, drop[0], drop[1], out=t, lib=lib)
        t = t.view_as(x)
        y = torch.mul(t, 0.5, out=out) if res is None else torch.add(res, t, alpha=0.5, out=out)
        return y, ((qkv, o, lse) if save else None)

    def _m3_attn_backward(self, blk, which, w_stack, g_stack, x_in, dy, add, att, smap, G, ws, drop, out=None):
        """dx = add + d/dx [0.5 * attention(LayerNorm(x))] (add: a grid laid out like x; may be `out`); the gradients of the stacked
        [W_q; W_k; W_v] operand go to g_stack, the layer's own (LayerNorm, to_out) through G"""
        c, lib = self._cfg, self._lib
        d, h, dh, sc = c["d"], self._m3_heads, self._m3_dh, self._m3_scale
        mode, per = self._m3_mode(smap)
        if mode == "fused":
            dx, _ = ops.attn_bwd_ex(x_in, dy, add, att[0], att[1], self._m3_params(blk, which, w_stack, self._p),
                                    self._m3_params(blk, which, g_stack, G), smap, d, h, dh, sc, 0.5, workspace=ws, out=out,
                                    arith=self._m3_arith(), dropout=drop, lib=lib)
            return dx
        names = blk[which]
        if mode == "grouped" and att[0] == "loop":
            wsl = self._workspace("attn_m3_groups", lib.size("rat_attn_bwd_groups_workspace", d, h, dh))
            dx, _ = ops.attn_bwd_groups(x_in, dy, add, att[1], att[2], self._m3_params(blk, which, w_stack, self._p),
                                        self._m3_params(blk, which, g_stack, G), smap, d, h, dh, sc, 0.5, workspace=wsl, out=out, dropout=drop,
                                        lib=lib)
            return dx
        if mode == "grouped":
            groups, ig = h // per, per * dh
            wsg = self._workspace("attn_m3g", lib.size("rat_attn_bwd_workspace", d, per, dh))
            g_lng, g_lnb, g_wout, g_bout = [G(n) for n in names]
            t_ln = torch.empty((2, groups, g_lng.numel()), dtype=torch.float32, device=dy.device)
            t_b = torch.empty_like(g_bout)
            t_w = torch.empty((groups, 3 * ig, d), dtype=torch.float32, device=dy.device)
            t_wo = torch.empty((groups, d, ig), dtype=torch.float32, device=dy.device)
            dx = out
            for g, (params_g, w_g, wo_g, zb, o, l) in enumerate(att):
                grads_g = ops.attn_params(t_ln[0, g], t_ln[1, g], t_w[g], t_wo[g], g_bout if g == 0 else t_b)
                dx, _ = ops.attn_bwd_ex(x_in, dy, add if g == 0 else dx, o, l, params_g, grads_g, smap, d, per, dh, sc, 0.5, workspace=wsg,
                                        out=dx, arith=self._m3_arith(per), dropout=drop, lib=lib)
            g_stack.view(3, groups, ig, d).copy_(t_w.view(groups, 3, ig, d).permute(1, 0, 2, 3))
            g_wout.view(d, groups, ig).copy_(t_wo.permute(1, 0, 2))
            torch.sum(t_ln[0], 0, out=g_lng)
            torch.sum(t_ln[1], 0, out=g_lnb)
            return dx
        inner, ntok = h * dh, x_in.numel() // d
        ln_g, ln_b, w_out, b_out = [self._p(n) if n else None for n in names]
        qkv, o, lse = att
        dev = dy.device
        xn = ops.layernorm_fwd(x_in, d, ntok, ln_g, ln_b, d, lib=lib)                                    # recomputed, not stored
        dyp = dy.reshape(ntok, d) if drop[0] == 0 else ops.dropout(dy.reshape(ntok, d), drop[0], drop[1], lib=lib)
        dyp = dyp * 0.5                                                                                   # the mean of the two attentions
        do = torch.empty((ntok, inner), dtype=torch.float32, device=dev)
        ops.sgemm(0, 0, ntok, inner, d, dyp, d, w_out, inner, do, inner, arith=self.gemm_arith, lib=lib)                 # dO = dy W_out
        ops.sgemm(1, 0, d, inner, ntok, dyp, d, o, inner, G(names[2]), inner, arith=self.gemm_arith, lib=lib)            # dW_out = dy^T O
        ops.colsum(dyp, d, G(names[3]), ntok, d, lib=lib)
        dqkv = ops.attn_core_bwd_map(qkv, o, lse, do, smap, h, dh, softmax_scale=sc, lib=lib)
        dxn = torch.empty((ntok, d), dtype=torch.float32, device=dev)
        ops.sgemm(0, 0, ntok, d, 3 * inner, dqkv, 3 * inner, w_stack, d, dxn, d, arith=self.gemm_arith, lib=lib)         # d(norm(x)) = dQKV W
        ops.sgemm(1, 0, 3 * inner, d, ntok, dqkv, 3 * inner, xn, d, g_stack, d, arith=self.gemm_arith, lib=lib)          # dW = dQKV^T norm(x)
        dx = ops.layernorm_bwd(x_in, d, dxn, ln_g, dxn, d, G(names[0]), G(names[1]), d, add=add, lib=lib).view_as(x_in)
        if out is not None and out.data_ptr() != dx.data_ptr():
            out.copy_(dx)
            dx = out
        return dx

    def _encoder_forward(self, x, x0, dims, save, saved):
        c, lib = self._cfg, self._lib
        B, T, L, S = dims
        d, H = c["d"], c["hidden"]
        imap, cmap = ops.intra_map(B, T, S), ops.cross_map(B, T, S)
        # (the composed form runs LayerNorm and the projections over the whole grid: nothing to gain from the last block's row maps)
        prune = self.prune_dead_tokens and "composed" not in (self._m3_mode(imap)[0], self._m3_mode(cmap)[0])
        last = len(self._blocks) - 1
        for bi, blk in enumerate(self._blocks):
            wq, wks, wvs, wkt, wvt = [self._p(n) for n in blk["proj"]]
            torch.cat([wq, wks, wvs], dim=0, out=blk["w_s"])
            torch.cat([wq, wkt, wvt], dim=0, out=blk["w_t"])
            # each of the two attentions has its own nn.Dropout behind to_out (RAT_m3.py:186-189 via Attention): y = 0.5 * Dropout(..)
            drop_on = self.training and c["attn_dropout"] > 0
            dr_s = (c["attn_dropout"], self._dropout_word()) if drop_on and blk["intra"][2] else (0.0, 0)
            dr_t = (c["attn_dropout"], self._dropout_word()) if drop_on and blk["cross"][2] else (0.0, 0)
            if bi == last and prune:
                # Dead-token pruning (RAT_m2._encoder_forward): the head reads x[:, 0][:, 0] only (RAT_m3.py:128-129), and in
                # the LAST block that token needs the intra-sample attention of the target sample's sequence and the cross-sample
                # attention of token position 0's sequence — B sequences each instead of B T and B S — and the MLP on one token.
                im0, cm0 = ops.intra_map_target_sample(B, T, S), ops.cross_map_label_token(B, T, S)
                out, a_s = self._m3_attn_forward(blk, "intra", blk["w_s"], x, None, im0, save, dr_s)
                out, a_t = self._m3_attn_forward(blk, "cross", blk["w_t"], x, out, cm0, save, dr_t, out=out)
                out_cls = out.view(B, T, S, d)[:, 0, 0, :].contiguous()
                x_cls = x.view(B, T, S, d)[:, 0, 0, :].contiguous()
                w1, b1, w2, b2 = [self._p(n) for n in blk["ffn"]]
                xc = ops.ffn_fwd_res(out_cls, x_cls, w1, b1, w2, b2, d, H, arith=self.arith, lib=lib)
                if save:
                    saved["blocks"].append((x, a_s, a_t, out_cls, dr_s, dr_t))
                    saved["pruned"] = True
                return xc, d
            out, a_s = self._m3_attn_forward(blk, "intra", blk["w_s"], x, None, imap, save, dr_s)                 # 0.5 * intra(x)
            out, a_t = self._m3_attn_forward(blk, "cross", blk["w_t"], x, out, cmap, save, dr_t, out=out)         # += 0.5 * cross(x)
            w1, b1, w2, b2 = [self._p(n) for n in blk["ffn"]]
            xn = ops.ffn_fwd_res(out, x, w1, b1, w2, b2, d, H, arith=self.arith, lib=lib)                         # mlp(out) + x
            if save:
                saved["blocks"].append((x, a_s, a_t, out, dr_s, dr_t))
            x = xn
        return x, T * S * d

    def _encoder_backward(self, saved, dx, G):
        c, lib = self._cfg, self._lib
        B, T, L, S = saved["dims"]
        d, H = c["d"], c["hidden"]
        inner = c["heads"] * c["dh"]
        h, dh = self._m3_heads, self._m3_dh
        imap, cmap = ops.intra_map(B, T, S), ops.cross_map(B, T, S)
        ws_attn = self._workspace("attn", lib.size("rat_attn_bwd_workspace", d, h, dh))
        ws_ffn = self._workspace("ffn", lib.size("rat_ffn_bwd_workspace", d, H))
        g_s = torch.empty((3 * inner, d), dtype=torch.float32, device=dx.device)
        g_t = torch.empty((3 * inner, d), dtype=torch.float32, device=dx.device)
        pruned = bool(saved.get("pruned"))
        for bi, (blk, (x_in, a_s, a_t, out, dr_s, dr_t)) in enumerate(zip(reversed(self._blocks), reversed(saved["blocks"]))):
            w1, b1, w2, b2 = [self._p(n) for n in blk["ffn"]]
            gw = [G(n) for n in blk["ffn"]]
            dout, _ = ops.ffn_bwd_res(out, dx, w1, b1, w2, b2, gw[0], gw[1], gw[2], gw[3], d, H, add_dy=False, workspace=ws_ffn,
                                      arith=self.arith, lib=lib)
            amap, bmap = cmap, imap
            if bi == 0 and pruned:        # the last block (see _encoder_forward): `out`, dx, dout are the class tokens' [B, d] rows
                dgrid = torch.zeros((B, T, S, d), dtype=torch.float32, device=dx.device)
                dgrid[:, 0, 0, :] = dx                              # the residual of x' = mlp(out) + x
                dog = torch.zeros((B, T, S, d), dtype=torch.float32, device=dx.device)
                dog[:, 0, 0, :] = dout
                dx, dout = dgrid, dog
                amap, bmap = ops.cross_map_label_token(B, T, S), ops.intra_map_target_sample(B, T, S)
                # in place: the rows of the two B-sequence maps receive their gradient, every other row stays zero
                dxn = self._m3_attn_backward(blk, "cross", blk["w_t"], g_t, x_in, dout, dx, a_t, amap, G, ws_attn, dr_t, out=dx)
            else:
                # dx = dy (the MLP residual) + cross backward + intra backward, accumulated in place
                dxn = self._m3_attn_backward(blk, "cross", blk["w_t"], g_t, x_in, dout, dx, a_t, amap, G, ws_attn, dr_t)
            dxn = self._m3_attn_backward(blk, "intra", blk["w_s"], g_s, x_in, dout, dxn, a_s, bmap, G, ws_attn, dr_s, out=dxn)
            gq, gks, gvs, gkt, gvt = [G(n) for n in blk["proj"]]
            torch.add(g_s[:inner], g_t[:inner], out=gq)                    # W_q is used by both attentions
            gks.copy_(g_s[inner:2 * inner])
            gvs.copy_(g_s[2 * inner:])
            gkt.copy_(g_t[inner:2 * inner])
            gvt.copy_(g_t[2 * inner:])
            dx = dxn
        return dx


class RAT_m0(RAT_m1):
    """RAT_m0 (fuxictr/pytorch/models/RAT_m0.py:24-141): ONE Transformer over the joint sequence of all T*S tokens of a sample
    ('b t n d -> b (t n) d'), class token = token (t=0, n=0).  Sequences of up to 64 tokens run on the fused attention kernel
    (K2a); longer ones (231 at the north-star shape) do not fit its LDS tile and take the composed path of
    RAT_m2._attn_layer_forward (K2c LayerNorm -> rat_sgemm -> K2d attention core -> rat_sgemm + bias + residual)."""

    def __init__(self, feature_map, model_id="RAT_m0", **kwargs):
        super().__init__(feature_map, model_id=model_id, **kwargs)

    def _make_encoder(self, d, num_heads, dim_head, dropout, depth, hidden):
        self.encoder = _Transformer(d, depth, num_heads, dim_head, hidden, dropout)               # RAT_m0.py:70

    def _build_encoder_descriptors(self):
        layers = []
        for i in range(self._cfg["depth"]):
            p = "encoder.layers.%d." % i
            layers.append(dict(attn=self._attn_descriptor(p + "0."), ln=[p + "1.norm.weight", p + "1.norm.bias"],
                               ffn=[p + "1.fn.net.0.weight", p + "1.fn.net.0.bias", p + "1.fn.net.3.weight", p + "1.fn.net.3.bias"]))
        self._stacks = {"encoder": (layers, ["encoder.norm.weight", "encoder.norm.bias"])}

    def _encoder_forward(self, x, x0, dims, save, saved):
        B, T, L, S = dims
        d = self._cfg["d"]
        xc = self._stack_forward("encoder", x, ops.intra_map(B, 1, T * S), B * T * S, T * S * d, B, save, saved)
        return xc, d

    def _encoder_backward(self, saved, dx, G):
        B, T, L, S = saved["dims"]
        d = self._cfg["d"]
        return self._stack_backward("encoder", saved, dx, ops.intra_map(B, 1, T * S), T * S * d, (B, T, S, d), G)
