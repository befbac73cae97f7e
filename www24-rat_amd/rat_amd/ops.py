"""Tensor-level wrappers over the C ABI (include/rat_hip.h).  No arithmetic happens here: every function checks
layout/dtype, takes raw ``data_ptr()``s plus the current HIP stream and calls into librat_hip.so.

``lib`` defaults to the process-wide HIP library; tests may pass another ``RatLib`` (the host-emulation build).
"""
import ctypes

import numpy as np
import torch

from ._lib import RatAttnParams, RatSeqMap, RatSplitJob, get_lib

FIELD_DTYPE = np.dtype([("table", "<u8"), ("col", "<i4"), ("ncols", "<i4"), ("vocab", "<i4"), ("padding_idx", "<i4")])
assert FIELD_DTYPE.itemsize == 24


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _stream(t):
    if t.is_cuda:
        return ctypes.c_void_p(torch.cuda.current_stream(t.device).cuda_stream)
    return None


def _chk(t, dtype=torch.float32, name="tensor"):
    if t is None:
        return
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)


def field_table(fields, tables, device):
    """Device array of RatField records.  fields: objects with .col/.ncols/.vocab/.padding_idx; tables: tensors."""
    arr = np.zeros(len(fields), dtype=FIELD_DTYPE)
    for i, (f, t) in enumerate(zip(fields, tables)):
        _chk(t, name="table")
        arr[i] = (t.data_ptr(), f.col, f.ncols, f.vocab, -1 if f.padding_idx is None else f.padding_idx)
    raw = torch.from_numpy(arr.view(np.uint8).copy())
    return raw.to(device)


def intra_map(B, T, S, queries=0):
    """queries: RatSeqMap.queries — 0 = every position; n: only the outputs of positions [0, n) of each sequence are wanted"""
    return RatSeqMap(nseq=B * T, L=S, queries=queries, q_div=B * T, hi_stride=0, lo_stride=S, pos_stride=1)


def cross_map(B, T, S):
    return RatSeqMap(nseq=B * S, L=T, q_div=S, hi_stride=T * S, lo_stride=1, pos_stride=S)


def cross_map_label_token(B, T, S, queries=0):
    """the cross-sample sequences of token position 0 (the label / class token) only: B sequences of T tokens, S tokens apart"""
    return RatSeqMap(nseq=B, L=T, queries=queries, q_div=1, hi_stride=T * S, lo_stride=0, pos_stride=S)


def intra_map_target_sample(B, T, S, queries=0):
    """the intra-sample sequences of retrieved-sample index 0 (the target) only: B sequences of S tokens"""
    return RatSeqMap(nseq=B, L=S, queries=queries, q_div=1, hi_stride=T * S, lo_stride=0, pos_stride=1)


def attn_params(ln_g, ln_b, w_qkv, w_out, b_out, planes=None):
    """planes: optional uint8 tensor of ``attn_planes_bytes`` bytes holding the bf16x3 fragment planes of these weights (filled by
    ``split_weights_batch`` from ``attn_split_jobs``); without it the bf16x3 entry points split the weights on every call."""
    for t in (ln_g, ln_b, w_qkv, w_out, b_out):
        _chk(t, name="attention parameter")
    return RatAttnParams(ln_g.data_ptr(), ln_b.data_ptr(), w_qkv.data_ptr(),
                         w_out.data_ptr() if w_out is not None else None,
                         b_out.data_ptr() if b_out is not None else None,
                         planes.data_ptr() if planes is not None else None)


# ----------------------------------------------------------------------------- bf16x3 weight planes, once per step
def attn_planes_bytes(d, heads, dim_head, lib=None):
    return (lib or get_lib()).size("rat_attn_planes_bytes", d, heads, dim_head)


def ffn_planes_bytes(d, hidden, lib=None):
    return (lib or get_lib()).size("rat_ffn_planes_bytes", d, hidden)


def attn_split_jobs(params, d, heads, dim_head, planes, lib=None):
    """-> list of RatSplitJob that fill ``planes`` (uint8 tensor, 16-byte aligned) from the weights of ``params``"""
    lib = lib or get_lib()
    jobs = (RatSplitJob * 4)()
    n = lib.cdll.rat_attn_split_jobs(ctypes.byref(params), d, heads, dim_head, _p(planes), jobs)
    if n < 0:
        raise RuntimeError("rat_attn_split_jobs: " + lib.last_error())
    return [jobs[i] for i in range(n)]


def attn_groups_planes_bytes(d, heads, dim_head, lib=None):
    """bytes of the per-group [W_qkv | W_out] fragment planes attn_fwd_groups reads; 0: that form does not serve these dimensions"""
    return (lib or get_lib()).size("rat_attn_groups_planes_bytes", d, heads, dim_head)


def attn_groups_split_jobs(params, d, heads, dim_head, planes, lib=None):
    """-> list of RatSplitJob (4 per head group) that fill ``planes`` from the FULL-width weights of ``params``, in place"""
    lib = lib or get_lib()
    jobs = (RatSplitJob * (4 * max(heads // 8, 1)))()
    n = lib.cdll.rat_attn_groups_split_jobs(ctypes.byref(params), d, heads, dim_head, _p(planes), jobs)
    if n < 0:
        raise RuntimeError("rat_attn_groups_split_jobs: " + lib.last_error())
    return [jobs[i] for i in range(n)]


def ffn_split_jobs(w1, w2, d, hidden, planes, lib=None):
    lib = lib or get_lib()
    jobs = (RatSplitJob * 3)()
    n = lib.cdll.rat_ffn_split_jobs(_p(w1), _p(w2), d, hidden, _p(planes), jobs)
    if n < 0:
        raise RuntimeError("rat_ffn_split_jobs: " + lib.last_error())
    return [jobs[i] for i in range(n)]


def split_job_array(jobs):
    """a ctypes array that can be kept and replayed every step (the pointers inside must stay valid)"""
    arr = (RatSplitJob * max(len(jobs), 1))()
    for i, j in enumerate(jobs):
        arr[i] = j
    return arr, len(jobs)


def split_weights_batch(job_array, njobs, like, lib=None):
    """ONE launch for every job; `like`: any tensor on the target device (stream selection)"""
    if njobs:
        (lib or get_lib()).call("rat_split_weights_batch", job_array, njobs, _stream(like))


# ----------------------------------------------------------------------------- K0
def batch_assemble(data_ids, data_labels, pool_ids, pool_labels, retr_indices, rows, lib=None):
    """Device-side Dataset.__getitem__ + collate: -> (idx [B,1+K,L] int32, label_ids [B,1+K] int32, y_true [B] fp32)."""
    lib = lib or get_lib()
    _chk(data_ids, torch.int32, "data_ids"), _chk(pool_ids, torch.int32, "pool_ids")
    _chk(data_labels, name="data_labels"), _chk(pool_labels, name="pool_labels")
    _chk(retr_indices, torch.int64, "retr_indices"), _chk(rows, torch.int64, "rows")
    Q, L = data_ids.shape
    N, K, B = pool_ids.shape[0], retr_indices.shape[1], rows.numel()
    dev = data_ids.device
    idx = torch.empty((B, K + 1, L), dtype=torch.int32, device=dev)
    label_ids = torch.empty((B, K + 1), dtype=torch.int32, device=dev)
    y_true = torch.empty((B,), dtype=torch.float32, device=dev)
    lib.call("rat_batch_assemble", _p(data_ids), _p(data_labels), _p(pool_ids), _p(pool_labels), _p(retr_indices), _p(rows),
             _p(idx), _p(label_ids), _p(y_true), Q, N, B, K, L, _stream(data_ids))
    return idx, label_ids, y_true


# ----------------------------------------------------------------------------- K1
_DTYPE_CODE = {torch.int32: 0, torch.int64: 1, torch.float32: 2, torch.float64: 3}       # RAT_DTYPE_* (include/rat_hip.h)


def batch_prepare(X, y, out=None, lib=None):
    """device tensors X [B,T,L] (int32 / int64 / fp32 / fp64), y [B,T] (fp32 / fp64) -> (idx int32, label ids int32, y_true fp32 [B]);
    `out`: three tensors of those shapes to write into (the static inputs of a captured step)"""
    lib = lib or get_lib()
    B, T, L = X.shape
    if X.dtype not in _DTYPE_CODE or y.dtype not in (torch.float32, torch.float64):
        raise TypeError("batch_prepare: X %s / y %s" % (X.dtype, y.dtype))
    X, y = X.contiguous(), y.contiguous()
    if out is None:
        out = (torch.empty((B, T, L), dtype=torch.int32, device=X.device), torch.empty((B, T), dtype=torch.int32, device=X.device),
               torch.empty((B,), dtype=torch.float32, device=X.device))
    idx, labels, y_true = out
    _chk(idx, torch.int32, "idx"), _chk(labels, torch.int32, "label_ids"), _chk(y_true, name="y_true")
    assert tuple(idx.shape) == (B, T, L) and tuple(labels.shape) == (B, T) and y_true.numel() == B
    lib.call("rat_batch_prepare", _p(X), _DTYPE_CODE[X.dtype], _p(y), _DTYPE_CODE[y.dtype], _p(idx), _p(labels), _p(y_true), B, T, L,
             _stream(X))
    return idx, labels, y_true


def gather_fwd(idx, label_ids, ftab, nfields, label_table, B, T, L, d, lib=None):
    lib = lib or get_lib()
    _chk(idx, torch.int32, "idx"), _chk(label_ids, torch.int32, "label_ids"), _chk(label_table, name="label_table")
    grid = torch.empty((B, T, nfields + 1, d), dtype=torch.float32, device=idx.device)
    lib.call("rat_gather_fwd", _p(idx), _p(label_ids), _p(ftab), nfields, _p(label_table), _p(grid), B, T, L, d, _stream(idx))
    return grid


def check_ids(idx, label_ids, ftab, nfields, counts, B, T, L, lib=None):
    """counts (device int32[2]) += [feature ids outside their table, label ids outside {0,1,2}] — see rat_check_ids."""
    lib = lib or get_lib()
    _chk(idx, torch.int32, "idx"), _chk(label_ids, torch.int32, "label_ids"), _chk(counts, torch.int32, "counts")
    lib.call("rat_check_ids", _p(idx), _p(label_ids), _p(ftab), nfields, B, T, L, _p(counts), _stream(idx))


def gather_bwd(dgrid, dflat, idx, label_ids, gftab, nfields, dlabel_table, B, T, L, d, lib=None):
    lib = lib or get_lib()
    _chk(dgrid, name="dgrid"), _chk(dflat, name="dflat"), _chk(dlabel_table, name="dlabel_table")
    lib.call("rat_gather_bwd", _p(dgrid), _p(dflat), _p(idx), _p(label_ids), _p(gftab), nfields, _p(dlabel_table),
             B, T, L, d, _stream(dgrid))


# ----------------------------------------------------------------------------- K1s (row-sparse / deterministic table gradients)
def col2field_table(fields, L, device):
    """int32 [L]: the field that owns each id column of idx (-1: no field reads it)"""
    arr = np.full(L, -1, dtype=np.int32)
    for i, f in enumerate(fields):
        arr[f.col:f.col + f.ncols] = i
    return torch.from_numpy(arr).to(device)


class SparsePlan:
    """Workspace + device-side unique-row count of one rat_sparse_plan_* call (what the matching reduce call needs)."""

    __slots__ = ("ws", "count", "n")

    def __init__(self, n, device, lib):
        self.n = int(n)
        self.ws = torch.empty(lib.size("rat_sparse_workspace", self.n), dtype=torch.uint8, device=device)
        self.count = torch.zeros(1, dtype=torch.int32, device=device)


def sparse_plan_ids(idx, ftab, col2field, nfields, flat_base, width, total_rows, B, T, L, target_only=False, plan=None, lib=None):
    """flat_base: the tensor whose first element is row 0 of the table block the RatField pointers of `ftab` point into."""
    lib = lib or get_lib()
    _chk(idx, torch.int32, "idx"), _chk(col2field, torch.int32, "col2field")
    n = (B if target_only else B * T) * L
    if plan is None or plan.n != n:
        plan = SparsePlan(n, idx.device, lib)
    lib.call("rat_sparse_plan_ids", _p(idx), _p(ftab), _p(col2field), nfields, _p(flat_base), int(width), int(total_rows), B, T, L,
             int(bool(target_only)), _p(plan.ws), plan.ws.numel(), _p(plan.count), _stream(idx))
    return plan


def sparse_plan_rows(rows, counts, cap, world, total_rows, plan=None, count_out=None, lib=None):
    """count_out: where the number of unique rows is written (default: the plan's own counter) — pass the same tensor to the reduce"""
    lib = lib or get_lib()
    _chk(rows, torch.int32, "rows"), _chk(counts, torch.int32, "counts")
    n = cap * world
    if plan is None or plan.n != n:
        plan = SparsePlan(n, rows.device, lib)
    lib.call("rat_sparse_plan_rows", _p(rows), _p(counts), int(cap), int(world), int(total_rows), _p(plan.ws), plan.ws.numel(),
             _p(plan.count if count_out is None else count_out), _stream(rows))
    return plan


def sparse_reduce_grid(plan, dgrid, dflat, col2field, B, T, L, nfields, d, out_rows=None, out_grads=None, dense_base=None,
                       target_only=False, lib=None):
    lib = lib or get_lib()
    _chk(dgrid, name="dgrid"), _chk(dflat, name="dflat"), _chk(out_grads, name="out_grads"), _chk(out_rows, torch.int32, "out_rows")
    lib.call("rat_sparse_reduce_grid", _p(plan.ws), _p(plan.count), _p(dgrid), _p(dflat), _p(col2field), B, T, L, nfields, d,
             int(bool(target_only)), _p(out_rows), _p(out_grads), _p(dense_base), _stream(dgrid))


def sparse_reduce_rows(plan, src_rows, cap, world, d, out_rows, out_grads, count=None, lib=None):
    lib = lib or get_lib()
    _chk(src_rows, name="src_rows"), _chk(out_grads, name="out_grads"), _chk(out_rows, torch.int32, "out_rows")
    lib.call("rat_sparse_reduce_rows", _p(plan.ws), _p(plan.count if count is None else count), _p(src_rows), int(cap), int(world), d,
             _p(out_rows), _p(out_grads), _stream(src_rows))


def sparse_reduce_scalar(plan, per_sample, B, L, out_rows=None, out_vals=None, dense_base=None, lib=None):
    lib = lib or get_lib()
    _chk(per_sample, name="per_sample"), _chk(out_vals, name="out_vals"), _chk(out_rows, torch.int32, "out_rows")
    lib.call("rat_sparse_reduce_scalar", _p(plan.ws), _p(plan.count), _p(per_sample), B, L, _p(out_rows), _p(out_vals), _p(dense_base),
             _stream(per_sample))


def sumsq_rows(grads, count, max_rows, d, out, lib=None):
    lib = lib or get_lib()
    lib.call("rat_sumsq_rows", _p(grads), _p(count), int(max_rows), d, _p(out), _stream(grads))


def adam_rows(w_base, m_base, v_base, rows, grads, count, max_rows, d, norm_sq, max_norm, lr, beta1, beta2, eps, step, lib=None):
    lib = lib or get_lib()
    lib.call("rat_adam_rows", _p(w_base), _p(m_base), _p(v_base), _p(rows), _p(grads), _p(count), int(max_rows), d, _p(norm_sq),
             float(max_norm), float(lr), float(beta1), float(beta2), float(eps), int(step), _stream(grads))


def label_grad(dgrid, label_ids, dlabel, nbt, S, d, lib=None):
    """deterministic gradient of the 3-row label table: per-block partial sums + fixed-order combine (no atomics)"""
    lib = lib or get_lib()
    ws = torch.empty(lib.size("rat_label_grad_workspace", d) // 4, dtype=torch.float32, device=dgrid.device)
    lib.call("rat_label_grad", _p(dgrid), _p(label_ids), _p(dlabel), _p(ws), int(nbt), S, d, _stream(dgrid))


# ----------------------------------------------------------------------------- K2
ARITH = {"f32": 0, "bf16x3": 1}       # RAT_ARITH_* of include/rat_hip.h


def attn_fwd(x, params, seqmap, d, heads, dim_head, save=False, eps=1e-5, out=None, arith="f32", dropout=(0.0, 0), lib=None):
    """PreNorm(Attention)(x) + x.  arith "f32": rat_attn_fwd (exact fp32 MFMA); "bf16x3": the split-operand bf16 MFMA kernels.
    dropout = (p, seed) of the Dropout behind the output projection (training only)."""
    if arith != "f32" or dropout[0] > 0:
        return attn_fwd_ex(x, x, params, seqmap, d, heads, dim_head, save=save, eps=eps, out=out, arith=arith, dropout=dropout, lib=lib)
    lib = lib or get_lib()
    _chk(x, name="x")
    y = out if out is not None else torch.empty_like(x)
    ntok = x.numel() // d
    o_save = lse = None
    if save:
        o_save = torch.empty((ntok, heads * dim_head), dtype=torch.float32, device=x.device)
        lse = torch.empty((ntok, heads), dtype=torch.float32, device=x.device)
    lib.call("rat_attn_fwd", _p(x), _p(y), _p(o_save), _p(lse), ctypes.byref(params), ctypes.byref(seqmap), d, heads,
             dim_head, eps, _stream(x))
    return y, o_save, lse


def attn_bwd(x, dy, o_save, lse, params, grads, seqmap, d, heads, dim_head, eps=1e-5, workspace=None, arith="f32", dropout=(0.0, 0),
             out=None, lib=None):
    """out: where dx goes (default: a new tensor); `out is dy` is allowed — a work-group reads the dy rows of its sequences before it
    writes their dx rows, and rows outside the map's sequences are left as they are."""
    if arith != "f32" or dropout[0] > 0:
        return attn_bwd_ex(x, dy, dy, o_save, lse, params, grads, seqmap, d, heads, dim_head, eps=eps, workspace=workspace,
                           out=out, arith=arith, dropout=dropout, lib=lib)
    lib = lib or get_lib()
    _chk(x, name="x"), _chk(dy, name="dy")
    need = lib.size("rat_attn_bwd_workspace", d, heads, dim_head)
    if workspace is None or workspace.numel() * 4 < need:
        workspace = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
    dx = out if out is not None else torch.empty_like(x)
    lib.call("rat_attn_bwd", _p(x), _p(dy), _p(o_save), _p(lse), _p(dx), ctypes.byref(params), ctypes.byref(grads),
             _p(workspace), workspace.numel() * 4, ctypes.byref(seqmap), d, heads, dim_head, eps, _stream(x))
    return dx, workspace


_FWD_WS = {}


def _attn_fwd_workspace(lib, d, heads, dim_head, device):
    """per (library, shape, device, stream) scratch for the pre-split weight fragments of the bf16x3 forward: launches on one
    stream are ordered, so consecutive layers may share it"""
    need = lib.size("rat_attn_fwd_workspace", d, heads, dim_head)
    if not need:
        return None
    key = (id(lib), d, heads, dim_head, str(device), torch.cuda.current_stream(device).cuda_stream if device.type == "cuda" else 0)
    ws = _FWD_WS.get(key)
    if ws is None:
        ws = _FWD_WS[key] = torch.empty((need + 3) // 4, dtype=torch.float32, device=device)
    return ws


def attn_fwd_ex(x, res, params, seqmap, d, heads, dim_head, softmax_scale=0.0, out_scale=1.0, save=False, eps=1e-5, out=None,
                arith="f32", dropout=(0.0, 0), lib=None):
    """y = out_scale * attention(LayerNorm(x)) + res (res: a tensor laid out like x, the output itself, or None)."""
    lib = lib or get_lib()
    _chk(x, name="x")
    y = out if out is not None else torch.empty_like(x)
    ntok = x.numel() // d
    o_save = lse = None
    if save:
        o_save = torch.empty((ntok, heads * dim_head), dtype=torch.float32, device=x.device)
        lse = torch.empty((ntok, heads), dtype=torch.float32, device=x.device)
    ws = _attn_fwd_workspace(lib, d, heads, dim_head, x.device) if arith != "f32" else None
    drop_p, drop_seed = _drop_args(params, dropout)
    lib.call("rat_attn_fwd_ex", _p(x), _p(res), _p(y), _p(o_save), _p(lse), ctypes.byref(params), ctypes.byref(seqmap), d, heads,
             dim_head, float(softmax_scale), float(out_scale), eps, drop_p, drop_seed, ARITH[arith],
             _p(ws), ws.numel() * 4 if ws is not None else 0, _stream(x))
    return y, o_save, lse


def attn_groups_supported(d, heads, dim_head, lib=None):
    """bit 0: attn_fwd_groups serves these dimensions (d = 64 on bf16x3 planes, or d <= 16 in place); bit 1: attn_bwd_groups does too"""
    return int((lib or get_lib()).size("rat_attn_groups_supported", int(d), int(heads), int(dim_head)))


def attn_bwd_groups(x, dy, add, o_save, lse, params, grads, seqmap, d, heads, dim_head, softmax_scale=0.0, out_scale=1.0, eps=1e-5,
                    workspace=None, out=None, dropout=(0.0, 0), lib=None):
    """Backward of attn_fwd_groups in ONE launch (small embedding dimensions): dx = add + LayerNorm-backward(...); `grads` receives the
    gradients in the layer's full-width layout."""
    lib = lib or get_lib()
    _chk(x, name="x"), _chk(dy, name="dy"), _chk(o_save, name="o_save"), _chk(lse, name="lse")
    need = lib.size("rat_attn_bwd_groups_workspace", d, heads, dim_head)
    if need == 0:
        raise ValueError("attn_bwd_groups does not serve d=%d heads=%d dim_head=%d" % (d, heads, dim_head))
    if workspace is None or workspace.numel() * 4 < need:
        workspace = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
    dx = out if out is not None else torch.empty_like(x)
    drop_p, drop_seed = _drop_args(params, dropout)
    lib.call("rat_attn_bwd_groups", _p(x), _p(dy), _p(add), _p(o_save), _p(lse), x.numel() // d, _p(dx), ctypes.byref(params),
             ctypes.byref(grads), _p(workspace), workspace.numel() * 4, ctypes.byref(seqmap), d, heads, dim_head, float(softmax_scale),
             float(out_scale), eps, drop_p, drop_seed, _stream(x))
    return dx, workspace


def attn_fwd_groups(x, res, params, planes, seqmap, d, heads, dim_head, softmax_scale=0.0, out_scale=1.0, save=False, eps=1e-5, out=None,
                    dropout=(0.0, 0), lib=None):
    """Wide heads (heads = G x 8 of width 10; at small embedding dimensions also G x 4 of width 20) in ONE launch: y = out_scale *
    Dropout(to_out(attention(LayerNorm(x)))) + res, the head groups looped over inside each chunk.  -> (y, o_save [G, ntok, 80],
    lse [G, ntok, heads per group]); slice g is what attn_bwd_ex takes for group g."""
    lib = lib or get_lib()
    _chk(x, name="x"), _chk(planes, torch.uint8, "planes")                    # planes: None at small embedding dimensions
    y = out if out is not None else torch.empty_like(x)
    per = 80 // dim_head                                     # heads per group: 8 x 10, or 4 x 20 (RAT_m3's halved head count; small d only)
    ntok, G = x.numel() // d, heads // per
    o_save = lse = None
    if save:
        o_save = torch.empty((G, ntok, 80), dtype=torch.float32, device=x.device)
        lse = torch.empty((G, ntok, per), dtype=torch.float32, device=x.device)
    drop_p, drop_seed = _drop_args(params, dropout)
    lib.call("rat_attn_fwd_groups", _p(x), _p(res), _p(y), _p(o_save), _p(lse), ntok, ctypes.byref(params), _p(planes),
             ctypes.byref(seqmap), d, heads, dim_head, float(softmax_scale), float(out_scale), eps, drop_p, drop_seed, _stream(x))
    return y, o_save, lse


def attn_bwd_ex(x, dy, add, o_save, lse, params, grads, seqmap, d, heads, dim_head, softmax_scale=0.0, out_scale=1.0, eps=1e-5,
                workspace=None, out=None, arith="f32", dropout=(0.0, 0), lib=None):
    """dx = add + LayerNorm-backward(... out_scale * dy ...) (add: tensor laid out like x, or None)."""
    lib = lib or get_lib()
    _chk(x, name="x"), _chk(dy, name="dy")
    need = lib.size("rat_attn_bwd_workspace", d, heads, dim_head)
    if workspace is None or workspace.numel() * 4 < need:
        workspace = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
    dx = out if out is not None else torch.empty_like(x)
    drop_p, drop_seed = _drop_args(params, dropout)
    lib.call("rat_attn_bwd_ex", _p(x), _p(dy), _p(add), _p(o_save), _p(lse), _p(dx), ctypes.byref(params), ctypes.byref(grads),
             _p(workspace), workspace.numel() * 4, ctypes.byref(seqmap), d, heads, dim_head, float(softmax_scale),
             float(out_scale), eps, drop_p, drop_seed, ARITH[arith], _stream(x))
    return dx, workspace


def attn_fused_supported(d, heads, dim_head, L, lib=None):
    """True when rat_attn_fwd / rat_attn_bwd serve these dimensions (else: the composed path, see model._attn_composed_*)."""
    lib = lib or get_lib()
    return bool(lib.size("rat_attn_fused_supported", int(d), int(heads), int(dim_head), int(L)))


def attn_core_fwd_map(qkv, seqmap, heads, dim_head, softmax_scale=0.0, save=True, lib=None):
    """attn_core_fwd with RatSeqMap addressing (strided sequences); qkv is [ntok, 3*heads*dim_head] over the WHOLE token grid."""
    lib = lib or get_lib()
    _chk(qkv, name="qkv")
    ntok = qkv.shape[0]
    o = torch.empty((ntok, heads * dim_head), dtype=torch.float32, device=qkv.device)
    lse = torch.empty((ntok, heads), dtype=torch.float32, device=qkv.device) if save else None
    lib.call("rat_attn_core_fwd_map", _p(qkv), _p(o), _p(lse), ctypes.byref(seqmap), heads, dim_head, float(softmax_scale), _stream(qkv))
    return o, lse


def attn_core_bwd_map(qkv, o, lse, dout, seqmap, heads, dim_head, softmax_scale=0.0, lib=None):
    lib = lib or get_lib()
    _chk(qkv, name="qkv"), _chk(o, name="o"), _chk(lse, name="lse"), _chk(dout, name="dout")
    dqkv = torch.empty_like(qkv)
    lib.call("rat_attn_core_bwd_map", _p(qkv), _p(o), _p(lse), _p(dout), _p(dqkv), ctypes.byref(seqmap), heads, dim_head,
             float(softmax_scale), _stream(qkv))
    return dqkv


def attn_core_fwd(qkv, nseq, L, heads, dim_head, softmax_scale=0.0, save=True, lib=None):
    """softmax(Q K^T * scale) V on projected rows qkv [nseq*L, 3*heads*dim_head] -> (o [ntok, heads*dim_head], lse [ntok, heads])."""
    lib = lib or get_lib()
    _chk(qkv, name="qkv")
    ntok = nseq * L
    o = torch.empty((ntok, heads * dim_head), dtype=torch.float32, device=qkv.device)
    lse = torch.empty((ntok, heads), dtype=torch.float32, device=qkv.device) if save else None
    lib.call("rat_attn_core_fwd", _p(qkv), _p(o), _p(lse), int(nseq), int(L), heads, dim_head, float(softmax_scale), _stream(qkv))
    return o, lse


def attn_core_bwd(qkv, o, lse, dout, nseq, L, heads, dim_head, softmax_scale=0.0, lib=None):
    lib = lib or get_lib()
    _chk(qkv, name="qkv"), _chk(o, name="o"), _chk(lse, name="lse"), _chk(dout, name="dout")
    dqkv = torch.empty_like(qkv)
    lib.call("rat_attn_core_bwd", _p(qkv), _p(o), _p(lse), _p(dout), _p(dqkv), int(nseq), int(L), heads, dim_head,
             float(softmax_scale), _stream(qkv))
    return dqkv


def ffn_fwd(x, w1, b1, w2, b2, d, hidden, out=None, arith="f32", lib=None):
    if arith != "f32":
        return ffn_fwd_res(x, x, w1, b1, w2, b2, d, hidden, out=out, arith=arith, lib=lib)
    lib = lib or get_lib()
    _chk(x, name="x")
    y = out if out is not None else torch.empty_like(x)
    lib.call("rat_ffn_fwd", _p(x), _p(y), _p(w1), _p(b1), _p(w2), _p(b2), x.numel() // d, d, hidden, _stream(x))
    return y


def ffn_bwd(x, dy, w1, b1, w2, b2, dw1, db1, dw2, db2, d, hidden, workspace=None, arith="f32", planes=None, lib=None):
    if arith != "f32":
        return ffn_bwd_res(x, dy, w1, b1, w2, b2, dw1, db1, dw2, db2, d, hidden, True, workspace=workspace, arith=arith, planes=planes,
                           lib=lib)
    lib = lib or get_lib()
    _chk(x, name="x"), _chk(dy, name="dy")
    need = lib.size("rat_ffn_bwd_workspace", d, hidden)
    if workspace is None or workspace.numel() * 4 < need:
        workspace = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    lib.call("rat_ffn_bwd", _p(x), _p(dy), _p(dx), _p(w1), _p(b1), _p(w2), _p(b2), _p(dw1), _p(db1), _p(dw2), _p(db2),
             _p(workspace), workspace.numel() * 4, x.numel() // d, d, hidden, _stream(x))
    return dx, workspace


def ffn_fwd_res(x, res, w1, b1, w2, b2, d, hidden, out=None, arith="f32", dropout=None, lib=None):
    """y = FFN(x) + res (res None: no residual) — rat_ffn_fwd_res.  dropout = (p, word1, word2): FeedForward's two Dropout layers in
    training mode (rat_ffn_fwd_drop; exact fp32, generic kernels)."""
    lib = lib or get_lib()
    _chk(x, name="x")
    if res is not None:
        _chk(res, name="res")
    y = out if out is not None else torch.empty_like(x)
    if dropout is not None and dropout[0] > 0:
        _chk(dropout[1], torch.int64, "seed word"), _chk(dropout[2], torch.int64, "seed word")
        lib.call("rat_ffn_fwd_drop", _p(x), _p(res), _p(y), _p(w1), _p(b1), _p(w2), _p(b2), x.numel() // d, d, hidden, float(dropout[0]),
                 _p(dropout[1]), _p(dropout[2]), _stream(x))
        return y
    lib.call("rat_ffn_fwd_res", _p(x), _p(res), _p(y), _p(w1), _p(b1), _p(w2), _p(b2), x.numel() // d, d, hidden, ARITH[arith],
             _stream(x))
    return y


def ffn_bwd_res(x, dy, w1, b1, w2, b2, dw1, db1, dw2, db2, d, hidden, add_dy, workspace=None, arith="f32", planes=None, dropout=None,
                lib=None):
    lib = lib or get_lib()
    _chk(x, name="x"), _chk(dy, name="dy")
    need = lib.size("rat_ffn_bwd_workspace", d, hidden)
    if workspace is None or workspace.numel() * 4 < need:
        workspace = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    if dropout is not None and dropout[0] > 0:
        lib.call("rat_ffn_bwd_drop", _p(x), _p(dy), _p(dx), _p(w1), _p(b1), _p(w2), _p(b2), _p(dw1), _p(db1), _p(dw2), _p(db2),
                 _p(workspace), workspace.numel() * 4, x.numel() // d, d, hidden, int(bool(add_dy)), float(dropout[0]), _p(dropout[1]),
                 _p(dropout[2]), _stream(x))
        return dx, workspace
    lib.call("rat_ffn_bwd_res", _p(x), _p(dy), _p(dx), _p(w1), _p(b1), _p(w2), _p(b2), _p(dw1), _p(db1), _p(dw2), _p(db2),
             _p(workspace), workspace.numel() * 4, _p(planes), x.numel() // d, d, hidden, int(bool(add_dy)), ARITH[arith], _stream(x))
    return dx, workspace


def ffn_bwd_rows_supported(d, hidden, arith, lib=None):
    return bool((lib or get_lib()).cdll.rat_ffn_bwd_rows_supported(int(d), int(hidden), ARITH[arith]))


def ffn_bwd_rows(x, dy_rows, period, w1, b1, w2, b2, dw1, db1, dw2, db2, d, hidden, workspace=None, arith="bf16x3", planes=None, lib=None):
    """ffn_bwd (residual from x itself) where dy is zero except on the token rows k * period, held compactly in dy_rows [*, d]"""
    lib = lib or get_lib()
    _chk(x, name="x"), _chk(dy_rows, name="dy_rows")
    ntok = x.numel() // d
    assert dy_rows.numel() == (ntok + period - 1) // period * d, (tuple(dy_rows.shape), ntok, period)
    need = lib.size("rat_ffn_bwd_workspace", d, hidden)
    if workspace is None or workspace.numel() * 4 < need:
        workspace = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    lib.call("rat_ffn_bwd_res_rows", _p(x), _p(dy_rows), int(period), _p(dx), _p(w1), _p(b1), _p(w2), _p(b2), _p(dw1), _p(db1), _p(dw2),
             _p(db2), _p(workspace), workspace.numel() * 4, _p(planes), ntok, d, hidden, 1, ARITH[arith], _stream(x))
    return dx, workspace


class deferred_reductions:
    """with deferred_reductions(tensor_on_the_stream, lib): ... — the gradient-slab reductions of the attention / feed-forward backward
    calls inside run as ONE launch at the end (rat_reduce_defer_begin / _end).  Every call inside needs a workspace of its own."""

    def __init__(self, like, lib=None, enabled=True):
        self.lib, self.like, self.enabled = lib or get_lib(), like, enabled

    def __enter__(self):
        if self.enabled:
            self.lib.call("rat_reduce_defer_begin")
        return self

    def __exit__(self, exc_type, exc, tb):
        if self.enabled:
            self.lib.call("rat_reduce_defer_end", _stream(self.like), 0 if exc_type is not None else 1)
        return False


# ----------------------------------------------------------------------------- K2c
def layernorm_fwd(x, x_stride, nrows, gamma, beta, d, eps=1e-5, lib=None):
    """LayerNorm of rows x[r * x_stride : r * x_stride + d] -> compact [nrows, d]."""
    lib = lib or get_lib()
    _chk(x, name="x")
    y = torch.empty((nrows, d), dtype=torch.float32, device=x.device)
    lib.call("rat_layernorm_fwd", _p(x), int(x_stride), _p(y), _p(gamma), _p(beta), int(nrows), d, float(eps), _stream(x))
    return y


def layernorm_bwd(x, x_stride, dy, gamma, dx, dx_stride, dgamma, dbeta, d, add=None, eps=1e-5, lib=None):
    """dx rows (stride dx_stride) = [add +] LN-backward(dy); dgamma / dbeta overwritten."""
    lib = lib or get_lib()
    _chk(x, name="x"), _chk(dy, name="dy"), _chk(dx, name="dx")
    nrows = dy.numel() // d
    need = lib.size("rat_layernorm_bwd_workspace", nrows, d)
    ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=x.device)
    lib.call("rat_layernorm_bwd", _p(x), int(x_stride), _p(dy), _p(gamma), _p(add), _p(dx), int(dx_stride), _p(dgamma),
             _p(dbeta), _p(ws), ws.numel() * 4, int(nrows), d, float(eps), _stream(x))
    return dx


# ----------------------------------------------------------------------------- K3
def sgemm(ta, tb, M, N, K, A, lda, Bm, ldb, C, ldc, bias=None, beta=0.0, arith="f32", lib=None):
    lib = lib or get_lib()
    need = lib.size("rat_sgemm_workspace", M, N, K)
    ws = torch.empty((need + 3) // 4, dtype=torch.float32, device=C.device) if need else None   # long-K / few-tile products: split-K
    lib.call("rat_sgemm_arith", int(ta), int(tb), M, N, K, _p(A), lda, _p(Bm), ldb, _p(C), ldc, _p(bias), float(beta),
             _p(ws), ws.numel() * 4 if ws is not None else 0, 1 if arith == "bf16x3" else 0, _stream(C))


def _bn_ws(N, device, lib):
    return torch.empty((lib.size("rat_bn_workspace", N) + 3) // 4, dtype=torch.float32, device=device)


ACT = {"relu": 0, "none": 1, "identity": 1, "sigmoid": 2, "tanh": 3, "leakyrelu": 4, "elu": 5}     # RAT_ACT_* (include/rat_hip.h)


def bn_relu_fwd(z, gamma, beta, running_mean, running_var, training, use_bn, eps=1e-5, momentum=0.1, act=0, lib=None):
    """[BatchNorm1d] + activation (act: a RAT_ACT_* code, 0 = ReLU)"""
    lib = lib or get_lib()
    _chk(z, name="z")
    M, N = z.shape
    a = torch.empty_like(z)
    save_mean = save_rstd = ws = None
    if use_bn and training:
        save_mean = torch.empty(N, dtype=torch.float32, device=z.device)
        save_rstd = torch.empty(N, dtype=torch.float32, device=z.device)
        ws = _bn_ws(N, z.device, lib)
    lib.call("rat_bn_relu_fwd", _p(z), _p(a), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(save_mean),
             _p(save_rstd), _p(ws), M, N, int(training), int(use_bn), eps, momentum, int(act), _stream(z))
    return a, save_mean, save_rstd


def bn_strip_ok(M, N, lib=None):
    return bool((lib or get_lib()).cdll.rat_bn_strip_ok(int(M), int(N)))


def bn_act_fwd_strip(z, gamma, beta, running_mean, running_var, training, use_bn, eps=1e-5, momentum=0.1, act=0, lib=None):
    """bn_relu_fwd in one launch (column strips; N % 4 == 0)"""
    lib = lib or get_lib()
    _chk(z, name="z")
    M, N = z.shape
    a = torch.empty_like(z)
    save_mean = save_rstd = None
    if use_bn and training:
        save = torch.empty((2, (N + 3) // 4 * 4), dtype=torch.float32, device=z.device)
        save_mean, save_rstd = save[0, :N], save[1, :N]
    lib.call("rat_bn_act_fwd_strip", _p(z), _p(a), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(save_mean),
             _p(save_rstd), M, N, int(training), int(use_bn), eps, momentum, int(act), _stream(z))
    return a, save_mean, save_rstd


def bn_act_bwd_strip(z, a, da, gamma, save_mean, save_rstd, dgamma, dbeta, dbias_lin, use_bn, act=0, lib=None):
    """bn_relu_bwd + colsum(dz) -> dbias_lin in one launch"""
    lib = lib or get_lib()
    M, N = z.shape
    dz = torch.empty_like(z)
    lib.call("rat_bn_act_bwd_strip", _p(z), _p(a), _p(da), _p(dz), _p(gamma), _p(save_mean), _p(save_rstd), _p(dgamma), _p(dbeta),
             _p(dbias_lin), M, N, int(use_bn), int(act), _stream(z))
    return dz


def bn_act_bwd_strip_outer(z, a, dl, w, dw, gamma, save_mean, save_rstd, dgamma, dbeta, dbias_lin, use_bn, act=0, lib=None):
    """bn_act_bwd_strip of the last hidden layer with da = dl (x) w formed in the kernel; dw <- sum_r dl[r] a[r][:]"""
    lib = lib or get_lib()
    M, N = z.shape
    dz = torch.empty_like(z)
    lib.call("rat_bn_act_bwd_strip_outer", _p(z), _p(a), _p(dl), _p(w), _p(dw), _p(dz), _p(gamma), _p(save_mean), _p(save_rstd),
             _p(dgamma), _p(dbeta), _p(dbias_lin), M, N, int(use_bn), int(act), _stream(z))
    return dz


def bn_relu_bwd(z, a, da, gamma, save_mean, save_rstd, dgamma, dbeta, use_bn, act=0, lib=None):
    lib = lib or get_lib()
    M, N = z.shape
    dz = torch.empty_like(z)
    ws = _bn_ws(N, z.device, lib) if use_bn else None
    lib.call("rat_bn_relu_bwd", _p(z), _p(a), _p(da), _p(dz), _p(gamma), _p(save_mean), _p(save_rstd), _p(dgamma), _p(dbeta),
             _p(ws), M, N, int(use_bn), int(act), _stream(z))
    return dz


def bn_relu_fwd_sync(z, gamma, beta, running_mean, running_var, all_gather, eps=1e-5, momentum=0.1, act=0, lib=None):
    """SyncBN training forward: local (mean, M2, count) -> `all_gather(stats) -> [world, 2N+1]` (the caller's
    collective: torch.distributed over RCCL / gloo) -> normalise with the GLOBAL batch statistics + ReLU."""
    lib = lib or get_lib()
    _chk(z, name="z")
    M, N = z.shape
    stats = torch.empty(2 * N + 1, dtype=torch.float32, device=z.device)
    ws = _bn_ws(N, z.device, lib)
    lib.call("rat_bn_local_stats", _p(z), _p(stats), _p(ws), M, N, _stream(z))
    all_stats = all_gather(stats)
    _chk(all_stats, name="all_stats")
    world = all_stats.numel() // (2 * N + 1)
    a = torch.empty_like(z)
    save_mean = torch.empty(N, dtype=torch.float32, device=z.device)
    save_rstd = torch.empty(N, dtype=torch.float32, device=z.device)
    lib.call("rat_bn_relu_fwd_sync", _p(z), _p(a), _p(gamma), _p(beta), _p(running_mean), _p(running_var), _p(save_mean),
             _p(save_rstd), _p(all_stats), world, M, N, eps, momentum, int(act), _stream(z))
    return a, save_mean, save_rstd, all_stats


def bn_relu_bwd_sync(z, a, da, gamma, save_mean, save_rstd, dgamma, dbeta, all_reduce_sum, all_stats, act=0, lib=None):
    """SyncBN backward: local (sum g, sum g*xhat) -> `all_reduce_sum(copy)` (caller's collective) -> dz; dgamma/dbeta = LOCAL sums."""
    lib = lib or get_lib()
    M, N = z.shape
    local = torch.empty(2 * N, dtype=torch.float32, device=z.device)
    ws = _bn_ws(N, z.device, lib)
    lib.call("rat_bn_bwd_local_sums", _p(z), _p(a), _p(da), _p(save_mean), _p(save_rstd), _p(local), _p(ws), M, N, int(act), _stream(z))
    glob = all_reduce_sum(local.clone())
    dz = torch.empty_like(z)
    lib.call("rat_bn_relu_bwd_sync", _p(z), _p(a), _p(da), _p(dz), _p(gamma), _p(save_mean), _p(save_rstd), _p(local), _p(glob),
             _p(dgamma), _p(dbeta), _p(all_stats), all_stats.numel() // (2 * N + 1), M, N, int(act), _stream(z))
    return dz


def colsum(a, lda, out, M, N, lib=None):
    lib = lib or get_lib()
    ws = torch.empty((lib.size("rat_colsum_workspace", M, N) + 3) // 4, dtype=torch.float32, device=out.device)
    lib.call("rat_colsum", _p(a), lda, _p(out), _p(ws), M, N, _stream(out))


def logit_fwd(cls, cls_stride, fc_w, fc_b, dnn_out, lr_ftab, nfields, idx, idx_stride, y_true, loss_sum, B, d, head=0, dnn_last=None,
              lib=None):
    """head: 0 = sigmoid + binary cross-entropy, 1 = no output activation + mean squared error (task "regression")
    dnn_last = (a [B][K], lda, K, w [1][K], b [1]): the DNN's one-output Linear is evaluated inside the launch (dnn_out must be None)"""
    lib = lib or get_lib()
    y_pred = torch.empty((B, 1), dtype=torch.float32, device=fc_w.device)
    if dnn_last is not None:
        assert dnn_out is None
        a, lda, K, w, b = dnn_last
        lib.call("rat_logit_fwd_dnn", _p(cls), cls_stride, _p(fc_w), _p(fc_b), _p(a), lda, _p(w), _p(b), K, _p(lr_ftab), nfields,
                 _p(idx), idx_stride, _p(y_true), _p(y_pred), _p(loss_sum), B, d, int(head), _stream(fc_w))
        return y_pred
    lib.call("rat_logit_fwd", _p(cls), cls_stride, _p(fc_w), _p(fc_b), _p(dnn_out), _p(lr_ftab), nfields, _p(idx),
             idx_stride, _p(y_true), _p(y_pred), _p(loss_sum), B, d, int(head), _stream(fc_w))
    return y_pred


def logit_bwd(y_pred, y_true, cls, cls_stride, fc_w, dcls, dcls_stride, dfc_w, dfc_b, lr_gftab, nfields, idx, idx_stride,
              gscale, B, d, gscale_dev=None, head=0, ddnn_b=None, lib=None):
    """gscale: host factor; gscale_dev: optional DEVICE scalar multiplied in by the kernel (autograd's incoming gradient).
    ddnn_b: also accumulate sum_b dlogit[b] there (bias gradient of the DNN's one-output Linear)"""
    lib = lib or get_lib()
    dlogit = torch.empty((B, 1), dtype=torch.float32, device=fc_w.device)
    if ddnn_b is not None:
        lib.call("rat_logit_bwd_dnn", _p(y_pred), _p(y_true), _p(cls), cls_stride, _p(fc_w), _p(dlogit), _p(dcls), dcls_stride,
                 _p(dfc_w), _p(dfc_b), _p(ddnn_b), _p(lr_gftab), nfields, _p(idx), idx_stride, float(gscale), _p(gscale_dev), B, d,
                 int(head), _stream(fc_w))
        return dlogit
    lib.call("rat_logit_bwd", _p(y_pred), _p(y_true), _p(cls), cls_stride, _p(fc_w), _p(dlogit), _p(dcls), dcls_stride,
             _p(dfc_w), _p(dfc_b), _p(lr_gftab), nfields, _p(idx), idx_stride, float(gscale), _p(gscale_dev), B, d, int(head),
             _stream(fc_w))
    return dlogit


# ----------------------------------------------------------------------------- K4 / K5
def l2_reg(w, g, lam, reg_out, lam_scale_dev=None, lib=None):
    lib = lib or get_lib()
    lib.call("rat_l2_reg", _p(w), _p(g), w.numel(), float(lam), _p(lam_scale_dev), _p(reg_out), _stream(w))


def sumsq(g, out, lib=None):
    lib = lib or get_lib()
    lib.call("rat_sumsq", _p(g), g.numel(), _p(out), _stream(g))


def clip_adam(w, g, m, v, norm_sq, max_norm, lr, beta1, beta2, eps, step, lib=None):
    lib = lib or get_lib()
    lib.call("rat_clip_adam", _p(w), _p(g), _p(m), _p(v), w.numel(), _p(norm_sq), float(max_norm), float(lr), float(beta1),
             float(beta2), float(eps), int(step), _stream(w))


def clip_opt(w, g, state, norm_sq, max_norm, lr, kind, p0, eps, lib=None):
    """SGD / Adagrad / RMSprop (kind 1 / 2 / 3) on a flat buffer, after the clip factor derived from norm_sq"""
    lib = lib or get_lib()
    lib.call("rat_clip_opt", _p(w), _p(g), _p(state), w.numel(), _p(norm_sq), float(max_norm), float(lr), int(kind), float(p0), float(eps),
             _stream(w))


def clip_opt_fused(w, g, state, n_split, lam_a, lam_b, norm_sq, max_norm, hyper, kind, p0, eps, zero_g=True, lib=None):
    lib = lib or get_lib()
    lib.call("rat_clip_opt_fused", _p(w), _p(g), _p(state), w.numel(), int(n_split), float(lam_a), float(lam_b), None, _p(norm_sq),
             float(max_norm), _p(hyper), int(kind), float(p0), float(eps), int(bool(zero_g)), _stream(w))


def dropout(x, p, seed, out=None, lib=None):
    """seed: an int (by value) or a one-element int64 DEVICE tensor — one of the words `dropout_seeds` refreshes per step"""
    lib = lib or get_lib()
    _chk(x, name="x")
    y = out if out is not None else torch.empty_like(x)
    if torch.is_tensor(seed):
        _chk(seed, torch.int64, "seed word")
        lib.call("rat_dropout_dev", _p(x), _p(y), x.numel(), float(p), _p(seed), _stream(x))
    else:
        lib.call("rat_dropout", _p(x), _p(y), x.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, _stream(x))
    return y


def dropout_seeds(words, base_seed, counter, lib=None):
    """counter += 1; words[i] = mix(base_seed, counter, i): the dropout state of one training step, on the device"""
    lib = lib or get_lib()
    _chk(words, torch.int64, "seed words"), _chk(counter, torch.int64, "counter")
    lib.call("rat_dropout_seeds", _p(words), words.numel(), int(base_seed) & 0xFFFFFFFFFFFFFFFF, _p(counter), _stream(words))


def _drop_args(params, dropout):
    """(p, seed) -> the by-value arguments; a tensor seed goes into the parameter struct (RatAttnParams.drop_seed_dev, ABI v6)"""
    p, seed = dropout
    if torch.is_tensor(seed):
        _chk(seed, torch.int64, "seed word")
        params.drop_seed_dev = seed.data_ptr()
        return float(p), 0
    params.drop_seed_dev = None
    return float(p), int(seed) & 0xFFFFFFFFFFFFFFFF


# ----------------------------------------------------------------------------- ABI v4: two-sweep optimizer, device-side clock
def adam_tick(step_dev, lr_dev, beta1, beta2, hyper, lib=None):
    """*step_dev += 1; hyper[0..2] = lr/(1-beta1^t), 1/sqrt(1-beta2^t), lr"""
    lib = lib or get_lib()
    _chk(step_dev, torch.int32, "step_dev"), _chk(lr_dev, name="lr_dev"), _chk(hyper, name="hyper")
    lib.call("rat_adam_tick", _p(step_dev), _p(lr_dev), float(beta1), float(beta2), _p(hyper), _stream(hyper))


def step_begin(step_dev, lr_dev, beta1, beta2, hyper, scalars, counters=None, lib=None):
    """adam_tick + scalars[:] = 0 + counters[:] += 1 (int64) in one launch"""
    lib = lib or get_lib()
    _chk(step_dev, torch.int32, "step_dev"), _chk(lr_dev, name="lr_dev"), _chk(hyper, name="hyper"), _chk(scalars, name="scalars")
    if counters is not None:
        _chk(counters, torch.int64, "counters")
    lib.call("rat_step_begin", _p(step_dev), _p(lr_dev), float(beta1), float(beta2), _p(hyper), _p(scalars), scalars.numel(),
             _p(counters), counters.numel() if counters is not None else 0, _stream(hyper))


def sumsq_reg(g, w, n_split, lam_a, lam_b, norm_sq_out, reg_out=None, lam_scale_dev=None, lib=None):
    lib = lib or get_lib()
    assert g.numel() == w.numel()
    lib.call("rat_sumsq_reg", _p(g), _p(w), g.numel(), int(n_split), float(lam_a), float(lam_b), _p(lam_scale_dev), _p(norm_sq_out),
             _p(reg_out), _stream(g))


def clip_adam_fused(w, g, m, v, n_split, lam_a, lam_b, norm_sq, max_norm, hyper, beta1, beta2, eps, zero_g=True, lam_scale_dev=None,
                    lib=None):
    lib = lib or get_lib()
    lib.call("rat_clip_adam_fused", _p(w), _p(g), _p(m), _p(v), w.numel(), int(n_split), float(lam_a), float(lam_b), _p(lam_scale_dev),
             _p(norm_sq), float(max_norm), _p(hyper), float(beta1), float(beta2), float(eps), 1 if zero_g else 0, _stream(w))


def adam_rows_dev(w_base, m_base, v_base, rows, grads, count, max_rows, d, norm_sq, max_norm, hyper, beta1, beta2, eps, lib=None):
    lib = lib or get_lib()
    lib.call("rat_adam_rows_dev", _p(w_base), _p(m_base), _p(v_base), _p(rows), _p(grads), _p(count), int(max_rows), d, _p(norm_sq),
             float(max_norm), _p(hyper), float(beta1), float(beta2), float(eps), _stream(grads))


def scatter_rows_lists(dense_base, rows, grads, counts, d, lib=None):
    """rows [lists, cap] int32, grads [lists, cap, d], counts [lists] int32 -> the zeroed dense gradient block"""
    lib = lib or get_lib()
    _chk(rows, torch.int32, "rows"), _chk(counts, torch.int32, "counts"), _chk(grads, name="grads")
    lib.call("rat_scatter_rows_lists", _p(dense_base), _p(rows), _p(grads), _p(counts), rows.shape[1], rows.shape[0], int(d), _stream(grads))


def scatter_rows(dense_base, rows, grads, count, d, lib=None):
    """merged (unique rows, gradient rows) lists -> the zeroed dense gradient block"""
    lib = lib or get_lib()
    _chk(rows, torch.int32, "rows"), _chk(count, torch.int32, "count")
    lib.call("rat_scatter_rows", _p(dense_base), _p(rows), _p(grads), _p(count), rows.numel(), int(d), _stream(grads))


# ---- owner-partitioned exchange of the row lists (data parallelism; include/rat_hip.h ABI v8) -----------------------------------------
def owner_counts(plan, rows_per_owner, world, out, lib=None):
    """out[k] (int32 [world]) = unique rows of `plan` in owner k's range [k per, (k + 1) per)"""
    lib = lib or get_lib()
    _chk(out, torch.int32, "out")
    assert out.numel() == world
    lib.call("rat_owner_counts", _p(plan.ws), _p(plan.count), int(plan.n), int(rows_per_owner), int(world), _p(out), _stream(out))


def owner_pack(mat, world, rank, d, rows_a, grads_a, rows_b, vals_b, max_pairs, wire, lib=None):
    lib = lib or get_lib()
    _chk(mat, torch.int32, "mat"), _chk(rows_a, torch.int32, "rows_a"), _chk(grads_a, name="grads_a"), _chk(wire, name="wire")
    lib.call("rat_owner_pack", _p(mat), int(world), int(rank), int(d), _p(rows_a), _p(grads_a), _p(rows_b), _p(vals_b), int(max_pairs),
             _p(wire), _stream(wire))


def owner_unpack(mat, world, rank, d, wire, max_pairs, rows_a, grads_a, rows_b, vals_b, totals, extra_src=None, extra_dst=None, lib=None):
    lib = lib or get_lib()
    _chk(mat, torch.int32, "mat"), _chk(rows_a, torch.int32, "rows_a"), _chk(grads_a, name="grads_a"), _chk(wire, name="wire")
    _chk(totals, torch.int32, "totals")
    n_extra = 0 if extra_src is None else extra_src.numel()
    lib.call("rat_owner_unpack", _p(mat), int(world), int(rank), int(d), _p(wire), int(max_pairs), _p(rows_a), _p(grads_a), _p(rows_b),
             _p(vals_b), _p(totals), _p(extra_src), _p(extra_dst), int(n_extra), _stream(wire))


def owner_scatter(dense_a, dense_b, extra_out, lists, stride, world, cap_a, cap_b, d, n_extra, lib=None):
    lib = lib or get_lib()
    _chk(lists, name="lists")
    lib.call("rat_owner_scatter", _p(dense_a), _p(dense_b), _p(extra_out), _p(lists), int(stride), int(world), int(cap_a), int(cap_b),
             int(d), int(n_extra), _stream(lists))
