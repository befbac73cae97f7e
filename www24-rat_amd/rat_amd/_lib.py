"""ctypes binding of librat_hip.so (include/rat_hip.h).

The product path has exactly one implementation: the HIP library built by ``www24-rat_amd/build.py``.  If it is
missing or fails to load, importing/using the ops raises — there is no CPU or eager-PyTorch fallback.
(``tests/emu`` may hand a host-emulation build of the same kernel sources to ``RatLib`` directly; nothing in this
package does.)
"""
import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(os.path.dirname(_HERE), "lib", "librat_hip.so")
ABI_VERSION = 9


class RatField(Structure):
    _fields_ = [("table", c_void_p), ("col", c_int32), ("ncols", c_int32), ("vocab", c_int32), ("padding_idx", c_int32)]


class RatSeqMap(Structure):
    _fields_ = [("nseq", c_int64), ("L", c_int32), ("queries", c_int32), ("q_div", c_int64), ("hi_stride", c_int64),
                ("lo_stride", c_int64), ("pos_stride", c_int64)]


class RatAttnParams(Structure):
    _fields_ = [("ln_g", c_void_p), ("ln_b", c_void_p), ("w_qkv", c_void_p), ("w_out", c_void_p), ("b_out", c_void_p),
                ("planes", c_void_p), ("drop_seed_dev", c_void_p)]


class RatSplitJob(Structure):
    _fields_ = [("w", c_void_p), ("out", c_void_p), ("N", c_int32), ("K", c_int32), ("ld", c_int32), ("transpose", c_int32),
                ("perm", c_int32), ("reserved", c_int32)]


class RatError(RuntimeError):
    pass


_P = c_void_p
_SIGNATURES = {
    "rat_version": (c_int, []),
    "rat_last_error": (c_char_p, []),
    "rat_gather_fwd": (c_int, [_P, _P, _P, c_int, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "rat_gather_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, _P, c_int, c_int, c_int, c_int, _P]),
    "rat_batch_prepare": (c_int, [_P, c_int, _P, c_int, _P, _P, _P, c_int, c_int, c_int, _P]),
    "rat_batch_assemble": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int64, c_int, c_int, c_int, _P]),
    "rat_attn_fwd": (c_int, [_P, _P, _P, _P, POINTER(RatAttnParams), POINTER(RatSeqMap), c_int, c_int, c_int, c_float, _P]),
    "rat_attn_bwd_workspace": (c_size_t, [c_int, c_int, c_int]),
    "rat_attn_bwd": (c_int, [_P, _P, _P, _P, _P, POINTER(RatAttnParams), POINTER(RatAttnParams), _P, c_size_t,
                             POINTER(RatSeqMap), c_int, c_int, c_int, c_float, _P]),
    "rat_ffn_fwd": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, _P]),
    "rat_attn_fwd_workspace": (c_size_t, [c_int, c_int, c_int]),
    "rat_attn_fwd_ex": (c_int, [_P, _P, _P, _P, _P, POINTER(RatAttnParams), POINTER(RatSeqMap), c_int, c_int, c_int, c_float, c_float,
                                c_float, c_float, ctypes.c_uint64, c_int, _P, c_size_t, _P]),
    "rat_attn_bwd_ex": (c_int, [_P, _P, _P, _P, _P, _P, POINTER(RatAttnParams), POINTER(RatAttnParams), _P, c_size_t,
                                POINTER(RatSeqMap), c_int, c_int, c_int, c_float, c_float, c_float, c_float, ctypes.c_uint64, c_int, _P]),
    "rat_attn_core_fwd": (c_int, [_P, _P, _P, c_int64, c_int, c_int, c_int, c_float, _P]),
    "rat_attn_core_bwd": (c_int, [_P, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, c_float, _P]),
    "rat_attn_core_fwd_map": (c_int, [_P, _P, _P, POINTER(RatSeqMap), c_int, c_int, c_float, _P]),
    "rat_attn_core_bwd_map": (c_int, [_P, _P, _P, _P, _P, POINTER(RatSeqMap), c_int, c_int, c_float, _P]),
    "rat_attn_fused_supported": (c_int, [c_int, c_int, c_int, c_int]),
    "rat_colsum_workspace": (c_size_t, [c_int, c_int]),
    "rat_bm25_topk": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int64, c_int, c_int, _P]),
    "rat_bm25_topk_grouped": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int64, c_int64, c_int, c_int, _P]),
    "rat_ffn_fwd_res": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_int, _P]),
    "rat_ffn_bwd_res": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P, c_int64, c_int, c_int, c_int, c_int, _P]),
    "rat_reduce_defer_begin": (c_int, []),
    "rat_reduce_defer_end": (c_int, [_P, c_int]),
    "rat_ffn_bwd_rows_supported": (c_int, [c_int, c_int, c_int]),
    "rat_ffn_bwd_res_rows": (c_int, [_P, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, _P, c_int64, c_int, c_int, c_int, c_int, _P]),
    "rat_ffn_fwd_drop": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int64, c_int, c_int, c_float, _P, _P, _P]),
    "rat_ffn_bwd_drop": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, c_int64, c_int, c_int, c_int, c_float, _P, _P, _P]),
    "rat_attn_planes_bytes": (c_size_t, [c_int, c_int, c_int]),
    "rat_attn_split_jobs": (c_int, [POINTER(RatAttnParams), c_int, c_int, c_int, _P, POINTER(RatSplitJob)]),
    "rat_attn_groups_planes_bytes": (c_size_t, [c_int, c_int, c_int]),
    "rat_attn_groups_split_jobs": (c_int, [POINTER(RatAttnParams), c_int, c_int, c_int, _P, POINTER(RatSplitJob)]),
    "rat_attn_fwd_groups": (c_int, [_P, _P, _P, _P, _P, c_int64, POINTER(RatAttnParams), _P, POINTER(RatSeqMap), c_int, c_int, c_int,
                                    c_float, c_float, c_float, c_float, ctypes.c_uint64, _P]),
    "rat_attn_groups_supported": (c_int, [c_int, c_int, c_int]),
    "rat_attn_bwd_groups_workspace": (c_size_t, [c_int, c_int, c_int]),
    "rat_attn_bwd_groups": (c_int, [_P, _P, _P, _P, _P, c_int64, _P, POINTER(RatAttnParams), POINTER(RatAttnParams), _P, c_size_t,
                                    POINTER(RatSeqMap), c_int, c_int, c_int, c_float, c_float, c_float, c_float, ctypes.c_uint64, _P]),
    "rat_ffn_planes_bytes": (c_size_t, [c_int, c_int]),
    "rat_ffn_split_jobs": (c_int, [_P, _P, c_int, c_int, _P, POINTER(RatSplitJob)]),
    "rat_split_weights_batch": (c_int, [POINTER(RatSplitJob), c_int, _P]),
    "rat_layernorm_fwd": (c_int, [_P, c_int64, _P, _P, _P, c_int64, c_int, c_float, _P]),
    "rat_layernorm_bwd_workspace": (c_size_t, [c_int64, c_int]),
    "rat_layernorm_bwd": (c_int, [_P, c_int64, _P, _P, _P, _P, c_int64, _P, _P, _P, c_size_t, c_int64, c_int, c_float, _P]),
    "rat_ffn_bwd_workspace": (c_size_t, [c_int, c_int]),
    "rat_ffn_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_size_t, c_int64, c_int, c_int, _P]),
    "rat_sgemm": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_float, _P]),
    "rat_sgemm_workspace": (c_size_t, [c_int, c_int, c_int]),
    "rat_sgemm_ws": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_float, _P, c_size_t, _P]),
    "rat_sgemm_arith": (c_int, [c_int, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P, c_int, _P, c_float, _P, c_size_t, c_int, _P]),
    "rat_bn_strip_ok": (c_int, [c_int, c_int]),
    "rat_bn_act_fwd_strip": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_int, _P]),
    "rat_bn_act_bwd_strip": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "rat_bn_act_bwd_strip_outer": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "rat_logit_fwd_dnn": (c_int, [_P, c_int64, _P, _P, _P, c_int64, _P, _P, c_int, _P, c_int, _P, c_int64, _P, _P, _P, c_int, c_int, c_int, _P]),
    "rat_logit_bwd_dnn": (c_int, [_P, _P, _P, c_int64, _P, _P, _P, c_int64, _P, _P, _P, _P, c_int, _P, c_int64, c_float, _P, c_int,
                                  c_int, c_int, _P]),
    "rat_bn_workspace": (c_size_t, [c_int]),
    "rat_bn_relu_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_int, _P]),
    "rat_bn_relu_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "rat_colsum": (c_int, [_P, c_int, _P, _P, c_int, c_int, _P]),
    "rat_bn_local_stats": (c_int, [_P, _P, _P, c_int, c_int, _P]),
    "rat_bn_relu_fwd_sync": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_float, c_float, c_int, _P]),
    "rat_bn_bwd_local_sums": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, _P]),
    "rat_bn_relu_bwd_sync": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "rat_logit_fwd": (c_int, [_P, c_int64, _P, _P, _P, _P, c_int, _P, c_int64, _P, _P, _P, c_int, c_int, c_int, _P]),
    "rat_logit_bwd": (c_int, [_P, _P, _P, c_int64, _P, _P, _P, c_int64, _P, _P, _P, c_int, _P, c_int64, c_float, _P, c_int,
                              c_int, c_int, _P]),
    "rat_l2_reg": (c_int, [_P, _P, c_int64, c_float, _P, _P, _P]),
    "rat_label_grad_workspace": (c_size_t, [c_int]),
    "rat_label_grad": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, _P]),
    "rat_sparse_workspace": (c_size_t, [c_int64]),
    "rat_sparse_plan_ids": (c_int, [_P, _P, _P, c_int, _P, c_int, c_int64, c_int, c_int, c_int, c_int, _P, c_size_t, _P, _P]),
    "rat_sparse_plan_rows": (c_int, [_P, _P, c_int64, c_int, c_int64, _P, c_size_t, _P, _P]),
    "rat_sparse_reduce_grid": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "rat_sparse_reduce_rows": (c_int, [_P, _P, _P, c_int64, c_int, c_int, _P, _P, _P]),
    "rat_sparse_reduce_scalar": (c_int, [_P, _P, _P, c_int, c_int, _P, _P, _P, _P]),
    "rat_sumsq_rows": (c_int, [_P, _P, c_int64, c_int, _P, _P]),
    "rat_adam_rows": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int, _P, c_float, c_float, c_float, c_float, c_float, c_int, _P]),
    "rat_check_ids": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "rat_sumsq": (c_int, [_P, c_int64, _P, _P]),
    "rat_dropout": (c_int, [_P, _P, c_int64, c_float, ctypes.c_uint64, _P]),
    "rat_dropout_dev": (c_int, [_P, _P, c_int64, c_float, _P, _P]),
    "rat_dropout_seeds": (c_int, [_P, c_int, ctypes.c_uint64, _P, _P]),
    "rat_clip_adam": (c_int, [_P, _P, _P, _P, c_int64, _P, c_float, c_float, c_float, c_float, c_float, c_int, _P]),
    "rat_adam_tick": (c_int, [_P, _P, c_float, c_float, _P, _P]),
    "rat_step_begin": (c_int, [_P, _P, c_float, c_float, _P, _P, c_int, _P, c_int, _P]),
    "rat_sumsq_reg": (c_int, [_P, _P, c_int64, c_int64, c_float, c_float, _P, _P, _P, _P]),
    "rat_clip_adam_fused": (c_int, [_P, _P, _P, _P, c_int64, c_int64, c_float, c_float, _P, _P, c_float, _P, c_float, c_float, c_float,
                                    c_int, _P]),
    "rat_clip_opt": (c_int, [_P, _P, _P, c_int64, _P, c_float, c_float, c_int, c_float, c_float, _P]),
    "rat_clip_opt_fused": (c_int, [_P, _P, _P, c_int64, c_int64, c_float, c_float, _P, _P, c_float, _P, c_int, c_float, c_float, c_int, _P]),
    "rat_scatter_rows": (c_int, [_P, _P, _P, _P, c_int64, c_int, _P]),
    "rat_scatter_rows_lists": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int, _P]),
    "rat_owner_counts": (c_int, [_P, _P, c_int64, c_int64, c_int, _P, _P]),
    "rat_owner_pack": (c_int, [_P, c_int, c_int, c_int, _P, _P, _P, _P, c_int64, _P, _P]),
    "rat_owner_unpack": (c_int, [_P, c_int, c_int, c_int, _P, c_int64, _P, _P, _P, _P, _P, _P, _P, c_int, _P]),
    "rat_owner_scatter": (c_int, [_P, _P, _P, _P, c_int64, c_int, c_int64, c_int64, c_int, c_int, _P]),
    "rat_adam_rows_dev": (c_int, [_P, _P, _P, _P, _P, _P, c_int64, c_int, _P, c_float, _P, c_float, c_float, c_float, _P]),
}
EXPORTED_SYMBOLS = tuple(_SIGNATURES)


class RatLib:
    """A loaded librat_hip.so with typed entry points; ``call(name, *args)`` raises RatError on a negative return."""

    def __init__(self, path=None):
        self.path = path or os.environ.get("RAT_HIP_LIBRARY", DEFAULT_LIB)
        if not os.path.exists(self.path):
            raise RatError("librat_hip.so not found at %s — build it with `python www24-rat_amd/build.py` "
                           "(hipcc --offload-arch=gfx950).  There is no fallback path." % self.path)
        import torch  # noqa: F401 — FIRST: torch loads its bundled libamdhip64 (soname libamdhip64.so.7); loading
        #                 librat_hip.so afterwards binds it to that same HIP runtime instead of a second copy.
        self.cdll = ctypes.CDLL(self.path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(self.cdll, name)          # AttributeError if the library does not export the symbol
            fn.restype = res
            fn.argtypes = args
        v = self.cdll.rat_version()
        if v != ABI_VERSION:
            raise RatError("librat_hip.so ABI version %d != expected %d" % (v, ABI_VERSION))

    def last_error(self):
        msg = self.cdll.rat_last_error()
        return msg.decode() if msg else ""

    def call(self, name, *args):
        rc = getattr(self.cdll, name)(*args)
        if rc != 0:
            raise RatError("%s failed (%d): %s" % (name, rc, self.last_error()))

    def size(self, name, *args):
        return int(getattr(self.cdll, name)(*args))


_default = None


def get_lib():
    """The process-wide HIP library (loaded after torch so that it binds to torch's HIP runtime)."""
    global _default
    if _default is None:
        import torch  # noqa: F401  (loads libamdhip64 first; librat_hip.so then resolves against the same runtime)
        _default = RatLib()
    return _default
