#!/usr/bin/env python3
"""Diagnostic: where does a chunk iteration of the attention / FFN kernels spend its cycles?

Loads the -DRAT_PROF build (www24-rat_amd/lib/librat_hip_prof.so: thread 0 of every work-group accumulates s_memtime
deltas per phase), runs a few training steps of the bench workload and prints the SHARE of each phase.  Read shares,
not absolute time (the stamps serialise a little).  Never used by the product or the tests."""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "www24-rat_amd")):
    sys.path.insert(0, p)
os.environ["RAT_HIP_LIBRARY"] = os.path.join(ROOT, "www24-rat_amd", "lib", "librat_hip_prof.so")
import build as _build  # noqa: E402

_build.build(prof=True)          # a no-op when the stamped library matches the sources (a stale one would fail the ABI check)

import torch  # noqa: E402
from rat_amd import synthetic  # noqa: E402
from rat_amd.base_model import seed_everything  # noqa: E402
from rat_amd.model import RAT_m2  # noqa: E402

PHASES = {
    0: ("attn_fwd", ["load+LN", "QKV gemm", "softmax(QK)V valu", "out-proj+store"]),
    12: ("attn_bwd", ["map rows", "load x,dy,O,lse", "LN", "QKV gemm (wave 0)", "dO gemm (wave 0)", "dWout+colsum+barrier", "pass1 (dQ)",
                      "pass2 (dK,dV) + copy", "dXn gemm (wave 0)", "dWqkv + barrier", "LN bwd + store"]),
    24: ("ffn_fwd", ["load", "W1 gemm + gelu", "W2 gemm + store"]),
    36: ("ffn_bwd", ["load", "W1 gemm + gelu", "dW2", "dh gemm", "dx gemm + dW1"]),
    48: ("attn_fwd bf16x3", ["LN -> planes", "QKV gemm (+ operand layouts)", "softmax(QK)V core", "O -> planes", "out-proj", "store"]),
    72: ("ffn_bwd_t4 (wave 0; RAT_FFN_BWD=t3: consume + stage | chain | dx partial | barrier 1 | dx store + prefetch | dW | barrier 2)",
         ["split x / dy -> planes", "barrier 1", "h / dh chain + gelu (hidden tile 0)", "barrier 2", "prefetch issue + dx gemm + store", "dW1, dW2",
          "barrier 3"]),
    84: ("attn_bwd wide heads, small d (attn_bwd_wide_kernel)",
         ["chunk: x, dy, LayerNorm", "group: O, lse, weights -> LDS", "QKV, dO, dW_out GEMMs", "pass 1 (dQ)", "pass 2 (dK, dV)", "dQ copy",
          "d(LN out) partials + dW_qkv", "partials -> registers", "chunk: LayerNorm backward + store"]),
    60: ("attn_bwd bf16x3", ["loads, LN, planes", "QKV gemm (wave 0)", "dO gemm (wave 0)", "dW_out + barrier", "pass1 (dQ)", "pass2 (dK,dV)",
                             "dQKV -> planes", "dXn gemm (wave 0)", "dW_qkv + barrier", "LN bwd + store"]),
}


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "synthetic_F20_V1M_K10_d64_B4096"
    spec = synthetic.WORKLOADS[name]
    fm = synthetic.feature_map_for(name, spec)
    seed_everything(2021)
    model = RAT_m2(fm, **synthetic.model_kwargs(spec, gpu=0))
    batch = synthetic.make_batch(spec, fm, seed=1, device=model.device)
    model.train()
    model.use_graph = False                    # eager launches: the stamps are read from a buffer the graph would not know about
    model.train_step(batch)
    buf = torch.zeros(96, dtype=torch.int64, device=model.device)
    model._lib.cdll.rat_debug_set_prof(ctypes.c_void_p(buf.data_ptr()))
    for _ in range(3):
        model.train_step(batch)
    torch.cuda.synchronize()
    vals = buf.cpu().tolist()
    for base, (kname, names) in PHASES.items():
        tot = sum(vals[base:base + len(names)]) or 1
        print("%s  (sum over work-groups: %.3e cycles)" % (kname, tot))
        for i, n in enumerate(names):
            print("   %-28s %5.1f %%" % (n, 100.0 * vals[base + i] / tot))


if __name__ == "__main__":
    main()
