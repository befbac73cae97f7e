#!/bin/bash
# Diagnostic: same-box A/B of FFN library variants (tools/variant.sh), 3 interleaved rounds of kbench on the bf16x3 kernels
for round in 1 2 3; do
  for v in "$@"; do
    echo "== $v (round $round)"
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python tools/kbench.py ffn_bwd ffn_fwd --arith bf16x3 --reps 20 2>&1 | grep -v amdgpu.ids
  done
done
