#!/bin/bash
# Diagnostic: same-box A/B of library variants (box-to-box spread is ~3-5 %, more than most single changes).
#   tools/ab_bench.sh "attn_bwd" base v1 v2      -> alternates www24-rat_amd/lib/librat_<name>.so, 3 rounds
what="$1"; shift
for round in 1 2 3; do
  for v in "$@"; do
    echo "== $v (round $round)"
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python tools/kbench.py $what --reps 20 2>&1 | grep -v amdgpu.ids
  done
done
