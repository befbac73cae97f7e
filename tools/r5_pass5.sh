#!/bin/bash
# round 5, fifth pass: dW_out of attn_bwd3 dealt over all 8 waves (hip) against round 4's assignment (dwr4)
out=gpurun_out/r5_pass5
export TMPDIR=/tmp
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn" > $out/pytest_attn.log 2>&1; echo "attn rc $?" > $out/rc.txt
for round in 1 2 3; do
  for v in hip dwr4; do
    echo "== $v (round $round)" >> $out/ab_attn.txt
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 tools/kbench.py attn_bwd --arith bf16x3 --reps 30 2>&1 | grep -v amdgpu.ids >> $out/ab_attn.txt
  done
done
cat $out/rc.txt; tail -2 $out/pytest_attn.log; cat $out/ab_attn.txt
