#!/bin/bash
# round 5: attn_fwd3's O -> planes pass in half pieces (hip) against whole pieces (op8 = -DRAT_O_PIECES, round 4)
out=gpurun_out/r5_ohalf
export TMPDIR=/tmp
mkdir -p $out
timeout 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn" > $out/pytest.log 2>&1; echo "rc $?" > $out/rc.txt
for round in 1 2 3; do
  for v in hip op8; do
    echo "== $v (round $round)" >> $out/ab.txt
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 tools/kbench.py attn_fwd --arith bf16x3 --reps 30 2>&1 | grep attn_fwd >> $out/ab.txt
  done
done
cat $out/rc.txt; tail -2 $out/pytest.log; cat $out/ab.txt
