#!/bin/bash
# round 5: sgemm3_kernel with two 32-wide k sub-tiles per trip (hip) against one (k1), the head's nine GEMM shapes at B = 4096 and B = 512
out=gpurun_out/r5_sgemm
export TMPDIR=/tmp
mkdir -p $out
timeout 600 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "sgemm" > $out/pytest.log 2>&1; echo "rc $?" > $out/rc.txt
for round in 1 2 3; do
  for v in hip k1; do
    for b in 4096 512; do
      echo "== $v B=$b (round $round)" >> $out/ab.txt
      KBENCH_B=$b RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 tools/kbench.py sgemm 2>&1 | grep "bf16x3" >> $out/ab.txt
    done
  done
done
cat $out/rc.txt; tail -2 $out/pytest.log; grep -A 10 "round 1" $out/ab.txt | head -80; echo ...; grep "total\|==" $out/ab.txt
