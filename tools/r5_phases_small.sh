#!/bin/bash
# round 5: in-kernel phase shares at the shipped small-d geometries (exact-fp32 kernels) and of the wide-head group loop
out=gpurun_out/r5_phases_small
export TMPDIR=/tmp
mkdir -p $out
for w in tmall_real_F9_K5_d10_h32_B4096 movielens_real_F3_K5_d10_B4096 tmall_like_F8_K30_d64_h32_B4096; do
  python3 tools/phase_profile.py $w > $out/$w.txt 2>&1
done
grep -v "0.0 %\|1.000e+00\|amdgpu.ids" $out/*.txt
