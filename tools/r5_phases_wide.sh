#!/bin/bash
out=gpurun_out/r5_phases_wide
export TMPDIR=/tmp
mkdir -p $out
python3 tools/phase_profile.py tmall_real_F9_K5_d10_h32_B4096 > $out/tmall_real.txt 2>&1
grep -v "0.0 %\|1.000e+00\|amdgpu.ids" $out/tmall_real.txt
