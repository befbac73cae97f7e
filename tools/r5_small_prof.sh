#!/bin/bash
# round 5: rocprofv3 kernel stats of the shipped small-d workloads (where does a step go besides the encoder kernels?)
out=gpurun_out/r5_small_prof
export TMPDIR=/tmp
mkdir -p $out
for w in movielens_real_F3_K5_d10_B4096 tmall_real_F9_K5_d10_h32_B4096; do
  rocprofv3 --kernel-trace --stats -d $out/prof_$w -o r --output-format csv -- python3 bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $out/bench_$w.json 2> $out/rocprof_$w.err
  cp $(find $out/prof_$w -name "*kernel_stats.csv" | head -1) $out/kernel_stats_$w.csv
done
ls $out
