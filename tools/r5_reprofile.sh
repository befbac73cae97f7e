#!/bin/bash
# round 5: the rocprofv3 kernel-stats passes once more on the final library (after the sumsq / slab-reduction / gather changes)
out=gpurun_out/r5_reprof
export TMPDIR=/tmp
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/prof -o r --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_under_rocprof.json 2> $out/rocprof.err
rocprofv3 --kernel-trace --stats -d $out/prof_b512 -o r --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --batch 512 > $out/bench_b512_under_rocprof.json 2>> $out/rocprof.err
rocprofv3 --kernel-trace --stats -d $out/prof_mltag -o r --output-format csv -- python3 bench.py --workload mltag_like_K10_d16_B256 --steps 50 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_mltag_under_rocprof.json 2>> $out/rocprof.err
rocprofv3 --kernel-trace --stats -d $out/prof_gather_v100m -o r --output-format csv -- python3 tools/kbench.py gather --reps 20 --B 1024 --gather-F 40 --gather-rows 2500000 > $out/kbench_gather_v100m.txt 2>> $out/rocprof.err
rocprofv3 --kernel-trace --stats -d $out/prof_gather -o r --output-format csv -- python3 tools/kbench.py gather --reps 20 > $out/kbench_gather.txt 2>> $out/rocprof.err
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err
find $out -name "*kernel_stats.csv"
