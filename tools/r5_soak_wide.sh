#!/bin/bash
# round 5: soak of the product default (graph replays, dead-token pruning) on the wide-head workloads + the pruned bench lines
out=gpurun_out/r5_soak_wide
export TMPDIR=/tmp
mkdir -p $out
for w in tmall_real_F9_K5_d10_h32_B4096 tmall_like_F8_K30_d64_h32_B4096 movielens_real_F3_K5_d10_B4096; do
  python3 tools/soak.py 300 $w > $out/soak_$w.txt 2>&1; echo "soak $w rc $?" >> $out/rc.txt
  python3 bench.py --workload $w --prune --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_prune_$w.json 2> $out/bench_prune_$w.err; echo "prune $w rc $?" >> $out/rc.txt
done
cat $out/rc.txt; tail -n 3 $out/soak_*.txt
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1][:-5], d["value"], d["ms_per_step"], d["config"].get("dead_token_pruning"))
    except Exception as e: print(f, "ERR", e)
PY
