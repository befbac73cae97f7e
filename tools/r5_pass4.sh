#!/bin/bash
# round 5, fourth pass: row-map variants of attn_bwd3 (b1: 32-bit in the backward only, b2: 64-bit without hoisting) against round 4's
# form (hip); the DP rehearsal with eager-first + graph attempt, and the watchdog's bail-out path
out=gpurun_out/r5_pass4
export TMPDIR=/tmp
mkdir -p $out
for round in 1 2 3; do
  for v in hip b1 b2; do
    echo "== $v (round $round)" >> $out/ab_attn.txt
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 tools/kbench.py attn_bwd --arith bf16x3 --reps 30 2>&1 | grep -v amdgpu.ids >> $out/ab_attn.txt
  done
done
python3 bench.py --steps 10 --warmup 3 --dp-rehearsal > $out/bench_rehearsal.json 2> $out/bench_rehearsal.err; echo "rehearsal rc $?" > $out/rc.txt
python3 bench.py --steps 10 --warmup 3 --dp-rehearsal --graph-attempt-timeout 0.05 > $out/bench_rehearsal_bail.json 2> $out/bench_rehearsal_bail.err; echo "bail rc $?" >> $out/rc.txt
python3 bench.py --steps 20 --warmup 3 --dp-rehearsal --batch 512 --scaling weak > $out/bench_rehearsal_b512.json 2> $out/bench_rehearsal_b512.err; echo "b512 rc $?" >> $out/rc.txt
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --batch 512 > $out/bench_b512.json 2> $out/bench_b512.err
python3 bench.py --workload mltag_like_K10_d16_B256 --no-cpu-baseline --no-extras --steps 50 > $out/bench_mltag.json 2> $out/bench_mltag.err
cat $out/rc.txt; cat $out/ab_attn.txt
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split('/')[-1], d["value"], d["ms_per_step"], d["step_mode"].get("attempts"), d["step_mode"].get("graph_attempt"), d.get("strong_scaling",{}).get("ms_per_step"))
    except Exception as e: print(f, "ERR", e)
PY
