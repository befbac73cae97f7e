#!/bin/bash
# round 5: sumsq_reg_kernel with 1 (product) / 2 / 4 pieces per trip; HIP events around every launch of the eager step
out=gpurun_out/r5_sumsq
export TMPDIR=/tmp
mkdir -p $out
for round in 1 2 3; do
  for v in hip su2 su4; do
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-graph --time-all-kernels > $out/bench_${v}_$round.json 2> $out/bench_${v}_$round.err
  done
done
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); ks={k["kernel"]:k["avg_ms"] for k in d["kernels"]}
        print(f.split('/')[-1], d["ms_per_step"], {k:v for k,v in ks.items() if "sumsq" in k or "clip_adam" in k or "reduce" in k})
    except Exception as e: print(f, "ERR", e)
PY
