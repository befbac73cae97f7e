#!/bin/bash
# round 5: parametric A/Bs of the HBM-bound kernels: clip_adam_fused pieces per trip (ca2), gather_bwd rows per wave (gb4 / gb16) and rows per trip (gt1 / gt4)
out=gpurun_out/r5_sweep
export TMPDIR=/tmp
mkdir -p $out
for round in 1 2 3; do
  for v in hip gb4 gb16 gt1 gt4; do
    echo "== $v (round $round)" >> $out/ab_gather_bwd.txt
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 tools/kbench.py gather --reps 30 2>&1 | grep gather_bwd | sed 's/^/N2 B4096:    /' >> $out/ab_gather_bwd.txt
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 tools/kbench.py gather --reps 30 --B 512 2>&1 | grep gather_bwd | sed 's/^/N2 B512:     /' >> $out/ab_gather_bwd.txt
  done
  for v in hip ca2; do
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-graph --time-all-kernels > $out/bench_${v}_$round.json 2> $out/bench_${v}_$round.err
  done
done
cat $out/ab_gather_bwd.txt
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); ks={k["kernel"]:k["avg_ms"] for k in d["kernels"]}
        print(f.split('/')[-1], d["ms_per_step"], {k:v for k,v in ks.items() if "sumsq" in k or "clip_adam" in k or "reduce_defer_end" in k})
    except Exception as e: print(f, "ERR", e)
PY
