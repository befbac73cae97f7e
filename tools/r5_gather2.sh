#!/bin/bash
out=gpurun_out/r5_gather2
export TMPDIR=/tmp
mkdir -p $out
for round in 1 2 3; do
  for v in hip g1 g2 g3; do
    echo "== $v (round $round)" >> $out/ab_gather.txt
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 tools/kbench.py gather --reps 30 --B 1024 --gather-F 40 --gather-rows 2500000 2>&1 | grep gather_ | sed 's/^/V100M B1024: /' >> $out/ab_gather.txt
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 tools/kbench.py gather --reps 30 2>&1 | grep gather_ | sed 's/^/N2 B4096:    /' >> $out/ab_gather.txt
  done
done
for round in 1 2; do
  for v in hip g1 g2; do
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_${v}_$round.json 2> $out/bench_${v}_$round.err
  done
done
cat $out/ab_gather.txt
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); ks={k["kernel"]:k["avg_ms"] for k in d["kernels"]}
        print(f.split('/')[-1], d["value"], d["ms_per_step"], {k:v for k,v in ks.items() if "gather" in k}, d["targets"]["rat_gather_fwd"]["frac_of_8TBps"])
    except Exception as e: print(f, "ERR", e)
PY
