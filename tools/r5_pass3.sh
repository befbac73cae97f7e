#!/bin/bash
# round 5, third pass: same-box A/B of the 32-bit row map (no scratch in attn_bwd3) against round 4's form (librat_map64.so)
out=gpurun_out/r5_pass3
export TMPDIR=/tmp
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn" > $out/pytest_attn.log 2>&1; echo "attn rc $?" > $out/rc.txt
bash tools/ab_attn.sh hip map64 > $out/ab_attn.txt 2>&1
for round in 1 2 3; do
  for v in hip map64; do
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_${v}_$round.json 2> $out/bench_${v}_$round.err
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --batch 512 > $out/bench512_${v}_$round.json 2>> $out/bench_${v}_$round.err
  done
done
cat $out/rc.txt; tail -2 $out/pytest_attn.log; grep -v "^$" $out/ab_attn.txt | tail -40
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); ks={k["kernel"]:k["avg_ms"] for k in d["kernels"]}
        print(f.split('/')[-1], d["value"], d["ms_per_step"], {k:v for k,v in ks.items() if "attn_bwd" in k})
    except Exception as e: print(f, "ERR", e)
PY
