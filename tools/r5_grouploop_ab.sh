#!/bin/bash
# round 5: does the group-loop template parameter leave the ordinary forward kernel alone?  Same-box A/B of the library against one whose
# attn.hip is the previous commit's (librat_oldfwd.so), headline workload and per-rank shape, three interleaved rounds
out=gpurun_out/r5_grouploop_ab
export TMPDIR=/tmp
mkdir -p $out
bash tools/ab_attn.sh new oldfwd > $out/ab_attn.txt 2>&1
for round in 1 2 3; do
  for v in new oldfwd; do
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_${v}_$round.json 2> $out/bench_${v}_$round.err
  done
done
grep -v "^$" $out/ab_attn.txt | grep -v amdgpu.ids | tail -30
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); ks={k["kernel"]:k["avg_ms"] for k in d["kernels"]}
        print(f.split('/')[-1], d["value"], d["ms_per_step"], {k:v for k,v in ks.items() if "attn_fwd" in k})
    except Exception as e: print(f, "ERR", e)
PY
