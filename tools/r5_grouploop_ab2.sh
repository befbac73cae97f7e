#!/bin/bash
# round 5: group-loop forward, second A/B — new2: the ordinary kernel reads W.qkv / W.out as before (run-time K steps), the group loop
# with compile-time steps; new3: run-time steps in the group loop too; oldfwd: previous commit's attn.hip
out=gpurun_out/r5_grouploop_ab2
export TMPDIR=/tmp
mkdir -p $out
bash tools/ab_attn.sh new2 oldfwd > $out/ab_attn.txt 2>&1
W=tmall_like_F8_K30_d64_h32_B4096
for round in 1 2 3; do
  for v in new2 oldfwd; do
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_${v}_$round.json 2> $out/bench_${v}_$round.err
  done
  for v in new2 new3; do
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/tmall_${v}_$round.json 2> $out/tmall_${v}_$round.err
  done
done
grep -v "^$" $out/ab_attn.txt | grep -v amdgpu.ids | tail -30
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); ks={k["kernel"]:k["avg_ms"] for k in d["kernels"]}
        print(f.split('/')[-1], d["value"], d["ms_per_step"], {k:v for k,v in ks.items() if "attn_" in k})
    except Exception as e: print(f, "ERR", e)
PY
