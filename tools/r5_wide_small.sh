#!/bin/bash
# round 5: the shipped Tmall head geometry (32 x 10 at d = 10) as one launch per layer and direction — parity, then a same-box A/B in the step
out=gpurun_out/r5_wide_small
export TMPDIR=/tmp
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wide_heads" > $out/pytest_kernels.log 2>&1; echo "kernels rc $?" > $out/rc.txt
timeout 1500 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu -k "tmall or wide_heads or grouped" > $out/pytest_model.log 2>&1; echo "model rc $?" >> $out/rc.txt
W=tmall_real_F9_K5_d10_h32_B4096
for round in 1 2 3; do
  python3 bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_loop_$round.json 2> $out/bench_loop_$round.err
  python3 bench.py --workload $W --steps 20 --warmup 3 --no-cpu-baseline --no-extras --no-group-loop > $out/bench_groups_$round.json 2> $out/bench_groups_$round.err
done
python3 bench.py --workload $W --no-cpu-baseline --no-extras --inference > $out/bench_inference.json 2> $out/bench_inference.err
cat $out/rc.txt; tail -n 3 $out/pytest_kernels.log $out/pytest_model.log
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); ks={k["kernel"]:(k["launches_per_step"],k["avg_ms"],k.get("frac")) for k in d["kernels"]}
        print(f.split('/')[-1], d["value"], d["ms_per_step"], (d.get("inference") or {}).get("value"), {k:v for k,v in ks.items() if "attn_" in k})
    except Exception as e: print(f, "ERR", e)
PY
