#!/bin/bash
# round 5: the wide-head forward group loop (rat_attn_fwd_groups) — parity on the GPU, then a same-box A/B inside the Tmall-like step
out=gpurun_out/r5_grouploop
export TMPDIR=/tmp
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wide_heads or attn_fwd or attn_ex" > $out/pytest_kernels.log 2>&1; echo "kernels rc $?" > $out/rc.txt
timeout 900 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu -k "wide_heads or grouped" > $out/pytest_model.log 2>&1; echo "model rc $?" >> $out/rc.txt
timeout 1500 python3 -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "tmall_like" > $out/pytest_configs.log 2>&1; echo "configs rc $?" >> $out/rc.txt
W=tmall_like_F8_K30_d64_h32_B4096
for round in 1 2 3; do
  python3 bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_loop_$round.json 2> $out/bench_loop_$round.err
  python3 bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-extras --no-group-loop > $out/bench_groups_$round.json 2> $out/bench_groups_$round.err
done
for round in 1 2; do
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_default_$round.json 2> $out/bench_default_$round.err
done
cat $out/rc.txt; tail -3 $out/pytest_kernels.log $out/pytest_model.log $out/pytest_configs.log
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); ks={k["kernel"]:(k["launches_per_step"],k["avg_ms"],k.get("frac")) for k in d["kernels"]}
        print(f.split('/')[-1], d["value"], d["ms_per_step"], {k:v for k,v in ks.items() if "attn_fwd" in k})
    except Exception as e: print(f, "ERR", e)
PY
