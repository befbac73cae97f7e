#!/bin/bash
out=gpurun_out/r5_mc2
export TMPDIR=/tmp
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn" > $out/pytest_attn.log 2>&1; echo "attn rc $?" > $out/rc.txt
for round in 1 2 3; do
  python3 tools/kbench.py attn_bwd --arith bf16x3 --reps 20 --T 31 --S 9 2>&1 | grep L31 >> $out/ab.txt
done
python3 bench.py --workload tmall_like_F8_K30_d64_h32_B4096 --no-cpu-baseline --no-extras --inference > $out/bench_tmall_like.json 2> $out/bench_tmall_like.err
python3 bench.py --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err
cat $out/rc.txt; tail -2 $out/pytest_attn.log; cat $out/ab.txt
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench*.json")):
    d=json.loads(open(f).read().strip().splitlines()[-1]); ks={k["kernel"]:(k["avg_ms"],k.get("frac")) for k in d["kernels"]}
    print(f.split('/')[-1], d["value"], d["ms_per_step"], {k:v for k,v in ks.items() if "attn_bwd" in k}, d.get("inference",{}).get("value"), d.get("inference",{}).get("with_dead_token_pruning"))
PY
