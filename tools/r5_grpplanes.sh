#!/bin/bash
# round 5: wide heads at d = 64 — per-group plane sets once per step for the backward's group launches too, PH form for groups 1 ... G - 1
# (new) against the previous commit (base: a worktree of it under _base/, removed afterwards): parity, then same-box A/B in the Tmall-like step and a check of the headline
out=gpurun_out/r5_grpplanes
export TMPDIR=/tmp
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "wide_heads or attn_ex or attn_fwd_bwd" > $out/pytest_kernels.log 2>&1; echo "kernels rc $?" > $out/rc.txt
timeout 1500 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py -x -q -m gpu -k "wide_heads or grouped or tmall" > $out/pytest_model.log 2>&1; echo "model rc $?" >> $out/rc.txt
W=tmall_like_F8_K30_d64_h32_B4096
for round in 1 2 3; do
  python3 bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/tmall_new_$round.json 2> $out/tmall_new_$round.err
  (cd _base && python3 bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-extras) > $out/tmall_base_$round.json 2> $out/tmall_base_$round.err
done
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/default_new.json 2> $out/default_new.err
(cd _base && python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras) > $out/default_base.json 2> $out/default_base.err
cat $out/rc.txt; tail -n 2 $out/pytest_kernels.log $out/pytest_model.log
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); ks={k["kernel"]:k["avg_ms"] for k in d["kernels"]}
        print(f.split('/')[-1][:-5], d["value"], d["ms_per_step"], {k:v for k,v in ks.items() if "attn_" in k})
    except Exception as e: print(f, "ERR", e)
PY
