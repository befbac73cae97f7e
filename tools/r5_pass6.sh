#!/bin/bash
# round 5, sixth pass: the matrix-pipe backward core inside attn_bwd3_kernel — parity at every length class, then the same-box A/B at
# BASELINE configs[4]'s cross-sample length (T = 31, S = 9: knob off = VALU passes, default = the host's rule, L 31 -> matrix core)
out=gpurun_out/r5_pass6
export TMPDIR=/tmp
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn" > $out/pytest_attn.log 2>&1; echo "attn rc $?" > $out/rc.txt
for round in 1 2 3; do
  echo "== matrix core by the host's rule (round $round)" >> $out/ab_core.txt
  python3 tools/kbench.py attn_bwd --arith bf16x3 --reps 20 --T 31 --S 9 2>&1 | grep -v amdgpu.ids >> $out/ab_core.txt
  echo "== VALU passes: RAT_ATTN_BWD_CORE=valu (round $round)" >> $out/ab_core.txt
  RAT_ATTN_BWD_CORE=valu python3 tools/kbench.py attn_bwd --arith bf16x3 --reps 20 --T 31 --S 9 2>&1 | grep -v amdgpu.ids >> $out/ab_core.txt
done
echo "== T 16 S 16: matrix core by rule / valu" >> $out/ab_core.txt
python3 tools/kbench.py attn_bwd --arith bf16x3 --reps 20 --T 16 --S 16 --B 2048 2>&1 | grep -v amdgpu.ids >> $out/ab_core.txt
RAT_ATTN_BWD_CORE=valu python3 tools/kbench.py attn_bwd --arith bf16x3 --reps 20 --T 16 --S 16 --B 2048 2>&1 | grep -v amdgpu.ids >> $out/ab_core.txt
python3 bench.py --workload tmall_like_F8_K30_d64_h32_B4096 --no-cpu-baseline --no-extras > $out/bench_tmall_like.json 2> $out/bench_tmall_like.err
RAT_ATTN_BWD_CORE=valu python3 bench.py --workload tmall_like_F8_K30_d64_h32_B4096 --no-cpu-baseline --no-extras > $out/bench_tmall_like_valu.json 2> $out/bench_tmall_like_valu.err
cat $out/rc.txt; tail -2 $out/pytest_attn.log; cat $out/ab_core.txt
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); ks={k["kernel"]:(k["avg_ms"],k.get("frac")) for k in d["kernels"]}
        print(f.split('/')[-1], d["value"], d["ms_per_step"], {k:v for k,v in ks.items() if "attn" in k})
    except Exception as e: print(f, "ERR", e)
PY
