#!/bin/bash
# round 5: RAT_m0's joint-sequence attention core (231 tokens) forward on the matrix pipe (core_fwd_mfma_kernel): parity, then A/B through the knob
out=gpurun_out/r5_m0_core
export TMPDIR=/tmp
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn_core" > $out/pytest_kernels.log 2>&1; echo "kernels rc $?" > $out/rc.txt
timeout 1500 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_fullsize.py -x -q -m gpu -k "m0 or composed" > $out/pytest_model.log 2>&1; echo "model rc $?" >> $out/rc.txt
for round in 1 2 3; do
  python3 bench.py --model RAT_m0 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --inference > $out/m0_mfma_$round.json 2> $out/m0_mfma_$round.err
  RAT_ATTN_FWD_CORE=valu python3 bench.py --model RAT_m0 --steps 10 --warmup 3 --no-cpu-baseline --no-extras --inference > $out/m0_valu_$round.json 2> $out/m0_valu_$round.err
done
cat $out/rc.txt; tail -n 2 $out/pytest_kernels.log $out/pytest_model.log
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); ks={k["kernel"]:(k["avg_ms"],k.get("frac")) for k in d["kernels"]}
        print(f.split('/')[-1][:-5], d["value"], d["ms_per_step"], (d.get("inference") or {}).get("value"), {k:v for k,v in ks.items() if "attn_core" in k})
    except Exception as e: print(f, "ERR", e)
PY
