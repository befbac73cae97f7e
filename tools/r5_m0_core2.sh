#!/bin/bash
# round 5: core_fwd_mfma_kernel variants — two accumulators, 2 (kb2) or 4 (kb4) key tiles per trip — against the VALU kernel (knob), RAT_m0 step
out=gpurun_out/r5_m0_core2
export TMPDIR=/tmp
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn_core" > $out/pytest_kernels.log 2>&1; echo "kernels rc $?" > $out/rc.txt
for round in 1 2 3; do
  for v in kb2 kb4; do
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 bench.py --model RAT_m0 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/m0_${v}_$round.json 2> $out/m0_${v}_$round.err
  done
  RAT_ATTN_FWD_CORE=valu python3 bench.py --model RAT_m0 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/m0_valu_$round.json 2> $out/m0_valu_$round.err
done
cat $out/rc.txt; tail -n 2 $out/pytest_kernels.log
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); ks={k["kernel"]:(k["avg_ms"],k.get("frac")) for k in d["kernels"]}
        print(f.split('/')[-1][:-5], d["value"], d["ms_per_step"], {k:v for k,v in ks.items() if "attn_core_fwd" in k})
    except Exception as e: print(f, "ERR", e)
PY
