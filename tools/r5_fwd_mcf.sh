#!/bin/bash
# round 5: the forward attention core on the matrix pipe, exact fp32 (attn_fwd3_kernel<.., MCF>): parity, then same-box A/Bs through the knob
# (RAT_ATTN_FWD_CORE=valu: the VALU loop everywhere; unset: matrix core for 28 <= L <= 32; mfma32: matrix core for every L <= 32)
out=gpurun_out/r5_fwd_mcf
export TMPDIR=/tmp
mkdir -p $out
timeout 1200 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn" > $out/pytest_kernels.log 2>&1; echo "kernels rc $?" > $out/rc.txt
timeout 1500 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_configs.py -x -q -m gpu -k "wide_heads or grouped or tmall" > $out/pytest_model.log 2>&1; echo "model rc $?" >> $out/rc.txt
W=tmall_like_F8_K30_d64_h32_B4096
for round in 1 2 3; do
  python3 bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/tmall_auto_$round.json 2> $out/tmall_auto_$round.err
  RAT_ATTN_FWD_CORE=valu python3 bench.py --workload $W --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/tmall_valu_$round.json 2> $out/tmall_valu_$round.err
done
for round in 1 2; do
  python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/default_auto_$round.json 2> $out/default_auto_$round.err
  RAT_ATTN_FWD_CORE=mfma32 python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/default_mfma32_$round.json 2> $out/default_mfma32_$round.err
done
python3 bench.py --workload $W --no-cpu-baseline --no-extras --inference > $out/tmall_inference.json 2> $out/tmall_inference.err
cat $out/rc.txt; tail -n 2 $out/pytest_kernels.log $out/pytest_model.log
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); ks={k["kernel"]:k["avg_ms"] for k in d["kernels"]}
        print(f.split('/')[-1][:-5], d["value"], d["ms_per_step"], (d.get("inference") or {}).get("value"), {k:v for k,v in ks.items() if "attn_fwd" in k})
    except Exception as e: print(f, "ERR", e)
PY
