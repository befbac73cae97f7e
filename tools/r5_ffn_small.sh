#!/bin/bash
# round 5: the generic feed-forward kernels at the shipped d = 10 geometries with the weights in LDS, one-sweep chunk loads and one GEMM task per
# tile (ffns) against the previous commit (base): parity, then same-box A/B on the shipped MovieLens and Tmall shapes
out=gpurun_out/r5_ffn_small
export TMPDIR=/tmp
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "ffn" > $out/pytest_kernels.log 2>&1; echo "kernels rc $?" > $out/rc.txt
timeout 900 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu -k "mltag or tmall" > $out/pytest_model.log 2>&1; echo "model rc $?" >> $out/rc.txt
for round in 1 2 3; do
  for v in ffns base; do
    for w in movielens_real_F3_K5_d10_B4096 tmall_real_F9_K5_d10_h32_B4096; do
      RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline --no-extras > $out/${w}_${v}_$round.json 2> $out/${w}_${v}_$round.err
    done
  done
done
cat $out/rc.txt; tail -n 2 $out/pytest_kernels.log $out/pytest_model.log
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); ks={k["kernel"]:k["avg_ms"] for k in d["kernels"]}
        print(f.split('/')[-1][:-5], d["value"], d["ms_per_step"], {k:v for k,v in ks.items() if "ffn" in k})
    except Exception as e: print(f, "ERR", e)
PY
