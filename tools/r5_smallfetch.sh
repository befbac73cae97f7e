#!/bin/bash
# round 5: one-round-trip chunk loads (SmallFetch) in the generic attention kernels with compile-time geometry — same-box A/B on the shipped
# MovieLens shape (d 10, 2 heads) and BASELINE configs[0] (d 16, 2 heads, B 256); base = the previous commit's library
out=${OUT:-gpurun_out/r5_smallfetch}
export TMPDIR=/tmp
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attn" > $out/pytest_kernels.log 2>&1; echo "kernels rc $?" > $out/rc.txt
timeout 900 python3 -m pytest tests/test_gpu_model.py -x -q -m gpu -k "mltag or tiny or kkbox" > $out/pytest_model.log 2>&1; echo "model rc $?" >> $out/rc.txt
for round in 1 2 3; do
  for v in ${VARIANTS:-sfetch base}; do
    for w in movielens_real_F3_K5_d10_B4096 mltag_like_K10_d16_B256; do
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
        print(f.split('/')[-1][:-5], d["value"], d["ms_per_step"], {k:v for k,v in ks.items() if "attn_" in k})
    except Exception as e: print(f, "ERR", e)
PY
