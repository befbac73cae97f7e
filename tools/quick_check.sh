#!/bin/bash
# Diagnostic: a short GPU pass after a kernel change (bash tools/quick_check.sh <outdir> [pytest -k expression])
out=${1:-gpurun_out/qc}
sel=${2:-"sgemm or strips or bn_relu or logit or colsum"}
export TMPDIR=/tmp
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "$sel" > $out/pytest_kernels.log 2>&1; echo "kernels rc $?" > $out/rc.txt
timeout 1500 python3 -m pytest tests/test_gpu_model.py tests/test_gpu_pipeline.py -x -q -m gpu > $out/pytest_model.log 2>&1; echo "model rc $?" >> $out/rc.txt
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_default.json 2> $out/bench_default.err
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --batch 512 > $out/bench_b512.json 2> $out/bench_b512.err
for wl in mltag_like_K10_d16_B256 movielens_real_F3_K5_d10_B4096 kkbox_real_F13_K5_d40_B4096; do
  python3 bench.py --workload $wl --no-cpu-baseline --no-extras > $out/bench_$wl.json 2> $out/bench_$wl.err
done
rocprofv3 --kernel-trace --stats -d $out/prof_b512 -o r --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --batch 512 > $out/bench_b512_under_rocprof.json 2> $out/rocprof.err
rocprofv3 --kernel-trace --stats -d $out/prof -o r --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_under_rocprof.json 2>> $out/rocprof.err
cat $out/rc.txt; tail -3 $out/pytest_kernels.log; tail -3 $out/pytest_model.log
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d["value"], d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
rocprofv3 --kernel-trace --stats -d $out/prof_mltag -o r --output-format csv -- python3 bench.py --workload mltag_like_K10_d16_B256 --steps 50 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_mltag_under_rocprof.json 2>> $out/rocprof.err
