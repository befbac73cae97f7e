#!/bin/bash
# round 5, first contact: the N > 1 path on one GPU with the new launcher bounds + today's baseline numbers
out=gpurun_out/r5_first
export TMPDIR=/tmp
mkdir -p $out
timeout 900 python3 -m pytest tests/test_gpu_rccl.py tests/test_gpu_dp.py tests/test_gpu_sparse.py -x -q -m gpu > $out/pytest_dp.log 2>&1; echo "dp rc $?" > $out/rc.txt
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?" >> $out/rc.txt
python3 bench.py --steps 10 --warmup 3 --dp-rehearsal --no-graph-dp > $out/bench_rehearsal.json 2> $out/bench_rehearsal.err; echo "rehearsal rc $?" >> $out/rc.txt
python3 bench.py --steps 10 --warmup 3 --dp-rehearsal > $out/bench_rehearsal_graph.json 2> $out/bench_rehearsal_graph.err; echo "rehearsal graph rc $?" >> $out/rc.txt
python3 bench.py --steps 20 --warmup 3 --dp-rehearsal --no-graph-dp --batch 512 --scaling weak > $out/bench_rehearsal_b512.json 2> $out/bench_rehearsal_b512.err
python3 bench.py --steps 20 --warmup 3 --dp-rehearsal --batch 512 --scaling weak > $out/bench_rehearsal_b512_graph.json 2> $out/bench_rehearsal_b512_graph.err
cat $out/rc.txt; tail -3 $out/pytest_dp.log
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d["value"], d["ms_per_step"], d.get("per_rank_shape",{}).get("ratio_to_headline"), d.get("strong_scaling",{}).get("ms_per_step"))
    except Exception as e: print(f, "ERR", e)
PY
