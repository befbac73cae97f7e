#!/bin/bash
# round 5, second pass: new GPU tests (doc snippet, batch_prepare, trajectory at both geometries, kernels with knobs) + bench lines
out=gpurun_out/r5_pass2
export TMPDIR=/tmp
mkdir -p $out
timeout 1500 python3 -m pytest tests/test_integration_doc.py tests/test_batch_assembly.py tests/test_gpu_kernels.py tests/test_gpu_model.py tests/test_gpu_pipeline.py -x -q -m gpu > $out/pytest_a.log 2>&1; echo "a rc $?" > $out/rc.txt
timeout 1500 python3 -m pytest tests/test_gpu_trajectory.py -x -q -m gpu -s > $out/pytest_traj.log 2>&1; echo "traj rc $?" >> $out/rc.txt
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?" >> $out/rc.txt
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --batch 512 > $out/bench_b512.json 2> $out/bench_b512.err
python3 bench.py --workload mltag_like_K10_d16_B256 --no-cpu-baseline --no-extras --inference --steps 50 > $out/bench_mltag.json 2> $out/bench_mltag.err
cat $out/rc.txt; tail -3 $out/pytest_a.log; tail -3 $out/pytest_traj.log
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d["value"], d["ms_per_step"], d.get("per_rank_shape",{}).get("ratio_to_headline"))
        inf=d.get("inference")
        if inf:
            for r in inf["shapes"]: print("   inference", {k:v for k,v in r.items() if k!="kernels"})
    except Exception as e: print(f, "ERR", e)
PY
