#!/bin/bash
out=gpurun_out/r5_guard
mkdir -p $out
timeout 1200 python3 -m pytest tests/test_gpu_rccl.py -x -q -m gpu -k "bench" > $out/pytest.log 2>&1; echo rc $?; tail -4 $out/pytest.log
python3 bench.py --dp-rehearsal --steps 5 --warmup 2 > $out/b.json 2> $out/b.err; echo bench rc $?
python3 -c "
import json; d=json.loads(open('$out/b.json').read().strip().splitlines()[-1]); print(d['value'], d['step_mode']['attempts'], d['strong_scaling']['ms_per_step'])"
