#!/bin/bash
# round 5: final confirmation — smoke, the whole GPU suite, the default bench line (as the driver runs it), B = 512, rehearsal
out=gpurun_out/r5_final
export TMPDIR=/tmp
mkdir -p $out
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc $?" > $out/rc.txt
timeout 3000 python3 -m pytest tests/ -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "gpu rc $?" >> $out/rc.txt
python3 bench.py > $out/bench_default.json 2> $out/bench_default.err; echo "bench rc $?" >> $out/rc.txt
python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --batch 512 > $out/bench_b512.json 2> $out/bench_b512.err
python3 bench.py --steps 10 --warmup 3 --dp-rehearsal > $out/bench_rehearsal.json 2> $out/bench_rehearsal.err; echo "rehearsal rc $?" >> $out/rc.txt
cat $out/rc.txt; tail -3 $out/pytest_gpu.log; tail -1 $out/smoke.log
python3 - <<PY
import json,glob
for f in sorted(glob.glob("$out/bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f.split('/')[-1], d["value"], d["ms_per_step"], d.get("per_rank_shape",{}).get("ratio_to_headline"), d.get("roofline",{}).get("frac"), d.get("targets",{}).get("rat_gather_fwd_V100M",{}).get("frac_of_8TBps"), d.get("targets",{}).get("cross_attention",{}).get("frac_of_f32_mfma_peak"))
    except Exception as e: print(f, "ERR", e)
PY
