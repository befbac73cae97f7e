#!/bin/bash
# round 5: the whole GPU suite + smoke, as the driver runs them
out=gpurun_out/r5_full
export TMPDIR=/tmp
mkdir -p $out
python3 -c "import __graft_entry__ as g; g.smoke()" > $out/smoke.log 2>&1; echo "smoke rc $?" > $out/rc.txt
timeout 3000 python3 -m pytest tests/ -x -q -m gpu > $out/pytest_gpu.log 2>&1; echo "gpu rc $?" >> $out/rc.txt
cat $out/rc.txt; tail -5 $out/pytest_gpu.log; tail -2 $out/smoke.log
