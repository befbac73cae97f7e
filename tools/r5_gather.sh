#!/bin/bash
# round 5: rows in flight per lane of the embedding gather (2 / 4 (product) / 8) at three shapes, same box, 3 interleaved rounds
out=gpurun_out/r5_gather
export TMPDIR=/tmp
mkdir -p $out
for round in 1 2 3; do
  for v in hip g8 g2; do
    echo "== $v (round $round)" >> $out/ab_gather.txt
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 tools/kbench.py gather --reps 30 --B 1024 --gather-F 40 --gather-rows 2500000 2>&1 | grep gather_fwd | sed 's/^/V100M B1024: /' >> $out/ab_gather.txt
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 tools/kbench.py gather --reps 30 2>&1 | grep gather_fwd | sed 's/^/N2 B4096:    /' >> $out/ab_gather.txt
    RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$v.so python3 tools/kbench.py gather --reps 30 --B 512 2>&1 | grep gather_fwd | sed 's/^/N2 B512:     /' >> $out/ab_gather.txt
  done
done
cat $out/ab_gather.txt
