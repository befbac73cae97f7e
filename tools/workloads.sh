#!/bin/bash
# One bench line per workload of rat_amd.synthetic.WORKLOADS other than the headline, and per model variant on the headline workload
# (what profiles/roundN/rN_bench_<workload>.json hold).   gpurun -- 'tools/workloads.sh OUT [extra bench.py flags]'
out=gpurun_out/$1; shift
export TMPDIR=/tmp
mkdir -p "$out"
for wl in mltag_like_K10_d16_B256 kkbox_like_F13_K10_d64_B4096 tmall_like_F8_K30_d64_h32_B4096 kkbox_real_F13_K5_d40_B4096 \
          synthetic_F40_V100M_K10_d64_B1024 movielens_real_F3_K5_d10_B4096 tmall_real_F9_K5_d10_h32_B4096; do
  timeout 900 python3 bench.py --workload $wl --no-cpu-baseline --no-extras --inference "$@" > "$out/bench_$wl.json" 2> "$out/bench_$wl.err"
done
for m in RAT_m0 RAT_m1 RAT_m3; do
  timeout 900 python3 bench.py --model $m --no-cpu-baseline --no-extras "$@" > "$out/bench_$m.json" 2> "$out/bench_$m.err"
done
python3 - "$out" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print("%-48s %10.1f %8.3f ms/step  inference %s" % (os.path.basename(f)[6:-5], d["value"], d["ms_per_step"], (d.get("inference") or {}).get("value")))
    except Exception as exc:
        print(f, "ERR", exc)
PY
