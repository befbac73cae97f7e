#!/bin/bash
# Diagnostic: same-box A/B of library variants INSIDE the training step, durations from rocprofv3 (not HIP events).
#   tools/ab_step.sh <outdir> "<kernel name fragments, |-separated>" base v1 v2 ...   (librat_<name>.so from tools/variant_multi.py;
#   "base" = the regular librat_hip.so)
out="$1"; pat="$2"; shift 2
export TMPDIR=/tmp
mkdir -p "$out"
for round in 1 2; do
  for v in "$@"; do
    lib=$PWD/www24-rat_amd/lib/librat_$v.so
    [ "$v" = "base" ] && lib=$PWD/www24-rat_amd/lib/librat_hip.so
    d=$out/${v}_r$round
    RAT_HIP_LIBRARY=$lib rocprofv3 --kernel-trace --stats -d $d -o s --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras ${AB_BENCH_ARGS} > $d.json 2> $d.err
    python3 - "$d" "$v" "$round" "$pat" <<'PY'
import csv, glob, json, sys
d, v, rnd, pat = sys.argv[1:5]
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)
line = json.loads(open(d + ".json").read().strip().splitlines()[-1])
out = ["%-10s r%s  step %.3f ms" % (v, rnd, line["ms_per_step"])]
for r in csv.DictReader(open(f[0])):
    if any(p in r["Name"] for p in pat.split("|")):
        out.append("%s %.1f us x%s" % (r["Name"].split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:28], float(r["AverageNs"]) / 1e3, r["Calls"]))
print(" | ".join(out), flush=True)
PY
  done
done
