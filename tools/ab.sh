#!/bin/bash
# Same-box A/B on the GPU box: interleaved rounds over VARIANTS of one measurement (round 5's 33 one-off r5_*.sh scripts were
# instances of this; they are in git history, `git log -- tools/r5_ohalf.sh`).  Run it through gpurun:
#
#   gpurun -- 'tools/ab.sh OUT [--rounds N] [--pytest "ARGS"] (--kbench "ARGS" | --bench "ARGS") [--grep RE] -- VARIANT...'
#
#   OUT        directory under gpurun_out/ for the logs (created)
#   --pytest   first run `python3 -m pytest ARGS` once (the parity gate of what is being measured); its tail goes to OUT/pytest.log
#   --kbench   the measurement is `python3 tools/kbench.py ARGS`   (one kernel, stand-alone; lines matching --grep are kept)
#   --bench    the measurement is `python3 bench.py ARGS`           (the step; the JSON line is kept and summarised at the end)
#   VARIANT    NAME[:ENV=VAL[,ENV=VAL...]] — NAME "hip" is the product library, any other NAME is www24-rat_amd/lib/librat_NAME.so
#              (built by tools/variant.sh NAME file.hip -DFLAG before the call); the ENV assignments are exported for that run only
#              (the library's knobs: RAT_ATTN_FWD_CORE=mfma32, RAT_ATTN_BWD_CORE=valu, ...)
#
#   e.g.  tools/ab.sh r6_core --rounds 3 --pytest "tests/test_gpu_kernels.py -x -q -m gpu -k attn" \
#             --kbench "attn_fwd --arith bf16x3 --reps 30" --grep attn_fwd -- hip hip:RAT_ATTN_FWD_CORE=mfma32
set -u
out=gpurun_out/$1; shift
rounds=3; pyt=""; kb=""; be=""; re="."
while [ $# -gt 0 ]; do
  case "$1" in
    --rounds) rounds=$2; shift 2;;
    --pytest) pyt=$2; shift 2;;
    --kbench) kb=$2; shift 2;;
    --bench) be=$2; shift 2;;
    --grep) re=$2; shift 2;;
    --) shift; break;;
    *) echo "unknown option $1" >&2; exit 2;;
  esac
done
export TMPDIR=/tmp
mkdir -p "$out"
if [ -n "$pyt" ]; then
  timeout 1200 python3 -m pytest $pyt > "$out/pytest.log" 2>&1; echo "pytest rc $?" | tee "$out/rc.txt"; tail -2 "$out/pytest.log"
fi
for round in $(seq 1 "$rounds"); do
  for v in "$@"; do
    name=${v%%:*}; envs=""; [ "$v" != "$name" ] && envs=${v#*:}
    tag=$(echo "$v" | tr ':=,/' '____')
    (
      [ "$name" != hip ] && export RAT_HIP_LIBRARY=$PWD/www24-rat_amd/lib/librat_$name.so
      IFS=,; for kv in $envs; do export "$kv"; done; unset IFS
      if [ -n "$kb" ]; then
        echo "== $v (round $round)" >> "$out/ab.txt"
        timeout 900 python3 tools/kbench.py $kb 2>&1 | grep -E "$re" >> "$out/ab.txt"
      else
        timeout 1500 python3 bench.py $be > "$out/bench_${tag}_$round.json" 2> "$out/bench_${tag}_$round.err"
      fi
    )
  done
done
[ -n "$kb" ] && cat "$out/ab.txt"
[ -n "$be" ] && python3 - "$out" <<'PY'
import glob, json, os, sys
for f in sorted(glob.glob(os.path.join(sys.argv[1], "bench_*.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        ks = {k["kernel"]: k["avg_ms"] for k in d.get("kernels", [])}
        print("%-44s %10.1f %s/s %8.3f ms/step  %s" % (os.path.basename(f)[6:-5], d["value"], d["unit"].split("/")[0], d["ms_per_step"],
                                                     {k: v for k, v in ks.items() if "attn" in k}))
    except Exception as exc:
        print(f, "ERR", exc)
PY
exit 0
