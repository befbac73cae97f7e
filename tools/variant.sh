#!/bin/bash
# Diagnostic: build www24-rat_amd/lib/librat_<name>.so with ONE source recompiled under extra flags (for tools/ab_bench.sh).
#   tools/variant.sh pk ffn.hip -DRAT_FFN_PK
set -e
name="$1"; src="$2"; shift 2
cd "$(dirname "$0")/../www24-rat_amd"
python build.py >/dev/null
extra=""
[ "$src" = "ffn.hip" ] && extra="-mllvm -amdgpu-sched-strategy=max-ilp"
mkdir -p lib/obj_$name
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function $extra "$@" -c csrc/$src -o lib/obj_$name/${src%.hip}.o 2>/dev/null
objs=""
for o in lib/obj/*.o; do
  b=$(basename $o)
  if [ "$b" = "${src%.hip}.o" ]; then objs="$objs lib/obj_$name/$b"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/librat_$name.so $objs
echo lib/librat_$name.so
