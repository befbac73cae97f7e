#!/bin/bash
# round 5: PMC passes over the shipped-Tmall step (the wide-head small-d kernels): MFMA / VALU / wait shares and HBM bytes per launch
out=gpurun_out/r5_pmc_wide
export TMPDIR=/tmp
mkdir -p $out
W=tmall_real_F9_K5_d10_h32_B4096
A="bench.py --workload $W --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-graph"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d $out/pmc_mfma -o m --output-format csv -- python3 $A > $out/pmc_mfma.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o f --output-format csv -- python3 $A > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -o w --output-format csv -- python3 $A > $out/pmc_write.log 2>&1
python3 tools/pmc_summary.py busy $out/pmc_mfma > $out/pmc_busy.json
python3 tools/pmc_summary.py traffic $out/pmc_fetch $out/pmc_write > $out/traffic_pmc.json
python3 - <<PY
import json
b=json.load(open("$out/pmc_busy.json")); t=json.load(open("$out/traffic_pmc.json"))
for k,v in b["kernels"].items(): print(k, v["kernel"], "mfma", v["mfma_busy_frac"], "valu", v["valu_active_frac"], "wait", v["wait_frac"])
for k,v in t["kernels"].items(): print(k, v["hbm_bytes_per_launch"], v["launches"])
PY
