#!/bin/bash
# Diagnostic: the measurement passes behind profiles/round6 (run on the GPU box from the repo root: bash tools/profile_round6.sh <outdir>)
out=${1:-gpurun_out/prof6}
export TMPDIR=/tmp
mkdir -p $out
python3 bench.py --steps 10 --warmup 3 > $out/bench_default.json 2> $out/bench_default.err
rocprofv3 --kernel-trace --stats -d $out/prof -o r --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_under_rocprof.json 2> $out/rocprof.err
rocprofv3 --kernel-trace --stats -d $out/prof_gather_v100m -o r --output-format csv -- python3 tools/kbench.py gather --reps 20 --B 1024 --gather-F 40 --gather-rows 2500000 > $out/kbench_gather_v100m.txt 2>> $out/rocprof.err
rocprofv3 --kernel-trace --stats -d $out/prof_m0 -o r --output-format csv -- python3 bench.py --model RAT_m0 --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_RAT_m0_under_rocprof.json 2>> $out/rocprof.err
# PMC passes on their own (never with --stats / trace domains other than --kernel-trace)
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o f --output-format csv -- python3 tools/kbench.py attn_bwd attn_fwd ffn_fwd ffn_bwd --arith bf16x3 --reps 2 > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -o w --output-format csv -- python3 tools/kbench.py attn_bwd attn_fwd ffn_fwd ffn_bwd --arith bf16x3 --reps 2 > $out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d $out/pmc_mfma -o m --output-format csv -- python3 tools/kbench.py attn_bwd attn_fwd ffn_fwd ffn_bwd --arith bf16x3 --reps 2 > $out/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d $out/pmc_core -o m --output-format csv -- python3 tools/kbench.py core --reps 2 > $out/pmc_core.log 2>&1
python3 tools/pmc_summary.py traffic $out/pmc_fetch $out/pmc_write > $out/traffic_pmc.json
python3 tools/pmc_summary.py busy $out/pmc_mfma > $out/pmc_busy.json
python3 tools/pmc_summary.py busy $out/pmc_core > $out/pmc_busy_core.json
# configs[3]'s intra-sample sequences (L = 41) on their own, beside the north-star lengths at the same batch
python3 tools/kbench.py attn_fwd attn_bwd --arith bf16x3 --reps 20 --B 1024 --T 11 --S 41 > $out/kbench_L41.txt 2>&1
python3 tools/kbench.py attn_fwd attn_bwd --arith bf16x3 --reps 20 --B 1024 --T 11 --S 21 >> $out/kbench_L41.txt 2>&1
python3 tools/kbench.py core --reps 10 > $out/kbench_core.txt 2>&1
tools/workloads.sh ${out#gpurun_out/}/wl > $out/workloads.txt 2>&1
python3 bench.py --steps 10 --warmup 3 --dp-rehearsal > $out/bench_dp_rehearsal.json 2> $out/bench_dp_rehearsal.err
find $out -name "*kernel_stats.csv" | head
ls $out
