#!/bin/bash
# Diagnostic: the rocprofv3 passes behind profiles/roundN (run on the GPU box from the repo root: bash tools/profile_round.sh <outdir>)
#   kernel-trace stats of the default bench line, then separate --pmc passes (never combined with other trace domains)
out=${1:-gpurun_out/prof}
export TMPDIR=/tmp
mkdir -p $out
python3 bench.py --steps 10 --warmup 3 > $out/bench_default.json 2> $out/bench_default.err
rocprofv3 --kernel-trace --stats -d $out/prof -o r --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline > $out/bench_under_rocprof.json 2> $out/rocprof.err
rocprofv3 --kernel-trace --stats -d $out/prof_f32 -o r --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --arith f32 > $out/bench_f32_under_rocprof.json 2>> $out/rocprof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o f --output-format csv -- python3 tools/kbench.py attn_bwd attn_fwd ffn_fwd ffn_bwd --arith both --reps 2 > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -o w --output-format csv -- python3 tools/kbench.py attn_bwd attn_fwd ffn_fwd ffn_bwd --arith both --reps 2 > $out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d $out/pmc_mfma -o m --output-format csv -- python3 tools/kbench.py attn_bwd attn_fwd ffn_fwd ffn_bwd --arith both --reps 2 > $out/pmc_mfma.log 2>&1
python3 tools/pmc_summary.py traffic $out/pmc_fetch $out/pmc_write > $out/traffic_pmc.json
python3 tools/pmc_summary.py busy $out/pmc_mfma > $out/pmc_busy.json
find $out -name "*kernel_stats.csv" | head
ls $out
