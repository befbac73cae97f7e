#!/bin/bash
# Diagnostic: the measurement passes behind profiles/round5 (run on the GPU box from the repo root: bash tools/profile_round5.sh <outdir>)
out=${1:-gpurun_out/prof5}
export TMPDIR=/tmp
mkdir -p $out
python3 bench.py --steps 10 --warmup 3 > $out/bench_default.json 2> $out/bench_default.err
rocprofv3 --kernel-trace --stats -d $out/prof -o r --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_under_rocprof.json 2> $out/rocprof.err
rocprofv3 --kernel-trace --stats -d $out/prof_b512 -o r --output-format csv -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-extras --batch 512 > $out/bench_b512_under_rocprof.json 2>> $out/rocprof.err
rocprofv3 --kernel-trace --stats -d $out/prof_gather -o r --output-format csv -- python3 tools/kbench.py gather --reps 20 > $out/kbench_gather.txt 2>> $out/rocprof.err
rocprofv3 --kernel-trace --stats -d $out/prof_gather_v100m -o r --output-format csv -- python3 tools/kbench.py gather --reps 20 --B 1024 --gather-F 40 --gather-rows 2500000 > $out/kbench_gather_v100m.txt 2>> $out/rocprof.err
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $out/pmc_fetch -o f --output-format csv -- python3 tools/kbench.py attn_bwd attn_fwd ffn_fwd ffn_bwd --arith bf16x3 --reps 2 > $out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $out/pmc_write -o w --output-format csv -- python3 tools/kbench.py attn_bwd attn_fwd ffn_fwd ffn_bwd --arith bf16x3 --reps 2 > $out/pmc_write.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d $out/pmc_mfma -o m --output-format csv -- python3 tools/kbench.py attn_bwd attn_fwd ffn_fwd ffn_bwd --arith bf16x3 --reps 2 > $out/pmc_mfma.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE --kernel-trace -d $out/pmc_coexec -o c --output-format csv -- python3 tools/kbench.py attn_bwd attn_fwd ffn_fwd ffn_bwd --arith bf16x3 --reps 2 > $out/pmc_coexec.log 2>&1
python3 tools/pmc_summary.py traffic $out/pmc_fetch $out/pmc_write > $out/traffic_pmc.json
python3 tools/pmc_summary.py coexec $out/pmc_coexec > $out/pmc_coexec.json
python3 tools/pmc_summary.py busy $out/pmc_mfma > $out/pmc_busy.json
python3 tools/phase_profile.py > $out/phase_shares.txt 2>&1
for wl in mltag_like_K10_d16_B256 kkbox_like_F13_K10_d64_B4096 tmall_like_F8_K30_d64_h32_B4096 kkbox_real_F13_K5_d40_B4096 synthetic_F40_V100M_K10_d64_B1024 movielens_real_F3_K5_d10_B4096 tmall_real_F9_K5_d10_h32_B4096; do
  python3 bench.py --workload $wl --no-cpu-baseline --no-extras > $out/bench_$wl.json 2> $out/bench_$wl.err
done
python3 bench.py --steps 10 --warmup 3 --dp-rehearsal > $out/bench_dp_rehearsal.json 2> $out/bench_dp_rehearsal.err
rocprofv3 --kernel-trace --stats -d $out/prof_mltag -o r --output-format csv -- python3 bench.py --workload mltag_like_K10_d16_B256 --steps 50 --warmup 3 --no-cpu-baseline --no-extras > $out/bench_mltag_under_rocprof.json 2>> $out/rocprof.err
for m in RAT_m0 RAT_m1 RAT_m3; do python3 bench.py --model $m --no-cpu-baseline --no-extras > $out/bench_$m.json 2> $out/bench_$m.err; done
find $out -name "*kernel_stats.csv" | head
ls $out
