cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/r2f
rocprofv3 --list-avail 2>/dev/null | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*\|SQ_INSTS_[A-Z_]*\|SQ_WAIT_[A-Z_]*\|SQ_ACTIVE_INST_[A-Z_]*\|SQ_INST_CYCLES_[A-Z_]*" | sort -u > gpurun_out/r2f/avail.txt
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_LDS_ADDR_CONFLICT --kernel-trace -d gpurun_out/r2f/pmc_lds -o l --output-format csv -- python3 tools/kbench.py attn_bwd attn_fwd --arith bf16x3 --reps 2 > gpurun_out/r2f/pmc_lds.log 2>&1
tail -3 gpurun_out/r2f/pmc_lds.log
