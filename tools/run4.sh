cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -oE "\bSQ_[A-Z0-9_]+\b|\bTCP_[A-Z0-9_]+\b|\bTCC_[A-Z0-9_]+\b|\bTA_[A-Z0-9_]+\b|\bGRBM_[A-Z0-9_]+\b|\bSPI_[A-Z0-9_]+\b" | sort -u > gpurun_out/counters.txt; wc -l gpurun_out/counters.txt
bash tools/pmc.sh pmc_eb1 "nt_encode_bwd" "SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" --steps 3 --warmup 1 --no-noisy > gpurun_out/pmc_eb1.txt 2>&1; tail -24 gpurun_out/pmc_eb1.txt
bash tools/pmc.sh pmc_eb2 "nt_encode_bwd" "SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_IFETCH" --steps 3 --warmup 1 --no-noisy > gpurun_out/pmc_eb2.txt 2>&1; tail -24 gpurun_out/pmc_eb2.txt
bash tools/pmc.sh pmc_eb3 "nt_encode_bwd" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU" --steps 3 --warmup 1 --no-noisy > gpurun_out/pmc_eb3.txt 2>&1; tail -24 gpurun_out/pmc_eb3.txt
