cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r03c13; mkdir -p $O
R="trace_q|nt_shade|nt_encode|nt_mlp|nt_assign|nt_mark|composite"
bash tools/pmc.sh r03c13/pmcA "$R" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --steps 3 --warmup 2 --no-noisy > $O/pmcA.txt
bash tools/pmc.sh r03c13/pmcB "$R" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" --steps 3 --warmup 2 --no-noisy > $O/pmcB.txt
bash tools/pmc.sh r03c13/pmcC "$R" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" --steps 3 --warmup 2 --no-noisy > $O/pmcC.txt
tail -5 $O/pmcA.txt
