cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r03c2; mkdir -p $O
STAGES="nt_encode_mlp_fwd" bash tools/ab_variants.sh fu_nogather fu_nomlp fu_nogather_nomlp fu_noxcd fu_b2 fu_b8 fu_b16 fu_wg2 2>&1 | tee $O/ab.txt
B='timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); s=d[\"stages_ms\"]; print(round(d[\"value\"],1), {k: round(s[k],4) for k in s if \"fwd\" in k})"'
echo base; eval $B | tee -a $O/ab.txt
(rocprofv3 -L > $O/counters_list.txt 2>&1 || true)
bash tools/pmc.sh r03c2/pmc1 "nt_encmlp" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --steps 3 --warmup 2 | tee $O/pmc1.txt
bash tools/pmc.sh r03c2/pmc2 "nt_encmlp" "SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_LDS SQ_WAVES SQ_INST_CYCLES_VMEM" --steps 3 --warmup 2 | tee $O/pmc2.txt
bash tools/pmc.sh r03c2/pmc3 "nt_encmlp" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" --steps 3 --warmup 2 | tee $O/pmc3.txt
bash tools/pmc.sh r03c2/pmc4 "nt_encmlp" "TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" --steps 3 --warmup 2 | tee $O/pmc4.txt
