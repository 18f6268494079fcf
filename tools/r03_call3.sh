cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r03c4; mkdir -p $O
timeout 900 python -m pytest tests/test_nt_fused.py -x -q 2>&1 | tail -15 | tee $O/pytest_fused.txt
B='timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); s=d[\"stages_ms\"]; print(round(d[\"value\"],1), {k: round(s[k],4) for k in s if \"fwd\" in k})"'
echo base; eval $B | tee -a $O/ab.txt
STAGES="nt_encode_mlp_fwd" bash tools/ab_variants.sh fu_nostage fu_st_nomlp 2>&1 | tee -a $O/ab.txt
echo base; eval $B | tee -a $O/ab.txt
bash tools/pmc.sh r03c4/pmc3 "nt_encmlp" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" --steps 3 --warmup 2 | tee $O/pmc3.txt
bash tools/pmc.sh r03c4/pmc1 "nt_encmlp" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" --steps 3 --warmup 2 | tee $O/pmc1.txt
