set -x
root=${GRAFT_REPO_ROOT:-$(pwd)}
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/r2z_tests.log; tail -3 gpurun_out/r2z_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python bench.py > gpurun_out/r2z_bench.json 2> gpurun_out/r2z_bench.err; tail -2 gpurun_out/r2z_bench.err
python bench.py --workload train > gpurun_out/r2z_train.json 2>/dev/null
python bench.py --workload train-permuto --steps 200 --warmup 50 > gpurun_out/r2z_trainp.json 2>/dev/null
python bench.py --workload dtu > gpurun_out/r2z_dtu.json 2>/dev/null
bash tools/prof.sh r2z_prof_frame --steps 20 --warmup 5 | tail -3
bash tools/prof.sh r2z_prof_train --workload train --steps 100 --warmup 30 | tail -3
bash tools/prof_bg.sh | tail -3
bash tools/traffic.sh | tail -12
bash tools/pmc.sh r2z_pmc_mlp "nt_mlp_bwd" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" --steps 3 --warmup 1 | tail -10
bash tools/pmc.sh r2z_pmc_mlp2 "nt_mlp_bwd" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" --steps 3 --warmup 1 | tail -10
