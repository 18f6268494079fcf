# usage (via gpurun): bash tools/collect_round.sh <tag>   e.g. r03 — the round's measurements into gpurun_out/<tag>_*
set -x
tag=${1:-r03}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/${tag}_tests.log; tail -3 gpurun_out/${tag}_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/${tag}_smoke.log
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -2 gpurun_out/${tag}_bench.err
python bench.py --workload train > gpurun_out/${tag}_train.json 2>/dev/null
python bench.py --workload train-permuto --steps 1000 --warmup 100 > gpurun_out/${tag}_trainp.json 2>/dev/null
python bench.py --workload dtu > gpurun_out/${tag}_dtu.json 2>/dev/null
python bench.py --workload render > gpurun_out/${tag}_render.json 2>/dev/null
VSA_NT_FUSED=1 python bench.py --no-cpu-baseline --no-noisy --steps 100 > gpurun_out/${tag}_bench_fused.json 2>/dev/null
VSA_NT_FUSED=0 python bench.py --workload render > gpurun_out/${tag}_render_unfused.json 2>/dev/null
bash tools/prof.sh ${tag}_prof_frame --steps 20 --warmup 5 --no-noisy | tail -3
bash tools/prof.sh ${tag}_prof_render --workload render --steps 20 --warmup 5 | tail -3
bash tools/prof.sh ${tag}_prof_train --workload train --steps 100 --warmup 30 | tail -3
bash tools/prof_bg.sh | tail -3
bash tools/traffic.sh | tail -12
bash tools/pmc.sh ${tag}_pmc_mlp "nt_mlp_bwd" "SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" --steps 3 --warmup 1 --no-noisy | tail -10
bash tools/pmc.sh ${tag}_pmc_mlp2 "nt_mlp_bwd" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY GRBM_GUI_ACTIVE" --steps 3 --warmup 1 --no-noisy | tail -10
