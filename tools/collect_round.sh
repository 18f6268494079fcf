# usage (via gpurun): bash tools/collect_round.sh <tag> [notests]   e.g. r04 — the round's measurements into gpurun_out/<tag>_*
set -x
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; export TMPDIR=/tmp
mkdir -p gpurun_out
if [ "$2" != "notests" ]; then
  rm -f gpurun_out/parity_report.json
  python -m pytest tests -m gpu -q --durations=25 2>&1 | tail -40 > gpurun_out/${tag}_tests.log; tail -3 gpurun_out/${tag}_tests.log
fi
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/${tag}_smoke.log
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -2 gpurun_out/${tag}_bench.err
python bench.py --workload train > gpurun_out/${tag}_train.json 2>/dev/null
python bench.py --workload train-permuto --steps 1000 --warmup 100 > gpurun_out/${tag}_trainp.json 2>/dev/null
python bench.py --workload dtu > gpurun_out/${tag}_dtu.json 2>/dev/null
python bench.py --workload render > gpurun_out/${tag}_render.json 2>/dev/null
python bench.py --res 1080 --width 1920 --shells 7 --subdiv 8 --no-cpu-baseline --no-noisy --steps 50 > gpurun_out/${tag}_bench_1080p_K7_subdiv8.json 2>/dev/null
bash tools/prof.sh ${tag}_prof_k7 --res 1080 --width 1920 --shells 7 --subdiv 8 --no-noisy --steps 10 --warmup 2 | tail -3
python bench.py --res 1080 --width 1920 --shells 7 --subdiv 8 --no-cpu-baseline --no-noisy --steps 50 --cold > gpurun_out/${tag}_bench_1080p_K7_subdiv8_cold.json 2>/dev/null
python bench.py --gpus 1 --force-dist --dist-backend nccl --no-cpu-baseline --no-noisy > gpurun_out/${tag}_bench_rccl_one_rank.json 2>/dev/null
bash tools/prof.sh ${tag}_prof_frame --steps 20 --warmup 5 --no-noisy | tail -3
bash tools/prof.sh ${tag}_prof_render --workload render --steps 20 --warmup 5 | tail -3
bash tools/prof.sh ${tag}_prof_train --workload train --steps 100 --warmup 30 | tail -3
bash tools/prof.sh ${tag}_prof_trainp --workload train-permuto --steps 100 --warmup 30 | tail -3
bash tools/prof.sh ${tag}_prof_dtu --workload dtu --steps 1 --warmup 1 | tail -3
bash tools/traffic.sh frame | tail -12
bash tools/traffic.sh k7 --res 1080 --width 1920 --shells 7 --subdiv 8 | tail -12
P="--steps 3 --warmup 1 --no-noisy"
bash tools/pmc.sh ${tag}_pmc_a "nt_mlp|nt_encode|nt_shade|trace_qf" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE" $P | tail -4
bash tools/pmc.sh ${tag}_pmc_b "nt_mlp|nt_encode|nt_shade|trace_qf" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" $P | tail -4
bash tools/pmc.sh ${tag}_pmc_c "nt_mlp|nt_encode|nt_shade|trace_qf" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" $P | tail -4
bash tools/pmc.sh ${tag}_pmc_d "nt_mlp|nt_encode|nt_shade|trace_qf" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_ACCESSES_sum" $P | tail -4
K7="--res 1080 --width 1920 --shells 7 --subdiv 8"
bash tools/pmc.sh ${tag}_pmck7_a "nt_mlp|nt_encode|nt_shade|trace_qf" "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" $P $K7 | tail -4
# the frame lines once more, now that traffic.json / pmc_summary.json can be rebuilt at these sources (on the box: only
# gpurun_out/ travels back; run tools/finish_round.py again at home)
python tools/finish_round.py $tag > /dev/null 2>&1
bash tools/collect_lines.sh $tag | tail -4
