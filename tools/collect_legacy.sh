# usage (via gpurun): bash tools/collect_legacy.sh : the legacy / background measurements of profiles/r02
# (dtu frame, config-3 training loop, their rocprof kernel statistics) into gpurun_out/legacy_*
set -x
root=${GRAFT_REPO_ROOT:-$(pwd)}
python bench.py --workload dtu > gpurun_out/legacy_dtu.json 2>/dev/null
python bench.py --workload train-permuto --steps 1000 --warmup 100 > gpurun_out/legacy_trainp.json 2>/dev/null
python bench.py --workload train > gpurun_out/legacy_train.json 2>/dev/null
python tools/bench_bg.py > gpurun_out/legacy_bg.json 2>/dev/null
bash tools/prof.sh legacy_prof_trainp --workload train-permuto --steps 30 --warmup 10 | tail -3
bash tools/prof_bg.sh | tail -3
tail -c 600 gpurun_out/legacy_dtu.json gpurun_out/legacy_trainp.json gpurun_out/legacy_bg.json
