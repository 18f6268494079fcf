# usage (via gpurun, AFTER tools/finish_round.py has rebuilt profiles/traffic.json and pmc_summary.json at the final kernel
# sources): bash tools/collect_lines.sh <tag> — the frame bench lines again, now carrying `roofline.traffic` / `pmc`
tag=${1:-r05}
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd $root; mkdir -p gpurun_out
python bench.py --no-cpu-baseline --no-noisy --steps 100 > /dev/null 2>&1      # a fresh box's first process runs slow: not the one that is kept
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err; tail -2 gpurun_out/${tag}_bench.err
python bench.py --res 1080 --width 1920 --shells 7 --subdiv 8 --no-cpu-baseline --no-noisy --steps 50 > gpurun_out/${tag}_bench_1080p_K7_subdiv8.json 2>/dev/null
python bench.py --res 1080 --width 1920 --shells 7 --subdiv 8 --no-cpu-baseline --no-noisy --steps 50 --cold > gpurun_out/${tag}_bench_1080p_K7_subdiv8_cold.json 2>/dev/null
python bench.py --gpus 1 --force-dist --dist-backend nccl --no-cpu-baseline --no-noisy > gpurun_out/${tag}_bench_rccl_one_rank.json 2>/dev/null
python - <<PY
import json
for f in ("bench", "bench_1080p_K7_subdiv8", "bench_rccl_one_rank"):
    d = json.load(open("gpurun_out/${tag}_%s.json" % f))
    r = d["roofline"]
    print(f, round(d["value"], 1), r["kernel"], round(r["frac"], 4), r.get("traffic"), (r.get("pmc") or {}).get("mfma_busy"))
PY
