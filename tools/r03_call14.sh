cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r03c14; mkdir -p $O
timeout 1200 python -m pytest tests/test_raytrace.py tests/test_raygen.py -q -m gpu 2>&1 | tail -8 | tee $O/pytest_trace.txt
for fmt in q16x4 q16; do
VSA_TRACE_NODES=$fmt timeout 300 python bench.py --no-cpu-baseline --no-noisy --steps 30 --warmup 5 2>&1 | tail -1 | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); s=d[\"stages_ms\"]; print('$fmt', round(d[\"value\"],1), {k: round(s[k],4) for k in s if \"trace\" in k})" | tee -a $O/ab.txt
done
VSA_TRACE_NODES=q16x4 python - <<'PY' 2>&1 | tee -a $O/ab.txt
import torch, time, sys, os
sys.path.insert(0, ".")
from volsurfs_amd.pipeline import KShellPipeline
for fmt in ("q16x4", "q16"):
  os.environ["VSA_TRACE_NODES"] = fmt
  for kw in (dict(K=5, subdiv=6, noise=0.05, atlas_charts=6), dict(K=7, subdiv=7, res=(1080, 1920))):
    p = KShellPipeline.synthetic(**kw)
    o, d = (p._o_t, p._d_t) if p.image_hw else (p.rays_o, p.rays_d)
    p.step()
    for _ in range(3): p.tracer.trace_all(o, d)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): p.tracer.trace_all(o, d)
    torch.cuda.synchronize(); print(fmt, kw, "trace ms", round((time.perf_counter() - t0) / 20 * 1e3, 4), "depth", p.tracer.max_depth, p.tracer.max_depth4)
    del p
PY
