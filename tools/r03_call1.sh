# round 3, first GPU call: fused encode+MLP forward — parity, then same-box A/B against the two-kernel path
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r03c1; mkdir -p $O
timeout 900 python -m pytest tests/test_nt_fused.py -x -q 2>&1 | tail -15 > $O/pytest_fused.txt
cat $O/pytest_fused.txt
for f in 0 1 0 1; do
  VSA_NT_FUSED=$f timeout 300 python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>$O/bench_f$f.err | tail -1 > $O/bench_f$f.json
  python - <<PY
import json
d=json.load(open("$O/bench_f$f.json")); s=d["stages_ms"]
print("fused=$f", round(d["value"],1), "Mrays/s", {k: round(v,4) for k,v in s.items() if "encode" in k or "mlp" in k})
PY
done
timeout 300 python bench.py --workload render --steps 30 2>/dev/null | tail -1 > $O/render_f1.json
VSA_NT_FUSED=0 timeout 300 python bench.py --workload render --steps 30 2>/dev/null | tail -1 > $O/render_f0.json
python - <<PY
import json
for f in (0,1):
    d=json.load(open("$O/render_f%d.json"%f)); print("render fused=%d"%f, round(d["value"],1), d["ms_per_step"])
PY
timeout 600 python -m pytest tests/test_bench_cli.py -x -q -m gpu 2>&1 | tail -5 > $O/pytest_bench.txt; cat $O/pytest_bench.txt
