cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r03c5; mkdir -p $O
timeout 600 python tools/diag_table_grad.py 3 3 56 2>&1 | tail -4 | tee $O/diag_table_grad.txt
timeout 2400 python -m pytest tests -q -m gpu -x --deselect tests/test_parity_report.py 2>&1 | tail -15 | tee $O/pytest_gpu.txt
timeout 900 python -m pytest tests/test_parity_report.py -q -m gpu 2>&1 | tail -15 | tee $O/pytest_parity.txt
cp gpurun_out/parity_report.json $O/ 2>/dev/null
for f in 0 1; do
  VSA_NT_FUSED=$f timeout 300 python bench.py --no-cpu-baseline --steps 50 --warmup 5 2>$O/bench_f$f.err | tail -1 > $O/bench_f$f.json
  python - <<PY
import json
d=json.load(open("$O/bench_f$f.json")); s=d["stages_ms"]
print("fused=$f", round(d["value"],1), "Mrays/s", {k: round(v,4) for k,v in s.items()})
PY
done
