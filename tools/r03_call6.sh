cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r03c6; mkdir -p $O
rm -f gpurun_out/parity_report.json
timeout 1800 python -m pytest tests/test_parity_report.py -q -m gpu 2>&1 | tail -40 | tee $O/pytest_parity.txt
cp gpurun_out/parity_report.json $O/ 2>/dev/null
timeout 2400 python -m pytest tests -q -m gpu --deselect tests/test_parity_report.py 2>&1 | tail -15 | tee $O/pytest_gpu.txt
timeout 600 python bench.py --steps 100 2>$O/bench.err | tail -1 > $O/bench.json
python - <<PY
import json
d=json.load(open("$O/bench.json")); s=d["stages_ms"]
print(round(d["value"],1), "Mrays/s; noisy", round(d["value_noisy"],1), d["noisy"], d["roofline"], d["cpu_baseline"])
print({k: round(v,4) for k,v in s.items()})
PY
