cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r03c7; mkdir -p $O
rm -f gpurun_out/parity_report.json
timeout 1800 python -m pytest tests/test_parity_report.py -q -m gpu 2>&1 | tail -30 | tee $O/pytest_parity.txt
cp gpurun_out/parity_report.json $O/ 2>/dev/null
timeout 1800 python -m pytest tests/test_pipeline_e2e.py tests/test_methods.py tests/test_parallel.py -q -m gpu -rP 2>&1 | grep -E "MEASURED|passed|failed" | tee $O/measured_bounds.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee -a $O/measured_bounds.txt
