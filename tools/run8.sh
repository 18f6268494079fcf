cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_pipeline_e2e.py -m gpu -x -q -s -k "stress or full_size" 2>&1 | grep -E "MEASURED|passed|failed|Error|assert|error" | head -20 > gpurun_out/t8.log; cat gpurun_out/t8.log
python -m pytest tests/test_parity_report.py -m gpu -x -q 2>&1 | tail -3
