cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r03c12; mkdir -p $O
timeout 1200 python -m pytest tests/test_parallel.py tests/test_optim.py tests/test_training_loop.py -q -m gpu 2>&1 | tail -15 | tee $O/pytest.txt


