cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r03c9; mkdir -p $O
timeout 1200 python -m pytest tests/test_packed.py tests/test_methods.py -q -m gpu 2>&1 | tail -15 | tee $O/pytest_bg.txt
timeout 600 python tools/bench_bg.py --steps 5 2>&1 | tail -3 | tee $O/bench_bg.txt
timeout 600 python bench.py --workload dtu --steps 2 2>/dev/null | tail -1 | tee $O/bench_dtu.json
