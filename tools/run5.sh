cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/parity_report.json
python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > gpurun_out/r04_tests.log; tail -15 gpurun_out/r04_tests.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r04_smoke.log
