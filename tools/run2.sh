cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_bench_cli.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/t1.log; tail -5 gpurun_out/t1.log
python -m pytest tests/test_methods.py -m gpu -x -q -k "config2" -s 2>&1 | grep -E "MEASURED|passed|failed|Error|assert" | head -20 > gpurun_out/t2.log; cat gpurun_out/t2.log
STAGES="nt_encode_fwd nt_encode_bwd" bash tools/ab2.sh pf1 pf2 u8 u8pf1 bnoat bnofl bpf2 > gpurun_out/ab_enc2.txt 2>&1; cat gpurun_out/ab_enc2.txt
