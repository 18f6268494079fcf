cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_nt_texels_encode.py tests/test_nt_backward.py tests/test_rebalance.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/t3.log; tail -5 gpurun_out/t3.log
STAGES="nt_encode_fwd nt_encode_bwd" bash tools/ab2.sh oldbwd noreuse > gpurun_out/ab_enc3.txt 2>&1; cat gpurun_out/ab_enc3.txt
python -m pytest tests/test_bench_cli.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/t1.log; tail -5 gpurun_out/t1.log
