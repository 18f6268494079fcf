cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_nt_texels_encode.py tests/test_nt_fused.py tests/test_nt_backward.py tests/test_pipeline_e2e.py tests/test_rebalance.py -m gpu -x -q 2>&1 | tail -5 > gpurun_out/t6.log; tail -5 gpurun_out/t6.log
STAGES="nt_encode_fwd nt_encode_bwd" bash tools/ab2.sh nopair pw22 pw30 flat > gpurun_out/ab_enc6.txt 2>&1; cat gpurun_out/ab_enc6.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
