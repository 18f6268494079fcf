cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_nt_texels_encode.py tests/test_nt_backward.py tests/test_pipeline_e2e.py tests/test_parallel.py -m gpu -x -q 2>&1 | tail -3
ROUNDS=3 STAGES="nt_encode_fwd nt_encode_bwd" bash tools/ab2.sh two 2>&1 | tee gpurun_out/ab_one.txt
