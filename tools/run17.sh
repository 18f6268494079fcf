cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_nt_mlp.py tests/test_nt_fused.py tests/test_pipeline_e2e.py -m gpu -x -q 2>&1 | tail -3
ROUNDS=3 STAGES="nt_mlp_fwd" bash tools/ab2.sh qold 2>&1 | tee gpurun_out/ab_q.txt
