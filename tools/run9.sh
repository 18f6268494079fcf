cd $GRAFT_REPO_ROOT
STAGES="nt_mlp_bwd nt_mlp_fwd" ROUNDS=2 bash tools/ab2.sh dabs s72 s76 s48 s72s48 > gpurun_out/ab_mlp9.txt 2>&1; cat gpurun_out/ab_mlp9.txt
