# timing-only variants of the fused MLP backward, rebuilt on the GPU box (results wrong by construction)
mkdir -p gpurun_out/r06
out=gpurun_out/r06/mlp_diag.txt; : > $out
for d in 0 3; do
  touch volsurfs_amd/csrc/mlp_f32.hip
  make -C volsurfs_amd/csrc EXTRA=-DFB_DIAG=$d > /dev/null 2>&1 || { echo "build failed $d" >> $out; continue; }
  echo "FB_DIAG=$d" >> $out
  python tools/bench_mlp.py --iters 3 >> $out 2>&1
done
touch volsurfs_amd/csrc/mlp_f32.hip; make -C volsurfs_amd/csrc > /dev/null 2>&1
cat $out
