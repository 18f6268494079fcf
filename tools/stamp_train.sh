mkdir -p gpurun_out/r06
touch volsurfs_amd/csrc/nt_mlp.hip; make -C volsurfs_amd/csrc EXTRA=-DNT_STAMP > /dev/null 2>&1
python tools/wg_timeline_train.py > gpurun_out/r06/mlp_bwd_train_stamps.txt 2>&1
touch volsurfs_amd/csrc/nt_mlp.hip; make -C volsurfs_amd/csrc > /dev/null 2>&1
cat gpurun_out/r06/mlp_bwd_train_stamps.txt
