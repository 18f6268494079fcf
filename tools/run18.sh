cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp volsurfs_amd/libvolsurfs_hip.so /tmp/base0.so
cp variants/lib_span.so volsurfs_amd/libvolsurfs_hip.so
VSA_NT_REBALANCE=0 timeout 300 python tools/span_profile.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/span_profile.txt
VSA_NT_REBALANCE=0 timeout 300 python tools/wg_span.py 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/span_profile.txt
cp /tmp/base0.so volsurfs_amd/libvolsurfs_hip.so
