# A/B of the experimental one-role MLP backward (variants/lib_t.so: tools/build_variant.sh t "-DNT_MLP_BWD_T -mllvm -amdgpu-mfma-vgpr-form=1";
# variants/lib_tstamp.so: the same + -DNT_STAMP) against the default library.  Run on the GPU box.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
cp volsurfs_amd/libvolsurfs_hip.so /tmp/base0.so
B='timeout 300 python bench.py --no-cpu-baseline --no-noisy --steps 30 --warmup 5 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d[\"stages_ms\"]; print(round(d[\"value\"],1), s[\"nt_mlp_bwd\"])"'
cp variants/lib_t.so volsurfs_amd/libvolsurfs_hip.so
VSA_NT_MLP_BWD=t python -m pytest tests/test_nt_backward.py tests/test_nt_mlp.py tests/test_pipeline_e2e.py -m gpu -x -q 2>&1 | tail -3
for r in 1 2; do
cp /tmp/base0.so volsurfs_amd/libvolsurfs_hip.so
echo -n "default (pc): "; bash -c "$B"
cp variants/lib_t.so volsurfs_amd/libvolsurfs_hip.so
echo -n "t: "; VSA_NT_MLP_BWD=t bash -c "$B"
done 2>&1 | tee gpurun_out/ab_mlp_t.txt
if [ -f variants/lib_tstamp.so ]; then
cp variants/lib_tstamp.so volsurfs_amd/libvolsurfs_hip.so
VSA_NT_MLP_BWD=t timeout 300 python tools/t_stamps.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/t_stamps.txt
fi
cp /tmp/base0.so volsurfs_amd/libvolsurfs_hip.so
