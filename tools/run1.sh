cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
./tools/ubench/valu_rate > gpurun_out/valu_rate.txt 2>&1; cat gpurun_out/valu_rate.txt
python -m pytest tests/test_optim.py tests/test_parallel.py tests/test_cabi.py tests/test_bench_cli.py "tests/test_fused_mlp.py::test_mlp_outside_the_fused_shapes_raises_unless_the_torch_path_is_asked_for" tests/test_raytrace.py -m gpu -x -q 2>&1 | tail -15 > gpurun_out/t1.log; tail -15 gpurun_out/t1.log
python -m pytest tests/test_methods.py -m gpu -x -q -k "config2" -s 2>&1 | grep -E "MEASURED|passed|failed|Error|assert" | head -20 > gpurun_out/t2.log; cat gpurun_out/t2.log
STAGES="nt_encode_fwd nt_encode_bwd" bash tools/ab_variants.sh fxy fnost fxyst bwin bxy > gpurun_out/ab_enc.txt 2>&1
cp volsurfs_amd/libvolsurfs_hip.so /tmp/keep.so
STAGES="nt_encode_fwd nt_encode_bwd" B0=1 bash -c 'timeout 300 python bench.py --no-cpu-baseline --no-noisy --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d[\"stages_ms\"]; print(\"base\", round(d[\"value\"],1), s[\"nt_encode_fwd\"], s[\"nt_encode_bwd\"])"' >> gpurun_out/ab_enc.txt 2>&1
cat gpurun_out/ab_enc.txt
timeout 900 python bench.py > gpurun_out/bench1.json 2> gpurun_out/bench1.err; tail -c 3000 gpurun_out/bench1.json; tail -3 gpurun_out/bench1.err
