mkdir -p gpurun_out/r06
python -m pytest tests/test_fused_mlp.py tests/test_legacy_models.py "tests/test_methods.py::test_dtu_config_full_size_learned_background" tests/test_methods.py::test_learned_background_path tests/test_methods.py::test_config2_permutohash_K5_noisy_shells_oracle_parity_gradients_and_training tests/test_pipeline_e2e.py::test_full_size_frame_properties -q -m gpu -x > gpurun_out/r06/mlp_tests.log 2>&1
tail -8 gpurun_out/r06/mlp_tests.log
bash tools/_diag_mlp.sh
python bench.py --workload dtu --steps 2 --warmup 1 > gpurun_out/r06/bench_dtu_fused.json 2> gpurun_out/r06/bench_dtu_fused.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/bench_dtu_fused.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_batch'])
for k,v in list(d['kernels_ms_per_batch'].items())[:8]: print(' ',k,v)
PY
