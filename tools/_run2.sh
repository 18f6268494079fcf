mkdir -p gpurun_out/r06
python -m pytest tests/test_fused_mlp.py tests/test_legacy_models.py "tests/test_methods.py::test_dtu_config_full_size_learned_background" tests/test_methods.py::test_learned_background_path -q -m gpu > gpurun_out/r06/mlp_tests.log 2>&1
tail -5 gpurun_out/r06/mlp_tests.log
bash tools/_diag_mlp.sh
