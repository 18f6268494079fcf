mkdir -p gpurun_out/r06
python -m pytest tests/test_methods.py::test_shared_appearance_models_on_the_neural_texture_branch_match_the_oracle -q -m gpu -s > gpurun_out/r06/new_tests.log 2>&1
grep "tex \|MEASURED shared" gpurun_out/r06/new_tests.log | grep -v print | tail -80
