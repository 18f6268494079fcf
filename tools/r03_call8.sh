cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r03c8; mkdir -p $O
for m in 1 16 256; do
rm -f gpurun_out/parity_report.json
VSA_TEST_GRAD_SCALE_MULT=$m timeout 900 python -m pytest "tests/test_parity_report.py::test_order_matched_rgb_and_f32_gradients[7-4-128-0.05]" -q -m gpu 2>&1 | tail -3
python - <<PY
import json
d=json.load(open("gpurun_out/parity_report.json"))["order_matched_K7_res128"]
print("mult=$m", "tables", d["grad_tables_err_rel_to_tensor_max"], "weights", d["grad_weights_err_rel_to_tensor_max"]["max"], "frac>1e-3", d["grad_tables_frac_over_1e-3"])
PY
done 2>&1 | tee $O/grad_scale_experiment.txt
