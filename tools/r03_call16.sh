cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r03c16; mkdir -p $O
for f in 0 1 0 1; do
VSA_NT_FUSED=$f timeout 300 python bench.py --workload train 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('train fused=$f', round(d['value'],1), 'it/s', round(d['ms_per_step'],4), 'fixed', round(d['fixed_ms_per_iter'],4))" | tee -a $O/train_ab.txt
done
for f in 0 1; do
VSA_NT_FUSED=$f timeout 300 python bench.py --workload render --steps 30 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('render fused=$f', round(d['value'],1), round(d['ms_per_step'],4))" | tee -a $O/train_ab.txt
done
