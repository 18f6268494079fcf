cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r03c15; mkdir -p $O
B='timeout 300 python bench.py --no-cpu-baseline --no-noisy --steps 30 --warmup 5 2>&1 | tail -1 | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); s=d[\"stages_ms\"]; print(round(d[\"value\"],1), {k: round(s[k],4) for k in s if \"shade\" in k})"'
echo base; eval $B | tee -a $O/ab.txt
STAGES="nt_shade_fwd nt_shade_bwd" bash tools/ab_variants.sh sf4 sf5 sb5 sb6 sb8 2>&1 | tee -a $O/ab.txt
echo base; eval $B | tee -a $O/ab.txt
