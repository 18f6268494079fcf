# usage: bash tools/ab_variants.sh name... : whole-step Mrays/s and the per-stage ms with each variants/lib_<name>.so
# (STAGES="a b" limits the stages printed)
B='timeout 300 python bench.py --no-cpu-baseline --no-noisy --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); s=d[\"stages_ms\"]; ks=os.environ.get(\"STAGES\",\"\").split() or list(s); print(round(d[\"value\"],1), {k: round(s[k],4) for k in ks if k in s})"'
cp volsurfs_amd/libvolsurfs_hip.so /tmp/base.so
for v in "$@"; do cp variants/lib_$v.so volsurfs_amd/libvolsurfs_hip.so; echo $v; eval $B; done
cp /tmp/base.so volsurfs_amd/libvolsurfs_hip.so
