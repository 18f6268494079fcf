# usage: [STAGES="a b"] [ROUNDS=2] bash tools/ab2.sh name... : like ab_variants.sh, but the built library
# ("base") runs beside the variants and the whole list is repeated ROUNDS times in one process tree on
# one box (interleaved rounds: guide rule 24), so that a drift of the box shows up as a spread of "base".
ROUNDS=${ROUNDS:-2}
B='timeout 300 python bench.py --no-cpu-baseline --no-noisy --steps 30 --warmup 5 2>&1 | tail -1 | python -c "import json,sys,os; d=json.loads(sys.stdin.read()); s=d[\"stages_ms\"]; ks=os.environ.get(\"STAGES\",\"\").split() or list(s); print(round(d[\"value\"],1), {k: round(s[k],4) for k in ks if k in s})"'
cp volsurfs_amd/libvolsurfs_hip.so /tmp/base.so
trap 'cp /tmp/base.so volsurfs_amd/libvolsurfs_hip.so' EXIT
for r in $(seq $ROUNDS); do
  for v in base "$@"; do
    if [ $v = base ]; then cp /tmp/base.so volsurfs_amd/libvolsurfs_hip.so; else cp variants/lib_$v.so volsurfs_amd/libvolsurfs_hip.so; fi
    echo -n "$v: "; eval $B
  done
done
cp /tmp/base.so volsurfs_amd/libvolsurfs_hip.so
