# usage: bash tools/ab_legacy.sh name... : the legacy / background paths (dtu frame, config-3 training loop) with each variant library ("base" = the built one)
cp volsurfs_amd/libvolsurfs_hip.so /tmp/base.so
for v in "$@"; do
  if [ "$v" != base ]; then cp variants/lib_$v.so volsurfs_amd/libvolsurfs_hip.so; else cp /tmp/base.so volsurfs_amd/libvolsurfs_hip.so; fi
  echo $v
  timeout 600 python bench.py --no-cpu-baseline --workload dtu --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  dtu', round(d['value'],3), d['unit'], round(d['ms_per_step'],1), 'ms/frame')"
  timeout 600 python bench.py --no-cpu-baseline --workload train-permuto --steps 100 --warmup 30 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  train-permuto', round(d['value'],1), d['unit'])"
done
cp /tmp/base.so volsurfs_amd/libvolsurfs_hip.so
