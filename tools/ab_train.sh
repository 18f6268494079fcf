cp volsurfs_amd/libvolsurfs_hip.so /tmp/base.so
for v in "$@"; do
  if [ "$v" != base ]; then cp variants/lib_$v.so volsurfs_amd/libvolsurfs_hip.so; else cp /tmp/base.so volsurfs_amd/libvolsurfs_hip.so; fi
  echo $v; timeout 300 python bench.py --no-cpu-baseline --workload train --steps 300 --warmup 100 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); k=d.get('kernels_ms') or {}; print(round(d['value'],1), round(d['ms_per_step'],4), {n[4:]: v['ms_per_iter'] for n, v in k.items() if v['ms_per_iter'] > 0.02})"
done
cp /tmp/base.so volsurfs_amd/libvolsurfs_hip.so
