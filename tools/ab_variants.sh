B='timeout 300 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d[\"stages_ms\"]; print(round(d[\"value\"],1), {k: s[k] for k in (\"nt_mlp_bwd\",\"nt_mlp_fwd\")})"'
cp volsurfs_amd/libvolsurfs_hip.so /tmp/base.so
for v in "$@"; do cp variants/lib_$v.so volsurfs_amd/libvolsurfs_hip.so; echo $v; eval $B; done
cp /tmp/base.so volsurfs_amd/libvolsurfs_hip.so
