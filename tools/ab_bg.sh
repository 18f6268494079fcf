# usage: tools/ab_bg.sh name...  : tools/bench_bg.py step time per prebuilt variants/lib_<name>.so (one box)
cp volsurfs_amd/libvolsurfs_hip.so /tmp/base.so
for v in "$@"; do cp variants/lib_$v.so volsurfs_amd/libvolsurfs_hip.so; echo -n "$v "; python tools/bench_bg.py 2>/dev/null | python -c "import json,sys; print(round(json.loads(sys.stdin.read())['ms_per_step'],2))"; done
cp /tmp/base.so volsurfs_amd/libvolsurfs_hip.so
