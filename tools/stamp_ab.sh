# usage: bash tools/stamp_ab.sh name... : runs tools/wg_timeline.py with each NT_STAMP variant library
cp volsurfs_amd/libvolsurfs_hip.so /tmp/base.so
for v in "$@"; do cp variants/lib_$v.so volsurfs_amd/libvolsurfs_hip.so; echo "== $v"; timeout 300 python tools/wg_timeline.py 2>&1 | grep -v amdgpu.ids | head -12; done
cp /tmp/base.so volsurfs_amd/libvolsurfs_hip.so
