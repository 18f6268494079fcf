# usage: bash tools/span_ab.sh name... : tools/wg_span.py with each NT_SPAN variant library
cp volsurfs_amd/libvolsurfs_hip.so /tmp/base.so
for v in "$@"; do cp variants/lib_$v.so volsurfs_amd/libvolsurfs_hip.so; echo "== $v"; timeout 300 python tools/wg_span.py 2>&1 | grep -v amdgpu.ids; done
cp /tmp/base.so volsurfs_amd/libvolsurfs_hip.so
