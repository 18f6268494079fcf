cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
cp volsurfs_amd/libvolsurfs_hip.so /tmp/base.so
for v in "$@"; do
  cp variants/lib_$v.so volsurfs_amd/libvolsurfs_hip.so; echo "== $v"
  timeout 600 python -m pytest tests/test_nt_fused.py -q 2>&1 | tail -8
done
cp /tmp/base.so volsurfs_amd/libvolsurfs_hip.so
