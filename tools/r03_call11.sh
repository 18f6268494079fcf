cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
export TMPDIR=/tmp
O=gpurun_out/r03c11; mkdir -p $O
python - <<'PY' 2>&1 | tee $O/composite_tiles.txt
# composite stage alone, kernel-only timing with events over many launches
import torch, sys, shutil, subprocess, os
sys.path.insert(0, ".")
PY
for v in base comp_t64 comp_t128 comp_t512; do
  if [ $v != base ]; then cp volsurfs_amd/libvolsurfs_hip.so /tmp/base.so; cp variants/lib_$v.so volsurfs_amd/libvolsurfs_hip.so; fi
  python - <<PY 2>&1 | tee -a $O/composite_tiles.txt
import torch, sys
sys.path.insert(0, ".")
from volsurfs_amd.composite import composite_fwd_bwd_l1_raw
N, K = 640000, 5
g = torch.Generator(device="cuda").manual_seed(0)
c = torch.rand(N, K, 3, device="cuda", generator=g); a = torch.rand(N, K, device="cuda", generator=g)
gt = torch.rand(N, 3, device="cuda", generator=g); bg = torch.ones(1, 3, device="cuda")
for _ in range(10): composite_fwd_bwd_l1_raw(c, a, bg, gt, 1e-6)
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); s.record()
for _ in range(200): composite_fwd_bwd_l1_raw(c, a, bg, gt, 1e-6)
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 200 * 1e3
print("$v", "us per fused composite+loss+bwd (incl. 3 torch.empty allocs):", round(us, 2), "GB/s", round(N * 184 / us / 1e3, 1))
PY
  if [ $v != base ]; then cp /tmp/base.so volsurfs_amd/libvolsurfs_hip.so; fi
done
