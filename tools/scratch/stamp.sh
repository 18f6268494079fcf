cp volsurfs_amd/libvolsurfs_hip.so /tmp/base.so
cp variants/lib_stamp.so volsurfs_amd/libvolsurfs_hip.so
python - <<'PY' 2>&1 | grep -E "wave|active|WG|cycles" | head -60
import torch, sys
sys.path.insert(0, ".")
from volsurfs_amd.pipeline import KShellPipeline
p = KShellPipeline.synthetic()
for _ in range(2):
    p.step()
torch.cuda.synchronize()
PY
python tools/wg_timeline.py 2>&1 | tail -12
cp /tmp/base.so volsurfs_amd/libvolsurfs_hip.so
