root=${GRAFT_REPO_ROOT:-$(pwd)}
cp $root/volsurfs_amd/libvolsurfs_hip.so /tmp/base.so
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" != base ]; then cp $root/variants/lib_$v.so $root/volsurfs_amd/libvolsurfs_hip.so; else cp /tmp/base.so $root/volsurfs_amd/libvolsurfs_hip.so; fi
  rm -rf /tmp/abp_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp_$v -- python3 $root/bench.py --no-cpu-baseline --workload train-permuto --steps 50 --warmup 20 > /tmp/abp_$v.log 2>&1
  python3 - <<PY
import csv,glob,os,json
f=max(glob.glob("/tmp/abp_$v/*/*kernel_stats.csv"), key=os.path.getmtime)
rows=list(csv.DictReader(open(f)))
out=[]
for key in ("permuto_bwd","permuto_fwd","mlp_fwd_kernel","mlp_dgrad","mlp_wgrad","mlp_pack","adam"):
    rr=[r for r in rows if key in r['Name']]
    out.append(f"{key} {sum(float(r['TotalDurationNs']) for r in rr)/max(1,sum(int(r['Calls']) for r in rr))/1e3:7.1f}us")
step=[l for l in open("/tmp/abp_$v.log") if l.startswith("{")]
val=json.loads(step[-1])["value"] if step else float("nan")
print(f"$v".ljust(8), " ".join(out), f"| {val:.1f} it/s (profiled)")
PY
done
cp /tmp/base.so $root/volsurfs_amd/libvolsurfs_hip.so
