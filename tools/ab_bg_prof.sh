# usage: bash tools/ab_bg_prof.sh name... : per-kernel averages of the fused fp32 MLP and hash-grid kernels in tools/bench_bg.py, per variant library ("base" = the built one), on one box
root=${GRAFT_REPO_ROOT:-$(pwd)}
cp $root/volsurfs_amd/libvolsurfs_hip.so /tmp/base.so
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" != base ]; then cp $root/variants/lib_$v.so $root/volsurfs_amd/libvolsurfs_hip.so; else cp /tmp/base.so $root/volsurfs_amd/libvolsurfs_hip.so; fi
  rm -rf /tmp/abp_$v
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abp_$v -- python3 $root/tools/bench_bg.py --steps 5 > /tmp/abp_$v.log 2>&1
  python3 - <<PY
import csv,glob,os,json
f=max(glob.glob("/tmp/abp_$v/*/*kernel_stats.csv"), key=os.path.getmtime)
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
out=[]
for key in ("mlp_fwd","mlp_dgrad","mlp_wgrad","mlp_reduce","grid_bin_count","grid_bin_scatter","grid_bin_accumulate","grid_encode_fwd"):
    rr=[r for r in rows if key in r['Name']]
    out.append(f"{key.replace('mlp_','').replace('grid_','')} {sum(float(r['TotalDurationNs']) for r in rr)/max(1,sum(int(r['Calls']) for r in rr))/1e3:7.1f}us")
step=[l for l in open("/tmp/abp_$v.log") if l.startswith("{")]
ms=json.loads(step[-1])["ms_per_step"] if step else float("nan")
print(f"$v".ljust(8), " ".join(out), f"| gpu total/step {tot/10/1e6:6.2f} ms  (profiled step {ms:.1f} ms)")
PY
done
cp /tmp/base.so $root/volsurfs_amd/libvolsurfs_hip.so
