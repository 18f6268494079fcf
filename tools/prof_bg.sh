root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/bg -- python3 $root/tools/bench_bg.py --steps 5 > $root/gpurun_out/bg.log 2>&1
python3 - <<PY
import csv,glob
import os; f=max(glob.glob("$root/gpurun_out/bg/*/*kernel_stats.csv"), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:40]:
    print(f"{r['Name'][:70]:70s} n={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} {r['Percentage']}%")
PY
