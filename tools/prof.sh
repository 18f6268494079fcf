#!/bin/bash
# usage (on the GPU box, via gpurun): tools/prof.sh <tag> [bench args...]
# rocprofv3 kernel trace of bench.py; prints the top kernels; CSVs land in gpurun_out/<tag>/
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $root/gpurun_out/$tag -- python3 $root/bench.py --no-cpu-baseline "$@" > $root/gpurun_out/$tag.log 2>&1
echo "rocprof rc=$?"
python3 - <<PY
import csv,glob
f=glob.glob("$root/gpurun_out/$tag/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:28]:
    print(f"{r['Name'][:64]:64s} n={r['Calls']:>4s} avg_us={float(r['AverageNs'])/1e3:9.1f} {r['Percentage']}%")
PY
