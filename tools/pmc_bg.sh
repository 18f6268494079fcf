#!/bin/bash
# usage: tools/pmc_bg.sh <tag> <kernel-regex> "<counters>"   (run via gpurun): PMC counters of tools/bench_bg.py
tag=$1; regex=$2; ctrs=$3
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --pmc $ctrs --kernel-include-regex "$regex" --output-format csv -d $root/gpurun_out/$tag -- python3 $root/tools/bench_bg.py --steps 2 > $root/gpurun_out/$tag.log 2>&1
echo "rocprof rc=$?"
python3 - <<PY
import csv,glob,collections
fs=glob.glob("$root/gpurun_out/$tag/*/*counter_collection.csv")
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in fs:
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:50]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    print(k)
    for c,vals in v.items():
        print(f"   {c:28s} mean={sum(vals)/len(vals):16.1f} n={len(vals)}")
PY
