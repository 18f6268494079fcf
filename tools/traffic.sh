#!/bin/bash
# PMC HBM traffic per kernel (separate passes for FETCH_SIZE and WRITE_SIZE, as
# /opt/skills/guides/MI355X_MICROARCH.md "HBM" prescribes).  Run via gpurun.
# usage: tools/traffic.sh [tag [bench args...]]   (tag names gpurun_out/traffic_<tag>_*; default "frame")
tag=${1:-frame}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --pmc $c --output-format csv -d $root/gpurun_out/traffic_${tag}_$c -- python3 $root/bench.py --no-cpu-baseline --no-noisy --steps 3 --warmup 1 "$@" > $root/gpurun_out/traffic_${tag}_$c.log 2>&1
  echo "$c rc=$?"
done
python3 - <<PY
import csv,glob,collections,json
out=collections.defaultdict(dict)
for c in ("FETCH_SIZE","WRITE_SIZE"):
    acc=collections.defaultdict(list)
    for f in glob.glob("$root/gpurun_out/traffic_${tag}_%s/*/*counter_collection.csv"%c):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name']==c: acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items():
        v=sorted(v); out[k][c]=v[len(v)//2]     # median per launch, KiB
out["_bench_args"]="$*"
json.dump(out, open("$root/gpurun_out/traffic_raw_${tag}.json","w"), indent=1)
for k,v in sorted(((k,v) for k,v in out.items() if isinstance(v,dict)), key=lambda kv:-sum(kv[1].values()))[:14]:
    print(f"{k[:60]:60s} fetch={v.get('FETCH_SIZE',0)/1024:9.1f} MiB write={v.get('WRITE_SIZE',0)/1024:9.1f} MiB")
PY
