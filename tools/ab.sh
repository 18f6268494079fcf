#!/bin/bash
# usage: tools/ab.sh ENVVAR v1 v2 ... : bench stage times per value of ENVVAR (eager, no CPU baseline)
var=$1; shift
for v in "$@"; do
  env $var=$v timeout 200 python bench.py --no-cpu-baseline --no-graph --steps 10 --warmup 3 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$var=$v', round(d['value'],1), d.get('stages_ms', d.get('stage_ms')))"
done
