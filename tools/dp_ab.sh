#!/bin/bash
# One-GPU price of the data-parallel step: the plain graph, the phased hash-grid
# backward without collectives, and the same through a one-rank RCCL group, interleaved on one box.
# Output: gpurun_out/dp/*.json + gpurun_out/dp/summary.txt
mkdir -p gpurun_out/dp
B="python bench.py --no-noisy --no-cpu-baseline --steps 200"
run() { name=$1; shift; timeout 600 $B "$@" > gpurun_out/dp/$name.json 2> gpurun_out/dp/$name.err || echo "FAILED $name" >> gpurun_out/dp/summary.txt; }
for round in 1 2; do
  run plain_$round
  run phased3_nocomm_$round --by-shell
  run rccl_phased3_$round --force-dist
  run torchdist_phased3_$round --force-dist --dp-comm torch
  run rccl_phased3_spin_$round --force-dist --dp-wait 2
  run rccl_phased2_$round --force-dist --dp-phases 3,5
  run rccl_phased5_$round --force-dist --dp-phases 1,2,3,4,5
  run rccl_phased1_$round --force-dist --dp-phases 5
done
run rccl_r4_split --force-dist --dp-split-launches
python - <<'PY' >> gpurun_out/dp/summary.txt
import json, glob, os
for f in sorted(glob.glob("gpurun_out/dp/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{os.path.basename(f):32s} {d['value']:8.2f} Mrays/s  {d['ms_per_step']:.4f} ms  launch={d['config']['launch']}  "
              f"phases={d['config'].get('dp_phases')}  comm={'rccl' if 'rccl' in (d['config'].get('dp_comm') or '') else d['config'].get('dp_comm')}  enc_bwd={d['stages_ms'].get('nt_encode_bwd')}  backend={d.get('dist_backend')}")
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
cat gpurun_out/dp/summary.txt
