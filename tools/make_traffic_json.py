#!/usr/bin/env python3
"""gpurun_out/traffic_raw.json (tools/traffic.sh) -> profiles/traffic.json (bytes per
launch and bench stage: 2*FETCH_SIZE + WRITE_SIZE, KiB -> B; gfx950 reports half of the
streamed read bytes, MI355X_MICROARCH.md 'HBM')."""
import json, os, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
raw = json.load(open(os.path.join(root, "gpurun_out", "traffic_raw.json")))
stages = {"nt_mlp_bwd": ["nt_mlp_bwd"], "nt_mlp_fwd": ["nt_mlp_fwd_kernel"],
          "nt_encode_bwd": ["nt_encode_bwd_kernelILb0", "nt_encode_bwd_kernelILb1", "nt_encode_bwd_both_kernel"],
          "nt_encode_fwd": ["nt_encode_fwd_kernelILb0", "nt_encode_fwd_kernelILb1", "nt_encode_fwd_both_kernel"],
          "nt_shade_bwd": ["nt_shade_bwd_kernel"], "nt_shade_fwd": ["nt_shade_fwd_kernel"],
          "trace": ["trace_qf_kernel", "trace_q_kernel", "trace_ww_kernel"],
          "composite_fwd_bwd": ["composite_dense_fwd_kernel", "composite_dense_bwd_kernel"]}   # the step's fused launch is the bwd kernel
out = {}
for st, subs in stages.items():
    b = 0
    for s in subs:
        for k, v in raw.items():
            if s in k:
                b += (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024
    out[st] = int(b)
sys.path.insert(0, root)
import bench  # noqa: E402  (kernel_source_hash: bench.py refuses the file once the kernels change)
out["_kernel_source_sha256"] = bench.kernel_source_hash()
out["_workload"] = bench.workload_key(bench.parse([]))      # tools/traffic.sh runs the default frame workload
json.dump(out, open(os.path.join(root, "profiles", "traffic.json"), "w"), indent=1)
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
shutil.copy(os.path.join(root, "gpurun_out", "traffic_raw.json"),
            os.path.join(root, "profiles", tag, "traffic_pmc_raw_KiB.json"))
print(json.dumps(out, indent=1))
