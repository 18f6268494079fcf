#!/usr/bin/env python3
"""gpurun_out/traffic_raw_<tag>.json (tools/traffic.sh <tag> [bench args]) -> profiles/traffic.json: bytes per
launch and bench stage (2*FETCH_SIZE + WRITE_SIZE, KiB -> B; gfx950 reports half of the streamed read bytes,
MI355X_MICROARCH.md 'HBM'), ONE ENTRY PER WORKLOAD (keyed by bench.workload_key of the collection's
arguments) with the hash of the kernel sources it was collected at: bench.py reports `traffic` only when both
match.  usage: tools/make_traffic_json.py <round tag, e.g. r05> <traffic tag> [<traffic tag> ...]"""
import json, os, shlex, shutil, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402
stages = {"nt_mlp_bwd": ["nt_mlp_bwd"], "nt_mlp_fwd": ["nt_mlp_fwd_kernel"],
          "nt_encode_bwd": ["nt_encode_bwd_kernelILb0", "nt_encode_bwd_kernelILb1", "nt_encode_bwd_both_kernel",
                            "nt_encode_bwd_phased_kernel"],
          "nt_encode_fwd": ["nt_encode_fwd_kernelILb0", "nt_encode_fwd_kernelILb1", "nt_encode_fwd_both_kernel"],
          "nt_shade_bwd": ["nt_shade_bwd_kernel"], "nt_shade_fwd": ["nt_shade_fwd_kernel"],
          "trace": ["trace_qf_kernel", "trace_q_kernel", "trace_ww_kernel"],
          "composite_fwd_bwd": ["composite_dense_fwd_kernel", "composite_dense_bwd_kernel"]}   # the step's fused launch is the bwd kernel
rnd = sys.argv[1]
path = os.path.join(root, "profiles", "traffic.json")
cur = json.load(open(path)) if os.path.exists(path) else {}
if "workloads" not in cur:
    cur = {"workloads": {}}
for tag in sys.argv[2:]:
    raw = json.load(open(os.path.join(root, "gpurun_out", f"traffic_raw_{tag}.json")))
    args = bench.parse(shlex.split(raw.pop("_bench_args", "")))
    out = {}
    for st, subs in stages.items():
        b = 0
        for s in subs:
            for k, v in raw.items():
                if s in k:
                    b += (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024
        out[st] = int(b)
    out["_kernel_source_sha256"] = bench.kernel_source_hash()
    cur["workloads"][bench.workload_key(args)] = out
    os.makedirs(os.path.join(root, "profiles", rnd), exist_ok=True)
    shutil.copy(os.path.join(root, "gpurun_out", f"traffic_raw_{tag}.json"),
                os.path.join(root, "profiles", rnd, f"traffic_pmc_raw_KiB_{tag}.json"))
json.dump(cur, open(path, "w"), indent=1)
print(json.dumps(cur, indent=1))
