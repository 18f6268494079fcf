#!/usr/bin/env python3
"""gpurun_out/<tag>_pmc_*/**/*counter_collection.csv (tools/pmc.sh passes, one counter group each) ->
profiles/pmc_summary.json: per workload (bench.workload_key) and kernel the mean counters per launch and the derived
pipe occupancies bench.py quotes as roofline.pmc:
    mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128)    (GRBM_GUI_ACTIVE sums the 8 XCDs' cycles; 1 024 SIMDs)
usage: tools/make_pmc_json.py <round tag, e.g. r05> <pmc tag prefix, e.g. r05_pmc> [bench args of the collection...]"""
import collections, csv, glob, json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import bench  # noqa: E402
rnd, prefix, bargs = sys.argv[1], sys.argv[2], sys.argv[3:]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(os.path.join(root, "gpurun_out", prefix + "*", "*", "*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = {"nt_mlp_bwd": "nt_mlp_bwd", "nt_mlp_fwd": "nt_mlp_fwd_kernel", "nt_encode_bwd": "nt_encode_bwd",
         "nt_encode_fwd": "nt_encode_fwd", "nt_shade_fwd": "nt_shade_fwd", "nt_shade_bwd": "nt_shade_bwd",
         "trace": "trace_qf_kernel"}
out = {}
for stage, sub in names.items():
    ks = [k for k in acc if sub in k]
    if not ks:
        continue
    c = {n: sum(v) / len(v) for n, v in acc[ks[0]].items()}
    e = {"counters_per_launch": {n: round(v) for n, v in sorted(c.items())}}
    if "GRBM_GUI_ACTIVE" in c:
        simd_cycles = c["GRBM_GUI_ACTIVE"] * 128.0
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            e["mfma_busy"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles, 4)
        if "SQ_ACTIVE_INST_VALU" in c:        # quad-cycles: one per issued vector instruction
            e["valu_issue_busy"] = round(4.0 * c["SQ_ACTIVE_INST_VALU"] / simd_cycles, 4)
        if "SQ_WAVE_CYCLES" in c and "SQ_WAIT_INST_ANY" in c:
            e["wave_cycles_waiting_on_operands"] = round(c["SQ_WAIT_INST_ANY"] / c["SQ_WAVE_CYCLES"], 4)
    # cache behaviour (separate passes): vector L1 (TCP) and L2 (TCC) hit rates, mean L1->L2 read round trip
    if c.get("TCP_TOTAL_CACHE_ACCESSES_sum") and "TCP_TCC_READ_REQ_sum" in c:
        e["l1_hit_rate"] = round(1.0 - c["TCP_TCC_READ_REQ_sum"] / c["TCP_TOTAL_CACHE_ACCESSES_sum"], 4)
    if c.get("TCP_TCC_READ_REQ_sum") and "TCP_TCC_READ_REQ_LATENCY_sum" in c:
        e["l2_read_round_trip_cycles"] = round(c["TCP_TCC_READ_REQ_LATENCY_sum"] / c["TCP_TCC_READ_REQ_sum"], 1)
    if c.get("TCC_HIT_sum") is not None and c.get("TCC_MISS_sum") is not None and (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]) > 0:
        e["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
    out[stage] = e
out["_kernel_source_sha256"] = bench.kernel_source_hash()
path = os.path.join(root, "profiles", "pmc_summary.json")
cur = json.load(open(path)) if os.path.exists(path) else {"workloads": {}}
cur["workloads"][bench.workload_key(bench.parse(bargs))] = out
json.dump(cur, open(path, "w"), indent=1)
os.makedirs(os.path.join(root, "profiles", rnd, "pmc"), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", rnd, "pmc", prefix + "_summary.json"), "w"), indent=1)
print(json.dumps({k: {kk: vv for kk, vv in v.items() if kk != "counters_per_launch"} for k, v in out.items() if not k.startswith("_")}, indent=1))
