#!/usr/bin/env python3
"""After `tools/collect_round.sh <tag>` came back from the GPU box: turn gpurun_out/<tag>_* into the tracked files —
profiles/<tag>/ (bench lines, kernel stats, smoke, tests, parity report), profiles/traffic.json and
profiles/pmc_summary.json (per workload, stamped with the kernel-source hash).  usage: tools/finish_round.py r05"""
import glob, json, os, shutil, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
R = os.path.join(root, "profiles", tag)
os.makedirs(R, exist_ok=True)
K7 = ["--res", "1080", "--width", "1920", "--shells", "7", "--subdiv", "8"]
subprocess.run([sys.executable, os.path.join(root, "tools", "make_traffic_json.py"), tag, "frame", "k7"], check=True,
               stdout=subprocess.DEVNULL)
subprocess.run([sys.executable, os.path.join(root, "tools", "make_pmc_json.py"), tag, f"{tag}_pmc_"], check=True, stdout=subprocess.DEVNULL)
subprocess.run([sys.executable, os.path.join(root, "tools", "make_pmc_json.py"), tag, f"{tag}_pmck7_", *K7], check=True,
               stdout=subprocess.DEVNULL)
names = {"bench.json": "bench_frame.json", "train.json": "bench_train.json", "trainp.json": "bench_train_permuto.json",
         "dtu.json": "bench_dtu.json", "render.json": "bench_render.json",
         "bench_1080p_K7_subdiv8.json": "bench_frame_1080p_K7_subdiv8.json",
         "bench_1080p_K7_subdiv8_cold.json": "bench_frame_1080p_K7_subdiv8_cold.json",
         "bench_rccl_one_rank.json": "bench_frame_rccl_one_rank.json", "smoke.log": "smoke.txt", "tests.log": "tests.log"}
for a, b in names.items():
    src = os.path.join(root, "gpurun_out", f"{tag}_{a}")
    if os.path.exists(src):
        shutil.copy(src, os.path.join(R, b))
pr = os.path.join(root, "gpurun_out", "parity_report.json")
if os.path.exists(pr):
    shutil.copy(pr, os.path.join(R, "parity_report.json"))
for t, name in (("prof_frame", "kernel_stats_frame.csv"), ("prof_k7", "kernel_stats_1080p_K7_subdiv8.csv"),
                ("prof_render", "kernel_stats_render.csv"), ("prof_train", "kernel_stats_train.csv"),
                ("prof_trainp", "kernel_stats_train_permuto.csv"), ("prof_dtu", "kernel_stats_dtu.csv")):
    fs = sorted(glob.glob(os.path.join(root, "gpurun_out", f"{tag}_{t}", "*", "*kernel_stats.csv")), key=os.path.getmtime)
    if fs:
        shutil.copy(fs[-1], os.path.join(R, name))
d = json.load(open(os.path.join(R, "bench_frame.json")))
print("frame %.1f Mrays/s %.4f ms | cold %.1f noisy %.1f stress %.1f / %.1f" % (d["value"], d["ms_per_step"], d["value_cold"],
      d["value_noisy"], d["value_stress"], d["value_stress_cold"]))
for f in ("bench_train", "bench_train_permuto", "bench_dtu", "bench_render", "bench_frame_1080p_K7_subdiv8",
          "bench_frame_1080p_K7_subdiv8_cold", "bench_frame_rccl_one_rank"):
    p = os.path.join(R, f + ".json")
    if os.path.exists(p):
        e = json.load(open(p))
        print("%-36s %9.2f %s  %.4f ms  %s" % (f, e["value"], e["unit"], e["ms_per_step"], (e.get("baked") or {}).get("Mrays/s", "")))
print(open(os.path.join(R, "tests.log")).read().strip().splitlines()[-1] if os.path.exists(os.path.join(R, "tests.log")) else "no tests.log")
