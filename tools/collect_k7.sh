#!/bin/bash
# BASELINE configs[4]'s per-GPU frame (1920x1080, K = 7, subdiv-8 shells): bench line, rocprofv3 kernel stats, PMC traffic
A="--res 1080 --width 1920 --shells 7 --subdiv 8 --no-noisy"
mkdir -p gpurun_out/k7
timeout 900 python bench.py $A --no-cpu-baseline --steps 50 > gpurun_out/k7/bench.json 2> gpurun_out/k7/bench.err
tools/prof.sh k7_prof $A --steps 10 --warmup 2
tools/traffic.sh k7 $A
