#!/bin/bash
# round 4, run r: the follower's workgroups inside the k_ldl_front launch (no second stream, no fork / join)
mkdir -p gpurun_out/r04r
timeout 1500 python -m pytest tests -q -m gpu -x > gpurun_out/r04r/gpu_suite.txt 2>&1; tail -4 gpurun_out/r04r/gpu_suite.txt
timeout 200 python tests/tools/soak_def.py 120 > gpurun_out/r04r/soak_def.txt 2>&1; tail -1 gpurun_out/r04r/soak_def.txt
for busy in 200 208 216 224 232 240; do python tools/starve_probe.py $busy 2000 >> gpurun_out/r04r/starve_probe.jsonl; done; cat gpurun_out/r04r/starve_probe.jsonl
