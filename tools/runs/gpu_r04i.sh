#!/bin/bash
# round 4, run i: the one-launch factor kernel (control07's shape: 56 + 66 workgroups with the inverse behind it) next to ANOTHER PROCESS
# that holds N compute units for 2 s (tests/gpuhog): how long sdm_plan_blkchol_wait takes and which path the plan is on afterwards
mkdir -p gpurun_out/r04i
for busy in 190 200 204 208 212 216 220 224 228 232 240 250; do python tools/starve_probe.py $busy 2000 >> gpurun_out/r04i/starve_probe.jsonl; done
timeout 300 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "starved_by_another_process" 2>&1 | tail -4
