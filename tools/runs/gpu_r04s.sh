#!/bin/bash
# round 4, run s: rank-deficient soak after the partial maxima of k_begin_factor became write-through / L2-bypassing
mkdir -p gpurun_out/r04s
timeout 400 python tests/tools/soak_def.py 300 > gpurun_out/r04s/soak_def.txt 2>&1; tail -3 gpurun_out/r04s/soak_def.txt
