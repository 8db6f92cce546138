#!/bin/bash
# round 4, run z: general soaks (sparse patterns, bordered blocks, mixed-cone iteration units) of the last build (MFMA_MIN_ROWS = 48)
mkdir -p gpurun_out/r04z
timeout 160 python tests/tools/soak.py 120 > gpurun_out/r04z/soak.txt 2>&1; tail -2 gpurun_out/r04z/soak.txt
timeout 130 python tests/tools/soak_ada.py 90 > gpurun_out/r04z/soak_ada.txt 2>&1; tail -1 gpurun_out/r04z/soak_ada.txt
