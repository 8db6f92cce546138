#!/bin/bash
# round 4, run o: soaks against the compiled reference with the round's last build (bounded waits with the shared give-up rule,
# refinement launches): rank-deficient fronts on both factor paths, the general factor / solve soak, ADA'
mkdir -p gpurun_out/r04o
timeout 200 python tests/tools/soak_def.py 150 > gpurun_out/r04o/soak_def.txt 2>&1; tail -1 gpurun_out/r04o/soak_def.txt
timeout 130 python tests/tools/soak_def.py 90 330 1344 panel > gpurun_out/r04o/soak_def_panel.txt 2>&1; tail -1 gpurun_out/r04o/soak_def_panel.txt
timeout 200 python tests/tools/soak.py 150 > gpurun_out/r04o/soak.txt 2>&1; tail -1 gpurun_out/r04o/soak.txt
timeout 160 python tests/tools/soak_ada.py 120 > gpurun_out/r04o/soak_ada.txt 2>&1; tail -1 gpurun_out/r04o/soak_ada.txt
