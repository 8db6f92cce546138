#!/bin/bash
# round 4, run k: blocks beyond the growth bound -- substitution against inverse + iterative refinement (timing, accuracy, tests)
mkdir -p gpurun_out/r04k
for wl in control07 maxcut4000 arch0; do python tools/time_refine.py $wl >> gpurun_out/r04k/refine.jsonl 2>gpurun_out/r04k/err_$wl.txt; done
cat gpurun_out/r04k/refine.jsonl | cut -c1-1500
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -s -k "refined or solve_widths or resident or golden" > gpurun_out/r04k/tests.txt 2>&1; tail -12 gpurun_out/r04k/tests.txt
