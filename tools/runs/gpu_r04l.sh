#!/bin/bash
# round 4, run l: the refinement launches switched by sweep numbers (async bench loop must stay on the refined path)
mkdir -p gpurun_out/r04l
timeout 600 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "follow_the_conditioning or refined or solve_widths or resident" 2>&1 | tail -3
python - <<'PY'
import json, bench
for refine in (1, 0):
    o = bench.measure_config("control07", 0, 100, 5, 20, 0, growth_max=0.0, refine=refine)
    print(refine, round(o["ms_per_step"], 4), o["phases_ms_per_step"], o["solve"]["us_per_solve"], o["solve"]["launches_per_solve"])
o = bench.measure_config("control07", 0, 100, 5, 20, 0)
print("default", round(o["ms_per_step"], 4), o["phases_ms_per_step"], o["solve"]["us_per_solve"], o["solve"]["launches_per_solve"])
PY
