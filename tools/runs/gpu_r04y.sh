#!/bin/bash
# round 4, run y: the build with MFMA_MIN_ROWS = 48 -- full GPU suite, soak of small rank-deficient fronts, bench line
mkdir -p gpurun_out/r04y
timeout 1500 python -m pytest tests -q -m gpu > gpurun_out/r04y/gpu_suite.txt 2>&1; tail -3 gpurun_out/r04y/gpu_suite.txt
timeout 100 python tests/tools/soak_def.py 60 100 340 > gpurun_out/r04y/soak_def_small.txt 2>&1; tail -1 gpurun_out/r04y/soak_def_small.txt
timeout 600 python bench.py > gpurun_out/r04y/bench_default.json 2> gpurun_out/r04y/bench_default.err
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04y/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["mex_inclusive"]["value"])
for o in d["other_configs"]:
    print(o.get("workload", "?")[:50], round(o.get("ms_per_step", 0), 4), {k: round(v, 4) for k, v in o.get("phases_ms_per_step", {}).items()}, (o.get("dominant_kernel") or {}).get("kernel"), o.get("error"))
PY
