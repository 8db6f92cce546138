#!/bin/bash
# round 4, run v: the last build -- bench line of the driver's command, the MEX-tier tests, a short soak of both factor paths, smoke
mkdir -p gpurun_out/r04v
timeout 600 python bench.py > gpurun_out/r04v/bench_default.json 2> gpurun_out/r04v/bench_default.err
timeout 300 python -m pytest tests/test_mexshims_gpu.py tests/test_gpu_parity.py -q -m gpu -k "mex or shim or iteration_units or golden or resident or one_launch_front_matches" 2>&1 | tail -2
timeout 100 python tests/tools/soak_def.py 60 2>&1 | tail -1
timeout 60 python tests/tools/soak_def.py 30 330 1344 panel 2>&1 | tail -1
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
python - <<'PY'
import json
d = json.load(open("gpurun_out/r04v/bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["cpu_baseline"]["value"], d["mex_inclusive"]["value"])
for o in d["other_configs"]:
    print(o.get("workload", "?")[:50], round(o.get("ms_per_step", 0), 4), o.get("mex_inclusive", {}).get("ms_per_step"), o.get("error"))
PY
