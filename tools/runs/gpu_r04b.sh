#!/bin/bash
# round 4, run b: optimistic diagonal block + device dpr1fact + MEX tier on hardware: full GPU suite, soaks of rank-deficient fronts, bench
mkdir -p gpurun_out/r04b
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r04b/build.log 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r04b/gpu_suite.txt 2>&1
tail -5 gpurun_out/r04b/gpu_suite.txt
timeout 200 python tests/tools/soak_def.py 90 5 > gpurun_out/r04b/soak_def.txt 2>&1; tail -2 gpurun_out/r04b/soak_def.txt
timeout 200 python tests/tools/soak.py 60 11 > gpurun_out/r04b/soak.txt 2>&1; tail -2 gpurun_out/r04b/soak.txt
timeout 900 python bench.py > gpurun_out/r04b/bench_default.json 2> gpurun_out/r04b/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04b/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "phases", d["phases_ms_per_step"]["ada_ms"], d["phases_ms_per_step"]["factor_ms"], d["phases_ms_per_step"]["solves_ms"])
print("roof", d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
print("mex", d["mex_inclusive"]["value"], d["mex_inclusive"]["stage_ms_per_unit"])
for o in d["other_configs"]:
    print(o.get("workload"), o.get("ms_per_step"), o.get("phases_ms_per_step"), o.get("error"))
PY
