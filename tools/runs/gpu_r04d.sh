#!/bin/bash
# round 4, run d: the optimistic diagonal block with LDS-only barriers: chain trace, factor tests, bench
mkdir -p gpurun_out/r04d
python tools/trace_front.py 666 > gpurun_out/r04d/trace_front_666.txt 2>&1
head -14 gpurun_out/r04d/trace_front_666.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "front or maxcut or pivot or factor or dense_front or golden or rank" > gpurun_out/r04d/gpu_factor_tests.txt 2>&1
tail -4 gpurun_out/r04d/gpu_factor_tests.txt
timeout 100 python tests/tools/soak_def.py 40 5 > gpurun_out/r04d/soak_def.txt 2>&1; tail -2 gpurun_out/r04d/soak_def.txt
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r04d/bench_default.json 2> gpurun_out/r04d/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04d/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"], "phases", d["phases_ms_per_step"]["ada_ms"], d["phases_ms_per_step"]["factor_ms"], d["phases_ms_per_step"]["solves_ms"])
print("roof", d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"])
for o in d["other_configs"]:
    print(o.get("workload"), o.get("ms_per_step"), o.get("phases_ms_per_step"), o.get("error"))
PY
