#!/bin/bash
# round 4, run a: the MEX tier on hardware (shim tests, bench with the mex_inclusive leg)
mkdir -p gpurun_out/r04a
python -c "import __graft_entry__ as g; g.build()" > gpurun_out/r04a/build.log 2>&1
timeout 900 python -m pytest tests/test_mexshims_gpu.py -x -q > gpurun_out/r04a/mexshims_gpu.txt 2>&1
tail -5 gpurun_out/r04a/mexshims_gpu.txt
timeout 900 python bench.py > gpurun_out/r04a/bench_default.json 2> gpurun_out/r04a/bench_default.err
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r04a/bench_default.json").read().strip().splitlines()[-1])
print("value", d["value"], "ms", d["ms_per_step"])
print("cpu", d["cpu_baseline"] and d["cpu_baseline"]["value"])
print("pcie", d["pcie_inclusive"])
print("mex", json.dumps(d["mex_inclusive"], indent=1))
for o in d["other_configs"]:
    print(o.get("workload"), o.get("ms_per_step"), o.get("phases_ms_per_step"), o.get("error"))
    if "mex_inclusive" in o: print(json.dumps(o["mex_inclusive"], indent=1))
PY
