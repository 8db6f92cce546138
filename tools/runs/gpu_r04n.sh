#!/bin/bash
# round 4, run n: the MEX tier with the host's memory policy (freed blocks kept: no mmap / munmap per returned array)
mkdir -p gpurun_out/r04n
python bench.py --steps 100 --warmup 10 --no-other-configs > gpurun_out/r04n/bench.json 2>gpurun_out/r04n/bench.err
MEXHOST_DEFAULT_MALLOC=1 python bench.py --steps 100 --warmup 10 --no-other-configs > gpurun_out/r04n/bench_default_malloc.json 2>gpurun_out/r04n/bench2.err
python - <<'PY'
import json
for f in ("bench", "bench_default_malloc"):
    d = json.load(open("gpurun_out/r04n/%s.json" % f))
    print(f, "value", round(d["value"], 1), "cpu", round(d["cpu_baseline"]["value"], 2), d["cpu_baseline"]["stage_ms_per_unit"], "mex", round(d["mex_inclusive"]["value"], 1), d["mex_inclusive"]["stage_ms_per_unit"], "first", d["mex_inclusive"]["first_unit_ms"], "pcie", round(d["pcie_inclusive"]["value"], 1))
    print("   traffic", d["roofline"].get("traffic_from_committed_profile"))
PY
