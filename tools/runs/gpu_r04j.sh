#!/bin/bash
# round 4, run j: the bench step eager against replayed from a captured hipGraph
mkdir -p gpurun_out/r04j
python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-other-configs > gpurun_out/r04j/bench_eager.json 2>gpurun_out/r04j/eager.err
python bench.py --steps 300 --warmup 20 --no-cpu-baseline --no-other-configs --graph > gpurun_out/r04j/bench_graph.json 2>gpurun_out/r04j/graph.err
python - <<'PY'
import json
for f in ("eager", "graph"):
    try:
        d = json.load(open("gpurun_out/r04j/bench_%s.json" % f)); print(f, d["value"], d["ms_per_step"], d["phases_ms_per_step"]["ada_ms"], d["phases_ms_per_step"]["factor_ms"], d["phases_ms_per_step"]["solves_ms"])
    except Exception as e: print(f, "failed", e)
PY
tail -2 gpurun_out/r04j/graph.err
