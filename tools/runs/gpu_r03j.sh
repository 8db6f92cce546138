#!/bin/bash
# data-tagged hand-over of the diagonal blocks in k_ldl_front
OUT=gpurun_out/r03j; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_launch or golden or control07 or deterministic" > $OUT/tests.txt 2>&1; echo "rc=$?"
tail -4 $OUT/tests.txt
timeout 400 python bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
b=json.load(open("gpurun_out/r03j/bench.json"))
print(b["value"], b["ms_per_step"], b["roofline"]["kernel"], b["roofline"]["avg_launch_us"], b["phases_ms_per_step"]["ada_ms"], b["phases_ms_per_step"]["factor_ms"], b["phases_ms_per_step"]["solves_ms"])
for o in b.get("other_configs", []):
    print(o.get("workload", o.get("config"))[:40] if isinstance(o, dict) else o)
PY
