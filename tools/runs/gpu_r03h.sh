#!/bin/bash
# chain workgroup of k_ldl_front: bit-identity with the panel launches, pivot rule, determinism; exit code of the shim tests; timing
OUT=gpurun_out/r03h; mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "one_launch or golden or control07 or deterministic" > $OUT/tests.txt 2>&1; echo "rc=$?"
tail -6 $OUT/tests.txt
timeout 300 python -m pytest tests/test_mexshims_gpu.py -m gpu -q > $OUT/t1.txt 2>&1; echo "mexshims rc=$?"; tail -2 $OUT/t1.txt
timeout 400 python bench.py --steps 50 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
python - <<'PY'
import json
b=json.load(open("gpurun_out/r03h/bench.json"))
print(b["value"], b["ms_per_step"], b["roofline"]["kernel"], b["roofline"]["avg_launch_us"], b["phases_ms_per_step"]["ada_ms"], b["phases_ms_per_step"]["factor_ms"], b["phases_ms_per_step"]["solves_ms"])
print({k: (v.get("ms_per_step"), v.get("phases_ms_per_step", {}).get("factor_ms")) for k, v in b.get("other_configs", {}).items()})
PY
