#!/bin/bash
# round 4, run c: where the chain of k_ldl_front goes with the optimistic diagonal block; solves with non-temporal loads; rest of the GPU suite
mkdir -p gpurun_out/r04c
python tools/trace_front.py 666 > gpurun_out/r04c/trace_front_666.txt 2>&1
python tools/phase_front.py 666 > gpurun_out/r04c/phase_front_666.txt 2>&1
cat gpurun_out/r04c/trace_front_666.txt gpurun_out/r04c/phase_front_666.txt
python tools/time_solves.py maxcut4000 0 1024 > gpurun_out/r04c/solves_maxcut4000_base.jsonl 2>&1
SDM_LIB=libsedumi_hip_nt.so python tools/time_solves.py maxcut4000 0 1024 > gpurun_out/r04c/solves_maxcut4000_nt.jsonl 2>&1
python tools/time_solves.py control07 0 > gpurun_out/r04c/solves_control07_base.jsonl 2>&1
SDM_LIB=libsedumi_hip_nt.so python tools/time_solves.py control07 0 > gpurun_out/r04c/solves_control07_nt.jsonl 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04c/solves_*.jsonl")):
    for l in open(f):
        try:
            d = json.loads(l); print(f.split("/")[-1], d["width"], d["us_per_solve"], d["launches_per_solve"], d["frac_of_hbm_peak"], d["factor_incl_inversion_ms"], d["kernel_us_with_events"])
        except Exception: print(f, l[:200])
PY
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_mexshims_gpu.py -q -m gpu > gpurun_out/r04c/gpu_suite_rest.txt 2>&1
tail -6 gpurun_out/r04c/gpu_suite_rest.txt
