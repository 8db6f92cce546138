#!/bin/bash
# round 4, run g: the one-launch solve (k_solve_chain) against the launch-per-stage form
mkdir -p gpurun_out/r04g
for wl in maxcut4000 control07 arch0; do
  python tools/time_solves.py $wl 0 > gpurun_out/r04g/solves_${wl}_chain.jsonl 2>&1
  SDM_CHAIN=0 python tools/time_solves.py $wl 0 > gpurun_out/r04g/solves_${wl}_launches.jsonl 2>&1
done
python tools/time_solves.py maxcut4000 1024 > gpurun_out/r04g/solves_maxcut4000_w1024_chain.jsonl 2>&1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04g/solves_*.jsonl")):
    for l in open(f):
        try:
            d = json.loads(l); print(f.split("/")[-1], d["width"], d["us_per_solve"], d["launches_per_solve"], d["frac_of_hbm_peak"], "relres", d["relres"], {k: v for k, v in d["kernel_us_with_events"].items() if k.startswith("k_s")})
        except Exception: print(f, l[:300])
PY
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "solve or golden or iteration or maxcut or width or resident" > gpurun_out/r04g/gpu_solve_tests.txt 2>&1
tail -4 gpurun_out/r04g/gpu_solve_tests.txt
