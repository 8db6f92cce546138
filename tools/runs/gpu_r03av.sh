#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03av; mkdir -p $OUT
for wl in control07 maxcut4000; do
  for lib in "" libsedumi_hip_base.so; do
    SDM_LIB=$lib timeout 150 python tools/time_solves.py $wl >> $OUT/$wl.jsonl 2>> $OUT/err.txt
  done
done
SDM_LIB="" timeout 100 python tools/time_solves.py control07 >> $OUT/control07.jsonl 2>> $OUT/err.txt
SDM_LIB=libsedumi_hip_base.so timeout 100 python tools/time_solves.py control07 >> $OUT/control07.jsonl 2>> $OUT/err.txt
python - <<'PY'
import json
for wl in ("control07","maxcut4000"):
    for l in open(f"gpurun_out/r03av/{wl}.jsonl"):
        j=json.loads(l); print(wl, j["lib"] or "new", j["factor_incl_inversion_ms"], {k:v for k,v in j["kernel_us_with_events"].items() if "ldl" in k}, j["relres"])
PY
