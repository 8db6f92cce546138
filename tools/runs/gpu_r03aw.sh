#!/bin/bash
# k_ldl_panel with its diagonal-block role as a call (product build) or inlined (-DSDM_NI_DIAG=__forceinline__), after not_tail_called
cd /root/repo
OUT=gpurun_out/r03aw; mkdir -p $OUT
for i in 1 2; do
  for lib in "" libsedumi_hip_diaginl.so; do
    SDM_LIB=$lib timeout 150 python tools/time_solves.py maxcut4000 >> $OUT/maxcut4000.jsonl 2>> $OUT/err.txt
  done
done
python - <<'PY'
import json
for l in open("gpurun_out/r03aw/maxcut4000.jsonl"):
    j=json.loads(l); print(j["lib"] or "role as a call", j["factor_incl_inversion_ms"], {k:v for k,v in j["kernel_us_with_events"].items() if "ldl" in k}, j["relres"])
PY
