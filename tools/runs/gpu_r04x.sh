#!/bin/bash
# round 4, run x: small fronts (arch0, nb) with the blocked matrix-core row solve / the one-launch path instead of the faithful
# row substitution (MFMA_MIN_ROWS 256 -> 96 / 48): parity against the reference, then the factor times
mkdir -p gpurun_out/r04x
for v in "" mfma96 mfma48; do
  if [ -z "$v" ]; then lib=""; else lib="libsedumi_hip_$v.so"; fi
  echo "== variant '$v'"
  SDM_LIB=$lib timeout 300 python tools/variant_check.py 2>&1 | tail -8
  for wl in arch0 nb; do SDM_LIB=$lib python tools/time_solves.py $wl 0 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['workload'], 'factor incl. inversion ms', d['factor_incl_inversion_ms'], 'solve us', d['us_per_solve'], 'relres', d['relres'], d['kernel_us_with_events'])"; done
done 2>&1 | tee gpurun_out/r04x/small_fronts.txt
