#!/bin/bash
# factor: width of the register sweep of the diagonal block (SW = 2 / 4 / 8 / 16)
cd /root/repo
OUT=gpurun_out/r03q; mkdir -p $OUT
for lib in "" libsedumi_hip_sw4.so libsedumi_hip_sw2.so libsedumi_hip_sw16.so; do
  SDM_LIB=$lib timeout 120 python tools/time_solves.py control07 0 >> $OUT/control07.jsonl 2>> $OUT/err.txt
  SDM_LIB=$lib timeout 300 python tools/time_solves.py maxcut4000 0 >> $OUT/maxcut4000.jsonl 2>> $OUT/err.txt
done
