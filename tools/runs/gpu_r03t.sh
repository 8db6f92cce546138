#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03t; mkdir -p $OUT
for lib in "" libsedumi_hip_niA.so libsedumi_hip_niB.so; do
  SDM_LIB=$lib timeout 300 python tools/time_solves.py maxcut4000 1024 >> $OUT/maxcut4000.jsonl 2>> $OUT/err.txt
done
