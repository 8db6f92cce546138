#!/bin/bash
# solves: loads in flight / row order / split rows (measurement variants built by python -m sedumi_amd.build --variant)
cd /root/repo
OUT=gpurun_out/r03p; mkdir -p $OUT
for lib in "" libsedumi_hip_np16.so libsedumi_hip_rev.so libsedumi_hip_np16rev.so libsedumi_hip_split1024.so libsedumi_hip_split256.so libsedumi_hip_split512np4.so; do
  SDM_LIB=$lib timeout 300 python tools/time_solves.py maxcut4000 1024 2048 >> $OUT/maxcut4000.jsonl 2>> $OUT/err.txt
  SDM_LIB=$lib timeout 120 python tools/time_solves.py control07 0 >> $OUT/control07.jsonl 2>> $OUT/err.txt
done
