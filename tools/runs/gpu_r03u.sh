#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03u; mkdir -p $OUT
for lib in "" libsedumi_hip_niJ.so; do
  SDM_LIB=$lib timeout 300 python tools/time_solves.py maxcut4000 1024 >> $OUT/maxcut4000.jsonl 2>> $OUT/err.txt
  SDM_LIB=$lib timeout 300 python tools/time_solves.py nb 0 >> $OUT/nb.jsonl 2>> $OUT/err.txt
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "front or panel or factor or iteration or maxcut or small" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
