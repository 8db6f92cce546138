#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03ao; mkdir -p $OUT
for wl in nb arch0 blockdiag maxcut4000; do
  timeout 300 python tools/time_solves.py $wl 0 >> $OUT/small.jsonl 2>> $OUT/err.txt
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_mexshims_gpu.py -q -x -k "panel or factor or iteration or small or many or golden or rank or skip or probe or shims or unit" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
