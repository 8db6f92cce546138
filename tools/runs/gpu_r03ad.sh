#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03ad; mkdir -p $OUT
for wl in arch0 nb blockdiag; do
  timeout 300 python tools/time_solves.py $wl 0 >> $OUT/small.jsonl 2>> $OUT/err.txt
done
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "panel or factor or iteration or small or many or blockdiag or golden" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
