#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03af; mkdir -p $OUT
timeout 300 python tools/time_ada.py maxcut4000 >> $OUT/ada.jsonl 2>> $OUT/err.txt
timeout 300 python tools/time_ada.py maxcut2000 >> $OUT/ada.jsonl 2>> $OUT/err.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "maxcut or ada or iteration" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
