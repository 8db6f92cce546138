#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03v; mkdir -p $OUT
timeout 300 python tools/time_solves.py control07 0 >> $OUT/control07.jsonl 2>> $OUT/err.txt
timeout 300 python tools/time_solves.py maxcut4000 1024 >> $OUT/maxcut4000.jsonl 2>> $OUT/err.txt
timeout 300 python tools/time_solves.py nb 0 >> $OUT/nb.jsonl 2>> $OUT/err.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "front or panel or factor or iteration or maxcut or small" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
