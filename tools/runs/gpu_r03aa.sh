#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03aa; mkdir -p $OUT
timeout 300 python tools/time_solves.py maxcut4000 0 1024 >> $OUT/maxcut4000.jsonl 2>> $OUT/err.txt
timeout 300 python tools/time_solves.py maxcut2000 0 >> $OUT/maxcut4000.jsonl 2>> $OUT/err.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "solve or width or maxcut or iteration or panel or factor" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
