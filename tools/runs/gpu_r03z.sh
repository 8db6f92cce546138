#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03z; mkdir -p $OUT
timeout 300 python tools/dpr1_fallback_rate.py 400 4000 6 > $OUT/dpr1_rate.jsonl 2> $OUT/err.txt
timeout 300 python tools/dpr1_fallback_rate.py 200 1500 3 >> $OUT/dpr1_rate.jsonl 2>> $OUT/err.txt
timeout 300 python tools/dpr1_fallback_rate.py 800 6000 12 >> $OUT/dpr1_rate.jsonl 2>> $OUT/err.txt
