#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03y; mkdir -p $OUT
timeout 120 python tools/time_ada.py control07 >> $OUT/ada3.jsonl 2>> $OUT/err.txt
timeout 120 python tools/time_ada.py blockdiag >> $OUT/ada3.jsonl 2>> $OUT/err.txt
timeout 120 python tools/time_ada.py maxcut4000 >> $OUT/ada3.jsonl 2>> $OUT/err.txt
