#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03ah; mkdir -p $OUT
timeout 300 python tools/time_ada.py blockdiag >> $OUT/ada.jsonl 2>> $OUT/err.txt
