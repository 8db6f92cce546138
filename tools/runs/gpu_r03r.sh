#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03r; mkdir -p $OUT
timeout 120 python tools/phase_front.py 666 > $OUT/phase_front_666.txt 2>&1
timeout 120 python tools/phase_front.py 1280 > $OUT/phase_front_1280.txt 2>&1
