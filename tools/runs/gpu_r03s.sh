#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03s; mkdir -p $OUT
timeout 200 python tools/phase_profile.py > $OUT/phase_profile.txt 2>&1
timeout 200 python tools/time_solves.py control07 0 > $OUT/control07.jsonl 2> $OUT/err.txt
timeout 200 python tools/time_solves.py maxcut4000 0 > $OUT/maxcut4000.jsonl 2>> $OUT/err.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "front or panel or factor or iteration or maxcut" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
