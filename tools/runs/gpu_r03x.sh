#!/bin/bash
cd /root/repo
OUT=gpurun_out/r03x; mkdir -p $OUT
timeout 200 python tools/phase_profile.py control07 > $OUT/phase_profile_control07.txt 2>&1
timeout 600 python bench.py --steps 100 --warmup 10 > $OUT/bench.json 2> $OUT/bench.err
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "ada or iteration or golden or getada or hermitian" > $OUT/tests.txt 2>&1
tail -3 $OUT/tests.txt
