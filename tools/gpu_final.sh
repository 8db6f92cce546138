#!/bin/bash
# final runs of the round: full GPU suite, bench line, rocprofv3 kernel stats + PMC passes of four workloads
cd /root/repo
OUT=gpurun_out/${TAG:-r04m}; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gpu_suite.txt 2>&1
tail -3 $OUT/gpu_suite.txt
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 python bench.py --steps 100 --warmup 10 > $OUT/bench_100.json 2> $OUT/bench_100.err
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1
for wl in control07 maxcut4000 blockdiag nb; do
  steps=100; [ $wl = maxcut4000 ] && steps=20; [ $wl = blockdiag ] && steps=40
  timeout 900 bash tools/profile_round.sh ${TAG:-r04m}_$wl $wl $steps > $OUT/prof_$wl.log 2>&1
done
ls gpurun_out/prof_${TAG:-r04m}_*/keep 2>/dev/null
