#!/bin/bash
# focused runs after a change of k_ldl_front: the rank-deficient soak, the factor tests of the GPU suite, the bench line,
# rocprofv3 kernel stats + PMC passes of the bench command (TAG = name of the round's run)
cd /root/repo
TAG=${TAG:-r03ax}
OUT=gpurun_out/$TAG; mkdir -p $OUT
timeout 100 python tests/tools/soak_def.py 30 > $OUT/soak_def.txt 2>&1
tail -2 $OUT/soak_def.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "one_launch_front or maxcut or sweeps or many_small or golden or iteration" > $OUT/factor_tests.txt 2>&1
tail -2 $OUT/factor_tests.txt
timeout 300 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 400 bash tools/profile_round.sh ${TAG}_control07 control07 100 > $OUT/prof_control07.log 2>&1
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/smoke.txt 2>&1
timeout 200 bash tools/profile_round.sh ${TAG}_maxcut4000 maxcut4000 20 nopmc > $OUT/prof_maxcut4000.log 2>&1
tail -1 $OUT/smoke.txt
head -c 600 $OUT/bench_default.json
