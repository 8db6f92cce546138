#!/bin/bash
# One parametrised command list for the measurement runs of a round on the GPU box (gpurun -- 'bash tools/gpu_run.sh <tag> <what> ...').
# Everything goes to gpurun_out/<tag>/; what is kept is copied to profiles/ by hand, with the tag as prefix.
#   tests <pytest args>     pytest -m gpu on the arguments (default: the whole suite)
#   bench [args]            python bench.py [args] -> bench.json
#   timeline <workload> <first kernel of a unit>     rocprofv3 kernel trace of tools/trace_units.py, as a per-launch timeline
#   prof <workload> [steps] tools/profile_round.sh (kernel stats + PMC passes incl. the MFMA counters)
#   mex <workload> [units]  tools/mex_counters.py (stage times and cache counters of the MEX tier)
#   run <command>           any command line (its output to run_<name>.txt)
TAG=$1; shift
OUT=gpurun_out/$TAG; mkdir -p $OUT
export TMPDIR=/tmp
while [ $# -gt 0 ]; do
  what=$1; shift
  case $what in
    tests) args=${1:-tests}; shift; f=$OUT/tests_$(echo $args | tr -c 'a-zA-Z0-9\n' _ | cut -c1-40).txt; eval "timeout 1500 python -m pytest $args -m gpu -q" > $f 2>&1; tail -n 4 $f ;;
    bench) args=${1:-}; shift; timeout 900 python bench.py $args > $OUT/bench$(echo "$args" | tr -c 'a-zA-Z0-9\n' _ | cut -c1-40).json 2> $OUT/bench.err; tail -c 400 $OUT/bench*.json ;;
    timeline) wl=$1; first=$2; shift 2
      (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$OUT/kt_$wl -o kt -- python $GRAFT_REPO_ROOT/tools/trace_units.py $wl > /dev/null 2> $GRAFT_REPO_ROOT/$OUT/err_$wl.txt)
      python tools/unit_timeline.py $OUT/kt_$wl/kt_kernel_trace.csv $first > $OUT/timeline_$wl.txt 2>&1; rm -rf $OUT/kt_$wl; head -80 $OUT/timeline_$wl.txt | cut -c1-120 ;;
    prof) wl=$1; steps=${2:-50}; shift 2; timeout 900 bash tools/profile_round.sh ${TAG}_$wl $wl $steps > $OUT/prof_$wl.log 2>&1; tail -3 $OUT/prof_$wl.log; cp gpurun_out/prof_${TAG}_$wl/keep/* $OUT/ 2>/dev/null ;;
    mex) wl=$1; units=${2:-6}; shift 2; timeout 600 python tools/mex_counters.py $wl $units > $OUT/mex_$wl.txt 2>&1; tail -12 $OUT/mex_$wl.txt ;;
    run) cmd=$1; shift; timeout 600 bash -c "$cmd" > $OUT/run_$(echo "$cmd" | tr -c 'a-zA-Z0-9\n' _ | cut -c1-40).txt 2>&1; tail -n 12 $OUT/run_*.txt ;;
    *) echo "unknown step $what"; exit 2 ;;
  esac
done
