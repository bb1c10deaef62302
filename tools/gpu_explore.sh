#!/bin/bash
# One parametrised exploration pass on the GPU box (replaces the one-off tools/explore_r05_*.sh / tools/gpu/g*.sh of round 5):
#   tools/gpu_explore.sh <tag> <what> [args ...]          (run through gpurun; output under gpurun_out/<tag>/)
# what:
#   run   <cmd ...>                      the command, stdout + stderr to gpurun_out/<tag>/run.txt
#   env   "<VAR=.. VAR=..>" <cmd ...>    the command once per ';'-separated environment set (A/B of library switches)
#   stats <cmd ...>                      rocprofv3 --kernel-trace --stats, kernel summary to kernel_stats.txt
#   pmc   "<COUNTER ...>" <kernel-substring> <cmd ...>   one --pmc pass, per-kernel means to counters.txt (appends)
# <cmd> is a python script with its arguments (rocprofv3 needs the program itself after --).
set -u
TAG=$1; WHAT=$2; shift 2
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $R
case $WHAT in
run)
    python3 "$@" > $OUT/run.txt 2>&1; tail -60 $OUT/run.txt ;;
env)
    SETS=$1; shift
    IFS=';' read -ra ARR <<< "$SETS"
    : > $OUT/env.txt
    for E in "${ARR[@]}"; do
        echo "######## $E" >> $OUT/env.txt
        ( for kv in $E; do export "$kv"; done; python3 "$@" >> $OUT/env.txt 2>&1 )
    done
    cat $OUT/env.txt | grep -v amdgpu.ids | tail -120 ;;
stats)
    rocprofv3 --kernel-trace --stats -d $OUT/kt -o kt -- python3 "$@" > $OUT/stats_run.txt 2>&1
    python3 tools/summarize_prof.py $OUT/kt > $OUT/kernel_stats.txt 2>&1 || true
    head -40 $OUT/kernel_stats.txt; rm -rf $OUT/kt ;;
pmc)
    CTRS=$1; KERN=$2; shift 2
    N=$(ls -d $OUT/pmc* 2>/dev/null | wc -l)
    rocprofv3 --pmc $CTRS -d $OUT/pmc$N -o p -- python3 "$@" > $OUT/pmc_run$N.txt 2>&1
    python3 tools/summarize_counters.py $OUT/pmc$N "$KERN" $OUT/counters$N.txt
    rm -rf $OUT/pmc$N ;;
*) echo "unknown: $WHAT"; exit 2 ;;
esac
