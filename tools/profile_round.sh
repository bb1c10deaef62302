#!/bin/bash
# One GPU-box pass that produces everything profiles/ is refreshed from (run through gpurun):
#   tools/profile_round.sh <tag>      e.g. r01_c
# Outputs land in gpurun_out/<tag>/ ; tools/summarize_prof.py + tools/summarize_pmc.py condense them.
set -u
TAG=${1:-rXX}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench.err
# kernel trace + stats of the same command line, extras off so that the rti kernel at B=4096 dominates
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
# HBM counters: separate passes, eager launches (one dispatch record per launch)
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- python3 bench.py --no-cpu-baseline --no-extras --no-graph --steps 20 --warmup 5 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- python3 bench.py --no-cpu-baseline --no-extras --no-graph --steps 20 --warmup 5 > /dev/null 2> $OUT/pmc_write.err
# phase stamps (diagnostic build path of the same kernel)
ALORE_NMPC_STAMPS=1 python bench.py --no-cpu-baseline --no-extras --no-graph --steps 20 --warmup 2 2>&1 | grep -A12 "alore_nmpc stamps" > $OUT/stamps_B4096.txt
ALORE_NMPC_STAMPS=1 python bench.py --batch 512 --no-cpu-baseline --no-extras --no-graph --steps 20 --warmup 2 2>&1 | grep -A12 "alore_nmpc stamps" > $OUT/stamps_B512.txt
find $OUT -name "*.csv" | head -20
tail -1 $OUT/bench.json | cut -c1-600
