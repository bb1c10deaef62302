#!/bin/bash
# instruction-cache counters of the whole-body kernels (the stage kernel is 107 KB of straight-line code)
set -u
TAG=${1:-r02_k}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $OUT/wb_ic1 -o ic1 -- python3 tools/wb_profile.py > /dev/null 2> $OUT/ic1.err
rocprofv3 --pmc SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $OUT/wb_ic2 -o ic2 -- python3 tools/wb_profile.py > /dev/null 2> $OUT/ic2.err
python3 tools/summarize_counters.py $OUT/wb_ic1 $OUT/wb_ic2 wb:: $OUT/wb_icache_counters.txt
