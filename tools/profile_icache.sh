#!/bin/bash
# instruction-cache counters of nmpc::rti_kernel
set -u
TAG=${1:-rXX}; B=${2:-4096}; L=${3:-0}
OUT=gpurun_out/$TAG/ic_B${B}_L$L
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
ARGS="bench.py --batch $B --lanes $L --no-cpu-baseline --no-extras --no-graph --steps 20 --warmup 5"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $OUT/p -o p -- python3 $ARGS > /dev/null 2> $OUT/p.err
tail -n 2 $OUT/p.err
