#!/bin/bash
# SQ issue counters of nmpc::rti_kernel (separate rocprofv3 --pmc passes, eager launches):
#   tools/profile_sq.sh <tag> [batch]
set -u
TAG=${1:-rXX}; B=${2:-4096}
OUT=gpurun_out/$TAG/sq_B$B
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 -L > $OUT/counters_available.txt 2>&1
ARGS="bench.py --batch $B --no-cpu-baseline --no-extras --no-graph --steps 20 --warmup 5"
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/p1 -o p1 -- python3 $ARGS > /dev/null 2> $OUT/p1.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/p2 -o p2 -- python3 $ARGS > /dev/null 2> $OUT/p2.err
ls -R $OUT | head -20; tail -3 $OUT/p1.err $OUT/p2.err
