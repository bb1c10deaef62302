#!/bin/bash
# Issue / occupancy counters of the secondary kernels (back_end, LTV-MPC) -- separate rocprofv3 --pmc passes.
set -u
TAG=${1:-r02_e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 --kernel-trace --stats -d $OUT/be_trace -o be -- python3 tools/be_profile.py > $OUT/be_run.txt 2> $OUT/be_trace.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVES -d $OUT/be_pmc -o bepmc -- python3 tools/be_profile.py > /dev/null 2> $OUT/be_pmc.err
rocprofv3 --kernel-trace --stats -d $OUT/ltv_trace -o ltv -- python3 tools/ltv_profile.py > $OUT/ltv_run.txt 2> $OUT/ltv_trace.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVES -d $OUT/ltv_pmc -o ltvpmc -- python3 tools/ltv_profile.py > /dev/null 2> $OUT/ltv_pmc.err
cat $OUT/be_run.txt $OUT/ltv_run.txt | tail -6
