#!/bin/bash
# LDS / wait-state counters of the whole-body kernels (separate rocprofv3 --pmc passes; no trace domains with --pmc).
set -u
TAG=${1:-r02_j}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*LDS[A-Z_0-9]*\|SQ_WAIT[A-Z_0-9]*\|SQ_INST_CYCLES[A-Z_0-9]*\|SQ_IFETCH[A-Z_0-9]*\|SQC_ICACHE[A-Z_0-9]*" | sort -u > $OUT/avail.txt
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/wb_lds1 -o l1 -- python3 tools/wb_profile.py > /dev/null 2> $OUT/l1.err
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES -d $OUT/wb_lds2 -o l2 -- python3 tools/wb_profile.py > /dev/null 2> $OUT/l2.err
rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_BUSY_CYCLES -d $OUT/wb_lds3 -o l3 -- python3 tools/wb_profile.py > /dev/null 2> $OUT/l3.err
python3 tools/summarize_counters.py $OUT/wb_lds1 $OUT/wb_lds2 $OUT/wb_lds3 wb:: $OUT/wb_lds_counters.txt
tail -2 $OUT/l1.err $OUT/l2.err $OUT/l3.err
