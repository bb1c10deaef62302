#!/bin/bash
# Round-2 GPU pass (through gpurun): bench line, rocprof kernel stats of the same command, HBM counters, whole-body
# kernel stats + MFMA counters.  Outputs in gpurun_out/<tag>/; tools/summarize_prof.py / summarize_pmc.py condense them.
set -u
TAG=${1:-r02_b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
python bench.py > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- python3 bench.py --no-cpu-baseline --no-extras --no-graph --steps 20 --warmup 5 > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- python3 bench.py --no-cpu-baseline --no-extras --no-graph --steps 20 --warmup 5 > /dev/null 2> $OUT/pmc_write.err
python3 tools/wb_profile.py > $OUT/wb_run.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/wb_trace -o wb -- python3 tools/wb_profile.py > /dev/null 2> $OUT/wb_trace.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/wb_pmc -o wbpmc -- python3 tools/wb_profile.py > /dev/null 2> $OUT/wb_pmc.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU -d $OUT/wb_pmc2 -o wbpmc2 -- python3 tools/wb_profile.py > /dev/null 2> $OUT/wb_pmc2.err
rocprofv3 --kernel-trace --stats -d $OUT/extras_trace -o extras -- python3 bench.py --no-cpu-baseline --steps 50 --warmup 5 > /dev/null 2> $OUT/extras_trace.err
find $OUT -name "*.csv" | head -40
cat $OUT/wb_run.txt
tail -3 $OUT/wb_pmc.err $OUT/wb_pmc2.err
