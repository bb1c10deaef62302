#!/bin/bash
# Round-6 GPU pass (through gpurun; profile_r05.sh + the round's additions): the bench lines, rocprof kernel stats of the same commands, HBM and SQ counters of the
# RTI kernel that ships (grid of many batches: the FULLN (4, 5) build; one batch at a time: (16, 2)), secondary kernel stats.
# Outputs in gpurun_out/<tag>/; condensed into gpurun_out/<tag>/summary by tools/summarize_r04.py (copy to profiles/).
#   usage: tools/profile_r06.sh <tag> [quick]
set -u
TAG=${1:-r06_x}
QUICK=${2:-}
OUT=gpurun_out/$TAG
mkdir -p $OUT
cd /tmp 2>/dev/null && export TMPDIR=/tmp && cd - >/dev/null
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_flags.json 2> $OUT/bench.err
python bench.py --no-extras --no-cpu-baseline > $OUT/bench.json 2>> $OUT/bench.err
# kernel trace of the SAME commands (driver flags; default flags)
rocprofv3 --kernel-trace --stats -d $OUT/trace_driver -o trace -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > $OUT/bench_driver_flags_under_rocprof.json 2> $OUT/trace.err
rocprofv3 --kernel-trace --stats -d $OUT/trace -o trace -- python3 bench.py --no-cpu-baseline --no-extras > $OUT/bench_under_rocprof.json 2>> $OUT/trace.err
# counters: every launch of the many-batch kernel serves 200 batches (timed, warm-up and spot-check alike)
PMC="bench.py --no-cpu-baseline --no-extras --no-converged --no-stress-check --no-graph --steps 200 --warmup 200"
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o fetch -- python3 $PMC > /dev/null 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o write -- python3 $PMC > /dev/null 2> $OUT/pmc_write.err
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/sq_p1 -o p1 -- python3 $PMC > /dev/null 2> $OUT/sq_p1.err
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d $OUT/sq_p2 -o p2 -- python3 $PMC > /dev/null 2> $OUT/sq_p2.err
if [ -z "$QUICK" ]; then
python bench.py --workload whole_body > $OUT/bench_whole_body.json 2> $OUT/bench_wb.err
python3 tools/wb_profile.py > $OUT/wb_run.txt 2>&1
rocprofv3 --kernel-trace --stats -d $OUT/wb_trace -o wb -- python3 tools/wb_profile.py > /dev/null 2> $OUT/wb_trace.err
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d $OUT/wb_pmc -o wbpmc -- python3 tools/wb_profile.py > /dev/null 2> $OUT/wb_pmc.err
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES -d $OUT/wb_pmc2 -o wbpmc2 -- python3 tools/wb_profile.py > /dev/null 2> $OUT/wb_pmc2.err
rocprofv3 --kernel-trace --stats -d $OUT/ltv_trace -o ltv -- python3 tools/ltv_profile.py > $OUT/ltv_run.txt 2> $OUT/ltv_trace.err
rocprofv3 --kernel-trace --stats -d $OUT/be_trace -o be -- python3 tools/be_profile.py > $OUT/be_run.txt 2> $OUT/be_trace.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVES -d $OUT/be_pmc -o bepmc -- python3 tools/be_profile.py > /dev/null 2> $OUT/be_pmc.err
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVES -d $OUT/ltv_pmc -o ltvpmc -- python3 tools/ltv_profile.py > /dev/null 2> $OUT/ltv_pmc.err
rocprofv3 --kernel-trace --stats -d $OUT/extras_trace -o extras -- python3 bench.py --no-cpu-baseline --steps 50 --warmup 5 > /dev/null 2> $OUT/extras_trace.err
fi
# round 5: per-SIMD timelines of the 20- and 200-batch grids (instrumented twin of the grid build), launch overhead, whole-body rows / exact mode
ALORE_NMPC_TRACE=$OUT/tr20 python3 tools/trace_grid.py 20 2 > $OUT/timeline_20_batches.txt 2>> $OUT/bench.err
ALORE_NMPC_TRACE=$OUT/tr200 python3 tools/trace_grid.py 200 1 > $OUT/timeline_200_batches.txt 2>> $OUT/bench.err
python3 tools/launch_overhead.py 20 > $OUT/launch_overhead.txt 2>> $OUT/bench.err
if [ -z "$QUICK" ]; then
ALORE_WB_STAMPS=1 python3 tools/wb_rows_time.py 600 > $OUT/wb_rows_and_exact_mode.txt 2>&1
fi
# round 6: the scanned backward sweep on and off (one binary, ALORE_NMPC_SCAN), the reference's horizon, one robot, closed loop; a traced two-phase grid
{
for sc in 1 0; do
  echo "== ALORE_NMPC_SCAN=$sc"
  ALORE_NMPC_SCAN=$sc python3 tools/horizon_in_flight.py 50 40 0 4096 -1 2>&1 | grep -v amdgpu
  ALORE_NMPC_SCAN=$sc python3 tools/horizon_in_flight.py 20 40 0x110 4096 -1 2>&1 | grep -v amdgpu
  ALORE_NMPC_SCAN=$sc python3 tools/horizon_in_flight.py 20 40 0x120 512 -1 2>&1 | grep -v amdgpu
  ALORE_NMPC_SCAN=$sc tools/micro/rti_latency 2>&1 | tail -1
  ALORE_NMPC_SCAN=$sc python3 tools/closed_loop_time.py 256 2>&1 | tail -1
  ALORE_NMPC_SCAN=$sc python3 tools/closed_loop_time.py 4096 2>&1 | tail -1
done
} > $OUT/scan_on_off.txt 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $OUT/sq_n50 -o p -- python3 tools/horizon_in_flight.py 50 40 0 4096 -1 > /dev/null 2>&1
python3 tools/summarize_counters.py $OUT/sq_n50 rti_block $OUT/sq_counters_n50_in_flight.txt > /dev/null 2>&1
python3 tools/two_phase_check.py 20 > $OUT/two_phase_check.txt 2>&1
ALORE_NMPC_TP_TRACE=$OUT/tptrace python3 tools/two_phase_trace.py 20 > /dev/null 2>&1
python3 tools/tp_trace.py $OUT/tptrace.2 --timeline > $OUT/two_phase_trace.txt 2>&1
python3 tools/summarize_r04.py $OUT $TAG
for f in scan_on_off.txt sq_counters_n50_in_flight.txt two_phase_check.txt two_phase_trace.txt; do [ -f $OUT/$f ] && cp $OUT/$f $OUT/summary/${TAG}_$f; done
for f in timeline_20_batches.txt timeline_200_batches.txt launch_overhead.txt wb_rows_and_exact_mode.txt; do [ -f $OUT/$f ] && cp $OUT/$f $OUT/summary/${TAG}_$f; done
# the rocpd databases are large: only the summaries travel back
find $OUT -mindepth 1 -maxdepth 1 ! -name summary -exec rm -rf {} +
ls $OUT/summary
