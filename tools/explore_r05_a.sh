#!/bin/bash
# round 5, first GPU pass: contract figure x3, launch overhead, per-SIMD traces of the 20- and 200-batch grids, stagger sweep
set -u
OUT=gpurun_out/${1:-r05_a}
mkdir -p $OUT
for i in 1 2 3; do
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-converged > $OUT/bench_driver_$i.json 2>> $OUT/err.txt
done
python tools/launch_overhead.py 20 > $OUT/launch_overhead.txt 2>> $OUT/err.txt
ALORE_NMPC_TRACE=$OUT/tr20 python tools/trace_grid.py 20 3 > $OUT/timeline_20.txt 2>> $OUT/err.txt
ALORE_NMPC_TRACE=$OUT/tr200 python tools/trace_grid.py 200 1 > $OUT/timeline_200.txt 2>> $OUT/err.txt
for ns in 0 5000 9000 20000; do
  ALORE_NMPC_STAGGER_NS=$ns ALORE_NMPC_TRACE=$OUT/st$ns python tools/trace_grid.py 20 2 > $OUT/timeline_20_stagger$ns.txt 2>> $OUT/err.txt
done
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "pinned or stress" > $OUT/pytest.txt 2>&1
tail -3 $OUT/pytest.txt
head -c 600 $OUT/bench_driver_1.json; echo
cat $OUT/launch_overhead.txt
