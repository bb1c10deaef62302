#!/bin/bash
set -u
OUT=gpurun_out/${1:-r05_d}
mkdir -p $OUT
for kv in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$kv" >> $OUT/dispatch_gap.txt
  HIP_FORCE_DEV_KERNARG=$kv tools/micro/dispatch_gap 2>&1 | head -3 >> $OUT/dispatch_gap.txt
  HIP_FORCE_DEV_KERNARG=$kv ALORE_NMPC_PERSIST=0 python tools/launch_overhead.py 20 > $OUT/launch_overhead_k$kv.txt 2>> $OUT/err.txt
  HIP_FORCE_DEV_KERNARG=$kv ALORE_NMPC_PERSIST=0 ALORE_NMPC_TRACE=$OUT/tr200k$kv python tools/trace_grid.py 200 1 > $OUT/timeline_200_k$kv.txt 2>> $OUT/err.txt
  HIP_FORCE_DEV_KERNARG=$kv ALORE_NMPC_PERSIST=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-converged > $OUT/bench_k$kv.json 2>> $OUT/err.txt
done
cat $OUT/dispatch_gap.txt
tail -n 3 $OUT/launch_overhead_k0.txt $OUT/launch_overhead_k1.txt
grep -E "duration|gap|lifetime \(" $OUT/timeline_200_k0.txt $OUT/timeline_200_k1.txt
python - <<PY
import json
for k in (0,1):
    d=json.load(open("$OUT/bench_k%d.json"%k))
    print("kernarg dev=%d"%k, "us/step %.2f"%(d["ms_per_step"]*1e3), "frac %.3f"%d["roofline"]["frac"], "kernel_ms %.4f"%d["roofline"]["kernel_ms_per_launch"], "steady %.3f"%(d["steady_state"]["ms_per_step"]*1e3), d["host_clock_breakdown_us"])
PY
