cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/g5
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/g5/trace -o trace -- python3 bench.py --no-cpu-baseline --no-extras --steps 20 --warmup 5 > gpurun_out/g5/bench_driver_under_rocprof.json 2> gpurun_out/g5/trace.err
python3 tools/summarize_prof.py gpurun_out/g5/trace gpurun_out/g5/kernel_stats.txt | grep -v "^(all" | cut -c1-260
rm -rf gpurun_out/g5/trace
python -c "
import json; d=json.loads(open('gpurun_out/g5/bench_driver_under_rocprof.json').read().strip().split('\n')[-1]); print(d['ms_per_step']*1e3, d['roofline']['kernel_ms_per_launch'], d['steady_state']['ms_per_step']*1e3)"
