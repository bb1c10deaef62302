cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
for rep in 1 2; do
for g in "" "--no-graph"; do
echo "graph='$g': $(python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-converged $g 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("host", round(d["ms_per_step"]*1e3,2), "events", round(d["roofline"]["kernel_ms_avg"]*1e3,2), "steady", round(d["steady_state"]["ms_per_step"]*1e3,2))')"
done; done | tee gpurun_out/g14_graph.txt
