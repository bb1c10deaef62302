cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
export TMPDIR=/tmp
for ns in 0 6000 10000 14000 20000 28000; do
echo "stagger $ns ns: $(ALORE_NMPC_STAGGER_NS=$ns python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("host", round(d["ms_per_step"]*1e3,2), "events", round(d["roofline"]["kernel_ms_avg"]*1e3,2), "steady", round(d["steady_state"]["ms_per_step"]*1e3,2), d["parity_spot_check"]["worst_rel"])')"
done | tee gpurun_out/g6_stagger.txt
